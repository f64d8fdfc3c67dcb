"""The N > 1 path on the device: image tiles dealt round-robin to the ranks, photon map replicated, the
partial films summed.  A one-GPU box runs the two ranks on GPU 0 over gloo (`--single-device`); the ranks are child
processes of their own (started before they touch the GPU), as bench.py's are under the driver."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "multi_rank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(nproc, script, *args, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script, *args]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-1000:]
    return out


def test_two_ranks_sum_to_the_single_rank_film(tmp_path):
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    out = subprocess.run([sys.executable, WORKER, "--out", one], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    _launch(2, WORKER, "--out", two, "--backend", "gloo", "--single-device")
    f1, f2 = np.load(one), np.load(two)
    assert f1.shape == f2.shape and np.abs(f1[0]).max() > 0 and np.abs(f1[1]).max() > 0
    # computeGradient adds one term from a pixel and one from its +x / +y neighbour, each non-zero on exactly one rank:
    # dx, dy and the throughput of the sharded run are the single-rank film's, up to the order in which the float
    # atomics of a pixel's work items land (two runs of ONE rank differ by as much: test_full_size_properties)
    scale = np.abs(f1[0]).max()
    assert np.allclose(f1, f2, rtol=2e-5, atol=2e-6 * scale)
    assert np.array_equal(f1[0] == 0, f2[0] == 0)  # the throughput's support: no pixel lost or doubled at a tile border


def test_library_rccl_collective_single_rank(tmp_path):
    """gvpm_comm_unique_id / gvpm_comm_init / gvpm_allreduce_film with a world of one: RCCL loaded, communicator
    created, the all-reduce of the film is the identity."""
    a, b = str(tmp_path / "a.npy"), str(tmp_path / "b.npy")
    for path, coll in ((a, "torch"), (b, "gvpm")):
        out = subprocess.run([sys.executable, WORKER, "--out", path, "--collective", coll, "--frame", "64", "--steps", "2"],
                             capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
    fa, fb = np.load(a), np.load(b)
    assert np.abs(fa).max() > 0 and np.allclose(fa, fb, rtol=2e-5, atol=2e-6 * np.abs(fa[0]).max())


def _gpu_count():
    # (torch.cuda.device_count() does not initialise the GPU on this image: the ranks are started as children afterwards)
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL with a world of two (VERDICT round 5, next 5)")
@pytest.mark.parametrize("collective", ["gvpm", "torch"])
def test_two_gpus_rccl_all_reduce_of_the_film(tmp_path, collective):
    """Two ranks on TWO GPUs over RCCL -- the library's own gvpm_comm_init / gvpm_allreduce_film (ncclUniqueId handed round
    by the host) and torch.distributed's nccl backend: the summed film is the single-GPU film.  The first multi-rank RCCL
    run of this library happens here, not in the driver's 8-GPU bench."""
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    out = subprocess.run([sys.executable, WORKER, "--out", one], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    _launch(2, WORKER, "--out", two, "--backend", "nccl", "--collective", collective)
    f1, f2 = np.load(one), np.load(two)
    scale = np.abs(f1[0]).max()
    assert scale > 0 and np.allclose(f1, f2, rtol=2e-5, atol=2e-6 * scale)
    assert np.array_equal(f1[0] == 0, f2[0] == 0)


def test_bench_two_ranks_is_the_strong_sharded_c4_shape():
    out = _launch(2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                  "--single-device", "--frame", "128", "--photons", "60000", "--distinct", "2")
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["scene"] == "fogroom" and line["config"]["frame"] == [128, 128]
    assert "round-robin over 2 ranks" in line["config"]["sharding"] and "all-reduce" in line["config"]["sharding"]
    assert line["roofline"]["kernel_avg_ms"] > 0 and line["config"]["pixels_per_gpu"] == 128 * 128 / 2


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` as the driver spells the N = 1 command (round 5): with WORLD_SIZE
    unset the script starts torch.distributed.run itself -- as a child process, before it touches the GPU -- and passes on
    rank 0's line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                          "--single-device", "--frame", "128", "--photons", "60000", "--distinct", "2"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and line["steps"] == 2


def test_bench_eight_ranks_run_to_completion_on_one_gpu():
    """The driver's 8-GPU launch is `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...`: the same
    command here with the eight ranks on GPU 0 over gloo and a small frame, so that the driver's run is not the first
    8-rank run of this code (sharding by 8, the film all-reduce, the max-over-ranks timing, rank 0's line)."""
    out = _launch(8, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                  "--single-device", "--frame", "128", "--photons", "40000", "--distinct", "2", timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 prints, nobody else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["value"] > 0 and line["steps"] == 2
    assert line["config"]["frame"] == [128, 128] and line["config"]["pixels_per_gpu"] == 128 * 128 / 8
    assert "round-robin over 8 ranks" in line["config"]["sharding"]
    # every evaluation of the frame is counted once: the eight shards' counts add up to a one-rank run's
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c4", "--steps", "2", "--warmup", "1",
                          "--frame", "128", "--photons", "40000", "--distinct", "2", "--only-timed"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert l1["config"]["evaluations"] == line["config"]["evaluations"]
