"""bench.py on the GPU box, at reduced sizes: every workload's line (c2 the default; c1 / c3 / c5 the G-VPM / G-Beams /
G-Planes configs) carries the contract keys with `roofline` and `cpu_baseline`, its parity leg agrees with the oracle,
and the default command line still finishes in minutes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def check_line(d, steps):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == steps and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel_avg_ms"] > 0 and r["launches"] == steps
    # achieved = algorithmic bytes per launch / the kernel's average duration
    assert abs(r["achieved"] - r["bytes_alg_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["build_s"] >= 0 and c["gather_s"] > 0
    assert d["parity"]["evaluations_device"] == d["parity"]["evaluations_oracle"] > 0
    assert d["parity_l2"] < 1e-3


@pytest.mark.parametrize("wl,extra", [
    ("c1", ["--frame", "64", "--photons", "20000"]),
    ("c3", ["--frame", "64", "--photons", "30000"]),
    ("c3", ["--frame", "64", "--photons", "30000", "--technique", "beams1d"]),
    ("c5", ["--frame", "64", "--photons", "4000"]),
])
def test_other_workloads_emit_the_same_line(wl, extra):
    d = run_bench("--workload", wl, "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5", *extra)
    check_line(d, 3)
    assert d["config"]["workload"].startswith("custom")  # reduced sizes are not the BASELINE config: the line says so
    assert d["roofline"]["kernel"] in ("vpm_find_kernel+vpm_eval_kernel", "evaluate_beams2_kernel", "gather_planes_kernel")


def test_default_workload_reduced():
    d = run_bench("--steps", "3", "--warmup", "1", "--frame", "96", "--photons", "50000", "--cpu-iters", "1")
    check_line(d, 3)
    assert d["roofline"]["kernel"] == "evaluate_bre_kernel" and "upload_inclusive" in d
    # the PCIe-inclusive leg replays the timed region's iterations (same input sets, same radii): same evaluation count
    u = d["upload_inclusive"]
    assert abs(u["evals_per_step"] - d["config"]["evals_per_iter_per_gpu"]) <= 0.01 * d["config"]["evals_per_iter_per_gpu"]
    assert u["evals_per_step_timed_region"] == d["config"]["evals_per_iter_per_gpu"]
    for leg in (u["packed"], u["soa"]):  # (two input sets instead of three: within the spread of the sets)
        assert abs(leg["evals_per_step"] - u["evals_per_step"]) <= 0.1 * u["evals_per_step"]
    assert u["sets"]["compact"] > 0 and u["sets"]["full"] == 0
    assert u["host_bytes_per_step"] < 0.75 * u["packed"]["host_bytes_per_step"] < u["soa"]["host_bytes_per_step"]


def test_headline_run_carries_the_other_workloads():
    """The default command line times C1 / C3 / C5 behind the headline's legs: one entry per
    workload with the figures of its own line -- here at reduced sizes, which the entries say."""
    d = run_bench("--steps", "3", "--warmup", "1", "--frame", "96", "--photons", "50000", "--cpu-iters", "1", "--other-frame", "64",
                  "--other-photons", "4000")
    check_line(d, 3)
    ow = d["other_workloads"]
    assert sorted(ow) == ["c1", "c3", "c5"]
    for w, kern in (("c1", "vpm_find_kernel+vpm_eval_kernel"), ("c3", "evaluate_beams2_kernel"), ("c5", "gather_planes_kernel")):
        o = ow[w]
        assert "error" not in o, o
        for k in ("metric", "value", "unit", "ms_per_step", "steps", "workload", "roofline_frac", "kernel", "kernel_avg_ms",
                  "evaluations", "csrc_sha"):
            assert k in o, (w, k)
        assert o["kernel"] == kern and o["value"] > 0 and o["ms_per_step"] > 0 and o["evaluations"] > 0
        assert o["workload"].startswith("custom") and o["csrc_sha"] == d["config"]["csrc_sha"]
