#!/usr/bin/env python3
"""Headline benchmark of the gvpm photon-gather + gradient-domain shift path on MI355X.

One "step" = one SPPM iteration of the hot path (device acceleration-structure build + beam ordering + traversal +
gather/shift evaluation, which folds into the running film) over one batch of synthetic input already resident in
HBM.  Metric: M photon-gather+shift evaluations / s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json `configs`):
  N = 1  configs[1] "C2": S-cbox + homogeneous medium, G-BRE 3D, 512x512, 1 M photons / iteration.
  N > 1  configs[3] "C4": S-fogroom, G-BRE 3D, 1024x1024, 4 M photons / iteration, STRONG scaling: the frame's 4x4-pixel
         tiles are dealt round-robin to the ranks (scripts/shard_balance.py: mean/max 0.99), the photon map is replicated,
         the film's three planes {throughput, dx, dy} are summed across ranks (RCCL through torch.distributed)
         once, inside the timed region, before reconstruction (gvpm.cpp:535).
  --workload c2|c4 forces one; --weak: every rank owns --tile^2 pixels of a larger frame of the same scene instead.
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    "c2": dict(scene="cbox", frame=512, photons=1000000, distinct=16, name="BASELINE configs[1]"),
    "c4": dict(scene="fogroom", frame=1024, photons=4000000, distinct=3, name="BASELINE configs[3]"),
    # the other configs of BASELINE.json at their stated sizes (parity cases: tests/test_configs_gpu.py), as bench lines
    # of the same shape (one GPU): `python bench.py --workload c1|c3|c5`
    "c1": dict(scene="cbox", frame=256, photons=100000, distinct=4, name="BASELINE configs[0]", technique="vpm", samples=40,
               scale=2.0),
    "c3": dict(scene="laser", frame=512, photons=2000000, distinct=3, name="BASELINE configs[2]", technique="beams3d", scale=1.0),
    "c5": dict(scene="laser_in", frame=256, photons=50000, distinct=4, name="BASELINE configs[4]", technique="planes0d",
               scale=1.0),
}
# per technique: kernel the roofline is quoted on, algorithmic bytes per evaluation / per map record
TECH = {
    # (round 6: the gather is the walk and the evaluation, two kernels back to back; the events bracket both)
    "vpm": dict(label="G-VPM (3D point kernel)", kernel="vpm_find_kernel+vpm_eval_kernel", rec=128, what="photons"),
    "beams3d": dict(label="G-Beams (beam x beam, 3D kernel)", kernel="evaluate_beams2_kernel", rec=160, what="beam segments"),
    "beams1d": dict(label="G-Beams (beam x beam, 1D kernel)", kernel="evaluate_beams2_kernel", rec=160, what="beam segments"),
    "planes0d": dict(label="G-Planes (0D kernel)", kernel="gather_planes_kernel", rec=176, what="planes"),
}


def tile_grid(n):
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(n, (n, 1))


def csrc_sha():
    """What the PMC traffic file must have been measured on: the sources of libgvpm_hip.so."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "gvpm_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h", ".cpp")) or os.path.basename(f) == "Makefile":
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto", "c1", "c2", "c3", "c4", "c5"],
                    help="auto: c2 (BASELINE configs[1]) on one GPU, c4 (configs[3], strong-sharded) on several; "
                         "c1 / c3 / c5: the G-VPM / G-Beams / G-Planes configs at their stated sizes (one GPU)")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: every rank owns --tile^2 pixels of a (tiles_x*tile) x (tiles_y*tile) frame")
    ap.add_argument("--frame", type=int, default=0, help="pixels per side of the whole frame (strong scaling; 0: the workload's)")
    ap.add_argument("--tile", type=int, default=512, help="--weak: pixels per side of one rank's share")
    ap.add_argument("--photons", type=int, default=0, help="photons per iteration (0: the workload's)")
    ap.add_argument("--scene", default="", help="synthetic scene (default: the workload's)")
    ap.add_argument("--technique", default="bre3d", choices=["bre3d", "bre2d", "beams1d"],
                    help="bre3d = the bench line; bre2d: the 2D-kernel BRE of the same path (probe); beams1d: with "
                         "--workload c3, the 1D beam kernel instead of the 3D one (probe)")
    ap.add_argument("--scale", type=float, default=0.0, help="initialScaleVolume (0: the workload's, 1.0 for c2 / c4)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="c1 / c3 / c5: CPU time the all-core oracle sample is sized for (a pilot window sets the size)")
    ap.add_argument("--distinct", type=int, default=0, help="distinct pre-generated iterations, cycled (0: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=8, help="iterations of the workload the all-core CPU baseline is timed on")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of step 1 (parity_l2 / relMSE)")
    ap.add_argument("--no-upload-inclusive", action="store_true", help="skip the PCIe-inclusive leg")
    ap.add_argument("--no-isolated", action="store_true", help="skip the single-stream leg (kernel_isolated_ms)")
    ap.add_argument("--other-frame", type=int, default=0, help="tests: frame side of the C1 / C3 / C5 leg (0: the stated sizes)")
    ap.add_argument("--other-photons", type=int, default=0, help="tests: map records of the C1 / C3 / C5 leg (0: the stated sizes)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the C1 / C3 / C5 leg of the headline run (`other_workloads` in the JSON line)")
    ap.add_argument("--only-timed", action="store_true",
                    help="profiling runs: nothing but warm-up + the timed steps touches the GPU (implies the four --no-* flags)")
    ap.add_argument("--emulate-gpus", type=int, default=0,
                    help="single-GPU run of rank 0's shard of an N-GPU frame (sizing probe); not a bench line")
    ap.add_argument("--device-gen", action="store_true",
                    help="probe (SURVEY 8f3): every step shoots its photons and generates its camera beams on the GPU "
                         "(gvpm_devgen_*) inside the timed region instead of reading pre-generated inputs from HBM")
    ap.add_argument("--primal", action="store_true",
                    help="probe (SURVEY 8f3): the PRIMAL beam radiance estimate of the reference's sppm integrator "
                         "(gvpm_gather_primal) on the same inputs instead of the gradient gather; not the bench line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only to smoke-test "
                                                      "the N > 1 code path on a one-GPU box together with --single-device")
    ap.add_argument("--single-device", action="store_true", help="every rank uses GPU 0 (smoke test, not a bench line)")
    args = ap.parse_args()
    if args.only_timed:
        args.no_cpu_baseline = args.no_parity = args.no_upload_inclusive = args.no_isolated = args.no_other_workloads = True
    if args.primal:
        args.no_cpu_baseline = args.no_upload_inclusive = args.no_isolated = True
    if args.workload in ("c1", "c3", "c5"):
        return main_technique(args)
    if args.scale == 0.0:
        args.scale = 1.0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU under torch.distributed.run as a CHILD process,
        # before anything in this process has touched the GPU (a process that has initialised the GPU must not exec another
        # program on this pool); rank 0's JSON line is the child's stdout, its exit code ours
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    # every GVPM_* variable changes what is measured: they are reported, and the evaluation-skipping development
    # switch of round 1 (compiled out since) is refused outright
    gvpm_env = {k: v for k, v in sorted(os.environ.items()) if k.startswith("GVPM_")}
    if "GVPM_DEBUG_FLAGS" in gvpm_env:
        raise SystemExit("bench.py refuses to run with GVPM_DEBUG_FLAGS set (development switches change the measured work)")

    import torch
    import torch.distributed as dist
    from gvpm_amd import abi, hip, metrics
    from gvpm_amd.host import SynthScene

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the gather path)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(args.backend)

    nshards = args.emulate_gpus if (args.emulate_gpus and world == 1) else world
    wl_key = args.workload if args.workload != "auto" else ("c2" if nshards == 1 else "c4")
    wl = WORKLOADS[wl_key]
    scene = args.scene or wl["scene"]
    photons = args.photons or wl["photons"]
    ndist_req = args.distinct or wl["distinct"]
    strong = not args.weak
    if strong:
        W = H = args.frame or wl["frame"]
    else:
        tx, ty = tile_grid(nshards)
        W, H = args.tile * tx, args.tile * ty
    sc = SynthScene(scene, W, H)
    p = sc.params()
    p.vol_technique = abi.GVPM_VOL_BRE3D if args.technique == "bre3d" else abi.GVPM_VOL_BRE2D
    if args.technique == "bre2d":
        p.use_shift_null = 0  # GPMConfig::load rejects useShiftNull for the 2D kernel (gvpm_struct.h:310-313)
    p.initial_scale_volume = args.scale
    if args.primal:
        p.path_set = 0  # (the primal pass has no checkerboard, sppm.cpp:882-1000)
    m, tris = sc.medium(), sc.triangles()
    ctx = hip.Context(p, device=local_rank)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)

    # ---- synthetic inputs, resident in HBM before the timed region ----
    K, Wu = args.steps, args.warmup
    ndist = max(1, min(ndist_req, max(K, Wu)))
    inputs, keep = [], []
    host0 = []  # host copies of the first iterations' inputs: the CPU baseline's and the parity check's sample
    gen = hip.DeviceGenerator(sc, device=local_rank) if args.device_gen else None
    # (the PCIe-inclusive leg replays the timed region from host memory: it needs every distinct input set on the host)
    n_host = max(args.cpu_iters if not args.no_cpu_baseline else 0, ndist if not args.no_upload_inclusive else 0, 1)
    for i in range(0 if gen else ndist):
        ph, nb = sc.shoot_photons(i + 1, photons)
        rays = sc.camera_beams_interleaved(i + 1, nshards, rank) if nshards > 1 else sc.camera_beams(i + 1)
        if i < n_host and rank == 0:
            host0.append((ph, nb, rays))
        soa = abi.PhotonSoA()
        for k in abi.PHOTON_VEC3 + abi.PHOTON_F1 + abi.PHOTON_U1:
            a = getattr(ph, k)
            t = torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
            keep.append(t)
            setattr(soa, k, t.data_ptr())
        soa.n = ph.n
        rt = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
        keep.append(rt)
        inputs.append((soa, ph.n, nb, rt.data_ptr(), rays.shape[0]))
    torch.cuda.synchronize()

    gen_sets, gen_ph = [], []

    def step(it):
        if gen:
            soa, nb = gen.shoot_photons((it - 1) % ndist + 1, photons)
            rptr, nsets = gen.camera_beams((it - 1) % ndist + 1, nshards, rank)
            gen_sets.append(nsets)
            gen_ph.append(int(soa.n))
        else:
            soa, nph, nb, rptr, nsets = inputs[(it - 1) % ndist]
        ctx.upload_photons_dev(soa)
        ctx.upload_camera_beams_dev(rptr, nsets)
        if args.primal:
            ctx.gather_primal(it, nb)
        else:
            ctx.gather(it, nb)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(1, Wu + 1):
        step(it)
    ctx.synchronize()
    ctx.reset()
    ctx.kernel_time()
    ctx.phase_time(1)
    ctx.phase_time(2)
    ev0 = ctx.stats()["evaluations"]
    film = torch.zeros(W * H * 9, dtype=torch.float32, device="cuda") if world > 1 else None
    if world > 1:
        dist.all_reduce(film)  # untimed: RCCL sets its channels and buffers up for this message size

    barrier()
    t0 = time.perf_counter()
    for it in range(1, K + 1):
        step(it)
    if world > 1:
        # one all-reduce of {throughput, dx, dy} before reconstruction (gvpm.cpp:535); each rank's
        # partial film is computeGradient over its own accumulators (zero elsewhere), the sum is the frame's
        ctx.download_film_dev(K, film.data_ptr())
        ctx.synchronize()  # the film kernel ran on the context's stream, the collective runs on torch's
        dist.all_reduce(film)
    ctx.synchronize()
    barrier()
    t1 = time.perf_counter()

    elapsed = t1 - t0
    st = ctx.stats()
    evals = st["evaluations"] - ev0
    kms, klaunches = ctx.kernel_time()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        te = torch.tensor([evals], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.SUM)
        evals_total = float(te.item())
    else:
        evals_total = float(evals)

    if rank == 0:
        nsets_avg = float(np.mean(gen_sets if gen else [x[4] for x in inputs]))
        nph_avg = float(np.mean(gen_ph if gen else [x[1] for x in inputs]))
        P = W * H / nshards  # pixels this rank owns
        # algorithmic bytes per gather-kernel launch (BASELINE.md / SURVEY 8d convention)
        bytes_alg = 128.0 * (evals / K) + 320.0 * nsets_avg + 108.0 * P + 128.0 * nph_avg
        achieved = bytes_alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        sha = csrc_sha()
        # HBM bytes per launch of the dominant kernel from the PMC passes of scripts/profile.sh (2 x FETCH_SIZE +
        # WRITE_SIZE as MI355X_MICROARCH.md prescribes): reported only when that file was measured on THIS workload with
        # THESE kernel sources
        traffic, traffic_src = None, None
        for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
            try:
                d = json.load(open(tj))
                meta = d.get("_meta", {})
                want = dict(technique=args.technique, scene=scene, scale=args.scale, frame=[W, H], photons=photons,
                            n_gpus=world, csrc_sha=sha)
                if all(meta.get(k) == v for k, v in want.items()):
                    traffic = d["evaluate_bre_kernel"]["hbm_bytes_per_launch"]
                    traffic_src = f"{os.path.relpath(tj, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/profile.sh)"
                    break
            except (KeyError, ValueError, OSError):
                pass
        out = {
            "metric": ("PROBE: primal beam-radiance-estimate evaluations per second (sppm volumePhotonPassBRE, %s kernel)"
                       if args.primal else "photon gather+shift evaluations per second (G-BRE %s)") % ("3D" if args.technique == "bre3d" else "2D"),
            "value": evals_total / elapsed / 1e6,
            "unit": "Mevals/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wu,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (generated on the device inside every step)" if gen else "synthetic",
            "config": {
                "workload": f"{wl['name'] if (scene == wl['scene'] and photons == wl['photons']) else 'custom'}: S-{scene} + homogeneous "
                            f"medium, G-BRE {args.technique[3:].upper()} kernel, {W}x{H} frame"
                            + (f" sharded over {nshards} GPUs (4x4-pixel tiles round-robin)" if nshards > 1 else "")
                            + f", {photons} photons/iter, {K} SPPM iters, initialScaleVolume {args.scale}",
                "technique": args.technique, "scene": scene, "frame": [W, H], "pixels_per_gpu": P,
                "photons_per_iter": photons, "iterations": K,
                "sharding": (f"4x4-pixel tiles round-robin over {nshards} ranks, photon map replicated, one film all-reduce "
                             f"in the timed region") if nshards > 1 else "none",
                "evaluations": evals_total, "evals_per_iter_per_gpu": evals / K,
                "tests_per_iter_per_gpu": st["candidates"] / K,
                "env": gvpm_env, "csrc_sha": sha,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "evaluate_primal_kernel" if args.primal else "evaluate_bre_kernel", "kernel_avg_ms": kms, "launches": klaunches,
                "traverse_avg_ms": ctx.phase_time(1)[0], "build_avg_ms": ctx.phase_time(2)[0],
                "note": "build + traversal of the next steps run on other streams while this kernel evaluates step N: "
                        "the durations include that sharing (kernel_isolated_ms: the same kernel alone, GVPM_PIPELINE=0)",
                "bytes_alg_per_launch": bytes_alg,
            },
            "stats": st,
        }
        if world == 1 and not gen and not args.no_upload_inclusive:
            out["upload_inclusive"] = upload_inclusive(hip, sc, p, m, tris, host0[:ndist], K, local_rank, evals / K)
        if world == 1 and not gen and not args.no_isolated:
            # the same kernel without the other streams' kernels beside it (a second handle with GVPM_PIPELINE=0,
            # a few untimed steps after the timed region): reported next to the live figure, which is the one `frac` uses
            os.environ["GVPM_PIPELINE"] = "0"
            try:
                iso = hip.Context(p, device=local_rank)
                iso.upload_scene(*tris)
                iso.upload_medium(m)
                for it in range(1, 7):
                    soa, nph, nb, rptr, nsets = inputs[(it - 1) % ndist]
                    if it == 3:
                        iso.synchronize()
                        iso.kernel_time()
                    iso.upload_photons_dev(soa)
                    iso.upload_camera_beams_dev(rptr, nsets)
                    iso.gather(it, nb)
                iso.synchronize()
                ims, _ = iso.kernel_time()
                iso.close()
                if ims > 0:
                    out["roofline"]["kernel_isolated_ms"] = ims
                    out["roofline"]["frac_isolated"] = bytes_alg / (ims * 1e-3) / 1e9 / 8000.0
            finally:
                os.environ.pop("GVPM_PIPELINE", None)
        if world == 1 and not gen and not args.no_parity and args.primal:
            out.update(parity_primal(hip, metrics, p, m, tris, host0[0], W, H))
        elif world == 1 and not gen and not args.no_parity:
            out.update(parity(hip, metrics, sc, p, m, tris, host0[0], W, H))
        if world == 1 and not args.no_cpu_baseline and not gen:
            out["cpu_baseline"] = cpu_baseline(p, m, tris, host0[:args.cpu_iters], W, H)
        if world == 1 and not gen and not args.primal and args.workload in ("auto", "c2") and not args.no_other_workloads \
                and (args.other_frame or not (args.photons or args.scene or args.frame or args.emulate_gpus or args.weak)):
            # the other configs of BASELINE.json (G-VPM, G-Beams, G-Planes at their stated sizes), timed in this process
            # behind the headline's legs: the same figures `--workload c1|c3|c5 --only-timed` prints (profiles/rNN_c*_bench_line.json)
            ctx.close()
            ctx = None
            out["other_workloads"] = other_workloads(args)
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main_technique(args, emit=True):
    """--workload c1 | c3 | c5: G-VPM / G-Beams / G-Planes at the size BASELINE.json states, one GPU, the same JSON shape
    as the G-BRE line: a step = one SPPM iteration of the hot path (device build + camera-beam ordering + gather/shift
    kernels + the fold into the film) on inputs already in HBM."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != 1 or args.gpus != 1:
        raise SystemExit("--workload c1 / c3 / c5 are one-GPU lines (the sharded workload is c4)")
    gvpm_env = {k: v for k, v in sorted(os.environ.items()) if k.startswith("GVPM_")}
    if "GVPM_DEBUG_FLAGS" in gvpm_env:
        raise SystemExit("bench.py refuses to run with GVPM_DEBUG_FLAGS set (development switches change the measured work)")
    import torch
    from gvpm_amd import abi, hip, metrics
    from gvpm_amd.host import SynthScene
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the gather path)")
    torch.cuda.set_device(0)
    wl = WORKLOADS[args.workload]
    tech = "beams1d" if (args.technique == "beams1d" and args.workload == "c3") else wl["technique"]
    T = TECH[tech]
    scene = args.scene or wl["scene"]
    nrec = args.photons or wl["photons"]
    W = H = args.frame or wl["frame"]
    scale = args.scale or wl["scale"]
    sc = SynthScene(scene, W, H)
    p = sc.params()
    p.initial_scale_volume = scale
    nsamp = wl.get("samples", 0)
    if tech == "vpm":
        p.vol_technique = abi.GVPM_DISTANCE
        p.nb_camera_samples = nsamp
    elif tech == "beams3d":
        p.vol_technique = abi.GVPM_BEAM_BEAM_3D_OPTIMIZED
    elif tech == "beams1d":
        p.vol_technique = abi.GVPM_BEAM_BEAM_1D
        p.use_shift_null = 0
    else:
        p.vol_technique = abi.GVPM_VOL_PLANE0D
        p.use_shift_null = 0  # GPMConfig::load rejects useShiftNull for planes (gvpm_struct.h:310-313)
        p.min_depth = 2       # gvpm.cpp:164-175
    m, tris = sc.medium(), sc.triangles()
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)

    K, Wu = args.steps, args.warmup
    ndist = max(1, min(args.distinct or wl["distinct"], max(K, Wu)))
    keep, inputs, host0 = [], [], []

    def dev(a):
        t = torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else np.ascontiguousarray(a)).cuda()
        keep.append(t)
        return t.data_ptr()

    def dev_soa(ph):
        soa = abi.PhotonSoA()
        for k in abi.PHOTON_VEC3 + abi.PHOTON_F1 + abi.PHOTON_U1:
            setattr(soa, k, dev(getattr(ph, k)))
        soa.n = ph.n
        return soa

    for i in range(ndist):
        it = i + 1
        rec = {}
        if tech == "vpm":
            ph, nb = sc.shoot_photons(it, nrec)
            rays, smp = sc.camera_beams_and_vpm_samples(it, nsamp)
            rec.update(host=(ph, nb, rays, smp), soa=dev_soa(ph), n=ph.n, nb=nb, smp=dev(smp.view(np.uint8).reshape(-1)),
                       nsmp=smp.shape[0])
        elif tech in ("beams3d", "beams1d"):
            ph, en, nb = sc.shoot_beams(it, nrec)
            rays = sc.camera_beams(it)
            rec.update(host=(ph, en, nb, rays), soa=dev_soa(ph), n=ph.n, nb=nb, en=dev(np.ascontiguousarray(en, np.float32)))
        else:
            ph, en, w1, l1, nb = sc.shoot_planes(it, nrec)
            rays = sc.camera_beams(it)
            rec.update(host=(ph, w1, l1, nb, rays), soa=dev_soa(ph), n=ph.n, nb=nb, w1=dev(np.ascontiguousarray(w1, np.float32)),
                       l1=dev(np.ascontiguousarray(l1, np.float32)))
        rec["rays"] = dev(rays.view(np.uint8).reshape(-1))
        rec["nsets"] = rays.shape[0]
        if i == 0:
            host0.append(rec["host"])
        rec.pop("host")
        inputs.append(rec)
    torch.cuda.synchronize()

    def step(it):
        r = inputs[(it - 1) % ndist]
        if tech == "vpm":
            ctx.upload_photons_dev(r["soa"])
        elif tech in ("beams3d", "beams1d"):
            ctx.upload_beams_dev(r["soa"], r["en"])
        else:
            ctx.upload_planes_dev(r["soa"], r["w1"], r["l1"])
        ctx.upload_camera_beams_dev(r["rays"], r["nsets"])
        if tech == "vpm":
            ctx.upload_vpm_samples_dev(r["smp"], r["nsmp"])
        ctx.gather(it, r["nb"])

    for it in range(1, Wu + 1):
        step(it)
    ctx.synchronize()
    ctx.reset()
    for ph_ in (0, 1, 2):
        ctx.phase_time(ph_)
    ev0 = ctx.stats()["evaluations"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, K + 1):
        step(it)
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    evals = st["evaluations"] - ev0
    kms, klaunches = ctx.kernel_time()
    trav_ms, build_ms = ctx.phase_time(1)[0], ctx.phase_time(2)[0]
    nsets_avg = float(np.mean([x["nsets"] for x in inputs]))
    nrec_avg = float(np.mean([x["n"] for x in inputs]))
    P = W * H
    # algorithmic bytes per launch of the dominant kernel (SURVEY 8d / BASELINE.md convention: record bytes per
    # evaluation + 320 B per camera-beam set + 108 B per pixel + one mandatory read of every map record; G-VPM: + 16 B per
    # camera sample)
    bytes_alg = T["rec"] * (evals / K) + 320.0 * nsets_avg + 108.0 * P + T["rec"] * nrec_avg
    if tech == "vpm":
        bytes_alg += 16.0 * float(np.mean([x["nsmp"] for x in inputs]))
    achieved = bytes_alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    sha = csrc_sha()
    traffic, traffic_src = None, None
    for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_%s.json" % args.workload)), reverse=True):
        try:
            d = json.load(open(tj))
            meta = d.get("_meta", {})
            want = dict(technique=tech, scene=scene, scale=scale, frame=[W, H], photons=nrec, n_gpus=1, csrc_sha=sha)
            if all(meta.get(k) == v for k, v in want.items()):
                traffic = sum(d[k]["hbm_bytes_per_launch"] for k in T["kernel"].split("+"))
                traffic_src = f"{os.path.relpath(tj, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/profile.sh)"
                break
        except (KeyError, ValueError, OSError):
            pass
    is_wl = scene == wl["scene"] and nrec == wl["photons"] and W == wl["frame"]
    out = {
        "metric": "photon gather+shift evaluations per second (%s)" % T["label"],
        "value": evals / elapsed / 1e6, "unit": "Mevals/s", "n_gpus": 1, "steps": K, "warmup": Wu,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{wl['name'] if is_wl else 'custom'}: S-{scene} + homogeneous medium, {T['label']}, {W}x{H} frame, "
                        f"{nrec} {T['what']}/iter" + (f", {nsamp} camera samples/pixel" if nsamp else "")
                        + f", {K} SPPM iters, initialScaleVolume {scale}",
            "technique": tech, "scene": scene, "frame": [W, H], "pixels_per_gpu": P, "records_per_iter": nrec,
            "iterations": K, "sharding": "none", "evaluations": float(evals), "evals_per_iter_per_gpu": evals / K,
            "tests_per_iter_per_gpu": st["candidates"] / K, "env": gvpm_env, "csrc_sha": sha,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "traffic": traffic, "traffic_source": traffic_src, "kernel": T["kernel"], "kernel_avg_ms": kms,
            "launches": klaunches, "traverse_avg_ms": trav_ms, "build_avg_ms": build_ms,
            "bytes_alg_per_launch": bytes_alg,
            "bytes_alg_formula": "%d*H + 320*B + 108*P + %d*N%s" % (T["rec"], T["rec"], " + 16*S" if tech == "vpm" else ""),
            "note": "one stream: the step's kernels run one after the other (build, traversal / plan, evaluation); "
                    "kernel_avg_ms brackets the dominant kernel alone (HIP events on its stream)"
                    + ("; G-VPM: the gather is vpm_find_kernel (the walk) and vpm_eval_kernel behind it -- the events bracket both, "
                       "their rocprof durations add up to kernel_avg_ms" if tech == "vpm" else ""),
        },
        "stats": st,
    }
    if not args.no_parity:
        out.update(parity_technique(hip, metrics, tech, p, m, tris, host0[0], W, H))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_technique(tech, p, m, tris, host0[0], W, H, args.cpu_seconds)
    if emit:
        print(json.dumps(out), flush=True)
    ctx.close()
    del keep[:]
    torch.cuda.empty_cache()
    return out


# steps per workload of the headline run's `other_workloads` leg (a G-Beams step is 20 ms, a G-VPM step half a millisecond)
OTHER_STEPS = {"c1": 24, "c3": 8, "c5": 12}


def other_workloads(args):
    """C1 / C3 / C5 behind the headline: warm-up + timed steps only (no parity, no CPU leg: tests/test_configs_gpu.py and
    `--workload cN` cover those), one entry per workload with the figures of its own bench line."""
    import copy
    res = {}
    for w in ("c1", "c3", "c5"):
        a = copy.copy(args)
        a.workload, a.steps, a.warmup = w, OTHER_STEPS[w], 2
        a.no_parity = a.no_cpu_baseline = a.no_upload_inclusive = a.no_isolated = a.only_timed = True
        a.photons, a.frame, a.distinct = args.other_photons, args.other_frame, 0  # (0: the workload's stated size)
        a.scene, a.scale, a.technique = "", 0.0, "bre3d"
        if args.other_frame:
            a.steps = 3
        t0 = time.perf_counter()
        try:
            o = main_technique(a, emit=False)
            res[w] = {"metric": o["metric"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"],
                      "steps": o["steps"], "warmup": o["warmup"], "workload": o["config"]["workload"],
                      "roofline_frac": o["roofline"]["frac"], "kernel": o["roofline"]["kernel"],
                      "kernel_avg_ms": o["roofline"]["kernel_avg_ms"], "evaluations": o["config"]["evaluations"],
                      "csrc_sha": o["config"]["csrc_sha"], "leg_seconds": time.perf_counter() - t0}
        except Exception as e:  # (the headline line must still be printed)
            res[w] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def _window(rays, x0, y0, w, h):
    px = rays["pixel"][:, 0] & 0xFFFF
    py = rays["pixel"][:, 0] >> 16
    return (px >= x0) & (px < x0 + w) & (py >= y0) & (py < y0 + h)


def _sub_vpm(rays, smp, sel):
    remap = np.full(len(rays), -1, np.int64)
    remap[np.nonzero(sel)[0]] = np.arange(int(sel.sum()))
    keepm = remap[smp["set"]] >= 0
    s2 = smp[keepm].copy()
    s2["set"] = remap[s2["set"]]
    return np.ascontiguousarray(rays[sel]), s2


def parity_technique(hip, metrics, tech, p, m, tris, first, W, H):
    """Step 1 of the workload on a centred 24x24-pixel window (the full map) against the fp64 oracle: evaluation counts
    and the per-pixel L2 of the 27 accumulators over the mean luminance."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    w = 24
    x0, y0 = (W - w) // 2 // 4 * 4, (H - w) // 2 // 4 * 4
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    if tech == "vpm":
        ph, nb, rays, smp = first
        wr, ws = _sub_vpm(rays, smp, _window(rays, x0, y0, w, w))
        ctx.upload_photons(ph); ctx.upload_camera_beams(wr); ctx.upload_vpm_samples(ws)
        ctx.gather(1, nb)
        ref, _, _, cnt, _ = O.gather_vpm(p, m, tris, ph, wr, ws, 64, use_accel=True)
    elif tech in ("beams3d", "beams1d"):
        ph, en, nb, rays = first
        wr = np.ascontiguousarray(rays[_window(rays, x0, y0, w, w)])
        ctx.upload_beams(ph, en); ctx.upload_camera_beams(wr)
        rad = ctx.radius()
        ctx.gather(1, nb)
        ref, cnt, _ = O.gather_beams(p, m, tris, ph, en, wr, rad, 1, nb, 64, use_accel=True)
    else:
        ph, w1, l1, nb, rays = first
        wr = np.ascontiguousarray(rays[_window(rays, x0, y0, w, w)])
        ctx.upload_planes(ph, w1, l1); ctx.upload_camera_beams(wr)
        ctx.gather(1, nb)
        ref, cnt, _ = O.gather_planes(p, m, tris, ph, w1, l1, wr, 1, nb, 64, use_accel=True)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    if tech == "vpm":
        ref = ref / 1.0  # plain sums on both sides (the film divides by the emitted count)
    win = (slice(y0, y0 + w), slice(x0, x0 + w))
    lum = max(ref[win][..., 0:3].mean(), 1e-30)
    l2 = metrics.l2_over_luminance(acc[win], ref[win], lum)
    return {"parity_l2": l2,
            "parity": {"what": f"step 1, centred {w}x{w}-pixel window, full map, fp64 oracle through the reference's accelerator",
                       "evaluations_device": st["evaluations"], "evaluations_oracle": cnt["evaluations"],
                       "l2_accumulators": l2, "bar": "L2 / mean luminance < 1e-3 (BASELINE.md), evaluation counts equal"}}


def cpu_baseline_technique(tech, p, m, tris, first, W, H, budget_s):
    """The oracle (fp32, fast-math) through the REFERENCE's accelerator (kd-tree for G-VPM, SubBeamBVH for G-Beams,
    PhotonPlaneBVH for G-Planes; the serial build included, as the reference pays it) on this box's host cores: a pilot
    window sets the size of the timed sample (about `budget_s` of CPU work), iteration 1 of the same workload, the full
    map.  Reported, not targeted."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores, hw_threads, physical = host_threads()
    r = float(np.float32(p.bsphere_radius) * np.float32(p.initial_scale_volume) * np.float32(0.01))

    def run(rows, threads):
        tm = {}
        y0 = max(0, (H - rows) // 2)
        if tech == "vpm":
            ph, nb, rays, smp = first
            wr, ws = _sub_vpm(rays, smp, _window(rays, 0, y0, W, rows))
            _, _, _, cnt, s = O.gather_vpm(p, m, tris, ph, wr, ws, 32, use_accel=True, threads=threads, fast=True, timing=tm)
        elif tech in ("beams3d", "beams1d"):
            ph, en, nb, rays = first
            wr = np.ascontiguousarray(rays[_window(rays, 0, y0, W, rows)])
            _, cnt, s = O.gather_beams(p, m, tris, ph, en, wr, r, 1, nb, 32, use_accel=True, threads=threads, fast=True, timing=tm)
        else:
            ph, w1, l1, nb, rays = first
            wr = np.ascontiguousarray(rays[_window(rays, 0, y0, W, rows)])
            _, cnt, s = O.gather_planes(p, m, tris, ph, w1, l1, wr, 1, nb, 32, use_accel=True, threads=threads, fast=True, timing=tm)
        return cnt["evaluations"], s, tm.get("build_s", 0.0), wr.shape[0]

    # pilot: a band of 8 pixel rows through the middle of the frame
    ev, s, b, _ = run(min(8, H), cores)
    gather_rate = ev / max(s - b, 1e-6)
    rows = H
    if gather_rate > 0:
        # evaluations per row vary over the frame: size by the pilot's rows, cap at the frame
        per_row = max(ev / min(8, H), 1.0)
        rows = int(min(H, max(8, (budget_s * gather_rate) / per_row)))
    ev, s, b, nsets = run(rows, cores)
    out = {"value": ev / s / 1e6, "unit": "Mevals/s", "cores": cores, "physical_cores": physical, "host_threads": hw_threads, "kind": "port",
           "build_s": b, "gather_s": s - b, "gather_only_value": ev / max(s - b, 1e-9) / 1e6,
           "sample": f"iteration 1, the middle {rows} of {H} pixel rows ({nsets} beam sets), the full map through the "
                     f"reference's accelerator; its serial build ({b:.2f} s) + the gather on {cores} threads ({s - b:.2f} s), "
                     f"{ev} evaluations"}
    ev1, s1, b1, n1 = run(min(2, H), 1)
    out["one_thread"] = {"value": ev1 / s1 / 1e6, "unit": "Mevals/s", "cores": 1, "build_s": b1, "gather_s": s1 - b1,
                         "sample": f"iteration 1, the middle {min(2, H)} pixel rows ({n1} beam sets), build included "
                                   f"({s1:.2f} s, {ev1} evaluations)"}
    return out


def upload_inclusive(hip, sc, p, m, tris, host0, K, device, evals_per_step_timed):
    """The same K steps fed from HOST memory: pinned buffers, copies on the handle's copy stream, the copy of step N+1 in
    flight while step N runs.  Never `value`: the PCIe-inclusive rate SURVEY 8d asks to have beside it.  The headline of
    this leg (`compact`) is what the shim uploads: packed photon records (76 bytes) + compact beam sets (60 bytes, rebuilt
    from the sensor on the device; deeper edges as 272-byte records); beside it the round-3 records (`packed`: 76 / 272) and
    the fp32 SoA entry points (120 / 320).  EVERY pass starts from gvpm_reset, so that its K gathers run at the radii of
    the timed region (gatherBRE shrinks the radius on every call: without the reset a later pass measures a lighter
    gather), and the headline cycles the same input sets as the timed region: its evaluation count must equal it."""
    table = hip.MaterialTable()
    sensor = sc.sensor()
    # (the packed records first: on a box whose GPU-side NUMA node is short of free memory the later pinned blocks land on
    # the far node and are read at ~33 GB/s instead of 55 -- seen on one box of the pool, after many processes had run)
    # (round 6: the photons of the headline as LINKED records, gvpm_pack_photons_linked -- 40 / 48 / 76 bytes by kind, ~50 a photon)
    compact = [hip.PinnedPacked(ph, rays, table, sensor=sensor, jitter=sc.jitter(i + 1, rays), linked=not os.environ.get("GVPM_BENCH_NO_LINKED"))
               for i, (ph, nb, rays) in enumerate(host0)]
    compact76 = [hip.PinnedPacked(ph, rays, table, sensor=sensor, jitter=sc.jitter(i + 1, rays)) for i, (ph, nb, rays) in enumerate(host0[:2])]
    packed = [hip.PinnedPacked(ph, rays, table) for ph, nb, rays in host0[:2]]
    soa = [(hip.PinnedPhotons(ph.n).fill(ph), hip.PinnedRays(rays)) for ph, nb, rays in host0[:2]]
    nbs = [nb for ph, nb, rays in host0]
    nbytes = {"compact": float(np.mean([c.nbytes for c in compact])), "compact76": float(compact76[0].nbytes), "packed": float(packed[0].nbytes),
              "soa": float(host0[0][0].n * 120 + host0[0][2].nbytes)}
    ctx = hip.Context(p, device=device)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    ctx.upload_materials(table)
    ctx.upload_sensor(sensor)
    res = {}
    for mode in os.environ.get("GVPM_BENCH_UPLOAD_MODES", "compact,compact76,packed,prefetch,serial").split(","):
        sets = compact if mode == "compact" else (compact76 if mode == "compact76" else (packed if mode == "packed" else soa))
        ns = len(sets)
        best = None
        for rep in range(4):  # first pass: allocations; then the fastest of three (see "how")
            ctx.reset()
            ctx.synchronize()
            ev0 = ctx.stats()["evaluations"]
            t0 = time.perf_counter()
            if mode in ("compact", "compact76", "packed"):
                ctx.upload_pinned_packed(sets[0])
            else:
                ctx.upload_pinned(*sets[0])
            for it in range(1, K + 1):
                if mode in ("compact", "compact76", "packed"):
                    if it < K:
                        ctx.prefetch_packed(sets[it % ns])
                elif mode == "prefetch":
                    if it < K:
                        ctx.prefetch(*sets[it % ns])
                elif it > 1:
                    ctx.upload_pinned(*sets[(it - 1) % ns])
                ctx.gather(it, nbs[(it - 1) % ns])
                if os.environ.get("GVPM_BENCH_UPLOAD_TRACE") == "2":
                    print("[upload]   %s rep %d it %d returned at %.3f ms" % (mode, rep, it, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            ev = ctx.stats()["evaluations"] - ev0
            if rep > 0 and (best is None or dt < best[0]):
                best = (dt, ev)
        dt, ev = best
        res[mode] = dict(ms_per_step=dt / K * 1e3, value=ev / dt / 1e6, evals_per_step=ev / K)
        if os.environ.get("GVPM_BENCH_UPLOAD_TRACE"):
            print("[upload] %s %.3f ms/step, %.0f evaluations/step" % (mode, dt / K * 1e3, ev / K), file=sys.stderr)
    ctx.close()
    for c in compact + compact76 + packed:
        c.close()
    for a, b in soa:
        a.close()
        b.close()
    head = res.get("compact") or next(iter(res.values()))
    # the leg exists to say what an iteration costs WITH its upload: it must be the timed region's iteration
    if "compact" in res and abs(head["evals_per_step"] - evals_per_step_timed) > 0.01 * evals_per_step_timed:
        raise SystemExit("upload_inclusive ran %.0f evaluations per step, the timed region %.0f: not the same workload"
                         % (head["evals_per_step"], evals_per_step_timed))

    def leg(mode, nb):
        r = res.get(mode)
        return None if r is None else dict(r, host_bytes_per_step=nb, pcie_gb_per_s_at_this_rate=nb / (r["ms_per_step"] * 1e-3) / 1e9)
    return {
        "value": head["value"], "unit": "Mevals/s", "ms_per_step": head["ms_per_step"],
        "evals_per_step": head["evals_per_step"], "evals_per_step_timed_region": evals_per_step_timed,
        "host_bytes_per_step": nbytes["compact"],
        "pcie_gb_per_s_at_this_rate": nbytes["compact"] / (head["ms_per_step"] * 1e-3) / 1e9,
        "sets": {"compact": compact[0].ncompact, "full": compact[0].nfull},
        "photon_bytes": float(np.mean([c.photon_bytes for c in compact])) / max(1, compact[0].n),
        "packed_photons_76": leg("compact76", nbytes["compact76"]),
        "packed": leg("packed", nbytes["packed"]),
        "soa": dict(leg("prefetch", nbytes["soa"]) or {}, ms_per_step_without_prefetch=(res.get("serial") or {}).get("ms_per_step")),
        "how": "pinned host buffers: gvpm_pack_photons_linked blobs (40 / 48 / 76 bytes a photon by kind: its parent is the "
               "previous photon, the emitter, anything else; `packed_photons_76`: round 3's 76-byte records beside the same sets) + "
               "gvpm_pack_camera_beams_compact sets (60 bytes "
               "for a sensor-adjacent edge, rebuilt from gvpm_upload_sensor on the device; 272 for deeper edges), decoded at the "
               "head of the consuming gather's build; gvpm_prefetch_* of step N+1 before gvpm_gather of step N; the input sets "
               "of the timed region, in its order, gvpm_reset before every pass.  `packed`: round 3's records (76 / 272 bytes), "
               "`soa`: the fp32 SoA entry points (gvpm_host_alloc_photons, 120 / 320 bytes), both over the first two input sets.  "
               "Each figure is the fastest of three passes of K steps after an allocation pass: on some boxes of the pool the "
               "host-to-device copies run at ~33 instead of 55 GB/s for stretches of 50-100 ms, which a single pass of 16 "
               "steps can fall into",
    }


def parity_primal(hip, metrics, p, m, tris, first, W, H):
    """--primal: step 1 on a centred 32x32-pixel window (the full photon map) against the fp64 oracle's literal restatement of
    the reference's sppm pass (oracle/gvpm_oracle_primal.hpp)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    ph, nb, rays = first
    w = 32
    x0, y0 = (W - w) // 2, (H - w) // 2
    wr = np.ascontiguousarray(rays[_window(rays, x0, y0, w, w)])
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    ctx.upload_photons(ph)
    ctx.upload_camera_beams(wr)
    r = ctx.radius()
    ctx.gather_primal(1, nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    ref, cnt = O.gather_primal_bre(p, m, tris, ph, wr, r, 1, nb, precision=64, use_accel=True)
    win = (slice(y0, y0 + w), slice(x0, x0 + w))
    lum = ref[win][..., 0:3].mean()
    l2 = metrics.l2_over_luminance(acc[win], ref[win], lum)
    return {"parity_l2": l2,
            "parity": {"what": f"step 1, centred {w}x{w}-pixel window, full photon map, fp64 oracle of the primal pass (walk over the hierarchy)",
                       "evaluations_device": st["evaluations"], "evaluations_oracle": cnt["evaluations"], "l2_accumulators": l2}}


def parity(hip, metrics, sc, p, m, tris, first, W, H):
    """Step 1 of the workload against the fp64 oracle on a centred 32x32-pixel window (the full photon map): per-pixel L2
    of the 27 accumulators and of throughput / dx / dy over the mean luminance, and relMSE (imageerrors.h:117-121)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    ph, nb, rays = first
    w = 32
    x0, y0 = (W - w) // 2, (H - w) // 2
    px = rays["pixel"][:, 0] & 0xFFFF
    py = rays["pixel"][:, 0] >> 16
    sel = (px >= x0) & (px < x0 + w) & (py >= y0) & (py < y0 + w)
    wr = np.ascontiguousarray(rays[sel])
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    ctx.upload_photons(ph)
    ctx.upload_camera_beams(wr)
    r = ctx.radius()
    ctx.gather(1, nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    film = ctx.download_film(1, True)
    ctx.close()
    ref, cnt, _ = O.gather_bre(p, m, tris, ph, wr, r, 1, nb, precision=64, use_accel=True)
    rfilm = O.assemble(ref, 1, True)
    win = (slice(y0, y0 + w), slice(x0, x0 + w))
    lum = ref[win][..., 0:3].mean()
    return {
        "parity_l2": max([metrics.l2_over_luminance(acc[win], ref[win], lum)] +
                         [metrics.l2_over_luminance(a[win], b[win], lum) for a, b in zip(film, rfilm)]),
        "parity": {
            "what": f"step 1, centred {w}x{w}-pixel window, full photon map, fp64 oracle (oracle/, reference BVH walk)",
            "evaluations_device": st["evaluations"], "evaluations_oracle": cnt["evaluations"],
            "l2_accumulators": metrics.l2_over_luminance(acc[win], ref[win], lum),
            "l2_throughput_dx_dy": [metrics.l2_over_luminance(a[win], b[win], lum) for a, b in zip(film, rfilm)],
            "relMSE_throughput_dx_dy": [metrics.rel_mse(a[win], b[win]) for a, b in zip(film, rfilm)],
            "bar": "L2 / mean luminance < 1e-3 (BASELINE.md), evaluation counts equal",
        },
    }


def host_threads():
    """(threads this process may actually run on, hardware threads of the host, physical cores).  The first is what the
    baseline uses: os.cpu_count() is the MACHINE's -- a container's affinity mask or CPU quota (cgroup v2 cpu.max, v1
    cfs_quota_us) can be far smaller, and a gather spread over 256 threads that share 16 cores measures the scheduler."""
    hw = os.cpu_count() or 1
    usable = hw
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        if quota:
            usable = max(1, min(usable, int(quota + 0.5)))
    except Exception:
        pass
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or hw
    except Exception:
        physical = hw
    return usable, hw, physical


def cpu_baseline(p, m, tris, host0, W, H):
    """The oracle (fp32, fast-math, kd-tree -> BVH walk as the reference) timed on this box's host cores on a bounded
    sample of the same workload.  All cores: the first --cpu-iters iterations (their own photon maps and camera beams),
    the full frame, every iteration including its kd-tree + BVH build as the reference pays it.  One thread: a centred
    pixel window of the first iteration.  Reported, not targeted."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    r = float(np.float32(p.bsphere_radius) * np.float32(p.initial_scale_volume) * np.float32(0.01))
    cores, hw_threads, physical = host_threads()
    evals, secs, nsets, build = 0, 0.0, 0, 0.0
    cpu0 = time.process_time()
    for ph, nb, rays in host0:
        tm = {}
        _, cnt, s = O.gather_bre(p, m, tris, ph, rays, r, 1, nb, precision=32, use_accel=True, threads=cores, fast=True,
                                 timing=tm)
        evals += cnt["evaluations"]
        secs += s
        build += tm["build_s"]
        nsets += rays.shape[0]
    cpu_s = time.process_time() - cpu0
    # one thread: iteration 1, a window sized for ~5-10 s (the whole 512x512 frame of C2)
    ph, nb, rays = host0[0]
    w1 = min(W, H, 512)
    x0, y0 = (W - w1) // 2, (H - w1) // 2
    px = rays["pixel"][:, 0] & 0xFFFF
    py = rays["pixel"][:, 0] >> 16
    sel = (px >= x0) & (px < x0 + w1) & (py >= y0) & (py < y0 + w1)
    tm1 = {}
    _, cnt1, s1 = O.gather_bre(p, m, tris, ph, np.ascontiguousarray(rays[sel]), r, 1, nb, precision=32, use_accel=True, threads=1,
                               fast=True, timing=tm1)
    return {
        "value": evals / secs / 1e6, "unit": "Mevals/s", "cores": cores, "kind": "port",
        # `cores` = the THREADS the gather ran on (what the process may use: affinity mask and CPU quota, host_threads());
        # the host's hardware threads and physical cores beside it, and how busy
        # the threads were during the gather (CPU seconds / (threads x gather seconds): the serial build counts for one)
        "physical_cores": physical, "host_threads": hw_threads,
        "thread_utilisation": (cpu_s - build) / max(cores * (secs - build), 1e-9),
        # the kd-tree + BVH build is serial in the reference (gvpm.cpp:450-454) and in the port: the split says how much of
        # the all-core figure is that one thread
        "build_s": build, "gather_s": secs - build, "gather_only_value": evals / max(secs - build, 1e-9) / 1e6,
        "sample": f"{len(host0)} iterations (each its own {host0[0][0].n}-photon map), the full {W}x{H} frame "
                  f"({nsets} beam sets in all), kd-tree + BVH builds included ({secs:.2f} s, {evals} evaluations); every "
                  f"iteration at the INITIAL radius (the GPU's radius shrinks with the iteration: its later steps find fewer "
                  f"photons per beam)",
        "one_thread": {"value": cnt1["evaluations"] / s1 / 1e6, "unit": "Mevals/s", "cores": 1,
                       "build_s": tm1["build_s"], "gather_s": s1 - tm1["build_s"],
                       "sample": f"iteration 1, centred {w1}x{w1}-pixel window ({int(sel.sum())} beam sets), the 1-thread "
                                 f"kd-tree + BVH build of the full {ph.n}-photon map included ({s1:.2f} s, "
                                 f"{cnt1['evaluations']} evaluations)"},
    }


if __name__ == "__main__":
    main()
