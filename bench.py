#!/usr/bin/env python3
"""Headline benchmark: G-BRE 3D photon gather + gradient-domain shift (BASELINE.json configs[1]).

One "step" = one SPPM iteration of the hot path (device acceleration-structure build + beam
ordering + gather/shift kernel + normalisation/APA fold) over one batch of synthetic input
already resident in HBM.  Metric: M photon-gather+shift evaluations / s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: image sharding (the frame's 4x4-pixel tiles dealt round-robin to the ranks), photon map
replicated, one all-reduce of the film's 3 planes {throughput, dx, dy} (RCCL through
torch.distributed) at the end of the timed region.  Weak scaling: every rank owns
512x512 pixels of a (tiles_x*512) x (tiles_y*512) frame of the same scene.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def tile_grid(n):
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(n, (n, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tile", type=int, default=512, help="pixels per side of one rank's tile")
    ap.add_argument("--photons", type=int, default=1000000)
    ap.add_argument("--scene", default="cbox")
    ap.add_argument("--technique", default="bre3d", choices=["bre3d", "bre2d"],
                    help="bre3d = BASELINE configs[1] (the bench line); bre2d: the 2D-kernel BRE of the same path (probe)")
    ap.add_argument("--scale", type=float, default=1.0, help="initialScaleVolume")
    ap.add_argument("--distinct", type=int, default=16, help="distinct pre-generated iterations (cycled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-window", type=int, default=512)
    ap.add_argument("--cpu-iters", type=int, default=8, help="iterations of the workload the CPU baseline is timed on")
    ap.add_argument("--emulate-gpus", type=int, default=0,
                    help="single-GPU run of rank 0's shard of an N-GPU frame (sizing probe, e.g. BASELINE configs[3]: "
                         "--emulate-gpus 8 --tile 362 --photons 4000000); not a bench line")
    ap.add_argument("--device-gen", action="store_true",
                    help="probe (SURVEY 8f3): every step shoots its photons and generates its camera beams on the GPU "
                         "(gvpm_devgen_*) inside the timed region instead of reading pre-generated inputs from HBM")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only to smoke-test "
                                                      "the N > 1 code path on a one-GPU box together with --single-device")
    ap.add_argument("--single-device", action="store_true", help="every rank uses GPU 0 (smoke test, not a bench line)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")

    import torch
    import torch.distributed as dist
    from gvpm_amd import abi, hip
    from gvpm_amd.host import SynthScene

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the gather path)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(args.backend)

    nshards = args.emulate_gpus if (args.emulate_gpus and world == 1) else world
    tx, ty = tile_grid(nshards)
    W, H = args.tile * tx, args.tile * ty  # weak scaling: tile^2 pixels per rank
    sc = SynthScene(args.scene, W, H)
    p = sc.params()
    p.vol_technique = abi.GVPM_VOL_BRE3D if args.technique == "bre3d" else abi.GVPM_VOL_BRE2D
    if args.technique == "bre2d":
        p.use_shift_null = 0  # GPMConfig::load rejects useShiftNull for the 2D kernel (gvpm_struct.h:310-313)
    p.initial_scale_volume = args.scale
    m, tris = sc.medium(), sc.triangles()
    ctx = hip.Context(p, device=local_rank)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)

    # ---- synthetic inputs, resident in HBM before the timed region ----
    K, Wu = args.steps, args.warmup
    ndist = max(1, min(args.distinct, max(K, Wu)))
    inputs = []
    keep = []
    host0 = []  # host copies of the first iterations' inputs: the CPU baseline's sample
    gen = hip.DeviceGenerator(sc, device=local_rank) if args.device_gen else None
    for i in range(0 if gen else ndist):
        ph, nb = sc.shoot_photons(i + 1, args.photons)
        # image sharding: the frame's 4x4-pixel tiles are dealt round-robin to the ranks -- contiguous blocks
        # split S-cbox 2.5:1 unevenly (scripts/shard_balance.py: mean/max 0.41 vs 0.99 interleaved)
        rays = sc.camera_beams_interleaved(i + 1, nshards, rank) if nshards > 1 else sc.camera_beams(i + 1)
        if i < args.cpu_iters and rank == 0 and not args.no_cpu_baseline:
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
            soa, nb = gen.shoot_photons((it - 1) % ndist + 1, args.photons)
            rptr, nsets = gen.camera_beams((it - 1) % ndist + 1, nshards, rank)
            gen_sets.append(nsets)
            gen_ph.append(int(soa.n))
        else:
            soa, nph, nb, rptr, nsets = inputs[(it - 1) % ndist]
        ctx.upload_photons_dev(soa)
        ctx.upload_camera_beams_dev(rptr, nsets)
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
        P = args.tile * args.tile
        # algorithmic bytes per gather-kernel launch (BASELINE.md / SURVEY 8d convention)
        bytes_alg = 128.0 * (evals / K) + 320.0 * nsets_avg + 108.0 * P + 128.0 * nph_avg
        achieved = bytes_alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the PMC passes of scripts/profile.sh on this same
        # command (2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes), recorded under profiles/
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tj) and args.tile == 512 and args.photons == 1000000:
            try:
                traffic = json.load(open(tj))["evaluate_bre_kernel"]["hbm_bytes_per_launch"]
                traffic_src = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/profile.sh)"
            except (KeyError, ValueError):
                pass
        out = {
            "metric": "photon gather+shift evaluations per second (G-BRE %s)" % ("3D" if args.technique == "bre3d" else "2D"),
            "value": evals_total / elapsed / 1e6,
            "unit": "Mevals/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wu,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (generated on the device inside every step)" if gen else "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: S-{args.scene} + homogeneous medium, G-BRE {args.technique[3:].upper()} kernel, "
                            f"{args.tile}x{args.tile} px per GPU ({W}x{H} frame), {args.photons} photons/iter, "
                            f"{K} SPPM iters, initialScaleVolume {args.scale}",
                "technique": args.technique, "frame": [W, H], "tile_per_gpu": [args.tile, args.tile],
                "photons_per_iter": args.photons, "iterations": K,
                "sharding": f"4x4-pixel tiles round-robin over {nshards} ranks" if nshards > 1 else "none",
                "evaluations": evals_total, "evals_per_iter_per_gpu": evals / K,
                "tests_per_iter_per_gpu": st["candidates"] / K,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "evaluate_bre_kernel", "kernel_avg_ms": kms, "launches": klaunches,
                "traverse_avg_ms": ctx.phase_time(1)[0], "build_avg_ms": ctx.phase_time(2)[0],
                "note": "build + traversal of step N+1 run on a second stream while this kernel evaluates step N: "
                        "the durations include that sharing (GVPM_PIPELINE=0 gives the isolated ones)",
                "bytes_alg_per_launch": bytes_alg,
            },
            "stats": st,
        }
        if world == 1 and not gen:
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
        if world == 1 and not args.no_cpu_baseline and not gen:
            out["cpu_baseline"] = cpu_baseline(sc, p, m, tris, host0, args)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(sc, p, m, tris, host0, args):
    """The oracle (fp32, fast-math, kd-tree -> BVH walk as the reference) timed on this box's host cores on a bounded
    sample of the same workload: the first --cpu-iters iterations (their own photon maps and camera beams), a centred
    pixel window, every iteration including its kd-tree + BVH build as the reference pays it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    w = min(args.cpu_window, args.tile)
    lo = (args.tile - w) // 2
    r = float(np.float32(p.bsphere_radius) * np.float32(p.initial_scale_volume) * np.float32(0.01))
    cores = os.cpu_count() or 1
    evals, secs, nsets = 0, 0.0, 0
    for ph, nb, rays in host0:
        px = rays["pixel"][:, 0] & 0xFFFF
        py = rays["pixel"][:, 0] >> 16
        sel = (px >= lo) & (px < lo + w) & (py >= lo) & (py < lo + w)
        sample = np.ascontiguousarray(rays[sel])
        _, cnt, s = O.gather_bre(p, m, tris, ph, sample, r, 1, nb, precision=32, use_accel=True, threads=cores, fast=True)
        evals += cnt["evaluations"]
        secs += s
        nsets += sample.shape[0]
    return {
        "value": evals / secs / 1e6, "unit": "Mevals/s", "cores": cores, "kind": "port",
        "sample": f"{len(host0)} iterations (each its own {host0[0][0].n}-photon map), centred {w}x{w} px window "
                  f"({nsets} beam sets in all), kd-tree + BVH builds included ({secs:.2f} s, {evals} evaluations, "
                  f"initial radius in every iteration)",
    }


if __name__ == "__main__":
    main()
