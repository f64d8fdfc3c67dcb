"""bench.py's command-line contract (no GPU needed): the driver calls `python bench.py --gpus N --steps K --warmup W`
and parses ONE JSON line; the keys that line must carry are spelled in the source."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_help_lists_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_json_line_has_the_contract_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert re.search(r'"%s"\s*[:\]]' % key, src), key
    for key in ("bound", "achieved", "peak", "frac", "traffic"):
        assert '"%s"' % key in src, key
    for key in ("cores", "kind", "sample"):
        assert '"%s"' % key in src, key
    assert '"workload"' in src and '"vs_baseline": None' in src
    # round 2: the PCIe-inclusive leg, the parity figures and the 1-thread CPU baseline ride on the same line
    for key in ("upload_inclusive", "parity_l2", "relMSE_throughput_dx_dy", "one_thread", "env", "csrc_sha"):
        assert '"%s"' % key in src, key
    # round 6: C1 / C3 / C5 ride on the headline's line, one entry each with the figures of their own lines
    assert 'out["other_workloads"] = other_workloads(args)' in src
    for key in ("roofline_frac", "kernel_avg_ms", "evaluations", "leg_seconds"):
        assert '"%s"' % key in src, key
    assert "--no-other-workloads" in src


def test_development_switches_are_refused():
    env = dict(os.environ, GVPM_DEBUG_FLAGS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and "GVPM_DEBUG_FLAGS" in (out.stderr + out.stdout)


def test_rel_mse_is_the_reference_formula():
    import numpy as np
    sys.path.insert(0, ROOT)
    from gvpm_amd import metrics
    rng = np.random.default_rng(3)
    ref = rng.random((5, 7, 3)) * 2
    img = ref + rng.normal(0, 0.1, ref.shape)
    # metricPix + errorNorm of scripts/rgbe/sources/imageerrors.h, pixel by pixel
    tot = 0.0
    for a, b in zip(img.reshape(-1, 3), ref.reshape(-1, 3)):
        diff = ((a[0] - b[0]) + (a[1] - b[1]) + (a[2] - b[2])) / 3
        gray = (b[0] + b[1] + b[2]) / 3
        tot += diff * diff / (gray * gray + 0.001)
    assert abs(metrics.rel_mse(img, ref) - tot / 35) < 1e-12
    assert metrics.rel_mse(ref, ref) == 0.0
    assert abs(metrics.l2_over_luminance(img, ref, 1.0) - np.sqrt(((img - ref) ** 2).mean())) < 1e-15


def test_without_a_gpu_it_refuses_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)


def test_cpu_baseline_leg_of_the_other_workloads_runs_on_small_frames():
    """`bench.py --workload c1|c3|c5`: the CPU leg (oracle through the reference's accelerator, build / gather split, a
    pilot window that sizes the sample) on tiny frames -- the GPU side of those lines is -m gpu (tests/test_bench_gpu.py)."""
    sys.path.insert(0, ROOT)
    import bench
    from gvpm_amd import abi
    from gvpm_amd.host import SynthScene
    for wl in ("c1", "c3", "c5"):
        assert wl in bench.WORKLOADS and bench.WORKLOADS[wl]["technique"] in bench.TECH
    for tech, scene in (("vpm", "cbox"), ("beams3d", "laser"), ("beams1d", "laser"), ("planes0d", "laser_in")):
        W = H = 16
        sc = SynthScene(scene, W, H)
        p = sc.params()
        p.initial_scale_volume = 3.0
        if tech == "vpm":
            p.vol_technique = abi.GVPM_DISTANCE
            p.nb_camera_samples = 4
            ph, nb = sc.shoot_photons(1, 3000)
            rays, smp = sc.camera_beams_and_vpm_samples(1, 4)
            first = (ph, nb, rays, smp)
        elif tech.startswith("beams"):
            p.vol_technique = abi.GVPM_BEAM_BEAM_3D_OPTIMIZED if tech == "beams3d" else abi.GVPM_BEAM_BEAM_1D
            if tech == "beams1d":
                p.use_shift_null = 0
            ph, en, nb = sc.shoot_beams(1, 3000)
            first = (ph, en, nb, sc.camera_beams(1))
        else:
            p.vol_technique = abi.GVPM_VOL_PLANE0D
            p.use_shift_null = 0
            p.min_depth = 2
            ph, en, w1, l1, nb = sc.shoot_planes(1, 2000)
            first = (ph, w1, l1, nb, sc.camera_beams(1))
        r = bench.cpu_baseline_technique(tech, p, sc.medium(), sc.triangles(), first, W, H, 0.2)
        assert r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0
        assert r["build_s"] >= 0 and r["gather_s"] > 0 and r["one_thread"]["value"] > 0
        assert "accelerator" in r["sample"]


def test_rel_mse_equals_the_references_own_metric_code():
    """gvpm_amd/metrics.rel_mse against the REFERENCE's metric() itself (scripts/rgbe/sources/imageerrors.h compiled into
    oracle/_ref by oracle/Makefile.ref): the reference sums floats, per pixel and over the image -- equal to that rounding."""
    import numpy as np
    import pytest
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from gvpm_amd import metrics
    if not os.path.exists(O.REF_IMAGEERRORS):
        pytest.skip("oracle/_ref not built (needs the reference tree)")
    rng = np.random.default_rng(11)
    for (H, W, noise) in ((1, 1, 0.1), (7, 5, 0.05), (64, 48, 0.01), (32, 32, 0.0)):
        ref = rng.random((H, W, 3)) * 3
        img = ref + rng.normal(0, noise, ref.shape)
        a, b = metrics.rel_mse(img, ref), O.ref_image_metric(img, ref, "relmse")
        assert abs(a - b) <= 2e-5 * max(a, 1e-30) + 1e-12, (H, W, a, b)
    # and the plain MSE of the same header against the L2 figure of the parity bar: mse = 3 * mean((img - ref)^2)
    ref = rng.random((16, 12, 3))
    img = ref + rng.normal(0, 0.02, ref.shape)
    assert abs(O.ref_image_metric(img, ref, "mse") - 3 * ((img - ref) ** 2).mean()) < 1e-6
