"""Screened-Poisson reconstruction timing (next row f1; not the bench.py metric): HIP solver vs the reference's
own OpenMP backend (oracle/_ref) on the host cores, 512x512 and 1024x1024, presets L2D and L1D."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from gvpm_amd import hip
from gvpm_amd.host import SynthScene
import oracle_lib as O

torch.cuda.init()
sc = SynthScene("cbox", 8, 8)
ctx = hip.Context(sc.params(), 0)
rng = np.random.default_rng(1)
for W in (512, 1024):
    img = rng.random((W, W, 3)).astype(np.float32)
    dx = np.zeros_like(img); dx[:, :-1] = img[:, 1:] - img[:, :-1]
    dy = np.zeros_like(img); dy[:-1] = img[1:] - img[:-1]
    tp = (img + 0.2 * rng.standard_normal(img.shape)).astype(np.float32)
    n3 = W * W * 3
    d = [torch.from_numpy(a.reshape(-1)).cuda() for a in (dx, dy, tp)]
    out = torch.zeros(n3, dtype=torch.float32, device="cuda")
    for preset in ("L2D", "L1D"):
        p = hip.poisson_preset(preset)
        import ctypes as C
        def run():
            ctx._check(hip.lib().gvpm_poisson_solve_dev(ctx._h, C.byref(p), W, W, d[0].data_ptr(), d[1].data_ptr(),
                                                        d[2].data_ptr(), None, out.data_ptr()))
            ctx.synchronize()
        run()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        gpu_ms = (time.perf_counter() - t0) / 5 * 1e3
        res = {"size": W, "preset": preset, "gpu_ms": round(gpu_ms, 3)}
        if os.path.exists(O.REF_POISSON):
            t0 = time.perf_counter()
            ref = O.ref_poisson_solve(dx, dy, tp, None, preset, 0.2, "OpenMP")
            res["reference_openmp_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            res["cores"] = os.cpu_count()
            got = out.cpu().numpy().reshape(W, W, 3)
            res["max_rel_diff_vs_reference_openmp"] = float(np.abs(got - ref).max() / np.abs(ref).max())
        print(json.dumps(res), flush=True)
ctx.close()
