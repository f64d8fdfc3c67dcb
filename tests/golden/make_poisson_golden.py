"""Generates tests/golden/poisson_*.npz with the REFERENCE's own solver (oracle/_ref/libref_poisson.so, built by
`make -C oracle -f Makefile.ref` from /root/reference/src/integrators/poisson_solver, naive backend): seeded
inputs and the outputs poisson::Solver returns for them.  Run in the build container (needs /root/reference)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402


def inputs(W, H, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([0.5 + 0.4 * np.sin(0.3 * xx + 0.2 * yy + c) + 0.2 * (xx > W // 2) for c in range(3)], -1)
    img = img.astype(np.float32)
    dx = np.zeros_like(img); dx[:, :-1] = img[:, 1:] - img[:, :-1]
    dy = np.zeros_like(img); dy[:-1] = img[1:] - img[:-1]
    dx += (0.02 * rng.standard_normal(img.shape)).astype(np.float32)
    dy += (0.02 * rng.standard_normal(img.shape)).astype(np.float32)
    # a few outliers: what the L1 solve is for
    k = rng.integers(0, W * H, 6)
    dx.reshape(-1, 3)[k] += 3.0
    tp = (img + 0.25 * rng.standard_normal(img.shape)).astype(np.float32)
    direct = (0.1 * rng.random(img.shape)).astype(np.float32)
    return dx, dy, tp, direct


if __name__ == "__main__":
    for name, W, H, seed, preset, alpha, use_direct in (("poisson_L2D", 37, 29, 11, "L2D", 0.2, False),
                                                        ("poisson_L1D", 37, 29, 12, "L1D", 0.2, True),
                                                        ("poisson_L1D_wide", 64, 9, 13, "L1D", 0.5, False)):
        dx, dy, tp, di = inputs(W, H, seed)
        out = O.ref_poisson_solve(dx, dy, tp, di if use_direct else None, preset, alpha, "Naive")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), dx=dx, dy=dy, throughput=tp,
                            direct=di if use_direct else np.zeros(0, np.float32), out=out,
                            preset=preset, alpha=np.float32(alpha))
        print(name, out.shape, float(np.abs(out).mean()))
