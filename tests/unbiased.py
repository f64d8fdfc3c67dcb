"""An estimator-level test that owes nothing to the oracle's text: the property the whole method
rests on.  For a fixed kernel radius (alpha = 1) the gradient estimate of a pixel pair is an unbiased estimate of the
difference of the two pixels' (kernel-blurred) throughputs:

    E[dx(x, y)] = E[throughput(x + 1, y)] - E[throughput(x, y)],      E[dy(x, y)] likewise for (x, y + 1)

(computeGradient, gvpm.cpp:1205-1306: the forward shift of (x, y) and the reverse shift of its neighbour, whose MIS weights
sum to one over every pair of paths the shift mapping identifies; shift_volume_photon.cpp:463-484,843-854).  A wrong
Jacobian, a pair of weights that does not sum to one, a mis-signed assembly term, a shifted pdf taken on the wrong ray: each
biases dx against the finite difference, and nothing but this test would notice if the oracle shared the mistake.

run(step, n): `step(k)` returns (throughput, dx, dy) [H, W, 3] of ONE iteration with fresh photons and camera samples; the
statistics are those of D = dx - (thr(x+1) - thr(x)) -- the two sides are strongly correlated within an iteration (same
photons), so D's standard error is far below either side's."""
import numpy as np


class Stats:
    def __init__(self, shape):
        self.n = 0
        self.s1 = np.zeros(shape)
        self.s2 = np.zeros(shape)

    def add(self, v):
        self.n += 1
        self.s1 += v
        self.s2 += v * v

    def mean(self):
        return self.s1 / self.n

    def sem(self):
        var = np.maximum(self.s2 / self.n - self.mean() ** 2, 0.0) * self.n / max(self.n - 1, 1)
        return np.sqrt(var / self.n)


def run(step, n):
    thr, dx, dy = (np.asarray(a, np.float64) for a in step(0))
    H, W, _ = thr.shape
    D = {k: Stats((H, W, 3)) for k in ("dx", "dy")}
    M = {k: Stats((H, W, 3)) for k in ("thr", "dx", "dy")}
    for k in range(n):
        if k:
            thr, dx, dy = (np.asarray(a, np.float64) for a in step(k))
        fdx = np.zeros_like(thr)
        fdy = np.zeros_like(thr)
        fdx[:, :-1] = thr[:, 1:] - thr[:, :-1]
        fdy[:-1] = thr[1:] - thr[:-1]
        # (the last column / row has no neighbour in the frame: the border rule gives those forward shifts weight 1, i.e. dx
        # there estimates the difference to a pixel OUTSIDE the frame -- nothing to compare it with: left out, D = 0)
        fdx[:, -1] = dx[:, -1]
        fdy[-1] = dy[-1]
        D["dx"].add(dx - fdx)
        D["dy"].add(dy - fdy)
        M["thr"].add(thr)
        M["dx"].add(dx)
        M["dy"].add(dy)
    out = {}
    mt = M["thr"].mean()
    for key, fd in (("dx", None), ("dy", None)):
        d = D[key]
        mean, sem = d.mean(), d.sem()
        g = M[key].mean()
        ref = g - mean  # = mean finite difference
        z = np.abs(mean) / np.maximum(sem, 1e-30)
        lit = sem > 0
        out[key] = dict(rel_l2=float(np.sqrt((mean ** 2).sum() / max((ref ** 2).sum(), 1e-300))),
                        zmax=float(z[lit].max()) if lit.any() else 0.0,
                        n_over4=int((z[lit] > 4).sum()), n_tests=int(lit.sum()),
                        # one number for a sign or scale error: the regression slope of mean dx on the mean finite difference
                        slope=float((g * ref).sum() / max((ref * ref).sum(), 1e-300)),
                        grad_over_thr=float(np.sqrt((ref ** 2).mean()) / max(mt.mean(), 1e-300)),
                        # the noise floor of rel_l2: what the standard errors alone would give
                        noise_l2=float(np.sqrt((sem ** 2).sum() / max((ref ** 2).sum(), 1e-300))),
                        z=z, mean=mean, ref=ref)
    return out
