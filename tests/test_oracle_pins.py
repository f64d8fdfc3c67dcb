"""The reference's own tests next to the hot path, applied to the oracle (SURVEY 8c "adjacent pins"): the oracle is
"parity unpinned" (no vector of the reference exercises the gather), but its kd-tree and its phase function are the
pieces the reference DOES test, and these are those tests."""
import numpy as np
import pytest
from scipy import stats

import oracle_lib as O


@pytest.mark.parametrize("precision", [32, 64])
def test_kdtree_radius_query_equals_brute_force(precision):
    """src/tests/test_kd.cpp:133-200 (test03_pointKDTree): 50 000 random points, the sliding-midpoint heuristic (the
    one gvpm builds, gvpm_accel.h:96), 20 random queries per neighbourhood size, results compared with a brute-force
    search.  The reference queries k nearest neighbours; the gather path uses the radius query (kdtree.h:675-731), so
    the radius is set to hold 1 ... 10 and then ~100 neighbours."""
    rng = np.random.default_rng(1234)
    n = 50000
    pos = rng.random((n, 3)).astype(np.float32)
    visited = []
    for k in list(range(1, 11)) + [100, 1000]:
        for _ in range(20):
            q = rng.random(3)
            d2 = ((pos.astype(np.float64) - q) ** 2).sum(1)
            r = float(np.sqrt(np.partition(d2, k)[k] * 0.999999))  # just inside the (k+1)-th neighbour
            got, vis = O.kd_radius_query(pos, q, r, precision)
            if precision == 64:
                want = np.nonzero(d2 < r * r)[0]
            else:
                qf = q.astype(np.float32)
                want = np.nonzero(((pos - qf) ** 2).sum(1, dtype=np.float32) < np.float32(r) * np.float32(r))[0]
            assert np.array_equal(got, np.sort(want).astype(np.uint32))
            visited.append(vis)
    assert max(visited) < n // 4  # a tree walk, not a scan


def test_kdtree_degenerate_inputs():
    pos = np.zeros((100, 3), np.float32)             # all points equal: the split cannot separate them
    got, _ = O.kd_radius_query(pos, [0, 0, 0], 0.1)
    assert got.size == 100
    got, _ = O.kd_radius_query(pos, [1, 0, 0], 0.1)
    assert got.size == 0
    one = np.array([[0.25, 0.5, 0.75]], np.float32)
    assert O.kd_radius_query(one, [0.25, 0.5, 0.75], 1e-3)[0].tolist() == [0]
    line = np.stack([np.linspace(0, 1, 1000), np.zeros(1000), np.zeros(1000)], 1).astype(np.float32)
    got, _ = O.kd_radius_query(line, [0.5, 0, 0], 0.0105)
    want = np.nonzero(np.abs(line[:, 0].astype(np.float64) - 0.5) < 0.0105)[0]
    assert np.array_equal(got, want) and got.size == 20


@pytest.mark.parametrize("g", [0.9, -0.3, 0.0, 0.7])
def test_hg_sampling_matches_its_pdf_chi_square(g):
    """src/tests/test_chisquare.cpp:508-573 (test02_PhaseFunction) for data/tests/test_phase.xml's Henyey-Greenstein
    instances (g = 0.9, -0.3), the isotropic one and the g = 0.7 the S-laser variants use: for 20 incident directions,
    the histogram of HGPhaseFunction::sample (hg.cpp:74-97) over 10 x 20 (theta, phi) bins against the integral of eval()
    (hg.cpp:107-110, the pdf) over the bins; bins with an expected frequency below 5 are pooled (libcore/chisquare.cpp),
    significance level 0.01 with the Sidak correction for the 20 tests."""
    rng = np.random.default_rng(7)
    theta_bins, phi_bins, n_wi, n_samples = 10, 20, 20, 200000
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / n_wi)
    for _ in range(n_wi):
        z = 1 - 2 * rng.random()
        ph = 2 * np.pi * rng.random()
        wi = np.array([np.sqrt(1 - z * z) * np.cos(ph), np.sqrt(1 - z * z) * np.sin(ph), z])
        u = rng.random((n_samples, 2))
        wo = np.array([O.hg_sample(g, wi, a, b) for a, b in u[:2000]])  # the literal routine on a subset ...
        # ... and its vectorised restatement for the bulk (checked against the subset)
        if abs(g) < 1e-4:
            ct = 1 - 2 * u[:, 0]
        else:
            sq = (1 - g * g) / (1 - g + 2 * g * u[:, 0])
            ct = (1 + g * g - sq * sq) / (2 * g)
        st_ = np.sqrt(np.maximum(0, 1 - ct * ct))
        n = -wi
        sign = np.copysign(1.0, n[2])
        a_ = -1.0 / (sign + n[2])
        b_ = n[0] * n[1] * a_
        s = np.array([1 + sign * n[0] * n[0] * a_, sign * b_, -sign * n[0]])
        t = np.array([b_, sign + n[1] * n[1] * a_, -n[1]])
        all_wo = (st_ * np.cos(2 * np.pi * u[:, 1]))[:, None] * s + (st_ * np.sin(2 * np.pi * u[:, 1]))[:, None] * t + ct[:, None] * n
        assert np.allclose(all_wo[:2000], wo, atol=1e-6)  # (coordinateSystemCoherent keeps float intermediates, util.cpp:592-599)
        assert np.allclose((all_wo ** 2).sum(1), 1, atol=1e-9)
        theta = np.arccos(np.clip(all_wo[:, 2], -1, 1))
        phi = np.arctan2(all_wo[:, 1], all_wo[:, 0]) % (2 * np.pi)
        obs, _, _ = np.histogram2d(theta, phi, bins=[theta_bins, phi_bins], range=[[0, np.pi], [0, 2 * np.pi]])
        # expected: the pdf integrated over every bin (midpoint rule on a 16 x 16 sub-grid, sin(theta) measure)
        sub = 16
        th = (np.arange(theta_bins * sub) + 0.5) * (np.pi / (theta_bins * sub))
        phs = (np.arange(phi_bins * sub) + 0.5) * (2 * np.pi / (phi_bins * sub))
        T, Pm = np.meshgrid(th, phs, indexing="ij")
        dirs = np.stack([np.sin(T) * np.cos(Pm), np.sin(T) * np.sin(Pm), np.cos(T)], -1)
        temp = 1 + g * g + 2 * g * (dirs @ wi)
        pdf = (1 / (4 * np.pi)) * (1 - g * g) / (temp * np.sqrt(temp))
        # spot-check the vectorised pdf against the oracle's phase eval
        for (i, j) in ((3, 5), (100, 200), (150, 17)):
            assert abs(pdf[i, j] - O.phase_eval(g, wi, dirs[i, j])) < 1e-12
        cell = (np.pi / (theta_bins * sub)) * (2 * np.pi / (phi_bins * sub))
        exp = (pdf * np.sin(T) * cell).reshape(theta_bins, sub, phi_bins, sub).sum((1, 3)) * n_samples
        assert abs(exp.sum() / n_samples - 1) < 1e-3
        o, e = obs.ravel(), exp.ravel()
        order = np.argsort(e)
        o, e = o[order], e[order]
        # pool the low-frequency cells
        cut = int(np.searchsorted(e, 5.0))
        if cut > 0:
            o = np.concatenate([[o[:cut].sum()], o[cut:]])
            e = np.concatenate([[e[:cut].sum()], e[cut:]])
        if e[0] < 5.0 and e.size > 1:
            o = np.concatenate([[o[0] + o[1]], o[2:]])
            e = np.concatenate([[e[0] + e[1]], e[2:]])
        e = e * (o.sum() / e.sum())
        chi2 = ((o - e) ** 2 / e).sum()
        pval = 1 - stats.chi2.cdf(chi2, df=e.size - 1)
        assert pval > alpha, (g, wi, chi2, pval)
