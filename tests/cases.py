"""Shared builders for seeded test cases (synthetic hosts) and numpy reference helpers."""
import ctypes as C

import numpy as np

from gvpm_amd import abi
from gvpm_amd.host import SynthScene


class Case:
    pass


def radius_of(p, scale=None):
    """breInitSize = R * globalScaleVolume * POURCENTAGE_BS in fp32 (gvpm.cpp:989)."""
    s = p.initial_scale_volume if scale is None else scale
    return float(np.float32(np.float32(p.bsphere_radius) * np.float32(s)) * np.float32(0.01))


def make_case(scene="cbox", W=24, H=20, nph=6000, scale=4.0, it=1, **overrides):
    c = Case()
    c.sc = SynthScene(scene, W, H)
    c.p = c.sc.params()
    c.p.initial_scale_volume = scale
    for k, v in overrides.items():
        setattr(c.p, k, v)
    c.m = c.sc.medium()
    c.tris = c.sc.triangles()
    c.ph, c.nb = c.sc.shoot_photons(it, nph)
    c.rays = c.sc.camera_beams(it)
    c.r = radius_of(c.p)
    c.it = it
    use_bsdfs(c)
    return c


def use_bsdfs(c):
    """The BSDF table of the case's scene (its glossy walls; empty for the Lambertian scenes) for everything that evaluates
    a parent's BSDF on the CPU: the oracle (both builds) and the independent numpy statements.  The device gets it through
    Context.upload_bsdfs(c.bsdfs)."""
    import indep_statements
    import oracle_lib
    c.bsdfs = c.sc.bsdfs()
    oracle_lib.set_bsdfs(c.bsdfs)
    indep_statements.set_bsdfs(c.bsdfs)


def rays_shift_equals_base(rays):
    """Replace the four shifted rays of every set by a copy of the base ray."""
    out = rays.copy()
    for k in range(1, 5):
        out[:, k] = rays[:, 0]
        out[:, k]["rand"] = 0
        out[:, k]["pixel"] = 0
    return out


def pixels_of(rays):
    px = (rays["pixel"][:, 0] & 0xFFFF).astype(np.int64)
    py = (rays["pixel"][:, 0] >> 16).astype(np.int64)
    return px, py


def numpy_base_flux(c, dtype=np.float64):
    """Independent numpy statement of the BRE-3D base estimator (shift_volume_photon.cpp:658-751
    + gvpm_accel.h:279-301): O(B*N) loop over beams, vectorised over photons.  Returns
    (flux[H,W,3] before /nb_paths, evaluation count)."""
    p, ph, rays, r = c.p, c.ph, c.rays, dtype(c.r)
    H, W = p.height, p.width
    out = np.zeros((H, W, 3), dtype)
    pos = ph.pos.astype(dtype)
    flux = ph.flux.astype(dtype)
    wi = ph.wi.astype(dtype)
    depth = ((ph.flags >> 8) & 0xFF).astype(np.int64)
    parity = (ph.path_id & 1).astype(np.int64)
    sig_t = dtype(c.m.sigma_t[0])
    sig_s = np.array(list(c.m.sigma_s), dtype)
    g = dtype(c.m.g)
    eps = dtype(p.epsilon)
    kv = dtype(4.0 / 3.0 * np.pi) * r ** 3
    evals = 0
    for s in range(rays.shape[0]):
        b = rays[s, 0]
        o = b["o"].astype(dtype)
        d = b["d"].astype(dtype)
        ln = dtype(b["len"])
        rnd = dtype(b["rand"])
        edge = (int(b["info"]) >> 8) & 0xFF
        px, py = int(b["pixel"]) & 0xFFFF, int(b["pixel"]) >> 16
        mint, maxt = eps, ln - eps
        w = pos - o
        disk = w @ d
        v = (o + np.outer(disk, d)) - pos
        d2 = (v * v).sum(1)
        hit = (disk > mint) & (d2 < r * r)
        # own-box slab test (aabb.h:310-340)
        near = np.full(pos.shape[0], -np.inf, dtype)
        far = np.full(pos.shape[0], np.inf, dtype)
        ok = np.ones(pos.shape[0], bool)
        for i in range(3):
            lo, hi = pos[:, i] - r, pos[:, i] + r
            if d[i] == 0:
                ok &= (o[i] >= lo) & (o[i] <= hi)
            else:
                t1, t2 = (lo - o[i]) * (1 / d[i]), (hi - o[i]) * (1 / d[i])
                near = np.maximum(near, np.minimum(t1, t2))
                far = np.minimum(far, np.maximum(t1, t2))
        ok &= (near <= far) & ~((far < mint) | (near > maxt))
        hit &= ok
        if p.max_depth > 0:
            hit &= (depth + edge) <= p.max_depth
        rr = 1.0
        if p.path_set:
            hit &= parity == ((px + py) % 2)
            rr = 2.0
        dT = np.sqrt(np.maximum(0, r * r - d2))
        tp = (disk - dT) + (dT * 2) * rnd
        hit &= ~((tp < mint) | (tp > ln))
        idx = np.nonzero(hit)[0]
        evals += idx.size
        if idx.size == 0:
            continue
        pdf = 1.0 / np.maximum(2 * dT[idx], 1e-4)
        tr = np.exp(-sig_t * (tp[idx] - mint))
        if g == 0:
            phase = np.full(idx.size, 1 / (4 * np.pi), dtype)
        else:
            temp = 1 + g * g + 2 * g * (wi[idx] @ (-d))
            phase = (1 / (4 * np.pi)) * (1 - g * g) / (temp * np.sqrt(temp))
        contrib = (tr * phase / (kv * pdf) * rr)[:, None] * flux[idx] * sig_s * b["eye"].astype(dtype)
        out[py, px] += contrib.sum(0)
    return out, evals


def tessellate(tris, levels):
    """Split every triangle (v0, e1, e2 arrays) into 4^levels congruent ones: the same occluder
    surface with many more primitives (exercises the occluder BVH)."""
    v0, e1, e2 = (np.asarray(t, np.float64) for t in tris)
    a, b, c = v0, v0 + e1, v0 + e2
    for _ in range(levels):
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        a, b, c = (np.concatenate(x) for x in ((a, ab, ca, ab), (ab, b, bc, bc), (ca, bc, c, ca)))
    f = np.float32
    return np.ascontiguousarray(a, f), np.ascontiguousarray(b - a, f), np.ascontiguousarray(c - a, f)


def upload_bsdfs(ctx, c):
    """the device's copy of the case's BSDF table (gvpm_upload_bsdfs); nothing for the Lambertian scenes"""
    if getattr(c, "bsdfs", None) is not None and c.bsdfs.size:
        ctx.upload_bsdfs(c.bsdfs)
