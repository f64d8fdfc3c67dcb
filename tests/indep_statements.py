"""Independent numpy statements of the gather + shift of G-BRE 3D -- base term, null shift, diffuse reconnection and MIS
weight -- written from the reference sources alone (never from oracle/):
    VolumeGradientBREQuery::operator()     gvpm/shift/shift_volume_photon.cpp:658-856
    shiftNull / shiftPhoton / ...Diffuse   shift_volume_photon.cpp:49-158,382-486
    getShiftPos                            shift_volume_photon.cpp:858-896
    diffuseReconnection                    gvpm/shift/operation/shift_diffuse.cpp:11-134
    GatherPoint::sensorMIS                 gvpm/gvpm_struct.h:608-631
    HomogeneousMedium::eval, phase eval    src/medium/homogeneous.cpp:432-513, src/phase/{isotropic,hg}.cpp
    area emitter evalDirection / pdf       src/emitters/area.cpp:132-150
    triangle intersection                  include/mitsuba/core/triangle.h (Moeller-Trumbore, two-sided)
They give the oracle a second statement for the SHIFTED terms (round 1 had one for the base term only): the tests
compare the 27 accumulators and the shift counters."""
import numpy as np

from gvpm_amd import abi

INV_PI = 1.0 / np.pi
INV_4PI = 1.0 / (4 * np.pi)


def phase(g, wi, wo):
    """isotropic.cpp:76-78 / hg.cpp:107-110; wi, wo: (..., 3), both pointing away from the vertex"""
    if g == 0:
        return np.full(np.broadcast_shapes(wi.shape[:-1], wo.shape[:-1]), INV_4PI)
    temp = 1 + g * g + 2 * g * (wi * wo).sum(-1)
    return INV_4PI * (1 - g * g) / (temp * np.sqrt(temp))


def any_hit(tris, o, d, mint, maxt):
    """scene->rayIntersect(Ray(o, d, mint, maxt)) for rays o[k], d[k], maxt[k]: (n,) bool"""
    v0, e1, e2 = (np.asarray(t, np.float64) for t in tris)
    hit = np.zeros(o.shape[0], bool)
    for a, b, c in zip(v0, e1, e2):
        p = np.cross(d, c)
        det = p @ b
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o - a
            u = (tv * p).sum(1) * inv
            q = np.cross(tv, b)
            v = (d * q).sum(1) * inv
            t = (q @ c) * inv
        hit |= (det != 0) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t >= mint) & (t <= maxt)
    return hit


def bre3d_full(c):
    """One iteration (it = 1) of G-BRE 3D over every beam set of the case: (accum[H, W, 27], counters)."""
    p, ph, rays, m = c.p, c.ph, c.rays, c.m
    H, W = p.height, p.width
    acc = np.zeros((H, W, 27))
    f8 = np.float64
    pos, wi, flux = ph.pos.astype(f8), ph.wi.astype(f8), ph.flux.astype(f8)
    ppos, pn, prefix = ph.parent_pos.astype(f8), ph.parent_n.astype(f8), ph.prefix_w.astype(f8)
    pscat, pwi = ph.parent_scat.astype(f8), ph.parent_wi.astype(f8)
    ppdf, epdf, prr, pg = (a.astype(f8) for a in (ph.parent_pdf, ph.edge_pdf, ph.parent_rr, ph.parent_g))
    fl = ph.flags
    ptype, stype, emed, depth, comp = fl & 3, (fl >> 2) & 7, (fl >> 5) & 1, (fl >> 8) & 0xFF, (fl >> 16) & 0xFFFF
    parity = ph.path_id & 1
    r = f8(c.r)
    eps, seps = f8(p.epsilon), f8(p.shadow_epsilon)
    sig_t, sig_s, g, msw = f8(m.sigma_t[0]), np.array(list(m.sigma_s), f8), f8(m.g), f8(m.medium_sampling_weight)
    kv = 4.0 / 3.0 * np.pi * r ** 3
    cnt = dict(evaluations=0, null_shifts=0, diffuse_shifts=0, failed_shifts=0)
    # computeVolumeContribution, shift_utilities.h:233-253
    mode = p.lighting_interaction_mode
    contributes = np.ones(ph.n, bool)
    if not ((mode & abi.GVPM_SURF2MEDIA) and (mode & abi.GVPM_MEDIA2MEDIA)):
        contributes &= np.where(ptype == abi.GVPM_PARENT_MEDIUM, bool(mode & abi.GVPM_MEDIA2MEDIA), bool(mode & abi.GVPM_SURF2MEDIA))
    if p.bsdf_interaction_mode != abi.GVPM_BSDF_ALL:
        contributes &= ~((comp > 0) & ((comp & p.bsdf_interaction_mode) == 0))
    if p.debug_shift not in (abi.GVPM_SHIFT_ALL, abi.GVPM_SHIFT_NULL):
        code = {1: abi.GVPM_SHIFT_DIFFUSE, 2: abi.GVPM_SHIFT_MEDIUM, 3: abi.GVPM_SHIFT_MANIFOLD, 0: abi.GVPM_SHIFT_INVALID}
        contributes &= np.array([code.get(int(s), abi.GVPM_SHIFT_INVALID) for s in stype]) == p.debug_shift

    for s in range(rays.shape[0]):
        b = rays[s, 0]
        o, d, ln, rnd = b["o"].astype(f8), b["d"].astype(f8), f8(b["len"]), f8(b["rand"])
        edge = (int(b["info"]) >> 8) & 0xFF
        px, py = int(b["pixel"]) & 0xFFFF, int(b["pixel"]) >> 16
        mint, maxt = eps, ln - eps
        # GradientBeamRadianceEstimator::query, gvpm_accel.h:279-301 (own sphere box + disk test)
        wv = pos - o
        disk = wv @ d
        perp = (o + np.outer(disk, d)) - pos
        d2 = (perp * perp).sum(1)
        hit = (disk > mint) & (d2 < r * r) & contributes
        near = np.full(ph.n, -np.inf)
        far = np.full(ph.n, np.inf)
        for k in range(3):
            lo, hi = pos[:, k] - r, pos[:, k] + r
            if d[k] == 0:
                hit &= (o[k] >= lo) & (o[k] <= hi)
            else:
                t1, t2 = (lo - o[k]) * (1 / d[k]), (hi - o[k]) * (1 / d[k])
                near = np.maximum(near, np.minimum(t1, t2))
                far = np.minimum(far, np.maximum(t1, t2))
        hit &= (near <= far) & ~((far < mint) | (near > maxt))
        if p.max_depth > 0:
            hit &= (depth.astype(np.int64) + edge) <= p.max_depth
        if p.min_depth != 0:
            hit &= (depth.astype(np.int64) + edge) >= p.min_depth
        rr = 1.0
        if p.path_set:
            hit &= parity == ((px + py) % 2)
            rr = 2.0
        dT = np.sqrt(np.maximum(0, r * r - d2))
        tp = (disk - dT) + (dT * 2) * rnd
        hit &= ~((tp < mint) | (tp > ln))
        idx = np.nonzero(hit)[0]
        if idx.size == 0:
            continue
        cnt["evaluations"] += idx.size
        tp, dT = tp[idx], dT[idx]
        P, WI, FL = pos[idx], wi[idx], flux[idx]
        pdf_cam = 1.0 / np.maximum(2 * dT, 1e-4)
        tr = np.exp(-sig_t * (tp - mint))
        tr = np.where(tr < 1e-20, 0.0, tr)
        base_pt = o + np.outer(tp, d)
        base_c = (tr * phase(g, WI, -d))[:, None] * FL * sig_s * b["eye"].astype(f8)
        norm = (rr / (kv * pdf_cam))[:, None]
        acc[py, px, 0:3] += (base_c * norm).sum(0)
        for i in range(4):
            sh = rays[s, 1 + i]
            w = np.ones(idx.size)
            sflux = np.zeros((idx.size, 3))
            if int(sh["info"]) & 1:
                so, sd, sl = sh["o"].astype(f8), sh["d"].astype(f8), f8(sh["len"])
                seye = sh["eye"].astype(f8)
                ratio, jac = f8(sh["pdf"]) / f8(b["pdf"]), f8(sh["jacobian"])
                if edge != 1:
                    jac *= f8(sh["gop"]) / f8(b["gop"])
                    ratio *= f8(b["gop"]) / f8(sh["gop"])
                sensor = ratio * jac
                sh_pt = so + np.outer(tp, sd)
                zp = ((sh_pt - P) ** 2).sum(1)
                is_null = np.zeros(idx.size, bool)
                if p.use_shift_null:
                    is_null = (zp < r * r) & (tp < sl)
                # ---- shiftNull ----
                if is_null.any():
                    k = np.nonzero(is_null)[0]
                    dk = (P[k] - so) @ sd
                    ds = (((so + np.outer(dk, sd)) - P[k]) ** 2).sum(1)
                    pdf_s = 1.0 / np.maximum(2.0 * np.sqrt(np.maximum(0, r * r - ds)), 1e-4)
                    sflux[k] = (tr[k] * phase(g, WI[k], -sd))[:, None] * FL[k] * sig_s * seye
                    w[k] = 0.5
                    if p.use_mis:
                        w[k] = 1.0 / (1.0 + sensor * pdf_s / pdf_cam[k])
                    cnt["null_shifts"] += k.size
                # ---- reconnection ----
                rec = ~is_null & (sl >= tp)
                if p.debug_shift == abi.GVPM_SHIFT_NULL:
                    rec[:] = False
                k = np.nonzero(rec)[0]
                if k.size:
                    off = sh_pt[k] + (P[k] - base_pt[k])
                    if p.use_shift_null:
                        inside = ((base_pt[k] - off) ** 2).sum(1) < r * r
                        dsh = sh_pt[k] - base_pt[k]
                        dsh = dsh / np.linalg.norm(dsh, axis=1)[:, None]
                        cosd = (dsh * -(off - sh_pt[k])).sum(1)
                        off = np.where(inside[:, None], off + dsh * (cosd * 2)[:, None], off)
                    dk = (off - so) @ sd
                    ds = (((so + np.outer(dk, sd)) - off) ** 2).sum(1)
                    pdf_s = 1.0 / np.maximum(2.0 * np.sqrt(np.maximum(0, r * r - ds)), 1e-4)
                    gi = idx[k]
                    st = stype[gi]
                    can = (st == 1) | (st == 2)   # EDiffuseShift, EMediumShift with noMediumShift; manifold: host-only -> fails
                    dproj = off - ppos[gi]
                    lproj = np.linalg.norm(dproj, axis=1)
                    dproj = dproj / lproj[:, None]
                    vmax = lproj * seps if p.visibility_as_written else lproj * (1 - seps)
                    ok = can & ~any_hit(c.tris, ppos[gi], dproj, eps, vmax)
                    is_med, is_surf = ptype[gi] == abi.GVPM_PARENT_MEDIUM, ptype[gi] == abi.GVPM_PARENT_SURFACE
                    n = pn[gi]
                    cos_wo = (n * dproj).sum(1)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        sign = cos_wo / (n * -WI[k]).sum(1)          # edge(c-1).d points parent -> photon = -wi
                    ok &= is_med | ~(sign < 0)
                    cos_wi = (n * pwi[gi]).sum(1)
                    lam = INV_PI * cos_wo
                    surf_ok = (cos_wi > 0) & (cos_wo > 0)
                    pmed = phase_vec(pg[gi], pwi[gi], dproj)
                    emit = INV_PI * np.maximum(cos_wo, 0)
                    pdf_val = np.where(is_med, pmed, np.where(is_surf, np.where(surf_ok, lam, 0.0), emit))
                    thr = np.where(is_med[:, None], pscat[gi] * pmed[:, None],
                                   np.where(is_surf[:, None], pscat[gi] * np.where(surf_ok, lam, 0.0)[:, None], emit[:, None] * np.ones(3)))
                    ok &= ~(is_surf & ~surf_ok)                      # the shading-normal test returns before the pdf is set
                    gop = 1.0 / (lproj * lproj)
                    spdf = pdf_val * gop
                    thr = thr * gop[:, None]
                    ok &= ppdf[gi] != 0
                    with np.errstate(divide="ignore", invalid="ignore"):
                        thr = thr / ppdf[gi][:, None] * prr[gi][:, None]
                        in_med = emed[gi] == 1
                        trl = np.exp(-sig_t * lproj)
                        trl = np.where(trl < 1e-20, 0.0, trl)
                        spdf = np.where(in_med, spdf * (sig_t * np.exp(-sig_t * lproj) * msw), spdf)
                        thr = np.where(in_med[:, None], thr * (trl / epdf[gi])[:, None], thr)
                    ok &= spdf != 0
                    contrib = sig_s * (prefix[gi] * thr) * phase(g, -dproj, -sd)[:, None]
                    sf = tr[k][:, None] * contrib * seye
                    wk = np.full(k.size, 0.5)
                    mis_ok = np.ones(k.size, bool)
                    if p.use_mis:
                        base_pdf = pdf_cam[k] * ppdf[gi] * epdf[gi]
                        off_pdf = spdf * pdf_s
                        mis_ok = ~((off_pdf == 0) | (base_pdf == 0))
                        with np.errstate(divide="ignore", invalid="ignore"):
                            x = sensor * (off_pdf / base_pdf)
                        wk = 1.0 / (1.0 + (x * x if p.power_heuristic else x))
                    good = ok & mis_ok
                    sflux[k] = np.where(ok[:, None], sf, 0.0)         # a failed MIS keeps its flux and takes weight 1
                    w[k] = np.where(good, wk, 1.0)
                    cnt["diffuse_shifts"] += int(good.sum())
                    cnt["failed_shifts"] += int((~good).sum())
            if (i == abi.GVPM_RIGHT and px == W - 1) or (i == abi.GVPM_TOP and py == H - 1):
                w[:] = 1.0
            acc[py, px, 15 + 3 * i:18 + 3 * i] += (base_c * (w[:, None] * norm)).sum(0)
            acc[py, px, 3 + 3 * i:6 + 3 * i] += (np.nan_to_num(sflux) * (w[:, None] * norm)).sum(0)
    return acc / c.nb, cnt


def phase_vec(g, wi, wo):
    """phase() with a per-element g (the parent's medium)"""
    temp = 1 + g * g + 2 * g * (wi * wo).sum(-1)
    return np.where(g == 0, INV_4PI, INV_4PI * (1 - g * g) / (temp * np.sqrt(temp)))
