"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference
algorithm.  TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

from gvpm_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIBS = {}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib(fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name not in _LIBS:
        path = os.path.join(ORACLE_DIR, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_gather_bre.argtypes = [
            C.POINTER(abi.Params), C.POINTER(abi.Medium), C.POINTER(abi.Triangles), C.POINTER(abi.PhotonSoA),
            C.c_void_p, C.c_uint64, C.c_double, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_gather_bre_timed.argtypes = L.oracle_gather_bre.argtypes + [C.POINTER(C.c_double)]
        L.oracle_scale_volume_apa.restype = C.c_double
        L.oracle_scale_volume_apa.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int]
        L.oracle_assemble.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_gather_vpm.argtypes = [
            C.POINTER(abi.Params), C.POINTER(abi.Medium), C.POINTER(abi.Triangles), C.POINTER(abi.PhotonSoA),
            C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_gather_vpm_timed.argtypes = L.oracle_gather_vpm.argtypes + [C.POINTER(C.c_double)]
        L.oracle_assemble_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_gather_beams.argtypes = [
            C.POINTER(abi.Params), C.POINTER(abi.Medium), C.POINTER(abi.Triangles), C.POINTER(abi.PhotonSoA),
            C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_int, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_gather_planes.argtypes = [
            C.POINTER(abi.Params), C.POINTER(abi.Medium), C.POINTER(abi.Triangles), C.POINTER(abi.PhotonSoA),
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_poisson_solve.argtypes = [C.c_char_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        L.oracle_set_bsdfs.argtypes = [C.c_void_p, C.c_uint32]
        _LIBS[name] = L
        if _BSDFS is not None:
            L.oracle_set_bsdfs(_BSDFS.ctypes.data if _BSDFS.size else None, _BSDFS.size)
    return _LIBS[name]


_BSDFS = None


def set_bsdfs(table):
    """The BSDF table of the scene's glossy surfaces (abi.BSDF_DTYPE) for every gather that follows, in both builds of
    the oracle: the counterpart of gvpm_upload_bsdfs.  An empty table restores the Lambertian-only closed set."""
    global _BSDFS
    _BSDFS = np.ascontiguousarray(table, abi.BSDF_DTYPE)
    for L in _LIBS.values():
        L.oracle_set_bsdfs(_BSDFS.ctypes.data if _BSDFS.size else None, _BSDFS.size)


COUNTER_NAMES = ("evaluations", "candidates", "null_shifts", "diffuse_shifts", "failed_shifts")


def gather_bre(params, medium, tris, photons, rays, radius, it=1, nb_paths=1, precision=64, use_accel=True,
               threads=0, accum=None, fast=False, timing=None):
    """One iteration of computeVolumeGradientPhotonBRE on the CPU.

    tris: (v0,e1,e2) arrays; photons: abi.Photons; rays: (n_sets,5) CAMERA_RAY_DTYPE.
    Returns (accum[H,W,27] float64, counters dict, seconds)."""
    tstruct, keep = abi.triangles_struct(*tris)
    soa = photons.soa()
    rays = np.ascontiguousarray(rays)
    P = params.width * params.height
    if accum is None:
        accum = np.zeros(P * 27, np.float64)
    else:
        accum = np.ascontiguousarray(accum, np.float64).reshape(-1).copy()
    counters = np.zeros(5, np.uint64)
    secs, bsecs = C.c_double(0), C.c_double(0)
    rc = lib(fast).oracle_gather_bre_timed(C.byref(params), C.byref(medium), C.byref(tstruct), C.byref(soa),
                                           rays.ctypes.data, rays.shape[0], float(radius), it, nb_paths, precision,
                                           1 if use_accel else 0, threads, accum.ctypes.data, counters.ctypes.data,
                                           C.byref(secs), C.byref(bsecs))
    if rc != 0:
        raise RuntimeError(f"oracle_gather_bre failed: {rc}")
    if timing is not None:
        timing["build_s"] = bsecs.value  # the serial kd-tree + BVH build, part of `seconds`
    return (accum.reshape(params.height, params.width, 27), dict(zip(COUNTER_NAMES, map(int, counters))),
            secs.value)


def standin_host_shifts(photons, requests):
    """The stand-in of the host's manifold walk (oracle/gvpm_oracle.hpp standinManifoldWalk) on downloaded requests."""
    requests = np.ascontiguousarray(requests)
    out = np.zeros(requests.size, abi.HOST_SHIFT_DTYPE)
    soa = photons.soa()
    L = lib()
    L.oracle_standin_host_shifts.argtypes = [C.POINTER(abi.PhotonSoA), C.c_void_p, C.c_uint64, C.c_void_p]
    rc = L.oracle_standin_host_shifts(C.byref(soa), requests.ctypes.data, requests.size, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("oracle_standin_host_shifts: a request names a photon outside the map")
    return out


def mirror_host_shifts(photons, requests):
    """The planar-mirror walk (oracle/gvpm_oracle.hpp mirrorManifoldWalk: the image construction) on downloaded requests."""
    requests = np.ascontiguousarray(requests)
    out = np.zeros(requests.size, abi.HOST_SHIFT_DTYPE)
    soa = photons.soa()
    L = lib()
    L.oracle_mirror_host_shifts.argtypes = [C.POINTER(abi.PhotonSoA), C.c_void_p, C.c_uint64, C.c_void_p]
    rc = L.oracle_mirror_host_shifts(C.byref(soa), requests.ctypes.data, requests.size, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("oracle_mirror_host_shifts: a request names a photon outside the map")
    return out


def set_manifold_walk(kind):
    """Which stand-in the oracle's G-BRE gathers answer manifold shifts with: 0 the smooth closed form, 1 the planar mirror."""
    if lib().oracle_set_manifold_walk(int(kind)) != 0:
        raise ValueError(kind)


def gather_vpm(params, medium, tris, photons, rays, samples, precision=64, use_accel=True, threads=0, accum=None,
               scale_vol=None, n_vol=None, fast=False, timing=None):
    """One iteration of computeVolumeGradientPhoton (G-VPM) on the CPU.
    Returns (accum[H,W,27] sums, scale_vol[H,W], n_vol[H,W], counters, seconds)."""
    tstruct, keep = abi.triangles_struct(*tris)
    soa = photons.soa()
    rays = np.ascontiguousarray(rays)
    samples = np.ascontiguousarray(samples)
    H, W = params.height, params.width
    accum = np.zeros(H * W * 27, np.float64) if accum is None else np.ascontiguousarray(accum, np.float64).reshape(-1).copy()
    scale_vol = (np.full(H * W, params.initial_scale_volume, np.float64) if scale_vol is None
                 else np.ascontiguousarray(scale_vol, np.float64).reshape(-1).copy())
    n_vol = np.zeros(H * W, np.float64) if n_vol is None else np.ascontiguousarray(n_vol, np.float64).reshape(-1).copy()
    counters = np.zeros(5, np.uint64)
    secs, bsecs = C.c_double(0), C.c_double(0)
    rc = lib(fast).oracle_gather_vpm_timed(C.byref(params), C.byref(medium), C.byref(tstruct), C.byref(soa),
                                           rays.ctypes.data, rays.shape[0], samples.ctypes.data, samples.shape[0],
                                           precision, 1 if use_accel else 0, threads, accum.ctypes.data,
                                           scale_vol.ctypes.data, n_vol.ctypes.data, counters.ctypes.data, C.byref(secs),
                                           C.byref(bsecs))
    if rc != 0:
        raise RuntimeError(f"oracle_gather_vpm failed: {rc}")
    if timing is not None:
        timing["build_s"] = bsecs.value
    return (accum.reshape(H, W, 27), scale_vol.reshape(H, W), n_vol.reshape(H, W),
            dict(zip(COUNTER_NAMES, map(int, counters))), secs.value)


def gather_beams(params, medium, tris, beams, end_n, rays, radius, it=1, nb_paths=1, precision=64, sub_beam_size=0.0,
                 threads=0, accum=None, fast=False, use_accel=False, timing=None):
    """One iteration of computeVolumeGradientBeams on the CPU -> (accum[H,W,27], counters, seconds).
    use_accel: through the reference's SubBeamBVH instead of the ENoAccel loop; timing (dict): receives build_s."""
    tstruct, keep = abi.triangles_struct(*tris)
    soa = beams.soa()
    end_n = np.ascontiguousarray(end_n, np.float32)
    rays = np.ascontiguousarray(rays)
    P = params.width * params.height
    accum = np.zeros(P * 27, np.float64) if accum is None else np.ascontiguousarray(accum, np.float64).reshape(-1).copy()
    counters = np.zeros(5, np.uint64)
    secs = C.c_double(0)
    bsecs = C.c_double(0)
    rc = lib(fast).oracle_gather_beams(C.byref(params), C.byref(medium), C.byref(tstruct), C.byref(soa),
                                       end_n.ctypes.data, rays.ctypes.data, rays.shape[0], float(radius), it,
                                       nb_paths, precision, float(sub_beam_size), int(bool(use_accel)), threads,
                                       accum.ctypes.data, counters.ctypes.data, C.byref(secs), C.byref(bsecs))
    if rc != 0:
        raise RuntimeError(f"oracle_gather_beams failed: {rc}")
    if timing is not None:
        timing["build_s"] = bsecs.value
    return (accum.reshape(params.height, params.width, 27), dict(zip(COUNTER_NAMES, map(int, counters))),
            secs.value)


def gather_planes(params, medium, tris, beams, w1, len1, rays, it=1, nb_paths=1, precision=64, threads=0, accum=None,
                  fast=False, use_accel=False, timing=None):
    """One iteration of computeVolumeGradientPlanes on the CPU -> (accum[H,W,27], counters, seconds).
    use_accel: through the reference's PhotonPlaneBVH instead of the loop over all planes; timing (dict): build_s."""
    tstruct, keep = abi.triangles_struct(*tris)
    soa = beams.soa()
    w1 = np.ascontiguousarray(w1, np.float32)
    len1 = np.ascontiguousarray(len1, np.float32)
    rays = np.ascontiguousarray(rays)
    P = params.width * params.height
    accum = np.zeros(P * 27, np.float64) if accum is None else np.ascontiguousarray(accum, np.float64).reshape(-1).copy()
    counters = np.zeros(5, np.uint64)
    secs = C.c_double(0)
    bsecs = C.c_double(0)
    rc = lib(fast).oracle_gather_planes(C.byref(params), C.byref(medium), C.byref(tstruct), C.byref(soa),
                                        w1.ctypes.data, len1.ctypes.data, rays.ctypes.data, rays.shape[0], it,
                                        nb_paths, precision, int(bool(use_accel)), threads, accum.ctypes.data,
                                        counters.ctypes.data, C.byref(secs), C.byref(bsecs))
    if rc != 0:
        raise RuntimeError(f"oracle_gather_planes failed: {rc}")
    if timing is not None:
        timing["build_s"] = bsecs.value
    return (accum.reshape(params.height, params.width, 27), dict(zip(COUNTER_NAMES, map(int, counters))),
            secs.value)


def scale_volume_apa(scale, it, alpha, technique):
    return lib().oracle_scale_volume_apa(scale, it, alpha, technique)


def assemble(accum, it=1, reuse_primal=True, emission=None, total_emitted=0.0):
    H, W = accum.shape[:2]
    a = np.ascontiguousarray(accum, np.float64)
    out = [np.zeros((H, W, 3), np.float64) for _ in range(3)]
    em = None if emission is None else np.ascontiguousarray(emission, np.float64)
    rc = lib().oracle_assemble_ex(W, H, it, 1 if reuse_primal else 0, float(total_emitted), a.ctypes.data,
                               None if em is None else em.ctypes.data, out[0].ctypes.data, out[1].ctypes.data,
                               out[2].ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_assemble failed: {rc}")
    return tuple(out)


def poisson_solve(dx, dy, throughput, direct=None, preset="L1D", alpha=0.2):
    """The reference's screened-Poisson solver restated (naive backend, sequential float sums)."""
    dx, dy, tp = (np.ascontiguousarray(a, np.float32) for a in (dx, dy, throughput))
    di = None if direct is None else np.ascontiguousarray(direct, np.float32)
    H, W = dx.shape[:2]
    out = np.zeros((H, W, 3), np.float32)
    rc = lib().oracle_poisson_solve(preset.encode(), alpha, W, H, dx.ctypes.data, dy.ctypes.data, tp.ctypes.data,
                                    None if di is None else di.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_poisson_solve failed: {rc}")
    return out


REF_POISSON = os.path.join(ROOT, "oracle", "_ref", "libref_poisson.so")
REF_IMAGEERRORS = os.path.join(ROOT, "oracle", "_ref", "libref_imageerrors.so")
REF_METRICS = {"mse": 0, "rmse": 1, "mse_log": 2, "rmse_log": 3, "tvi": 4, "relative": 5, "relmse": 6}


def ref_image_metric(img, ref, which="relmse"):
    """The REFERENCE's metric() (scripts/rgbe/sources/imageerrors.h, built into oracle/_ref by oracle/Makefile.ref)."""
    L = C.CDLL(REF_IMAGEERRORS)
    L.ref_image_metric.restype = C.c_float
    L.ref_image_metric.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    a = np.ascontiguousarray(img, np.float64)
    b = np.ascontiguousarray(ref, np.float64)
    assert a.shape == b.shape and a.shape[-1] == 3
    return float(L.ref_image_metric(a.ctypes.data, b.ctypes.data, a.shape[1], a.shape[0], REF_METRICS[which]))
_REF = None


def ref_poisson_solve(dx, dy, throughput, direct=None, preset="L1D", alpha=0.2, backend="Naive"):
    """The REFERENCE's own solver (oracle/_ref, built from the reference sources by oracle/Makefile.ref)."""
    global _REF
    if _REF is None:
        _REF = C.CDLL(REF_POISSON)
        _REF.ref_poisson_solve.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 5
    dx, dy, tp = (np.array(a, np.float32, order="C") for a in (dx, dy, throughput))
    di = None if direct is None else np.array(direct, np.float32, order="C")
    H, W = dx.shape[:2]
    out = np.zeros((H, W, 3), np.float32)
    rc = _REF.ref_poisson_solve(preset.encode(), backend.encode(), alpha, W, H, dx.ctypes.data, dy.ctypes.data,
                                tp.ctypes.data, None if di is None else di.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"ref_poisson_solve failed: {rc}")
    return out


# ---- pins ported from the reference's adjacent tests + camera-path pieces (oracle_api.cpp) -------------------------
def kd_radius_query(pos, query, radius, precision=64):
    """Indices (into pos) of the points within `radius` of `query` by the oracle's kd-tree, and the nodes visited."""
    L = lib()
    L.oracle_kd_radius_query.restype = C.c_int64
    L.oracle_kd_radius_query.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64)]
    pos = np.ascontiguousarray(pos, np.float32)
    q = np.ascontiguousarray(query, np.float64)
    out = np.zeros(pos.shape[0], np.uint32)
    vis = C.c_uint64(0)
    n = L.oracle_kd_radius_query(pos.ctypes.data, pos.shape[0], q.ctypes.data, float(radius), precision, out.ctypes.data,
                                 out.size, C.byref(vis))
    assert n >= 0
    return np.sort(out[:n]), int(vis.value)


def phase_eval(g, wi, wo):
    L = lib()
    L.oracle_phase_eval.restype = C.c_double
    L.oracle_phase_eval.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
    a, b = np.ascontiguousarray(wi, np.float64), np.ascontiguousarray(wo, np.float64)
    return L.oracle_phase_eval(float(g), a.ctypes.data, b.ctypes.data)


def hg_sample(g, wi, u1, u2):
    L = lib()
    L.oracle_hg_sample.argtypes = [C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    a = np.ascontiguousarray(wi, np.float64)
    out = np.zeros(3, np.float64)
    L.oracle_hg_sample(float(g), a.ctypes.data, float(u1), float(u2), out.ctypes.data)
    return out


def half_vector_shift(main_wi, main_wo, shifted_wi, main_eta=1.0, shifted_eta=1.0):
    """halfVectorShift of shift_utilities.h:42-110 -> (success, wo[3], jacobian)"""
    L = lib()
    L.oracle_half_vector_shift.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    a, b, c = (np.ascontiguousarray(v, np.float64) for v in (main_wi, main_wo, shifted_wi))
    out = np.zeros(4, np.float64)
    ok = L.oracle_half_vector_shift(a.ctypes.data, b.ctypes.data, c.ctypes.data, float(main_eta), float(shifted_eta),
                                    out.ctypes.data)
    return bool(ok), out[:3].copy(), float(out[3])


def sensor_mis(id_vertex, s_pdf, s_jac, s_g, b_pdf, b_g, s_dist=1.0, b_dist=1.0):
    L = lib()
    L.oracle_sensor_mis.restype = C.c_double
    L.oracle_sensor_mis.argtypes = [C.c_uint] + [C.c_double] * 7
    return L.oracle_sensor_mis(int(id_vertex), s_pdf, s_jac, s_g, b_pdf, b_g, s_dist, b_dist)


def phong_eval_pdf(bsdf, kd, n, wi, wo):
    """Phong::eval (x cos) and Phong::pdf of one table entry (numpy record of abi.BSDF_DTYPE), world-space unit vectors"""
    L = lib()
    L.oracle_phong_eval_pdf.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    b = np.ascontiguousarray(np.atleast_1d(bsdf), abi.BSDF_DTYPE)
    a = [np.ascontiguousarray(x, np.float64) for x in (kd, n, wi, wo)]
    f, pdf = np.zeros(3), np.zeros(1)
    L.oracle_phong_eval_pdf(b.ctypes.data, *[x.ctypes.data for x in a], f.ctypes.data, pdf.ctypes.data)
    return f, float(pdf[0])


def phong_sample(bsdf, n, wi, u1, u2):
    """Phong::sample with both components (phong.cpp:188-247): the sampled world direction, or None (below the horizon)"""
    L = lib()
    L.oracle_phong_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    b = np.ascontiguousarray(np.atleast_1d(bsdf), abi.BSDF_DTYPE)
    n, wi = np.ascontiguousarray(n, np.float64), np.ascontiguousarray(wi, np.float64)
    wo = np.zeros(3)
    ok = L.oracle_phong_sample(b.ctypes.data, n.ctypes.data, wi.ctypes.data, float(u1), float(u2), wo.ctypes.data)
    return wo if ok else None


def bsdf_eval_pdf(bsdf, kd, n, wi, wo):
    """eval (x cos) and pdf of one table entry whatever its kind (glossyEvalPdf), world-space unit vectors"""
    L = lib()
    L.oracle_bsdf_eval_pdf.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    b = np.ascontiguousarray(np.atleast_1d(bsdf), abi.BSDF_DTYPE)
    a = [np.ascontiguousarray(x, np.float64) for x in (kd, n, wi, wo)]
    f, pdf = np.zeros(3), np.zeros(1)
    L.oracle_bsdf_eval_pdf(b.ctypes.data, *[x.ctypes.data for x in a], f.ctypes.data, pdf.ctypes.data)
    return f, float(pdf[0])


def ward_sample(bsdf, n, wi, u1, u2):
    """Ward::sample with both components (ward.cpp:268-327): the sampled world direction, or None (below the horizon)"""
    L = lib()
    L.oracle_ward_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    b = np.ascontiguousarray(np.atleast_1d(bsdf), abi.BSDF_DTYPE)
    n, wi = np.ascontiguousarray(n, np.float64), np.ascontiguousarray(wi, np.float64)
    wo = np.zeros(3)
    ok = L.oracle_ward_sample(b.ctypes.data, n.ctypes.data, wi.ctypes.data, float(u1), float(u2), wo.ctypes.data)
    return wo if ok else None


def roughconductor_sample(bsdf, n, wi, u1, u2):
    """RoughConductor::sample without visible-normal sampling: (wo, weight[3], pdf), or None"""
    L = lib()
    L.oracle_roughconductor_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
    b = np.ascontiguousarray(np.atleast_1d(bsdf), abi.BSDF_DTYPE)
    n, wi = np.ascontiguousarray(n, np.float64), np.ascontiguousarray(wi, np.float64)
    wo, wgt, pdf = np.zeros(3), np.zeros(3), np.zeros(1)
    ok = L.oracle_roughconductor_sample(b.ctypes.data, n.ctypes.data, wi.ctypes.data, float(u1), float(u2), wo.ctypes.data,
                                        wgt.ctypes.data, pdf.ctypes.data)
    return (wo, wgt, float(pdf[0])) if ok else None


def gather_primal_bre(params, medium, tris, photons, rays, radius, it=1, nb_paths=1, precision=64, use_accel=True, threads=0,
                      accum=None):
    """One iteration of the primal sppm integrator's volumePhotonPassBRE (oracle/gvpm_oracle_primal.hpp).
    Returns (accum[H,W,27] with fluxVol in [..., 0:3], counters dict)."""
    L = lib()
    L.oracle_gather_primal_bre.argtypes = [
        C.POINTER(abi.Params), C.POINTER(abi.Medium), C.POINTER(abi.Triangles), C.POINTER(abi.PhotonSoA),
        C.c_void_p, C.c_uint64, C.c_double, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    tstruct, keep = abi.triangles_struct(*tris)
    soa = photons.soa()
    rays = np.ascontiguousarray(rays)
    P = params.width * params.height
    accum = np.zeros(P * 27, np.float64) if accum is None else np.ascontiguousarray(accum, np.float64).reshape(-1).copy()
    counters = np.zeros(5, np.uint64)
    rc = L.oracle_gather_primal_bre(C.byref(params), C.byref(medium), C.byref(tstruct), C.byref(soa), rays.ctypes.data,
                                    rays.shape[0], float(radius), it, nb_paths, precision, 1 if use_accel else 0, threads,
                                    accum.ctypes.data, counters.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_gather_primal_bre failed: {rc}")
    return accum.reshape(params.height, params.width, 27), dict(zip(COUNTER_NAMES, map(int, counters)))


def gather_primal_vpm(params, medium, tris, photons, rays, samples, precision=64, use_accel=True, threads=0, accum=None,
                      scale_vol=None, n_vol=None):
    """One iteration of the sppm integrator's point estimate (EDistance; oracle/gvpm_oracle_primal.hpp).
    Returns (accum[H,W,27] with the fluxVol sums in [..., 0:3], scale_vol[H,W], n_vol[H,W], counters)."""
    L = lib()
    L.oracle_gather_primal_vpm.argtypes = [
        C.POINTER(abi.Params), C.POINTER(abi.Medium), C.POINTER(abi.Triangles), C.POINTER(abi.PhotonSoA),
        C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    tstruct, keep = abi.triangles_struct(*tris)
    soa = photons.soa()
    rays = np.ascontiguousarray(rays)
    samples = np.ascontiguousarray(samples)
    H, W = params.height, params.width
    accum = np.zeros(H * W * 27, np.float64) if accum is None else np.ascontiguousarray(accum, np.float64).reshape(-1).copy()
    scale_vol = (np.full(H * W, params.initial_scale_volume, np.float64) if scale_vol is None
                 else np.ascontiguousarray(scale_vol, np.float64).reshape(-1).copy())
    n_vol = np.zeros(H * W, np.float64) if n_vol is None else np.ascontiguousarray(n_vol, np.float64).reshape(-1).copy()
    counters = np.zeros(5, np.uint64)
    rc = L.oracle_gather_primal_vpm(C.byref(params), C.byref(medium), C.byref(tstruct), C.byref(soa), rays.ctypes.data,
                                    rays.shape[0], samples.ctypes.data, samples.shape[0], precision, 1 if use_accel else 0,
                                    threads, accum.ctypes.data, scale_vol.ctypes.data, n_vol.ctypes.data, counters.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_gather_primal_vpm failed: {rc}")
    return (accum.reshape(H, W, 27), scale_vol.reshape(H, W), n_vol.reshape(H, W), dict(zip(COUNTER_NAMES, map(int, counters))))


def gather_primal_beams(params, medium, tris, beams, end_n, rays, radius, it=1, nb_paths=1, precision=64, sub_beam_size=0.0,
                        threads=0, accum=None, use_accel=False):
    """One iteration of the sppm integrator's beam x beam pass (oracle/gvpm_oracle_primal.hpp) -> (accum[H,W,27] with
    fluxVol in [..., 0:3], counters)."""
    L = lib()
    L.oracle_gather_primal_beams.argtypes = [
        C.POINTER(abi.Params), C.POINTER(abi.Medium), C.POINTER(abi.Triangles), C.POINTER(abi.PhotonSoA),
        C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_int, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int,
        C.c_void_p, C.c_void_p]
    tstruct, keep = abi.triangles_struct(*tris)
    soa = beams.soa()
    end_n = np.ascontiguousarray(end_n, np.float32)
    rays = np.ascontiguousarray(rays)
    P = params.width * params.height
    accum = np.zeros(P * 27, np.float64) if accum is None else np.ascontiguousarray(accum, np.float64).reshape(-1).copy()
    counters = np.zeros(5, np.uint64)
    rc = L.oracle_gather_primal_beams(C.byref(params), C.byref(medium), C.byref(tstruct), C.byref(soa), end_n.ctypes.data,
                                      rays.ctypes.data, rays.shape[0], float(radius), it, nb_paths, precision,
                                      float(sub_beam_size), int(bool(use_accel)), threads, accum.ctypes.data, counters.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_gather_primal_beams failed: {rc}")
    return accum.reshape(params.height, params.width, 27), dict(zip(COUNTER_NAMES, map(int, counters)))
