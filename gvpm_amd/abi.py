"""ctypes mirror of include/gvpm_hip.h (POD structs and enums only).

Kept free of any library loading so that tests/, bench.py and the oracle
harness can share the struct definitions.
"""
import ctypes as C

import numpy as np

GVPM_ABI_VERSION = 3

# gvpm_status
GVPM_OK = 0
GVPM_ERR_INVALID_ARG = -1
GVPM_ERR_NO_DEVICE = -2
GVPM_ERR_HIP = -3
GVPM_ERR_STATE = -4
GVPM_ERR_UNSUPPORTED = -5
GVPM_ERR_COMM = -6

# gvpm_technique (EVolumeTechnique, src/integrators/volume_utils.h:12-21)
GVPM_VOL_BRE2D = 0
GVPM_VOL_BRE3D = 1
GVPM_DISTANCE = 2
GVPM_BEAM_BEAM_1D = 3
GVPM_BEAM_BEAM_3D_NAIVE = 4
GVPM_BEAM_BEAM_3D_EGSR = 5
GVPM_BEAM_BEAM_3D_OPTIMIZED = 6
GVPM_VOL_PLANE0D = 7

GVPM_LEFT, GVPM_RIGHT, GVPM_TOP, GVPM_BOTTOM = 0, 1, 2, 3

GVPM_SHIFT_ALL = 0
GVPM_SHIFT_DIFFUSE = 1 << 1
GVPM_SHIFT_MANIFOLD = 1 << 2
GVPM_SHIFT_NULL = 1 << 4
GVPM_SHIFT_MEDIUM = 1 << 5
GVPM_SHIFT_INVALID = 1 << 6

GVPM_SURF2SURF = 1 << 1
GVPM_SURF2MEDIA = 1 << 2
GVPM_MEDIA2SURF = 1 << 3
GVPM_MEDIA2MEDIA = 1 << 4

GVPM_BSDF_DIFFUSE_REFLECTION = 0x00002
GVPM_BSDF_ALL = 0x1FFFF

GVPM_PARENT_EMITTER, GVPM_PARENT_SURFACE, GVPM_PARENT_MEDIUM, GVPM_PARENT_SURFACE_BSDF = 0, 1, 2, 3
GVPM_BSDF_PHONG = 1
GVPM_ACCUM_FLOATS = 27


def pf_make(parent, shift, edge_medium, depth, comp):
    return ((parent & 3) | ((shift & 7) << 2) | ((edge_medium & 1) << 5)
            | ((depth & 0xFF) << 8) | ((comp & 0xFFFF) << 16))


def ray_info(valid, edge):
    return (valid & 1) | ((edge & 0xFF) << 8)


class Params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("vol_technique", C.c_int32), ("max_depth", C.c_int32), ("min_depth", C.c_int32),
        ("use_mis", C.c_int32), ("use_shift_null", C.c_int32), ("path_set", C.c_int32),
        ("power_heuristic", C.c_int32), ("no_medium_shift", C.c_int32),
        ("use_manifold", C.c_int32), ("debug_shift", C.c_int32),
        ("lighting_interaction_mode", C.c_int32), ("bsdf_interaction_mode", C.c_int32),
        ("nb_camera_samples", C.c_int32), ("visibility_as_written", C.c_int32),
        ("alpha", C.c_float), ("initial_scale_volume", C.c_float),
        ("bsphere_radius", C.c_float), ("epsilon", C.c_float), ("shadow_epsilon", C.c_float),
        ("reserved", C.c_int32 * 8),
    ]

    def copy(self):
        p = Params()
        C.memmove(C.byref(p), C.byref(self), C.sizeof(Params))
        return p


class Medium(C.Structure):
    _fields_ = [
        ("sigma_a", C.c_float * 3), ("sigma_s", C.c_float * 3), ("sigma_t", C.c_float * 3),
        ("g", C.c_float), ("medium_sampling_weight", C.c_float), ("reserved", C.c_float * 5),
    ]


class Triangles(C.Structure):
    _fields_ = [("v0", C.c_void_p), ("e1", C.c_void_p), ("e2", C.c_void_p), ("n", C.c_uint32)]


PHOTON_VEC3 = ["pos", "wi", "flux", "parent_pos", "parent_n", "prefix_w", "parent_scat", "parent_wi"]
PHOTON_F1 = ["parent_pdf", "edge_pdf", "parent_rr", "parent_g"]
PHOTON_U1 = ["flags", "path_id"]


class PhotonSoA(C.Structure):
    _fields_ = ([(k, C.c_void_p) for k in PHOTON_VEC3 + PHOTON_F1 + PHOTON_U1] + [("n", C.c_uint64)])


class Stats(C.Structure):
    _fields_ = [
        ("evaluations", C.c_uint64), ("candidates", C.c_uint64), ("null_shifts", C.c_uint64),
        ("diffuse_shifts", C.c_uint64), ("failed_shifts", C.c_uint64), ("dropped_pairs", C.c_uint64),
        ("reserved", C.c_uint64 * 2),
    ]


# gvpm_camera_ray, 64 bytes
CAMERA_RAY_DTYPE = np.dtype([
    ("o", np.float32, 3), ("len", np.float32), ("d", np.float32, 3), ("pdf", np.float32),
    ("eye", np.float32, 3), ("jacobian", np.float32), ("gop", np.float32), ("info", np.uint32),
    ("rand", np.float32), ("pixel", np.uint32),
])
assert CAMERA_RAY_DTYPE.itemsize == 64


# gvpm_vpm_sample, 16 bytes
VPM_SAMPLE_DTYPE = np.dtype([("set", np.uint32), ("rand", np.float32), ("pdf_sel", np.float32),
                             ("reserved", np.uint32)])
assert VPM_SAMPLE_DTYPE.itemsize == 16


# packed upload records (include/gvpm_hip.h, "packed uploads")
MATERIAL_DTYPE = np.dtype([("scat", np.float32, 3), ("g", np.float32)])
PHOTON_PACKED_DTYPE = np.dtype([
    ("pos", np.float32, 3), ("parent_pdf", np.float32), ("parent_pos", np.float32, 3), ("edge_pdf", np.float32),
    ("flux", np.float32, 3), ("parent_rr", np.float32), ("prefix_w", np.float32, 3), ("parent_n_oct", np.uint32),
    ("parent_wi_oct", np.uint32), ("flags", np.uint32), ("material", np.uint32)])
SHIFT_REQUEST_DTYPE = np.dtype([
    ("photon", np.uint32), ("set", np.uint32), ("shift", np.uint32), ("reserved", np.uint32), ("offset_pos", np.float32, 3),
    ("radius", np.float32), ("base_point", np.float32, 3), ("t", np.float32), ("shift_point", np.float32, 3),
    ("reserved2", np.float32)])
HOST_SHIFT_DTYPE = np.dtype([
    ("ok", np.uint32), ("throughput", np.float32, 3), ("wi", np.float32, 3), ("pdf", np.float32), ("det_ratio", np.float32),
    ("base_pdf", np.float32)])
assert SHIFT_REQUEST_DTYPE.itemsize == 64 and HOST_SHIFT_DTYPE.itemsize == 40
# linked photon records (include/gvpm_hip.h, "linked photon records")
PHOTON_EMIT_DTYPE = np.dtype([("pos", np.float32, 3), ("parent_pdf", np.float32), ("parent_pos", np.float32, 3), ("edge_pdf", np.float32),
                              ("flux", np.float32, 3), ("flags", np.uint32)])
PHOTON_CHAIN_DTYPE = np.dtype([("pos", np.float32, 3), ("parent_pdf", np.float32), ("flux", np.float32, 3), ("edge_pdf", np.float32),
                               ("parent_rr", np.float32), ("flags", np.uint32)])
EMITTER_ENTRY_DTYPE = np.dtype([("prefix_w", np.float32, 3), ("parent_rr", np.float32), ("parent_n", np.float32, 3), ("parent_g", np.float32)])
assert PHOTON_EMIT_DTYPE.itemsize == 48 and PHOTON_CHAIN_DTYPE.itemsize == 40 and EMITTER_ENTRY_DTYPE.itemsize == 32
RAY_PACKED_DTYPE = np.dtype([
    ("o", np.float32, 3), ("len", np.float32), ("d", np.float32, 3), ("pdf", np.float32), ("eye", np.float32, 3),
    ("jacobian", np.float32), ("gop", np.float32)])
assert PHOTON_PACKED_DTYPE.itemsize == 76 and RAY_PACKED_DTYPE.itemsize == 52


# gvpm_bsdf: a glossy surface parent's BSDF (gvpm_upload_bsdfs)
BSDF_DTYPE = np.dtype([("kind", np.int32), ("specular", np.float32, 3), ("exponent", np.float32),
                       ("specular_sampling_weight", np.float32), ("distribution", np.int32), ("sample_visible", np.int32),
                       ("eta", np.float32, 3), ("k", np.float32, 3), ("reserved", np.float32, 2)])
assert BSDF_DTYPE.itemsize == 64
GVPM_BSDF_PHONG, GVPM_BSDF_ROUGHCONDUCTOR, GVPM_BSDF_WARD = 1, 2, 3
GVPM_WARD_WARD, GVPM_WARD_DUER, GVPM_WARD_BALANCED = 0, 1, 2
GVPM_MICROFACET_BECKMANN, GVPM_MICROFACET_GGX = 0, 1

# compact camera-beam sets (include/gvpm_hip.h, "compact camera-beam sets")
BEAM_SET_COMPACT_DTYPE = np.dtype([
    ("pixel", np.uint32), ("jitter", np.float32, 2), ("rand", np.float32), ("info", np.uint32), ("t0", np.float32, 5),
    ("len", np.float32, 5)])
assert BEAM_SET_COMPACT_DTYPE.itemsize == 60


class Sensor(C.Structure):
    """gvpm_sensor: the perspective sensor the compact beam sets are decoded with"""
    _fields_ = [("pos", C.c_double * 3), ("to_world", C.c_double * 9), ("tan_half_fov_x", C.c_double),
                ("tan_half_fov_y", C.c_double), ("width", C.c_int32), ("height", C.c_int32), ("reserved", C.c_int32 * 4)]


class Photons:
    """Host-side photon SoA as numpy arrays (owning), convertible to gvpm_photon_soa."""

    def __init__(self, n=0):
        self.n = n
        for k in PHOTON_VEC3:
            setattr(self, k, np.zeros((n, 3), np.float32))
        for k in PHOTON_F1:
            setattr(self, k, np.zeros(n, np.float32))
        for k in PHOTON_U1:
            setattr(self, k, np.zeros(n, np.uint32))

    @staticmethod
    def from_soa(soa):
        """Copy out of a gvpm_photon_soa whose pointers are host memory."""
        n = int(soa.n)
        p = Photons(0)
        p.n = n
        for k in PHOTON_VEC3:
            buf = (C.c_float * (3 * n)).from_address(getattr(soa, k)) if n else []
            setattr(p, k, np.array(buf, np.float32).reshape(n, 3).copy())
        for k in PHOTON_F1:
            buf = (C.c_float * n).from_address(getattr(soa, k)) if n else []
            setattr(p, k, np.array(buf, np.float32).copy())
        for k in PHOTON_U1:
            buf = (C.c_uint32 * n).from_address(getattr(soa, k)) if n else []
            setattr(p, k, np.array(buf, np.uint32).copy())
        return p

    def soa(self):
        s = PhotonSoA()
        for k in PHOTON_VEC3 + PHOTON_F1 + PHOTON_U1:
            a = np.ascontiguousarray(getattr(self, k))
            setattr(self, k, a)
            setattr(s, k, a.ctypes.data)
        s.n = self.n
        return s

    def subset(self, idx):
        q = Photons(0)
        idx = np.asarray(idx)
        q.n = int(idx.size) if idx.dtype != bool else int(idx.sum())
        for k in PHOTON_VEC3 + PHOTON_F1 + PHOTON_U1:
            setattr(q, k, np.ascontiguousarray(getattr(self, k)[idx]))
        return q

    def save(self, path):
        np.savez_compressed(path, **{k: getattr(self, k) for k in PHOTON_VEC3 + PHOTON_F1 + PHOTON_U1})

    @staticmethod
    def load(npz):
        p = Photons(0)
        for k in PHOTON_VEC3 + PHOTON_F1 + PHOTON_U1:
            setattr(p, k, np.ascontiguousarray(npz[k]))
        p.n = int(p.flags.shape[0])
        return p


def triangles_struct(v0, e1, e2):
    """Build a gvpm_triangles from three (n,3) float32 arrays; returns (struct, keepalive)."""
    v0 = np.ascontiguousarray(v0, np.float32)
    e1 = np.ascontiguousarray(e1, np.float32)
    e2 = np.ascontiguousarray(e2, np.float32)
    t = Triangles()
    t.v0, t.e1, t.e2, t.n = v0.ctypes.data, e1.ctypes.data, e2.ctypes.data, v0.shape[0]
    return t, (v0, e1, e2)


class PoissonParams(C.Structure):
    """gvpm_poisson_params: Solver::Params' solver configuration (poisson_solver/Solver.hpp:81-88)."""
    _fields_ = [("alpha", C.c_float), ("irls_iter_max", C.c_int32), ("irls_reg_init", C.c_float),
                ("irls_reg_iter", C.c_float), ("cg_iter_max", C.c_int32), ("cg_iter_check", C.c_int32),
                ("cg_precond", C.c_int32), ("cg_tolerance", C.c_float)]


class DevgenScene(C.Structure):
    """gvpm_devgen_scene: a closed-form scene as the device-side generators take it."""
    _fields_ = [("n_tris", C.c_uint32), ("n_mats", C.c_uint32), ("tris", C.c_void_p), ("tri_mat", C.c_void_p),
                ("mat_kind", C.c_void_p), ("mat_albedo", C.c_void_p),
                ("light_c", C.c_double * 3), ("light_u", C.c_double * 3), ("light_v", C.c_double * 3),
                ("light_n", C.c_double * 3), ("radiance", C.c_double * 3), ("light_area", C.c_double),
                ("medium", Medium), ("cam_pos", C.c_double * 3), ("tan_half_fov_x", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32), ("seed", C.c_uint32), ("camera_inside", C.c_int32),
                ("max_depth", C.c_int32), ("rr_depth", C.c_int32), ("min_depth", C.c_int32),
                ("camera_sphere", C.c_double), ("cam_to_world", C.c_double * 9)]
