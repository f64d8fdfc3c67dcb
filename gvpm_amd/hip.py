"""ctypes binding of libgvpm_hip.so -- the C ABI of include/gvpm_hip.h.

There is no CPU fallback: if the HIP library is missing or no gfx950 device is
visible, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GVPM_HIP_LIB", os.path.join(_HERE, "libgvpm_hip.so"))  # override: A/B probes only
_LIB = None

# every symbol include/gvpm_hip.h declares
SYMBOLS = [
    "gvpm_create", "gvpm_destroy", "gvpm_last_error", "gvpm_abi_version", "gvpm_reset",
    "gvpm_upload_scene", "gvpm_upload_medium", "gvpm_upload_photons", "gvpm_upload_camera_beams",
    "gvpm_upload_photons_dev", "gvpm_upload_camera_beams_dev", "gvpm_upload_vpm_samples",
    "gvpm_upload_beams", "gvpm_upload_beams_dev", "gvpm_upload_planes", "gvpm_upload_planes_dev",
    "gvpm_upload_vpm_samples_dev", "gvpm_download_vpm_state", "gvpm_gather", "gvpm_get_radius",
    "gvpm_set_global_scale", "gvpm_get_stats", "gvpm_get_exact_shift_count", "gvpm_get_kernel_time", "gvpm_get_phase_time", "gvpm_download_accum",
    "gvpm_download_accum_dev", "gvpm_download_film", "gvpm_synchronize", "gvpm_comm_unique_id", "gvpm_comm_init",
    "gvpm_allreduce_accum", "gvpm_allreduce_film", "gvpm_download_film_dev", "gvpm_devgen_create",
    "gvpm_devgen_destroy", "gvpm_devgen_shoot_photons", "gvpm_devgen_shoot_beams", "gvpm_devgen_camera_beams", "gvpm_devgen_read",
    "gvpm_poisson_preset", "gvpm_poisson_solve", "gvpm_poisson_solve_dev",
    "gvpm_host_alloc", "gvpm_host_alloc_photons", "gvpm_host_free", "gvpm_prefetch_photons", "gvpm_prefetch_camera_beams",
    "gvpm_pack_photons", "gvpm_unpack_photons", "gvpm_pack_camera_beams", "gvpm_unpack_camera_beams", "gvpm_upload_materials",
    "gvpm_upload_photons_packed", "gvpm_prefetch_photons_packed", "gvpm_upload_camera_beams_packed",
    "gvpm_prefetch_camera_beams_packed",
    "gvpm_linked_photons_bound", "gvpm_pack_photons_linked", "gvpm_unpack_photons_linked", "gvpm_upload_photons_linked",
    "gvpm_prefetch_photons_linked",
    "gvpm_enable_host_shifts", "gvpm_download_shift_requests", "gvpm_upload_host_shifts",
    "gvpm_upload_sensor", "gvpm_pack_camera_beams_compact", "gvpm_unpack_camera_beams_compact",
    "gvpm_upload_camera_beams_compact", "gvpm_prefetch_camera_beams_compact", "gvpm_upload_bsdfs", "gvpm_gather_primal",
]


class GvpmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gvpm error {code}: {msg}")
        self.code = code


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing -- the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gvpm_amd/csrc`). "
                "There is no CPU fallback for the gather path.")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.gvpm_create.argtypes = [C.POINTER(abi.Params), C.c_int, C.POINTER(vp)]
        L.gvpm_destroy.argtypes = [vp]
        L.gvpm_last_error.argtypes = [vp]
        L.gvpm_last_error.restype = C.c_char_p
        L.gvpm_reset.argtypes = [vp]
        L.gvpm_upload_scene.argtypes = [vp, C.POINTER(abi.Triangles)]
        L.gvpm_upload_medium.argtypes = [vp, C.POINTER(abi.Medium)]
        L.gvpm_upload_photons.argtypes = [vp, C.POINTER(abi.PhotonSoA)]
        L.gvpm_upload_photons_dev.argtypes = [vp, C.POINTER(abi.PhotonSoA)]
        L.gvpm_upload_camera_beams.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_upload_camera_beams_dev.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_upload_beams.argtypes = [vp, C.POINTER(abi.PhotonSoA), vp]
        L.gvpm_upload_beams_dev.argtypes = [vp, C.POINTER(abi.PhotonSoA), vp]
        L.gvpm_upload_planes.argtypes = [vp, C.POINTER(abi.PhotonSoA), vp, vp]
        L.gvpm_upload_planes_dev.argtypes = [vp, C.POINTER(abi.PhotonSoA), vp, vp]
        L.gvpm_upload_vpm_samples.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_upload_vpm_samples_dev.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_download_vpm_state.argtypes = [vp, vp, vp]
        L.gvpm_gather.argtypes = [vp, C.c_int, C.c_uint64]
        L.gvpm_gather_primal.argtypes = [vp, C.c_int, C.c_uint64]
        L.gvpm_get_radius.argtypes = [vp, C.POINTER(C.c_float)]
        L.gvpm_set_global_scale.argtypes = [vp, C.c_float]
        L.gvpm_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
        L.gvpm_get_kernel_time.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.gvpm_get_phase_time.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.gvpm_download_accum.argtypes = [vp, vp]
        L.gvpm_download_accum_dev.argtypes = [vp, vp]
        L.gvpm_download_film.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
        L.gvpm_synchronize.argtypes = [vp]
        L.gvpm_linked_photons_bound.argtypes = [C.c_uint64]
        L.gvpm_linked_photons_bound.restype = C.c_size_t
        L.gvpm_pack_photons_linked.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint32, vp, vp]
        L.gvpm_unpack_photons_linked.argtypes = [vp, C.c_size_t, vp, C.c_uint32, vp]
        L.gvpm_upload_photons_linked.argtypes = [vp, vp, C.c_size_t]
        L.gvpm_prefetch_photons_linked.argtypes = [vp, vp, C.c_size_t]
        L.gvpm_comm_unique_id.argtypes = [vp]
        L.gvpm_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
        L.gvpm_allreduce_accum.argtypes = [vp]
        L.gvpm_allreduce_film.argtypes = [vp, vp]
        L.gvpm_devgen_create.argtypes = [C.POINTER(abi.DevgenScene), C.c_int, C.POINTER(vp)]
        L.gvpm_devgen_destroy.argtypes = [vp]
        L.gvpm_devgen_shoot_photons.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(abi.PhotonSoA), C.POINTER(C.c_uint64)]
        L.gvpm_devgen_shoot_beams.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(abi.PhotonSoA), C.POINTER(vp),
                                              C.POINTER(C.c_uint64)]
        L.gvpm_devgen_camera_beams.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.gvpm_devgen_read.argtypes = [vp, vp, vp, C.c_uint64]
        L.gvpm_download_film_dev.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        L.gvpm_poisson_preset.argtypes = [C.c_char_p, C.POINTER(abi.PoissonParams)]
        L.gvpm_poisson_solve.argtypes = [vp, C.POINTER(abi.PoissonParams), C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.gvpm_poisson_solve_dev.argtypes = [vp, C.POINTER(abi.PoissonParams), C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.gvpm_host_alloc.argtypes = [C.c_uint64, C.POINTER(vp)]
        L.gvpm_enable_host_shifts.argtypes = [vp, C.c_uint64]
        L.gvpm_download_shift_requests.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.gvpm_upload_host_shifts.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_pack_photons.argtypes = [C.POINTER(abi.PhotonSoA), vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        L.gvpm_unpack_photons.argtypes = [vp, C.c_uint64, vp, C.c_uint32, C.POINTER(abi.PhotonSoA)]
        L.gvpm_pack_camera_beams.argtypes = [vp, C.c_uint64, vp]
        L.gvpm_unpack_camera_beams.argtypes = [vp, C.c_uint64, vp]
        L.gvpm_upload_materials.argtypes = [vp, vp, C.c_uint32]
        L.gvpm_upload_photons_packed.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_prefetch_photons_packed.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_upload_camera_beams_packed.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_prefetch_camera_beams_packed.argtypes = [vp, vp, C.c_uint64]
        L.gvpm_upload_bsdfs.argtypes = [vp, vp, C.c_uint32]
        L.gvpm_upload_sensor.argtypes = [vp, C.POINTER(abi.Sensor)]
        L.gvpm_pack_camera_beams_compact.argtypes = [C.POINTER(abi.Sensor), vp, vp, C.c_uint64, vp, C.POINTER(C.c_uint64), vp,
                                                     C.POINTER(C.c_uint64), vp]
        L.gvpm_unpack_camera_beams_compact.argtypes = [C.POINTER(abi.Sensor), vp, C.c_uint64, vp]
        L.gvpm_upload_camera_beams_compact.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
        L.gvpm_prefetch_camera_beams_compact.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
        L.gvpm_host_alloc_photons.argtypes = [C.c_uint64, C.POINTER(abi.PhotonSoA), C.POINTER(vp)]
        L.gvpm_host_free.argtypes = [vp]
        L.gvpm_prefetch_photons.argtypes = [vp, C.POINTER(abi.PhotonSoA)]
        L.gvpm_prefetch_camera_beams.argtypes = [vp, vp, C.c_uint64]
        _LIB = L
    return _LIB


class PinnedPhotons:
    """A photon SoA in ONE block of pinned host memory (gvpm_host_alloc_photons): uploads from it are a single packed,
    asynchronous copy.  fill() copies an abi.Photons into it."""

    def __init__(self, n):
        self.soa = abi.PhotonSoA()
        self._block = C.c_void_p()
        rc = lib().gvpm_host_alloc_photons(n, C.byref(self.soa), C.byref(self._block))
        if rc != 0:
            raise GvpmError(rc, "gvpm_host_alloc_photons failed")
        self.n = n

    def fill(self, ph):
        assert ph.n == self.n
        for k in abi.PHOTON_VEC3 + abi.PHOTON_F1 + abi.PHOTON_U1:
            a = np.ascontiguousarray(getattr(ph, k))
            C.memmove(getattr(self.soa, k), a.ctypes.data, a.nbytes)
        return self

    def close(self):
        if self._block:
            lib().gvpm_host_free(self._block)
            self._block = C.c_void_p()

    __del__ = close


class PinnedRays:
    """Camera-beam sets in pinned host memory (gvpm_host_alloc)."""

    def __init__(self, rays):
        rays = np.ascontiguousarray(rays)
        self.nsets = rays.shape[0]
        self._p = C.c_void_p()
        rc = lib().gvpm_host_alloc(max(rays.nbytes, 64), C.byref(self._p))
        if rc != 0:
            raise GvpmError(rc, "gvpm_host_alloc failed")
        C.memmove(self._p, rays.ctypes.data, rays.nbytes)

    @property
    def ptr(self):
        return self._p

    def close(self):
        if self._p:
            lib().gvpm_host_free(self._p)
            self._p = C.c_void_p()

    __del__ = close


class MaterialTable:
    """The table the packed photon records index (gvpm_material): grows as gvpm_pack_photons meets new materials."""

    def __init__(self, cap=256):
        self.table = np.zeros(cap, abi.MATERIAL_DTYPE)
        self.n = 0


def pack_photons(ph, table, out=None):
    """gvpm_pack_photons: abi.Photons -> packed records (numpy, PHOTON_PACKED_DTYPE), in `out` (e.g. a view of pinned
    memory) when given.  Plain host code: works without a GPU."""
    if out is None:
        out = np.zeros(ph.n, abi.PHOTON_PACKED_DTYPE)
    assert out.dtype == abi.PHOTON_PACKED_DTYPE and out.size == ph.n and out.flags["C_CONTIGUOUS"]
    soa = ph.soa()
    cnt = C.c_uint32(table.n)
    rc = lib().gvpm_pack_photons(C.byref(soa), out.ctypes.data, table.table.ctypes.data, table.table.size, C.byref(cnt))
    if rc != 0:
        raise GvpmError(rc, "gvpm_pack_photons failed (more materials than the table holds?)")
    table.n = cnt.value
    return out


def unpack_photons(packed, table):
    """gvpm_unpack_photons: what the device makes of the records, as abi.Photons"""
    ph = abi.Photons(packed.size)
    soa = ph.soa()
    packed = np.ascontiguousarray(packed)
    rc = lib().gvpm_unpack_photons(packed.ctypes.data, packed.size, table.table.ctypes.data, table.n, C.byref(soa))
    if rc != 0:
        raise GvpmError(rc, "gvpm_unpack_photons failed")
    return ph


def linked_photons_bound(n):
    return int(lib().gvpm_linked_photons_bound(n))


def pack_photons_linked(ph, table, out=None):
    """gvpm_pack_photons_linked: abi.Photons -> one blob of linked records (uint8 array of the blob's size; a view of `out`
    -- e.g. pinned memory of linked_photons_bound(n) bytes -- when given).  Plain host code: works without a GPU."""
    cap = linked_photons_bound(ph.n)
    if out is None:
        out = np.zeros(cap, np.uint8)
    assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
    soa = ph.soa()
    cnt, nbytes = C.c_uint32(table.n), C.c_size_t(0)
    rc = lib().gvpm_pack_photons_linked(C.byref(soa), out.ctypes.data, out.size, table.table.ctypes.data, table.table.size,
                                        C.byref(cnt), C.byref(nbytes))
    if rc != 0:
        raise GvpmError(rc, "gvpm_pack_photons_linked failed (blob capacity or a table exhausted?)")
    table.n = cnt.value
    return out[:nbytes.value]


def linked_header(blob):
    """the gvpm_linked_header of a blob as a dict"""
    w = np.frombuffer(blob[:64].tobytes(), np.uint32)
    names = ("magic", "n", "n_full", "n_emit", "n_chain", "n_emitters", "off_kinds", "off_groups", "off_emitters", "off_full",
             "off_emit", "off_chain", "bytes")
    return {k: int(w[i]) for i, k in enumerate(names)}


def unpack_photons_linked(blob, table):
    """gvpm_unpack_photons_linked: what the device makes of a blob, as abi.Photons"""
    blob = np.ascontiguousarray(blob)
    ph = abi.Photons(linked_header(blob)["n"])
    soa = ph.soa()
    rc = lib().gvpm_unpack_photons_linked(blob.ctypes.data, blob.size, table.table.ctypes.data, table.n, C.byref(soa))
    if rc != 0:
        raise GvpmError(rc, "gvpm_unpack_photons_linked failed")
    return ph


def pack_camera_beams(rays, out=None):
    """gvpm_pack_camera_beams: (nsets, 5) camera rays -> nsets records of 272 bytes (uint8 array (nsets, 272))"""
    rays = np.ascontiguousarray(rays)
    n = rays.size // 5
    if out is None:
        out = np.zeros((n, 272), np.uint8)
    assert out.nbytes == n * 272 and out.flags["C_CONTIGUOUS"]
    rc = lib().gvpm_pack_camera_beams(rays.ctypes.data, n, out.ctypes.data)
    if rc != 0:
        raise GvpmError(rc, "gvpm_pack_camera_beams failed (a shifted ray on another edge than its base?)")
    return out


def unpack_camera_beams(packed):
    packed = np.ascontiguousarray(packed)
    n = packed.nbytes // 272
    rays = np.zeros((n, 5), abi.CAMERA_RAY_DTYPE)
    rc = lib().gvpm_unpack_camera_beams(packed.ctypes.data, n, rays.ctypes.data)
    if rc != 0:
        raise GvpmError(rc, "gvpm_unpack_camera_beams failed")
    return rays


def pack_camera_beams_compact(sensor, rays, jitter, out_compact=None, out_full=None):
    """gvpm_pack_camera_beams_compact: (nsets, 5) camera rays + (nsets, 2) film offsets -> (compact records
    [BEAM_SET_COMPACT_DTYPE], full records [uint8 (nfull, 272)], new_index [uint32 per input set]).  out_*: buffers with
    room for nsets records each (e.g. views of pinned memory); the returned arrays are their filled prefixes."""
    rays = np.ascontiguousarray(rays)
    n = rays.size // 5
    jitter = np.ascontiguousarray(jitter, np.float32)
    assert jitter.size == 2 * n
    if out_compact is None:
        out_compact = np.zeros(n, abi.BEAM_SET_COMPACT_DTYPE)
    if out_full is None:
        out_full = np.zeros((n, 272), np.uint8)
    assert out_compact.nbytes >= n * 60 and out_full.nbytes >= n * 272
    nc, nf = C.c_uint64(0), C.c_uint64(0)
    idx = np.zeros(n, np.uint32)
    rc = lib().gvpm_pack_camera_beams_compact(C.byref(sensor), rays.ctypes.data, jitter.ctypes.data, n, out_compact.ctypes.data,
                                              C.byref(nc), out_full.ctypes.data, C.byref(nf), idx.ctypes.data)
    if rc != 0:
        raise GvpmError(rc, "gvpm_pack_camera_beams_compact failed")
    return out_compact[:nc.value], out_full[:nf.value], idx


def unpack_camera_beams_compact(sensor, compact):
    compact = np.ascontiguousarray(compact)
    n = compact.size
    rays = np.zeros((n, 5), abi.CAMERA_RAY_DTYPE)
    rc = lib().gvpm_unpack_camera_beams_compact(C.byref(sensor), compact.ctypes.data, n, rays.ctypes.data)
    if rc != 0:
        raise GvpmError(rc, "gvpm_unpack_camera_beams_compact failed")
    return rays


class PinnedPacked:
    """One iteration's inputs as packed records in pinned host memory: what a pipelined producer hands to
    gvpm_upload_*_packed / gvpm_prefetch_*_packed.  With sensor + jitter the beam sets are split into compact records
    (60 bytes) and full ones (272) as gvpm_pack_camera_beams_compact does; otherwise every set is a full record."""

    def __init__(self, ph, rays, table, sensor=None, jitter=None, linked=False):
        self.n = ph.n
        self.nsets = np.asarray(rays).size // 5
        self.compact = sensor is not None
        self.linked = bool(linked) and ph.n > 0
        self._pp, self._pr, self._pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        pcap = linked_photons_bound(self.n) if self.linked else self.n * 76
        for ptr, nbytes in ((self._pp, pcap), (self._pr, self.nsets * 272)) + (((self._pc, self.nsets * 60),) if self.compact else ()):
            rc = lib().gvpm_host_alloc(max(nbytes, 64), C.byref(ptr))
            if rc != 0:
                raise GvpmError(rc, "gvpm_host_alloc failed")
        rv = np.frombuffer((C.c_char * (self.nsets * 272)).from_address(self._pr.value), np.uint8).reshape(self.nsets, 272) if self.nsets else np.zeros((0, 272), np.uint8)
        if self.linked:
            # one blob of linked records (gvpm_pack_photons_linked): 40 / 48 / 76 bytes a photon
            bv = np.frombuffer((C.c_char * pcap).from_address(self._pp.value), np.uint8)
            self.photon_bytes = int(pack_photons_linked(ph, table, out=bv).size)
        else:
            pv = np.frombuffer((C.c_char * (self.n * 76)).from_address(self._pp.value), abi.PHOTON_PACKED_DTYPE) if self.n else np.zeros(0, abi.PHOTON_PACKED_DTYPE)
            pack_photons(ph, table, out=pv)
            self.photon_bytes = self.n * 76
        if self.compact:
            cv = np.frombuffer((C.c_char * (self.nsets * 60)).from_address(self._pc.value), abi.BEAM_SET_COMPACT_DTYPE) if self.nsets else np.zeros(0, abi.BEAM_SET_COMPACT_DTYPE)
            c, f, self.new_index = pack_camera_beams_compact(sensor, rays, jitter, out_compact=cv, out_full=rv)
            self.ncompact, self.nfull = int(c.size), int(f.shape[0])
            self.nbytes = self.photon_bytes + self.ncompact * 60 + self.nfull * 272
        else:
            pack_camera_beams(rays, out=rv)
            self.ncompact, self.nfull = 0, self.nsets
            self.nbytes = self.photon_bytes + self.nsets * 272

    def close(self):
        for ptr in (self._pp, self._pr, self._pc):
            if ptr:
                lib().gvpm_host_free(ptr)
        self._pp, self._pr, self._pc = C.c_void_p(), C.c_void_p(), C.c_void_p()

    __del__ = close


class Context:
    """One gvpm handle (single owner; calls are stream ordered)."""

    def __init__(self, params, device=0):
        self._h = C.c_void_p()
        self.params = params.copy()
        rc = lib().gvpm_create(C.byref(self.params), device, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise GvpmError(rc, "gvpm_create failed")
        self._keep = []

    def _check(self, rc):
        if rc != 0:
            raise GvpmError(rc, lib().gvpm_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().gvpm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._check(lib().gvpm_reset(self._h))

    def upload_scene(self, v0, e1, e2):
        t, keep = abi.triangles_struct(v0, e1, e2)
        self._check(lib().gvpm_upload_scene(self._h, C.byref(t)))

    def upload_medium(self, medium):
        self._check(lib().gvpm_upload_medium(self._h, C.byref(medium)))

    def upload_photons(self, photons):
        soa = photons.soa()
        self._check(lib().gvpm_upload_photons(self._h, C.byref(soa)))

    def upload_photons_dev(self, soa):
        """soa: abi.PhotonSoA whose pointers are device addresses; the caller keeps them alive."""
        self._check(lib().gvpm_upload_photons_dev(self._h, C.byref(soa)))

    def upload_camera_beams(self, rays):
        rays = np.ascontiguousarray(rays)
        assert rays.dtype == abi.CAMERA_RAY_DTYPE
        n = rays.size // 5
        self._check(lib().gvpm_upload_camera_beams(self._h, rays.ctypes.data if n else None, n))

    def upload_camera_beams_dev(self, dev_ptr, n_sets):
        self._check(lib().gvpm_upload_camera_beams_dev(self._h, dev_ptr, n_sets))

    # host buffers in pinned memory (PinnedPhotons / PinnedRays): asynchronous copies on the handle's copy stream
    def upload_pinned(self, photons=None, rays=None):
        if photons is not None:
            self._check(lib().gvpm_upload_photons(self._h, C.byref(photons.soa)))
        if rays is not None:
            self._check(lib().gvpm_upload_camera_beams(self._h, rays.ptr, rays.nsets))

    # manifold shifts through the host (gvpm_enable_host_shifts)
    def enable_host_shifts(self, capacity):
        self._check(lib().gvpm_enable_host_shifts(self._h, capacity))

    def download_shift_requests(self, cap):
        """(requests recorded by the last gather [numpy SHIFT_REQUEST_DTYPE, at most cap], their total number)"""
        out = np.zeros(cap, abi.SHIFT_REQUEST_DTYPE)
        n = C.c_uint64(0)
        self._check(lib().gvpm_download_shift_requests(self._h, out.ctypes.data if cap else None, cap, C.byref(n)))
        return out[:min(cap, n.value)], n.value

    def upload_host_shifts(self, results):
        results = np.ascontiguousarray(results)
        assert results.dtype == abi.HOST_SHIFT_DTYPE
        self._check(lib().gvpm_upload_host_shifts(self._h, results.ctypes.data if results.size else None, results.size))

    # packed records (gvpm_upload_*_packed): 76 bytes a photon, 272 a beam set
    def upload_materials(self, table):
        self._check(lib().gvpm_upload_materials(self._h, table.table.ctypes.data, table.n))

    def upload_photons_packed(self, packed):
        packed = np.ascontiguousarray(packed)
        assert packed.dtype == abi.PHOTON_PACKED_DTYPE
        self._check(lib().gvpm_upload_photons_packed(self._h, packed.ctypes.data if packed.size else None, packed.size))

    def upload_camera_beams_packed(self, packed):
        packed = np.ascontiguousarray(packed)
        n = packed.nbytes // 272
        self._check(lib().gvpm_upload_camera_beams_packed(self._h, packed.ctypes.data if n else None, n))

    def upload_bsdfs(self, table):
        """table: numpy array of abi.BSDF_DTYPE (the scene's glossy surfaces)"""
        table = np.ascontiguousarray(table, abi.BSDF_DTYPE)
        self._check(lib().gvpm_upload_bsdfs(self._h, table.ctypes.data if table.size else None, table.size))

    def upload_sensor(self, sensor):
        self._check(lib().gvpm_upload_sensor(self._h, C.byref(sensor)))

    def upload_camera_beams_compact(self, compact, full):
        """compact: BEAM_SET_COMPACT_DTYPE records; full: uint8 (nfull, 272) packed records (gvpm_pack_camera_beams_compact)"""
        compact = np.ascontiguousarray(compact)
        full = np.ascontiguousarray(full)
        nc, nf = compact.size, full.nbytes // 272
        self._check(lib().gvpm_upload_camera_beams_compact(self._h, compact.ctypes.data if nc else None, nc,
                                                           full.ctypes.data if nf else None, nf))

    def upload_photons_linked(self, blob):
        blob = np.ascontiguousarray(blob)
        self._check(lib().gvpm_upload_photons_linked(self._h, blob.ctypes.data, blob.size))

    def upload_pinned_packed(self, pk):
        if pk.linked:
            self._check(lib().gvpm_upload_photons_linked(self._h, pk._pp, pk.photon_bytes))
        else:
            self._check(lib().gvpm_upload_photons_packed(self._h, pk._pp, pk.n))
        if pk.compact:
            self._check(lib().gvpm_upload_camera_beams_compact(self._h, pk._pc if pk.ncompact else None, pk.ncompact,
                                                               pk._pr if pk.nfull else None, pk.nfull))
        else:
            self._check(lib().gvpm_upload_camera_beams_packed(self._h, pk._pr, pk.nsets))

    def prefetch_packed(self, pk):
        if pk.linked:
            self._check(lib().gvpm_prefetch_photons_linked(self._h, pk._pp, pk.photon_bytes))
        else:
            self._check(lib().gvpm_prefetch_photons_packed(self._h, pk._pp, pk.n))
        if pk.compact:
            self._check(lib().gvpm_prefetch_camera_beams_compact(self._h, pk._pc if pk.ncompact else None, pk.ncompact,
                                                                 pk._pr if pk.nfull else None, pk.nfull))
        else:
            self._check(lib().gvpm_prefetch_camera_beams_packed(self._h, pk._pr, pk.nsets))

    def prefetch(self, photons=None, rays=None):
        """The inputs of the step after the coming gather (gvpm_prefetch_*)."""
        if photons is not None:
            self._check(lib().gvpm_prefetch_photons(self._h, C.byref(photons.soa)))
        if rays is not None:
            self._check(lib().gvpm_prefetch_camera_beams(self._h, rays.ptr, rays.nsets))

    def upload_beams(self, beams, end_n):
        """beams: abi.Photons re-read as photon beams; end_n: (n,3) float32"""
        soa = beams.soa()
        end_n = np.ascontiguousarray(end_n, np.float32)
        self._check(lib().gvpm_upload_beams(self._h, C.byref(soa), end_n.ctypes.data if beams.n else None))

    def upload_planes(self, beams, w1, len1):
        """beams: abi.Photons re-read as photon beams; w1: (n,3), len1: (n,) float32 second plane edge"""
        soa = beams.soa()
        w1 = np.ascontiguousarray(w1, np.float32)
        len1 = np.ascontiguousarray(len1, np.float32)
        assert w1.size == 3 * beams.n and len1.size == beams.n
        self._check(lib().gvpm_upload_planes(self._h, C.byref(soa), w1.ctypes.data if beams.n else None,
                                             len1.ctypes.data if beams.n else None))

    def upload_beams_dev(self, soa, end_n_dev_ptr):
        """soa: abi.PhotonSoA of device addresses (photon beams); end_n_dev_ptr: 3 floats per beam, device."""
        self._check(lib().gvpm_upload_beams_dev(self._h, C.byref(soa), end_n_dev_ptr))

    def upload_planes_dev(self, soa, w1_dev_ptr, len1_dev_ptr):
        self._check(lib().gvpm_upload_planes_dev(self._h, C.byref(soa), w1_dev_ptr, len1_dev_ptr))

    def upload_vpm_samples(self, samples):
        samples = np.ascontiguousarray(samples)
        assert samples.dtype == abi.VPM_SAMPLE_DTYPE
        self._check(lib().gvpm_upload_vpm_samples(self._h, samples.ctypes.data if samples.size else None, samples.size))

    def upload_vpm_samples_dev(self, dev_ptr, n):
        self._check(lib().gvpm_upload_vpm_samples_dev(self._h, dev_ptr, n))

    def download_vpm_state(self):
        H, W = self.params.height, self.params.width
        sv, nv = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
        self._check(lib().gvpm_download_vpm_state(self._h, sv.ctypes.data, nv.ctypes.data))
        return sv, nv

    def gather(self, it, nb_paths):
        self._check(lib().gvpm_gather(self._h, it, nb_paths))

    def gather_primal(self, it, nb_paths):
        """the primal beam radiance estimate (sppm's volumePhotonPassBRE): fluxVol in the first three accumulators"""
        self._check(lib().gvpm_gather_primal(self._h, it, nb_paths))

    def radius(self):
        r = C.c_float()
        self._check(lib().gvpm_get_radius(self._h, C.byref(r)))
        return r.value

    def set_global_scale(self, s):
        self._check(lib().gvpm_set_global_scale(self._h, s))

    def stats(self):
        s = abi.Stats()
        self._check(lib().gvpm_get_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k in
                ("evaluations", "candidates", "null_shifts", "diffuse_shifts", "failed_shifts")}

    def exact_shifts(self):
        """(evaluated, lost): the shifts the exact fp64 pass has taken since the last reset (gvpm_get_exact_shift_count)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(lib().gvpm_get_exact_shift_count(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def grid_info(self):
        """G-BRE: (kind, cells) of the last build's photon cells -- kind 1: the ray-bundle cells (gvpm_stats.reserved[0])"""
        s = abi.Stats()
        self._check(lib().gvpm_get_stats(self._h, C.byref(s)))
        return int(s.reserved[0]) >> 56, int(s.reserved[0]) & 0xFFFFFFFF

    def refused_steps(self):
        """G-BRE: optimistic steps whose traversal / evaluation the build's guard refused and the host queued again
        (gvpm_stats.reserved[0] bits 32-55; GVPM_OPTIMISTIC_REFUSE forces them)"""
        s = abi.Stats()
        self._check(lib().gvpm_get_stats(self._h, C.byref(s)))
        return (int(s.reserved[0]) >> 32) & 0xFFFFFF

    def kernel_time(self):
        ms, n = C.c_float(), C.c_uint32()
        self._check(lib().gvpm_get_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def phase_time(self, phase):
        ms, n = C.c_float(), C.c_uint32()
        self._check(lib().gvpm_get_phase_time(self._h, phase, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def download_accum(self):
        out = np.zeros((self.params.height, self.params.width, 27), np.float32)
        self._check(lib().gvpm_download_accum(self._h, out.ctypes.data))
        return out

    def download_accum_dev(self, dev_ptr):
        self._check(lib().gvpm_download_accum_dev(self._h, dev_ptr))

    def download_film(self, it, reuse_primal=True, emission=None):
        H, W = self.params.height, self.params.width
        thr, dx, dy = (np.zeros((H, W, 3), np.float32) for _ in range(3))
        em = None if emission is None else np.ascontiguousarray(emission, np.float32)
        self._check(lib().gvpm_download_film(self._h, it, 1 if reuse_primal else 0,
                                             None if em is None else em.ctypes.data,
                                             thr.ctypes.data, dx.ctypes.data, dy.ctypes.data))
        return thr, dx, dy

    def download_film_dev(self, it, film_dev_ptr, reuse_primal=False, emission_dev_ptr=None):
        """throughput | dx | dy (9*W*H floats) into device memory, on the context's stream."""
        self._check(lib().gvpm_download_film_dev(self._h, it, 1 if reuse_primal else 0, emission_dev_ptr, film_dev_ptr))

    def allreduce_film(self, film_dev_ptr):
        self._check(lib().gvpm_allreduce_film(self._h, film_dev_ptr))

    def synchronize(self):
        self._check(lib().gvpm_synchronize(self._h))

    def comm_init(self, id128, rank, world):
        buf = (C.c_char * 128).from_buffer_copy(id128)
        self._check(lib().gvpm_comm_init(self._h, buf, rank, world))

    def poisson_solve(self, dx, dy, throughput, direct=None, preset="L1D", alpha=0.2, params=None):
        """Screened-Poisson reconstruction (H, W, 3) float32 -> (H, W, 3); params overrides the preset."""
        if params is None:
            params = poisson_preset(preset)
            params.alpha = alpha
        dx, dy = (np.ascontiguousarray(a, np.float32) for a in (dx, dy))
        tp = None if throughput is None else np.ascontiguousarray(throughput, np.float32)
        di = None if direct is None else np.ascontiguousarray(direct, np.float32)
        H, W = dx.shape[:2]
        out = np.zeros((H, W, 3), np.float32)
        self._check(lib().gvpm_poisson_solve(self._h, C.byref(params), W, H, dx.ctypes.data, dy.ctypes.data,
                                             None if tp is None else tp.ctypes.data,
                                             None if di is None else di.ctypes.data, out.ctypes.data))
        return out

    def allreduce_accum(self):
        self._check(lib().gvpm_allreduce_accum(self._h))


class DeviceGenerator:
    """Device-side photon shooting and camera-beam generation for a closed-form scene (gvpm_devgen_*).
    Outputs are device pointers owned by the generator; it rotates three output buffers per kind, so an output stays
    untouched for the two following calls of the same kind (the kernels of up to three steps are in flight)."""

    def __init__(self, synth_scene, device=0):
        self._scene = synth_scene  # keeps the host arrays alive during create
        self._h = C.c_void_p()
        d = synth_scene.devgen_scene()
        rc = lib().gvpm_devgen_create(C.byref(d), device, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise GvpmError(rc, "gvpm_devgen_create failed")  # (status kept: GVPM_ERR_NO_DEVICE / _HIP / _INVALID_ARG)

    def close(self):
        if getattr(self, "_h", None):
            lib().gvpm_devgen_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def shoot_photons(self, iteration, capacity):
        """-> (abi.PhotonSoA of device pointers, nb_paths)"""
        soa, nb = abi.PhotonSoA(), C.c_uint64(0)
        rc = lib().gvpm_devgen_shoot_photons(self._h, iteration, capacity, C.byref(soa), C.byref(nb))
        if rc != 0:
            raise GvpmError(rc, "gvpm_devgen_shoot_photons failed")
        return soa, int(nb.value)

    def shoot_beams(self, iteration, capacity):
        """-> (abi.PhotonSoA of device pointers, end_n device pointer, nb_paths)"""
        soa, nb, en = abi.PhotonSoA(), C.c_uint64(0), C.c_void_p()
        rc = lib().gvpm_devgen_shoot_beams(self._h, iteration, capacity, C.byref(soa), C.byref(en), C.byref(nb))
        if rc != 0:
            raise GvpmError(rc, "gvpm_devgen_shoot_beams failed")
        return soa, en.value, int(nb.value)

    def read(self, dev_ptr, count, dtype):
        """Copy `count` items of `dtype` of a generator output to a numpy array."""
        out = np.empty(count, dtype)
        rc = lib().gvpm_devgen_read(self._h, dev_ptr, out.ctypes.data, out.nbytes)
        if rc != 0:
            raise GvpmError(rc, "gvpm_devgen_read failed")
        return out

    def camera_beams(self, iteration, tile_mod=1, tile_rem=0):
        """-> (device pointer to n_sets * 5 gvpm_camera_ray, n_sets)"""
        ptr, n = C.c_void_p(), C.c_uint64(0)
        rc = lib().gvpm_devgen_camera_beams(self._h, iteration, tile_mod, tile_rem, C.byref(ptr), C.byref(n))
        if rc != 0:
            raise GvpmError(rc, "gvpm_devgen_camera_beams failed")
        return ptr.value, int(n.value)


def poisson_preset(name):
    p = abi.PoissonParams()
    rc = lib().gvpm_poisson_preset(name.encode(), C.byref(p))
    if rc != 0:
        raise GvpmError(rc, f"unknown Poisson preset {name}")
    return p


def comm_unique_id():
    buf = (C.c_char * 128)()
    rc = lib().gvpm_comm_unique_id(buf)
    if rc != 0:
        raise GvpmError(rc, "gvpm_comm_unique_id failed")
    return bytes(buf)
