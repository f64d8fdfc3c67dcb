"""ctypes binding of libgvpm_host.so (synthetic hosts, host_api.h)."""
import ctypes as C
import os

import numpy as np

from .. import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgvpm_host.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C gvpm_amd/csrc`) first")
        L = C.CDLL(path)
        L.gvpm_synth_create.restype = C.c_void_p
        L.gvpm_synth_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint32]
        L.gvpm_synth_destroy.argtypes = [C.c_void_p]
        L.gvpm_synth_devgen_scene.argtypes = [C.c_void_p, C.POINTER(abi.DevgenScene)]
        L.gvpm_synth_params.argtypes = [C.c_void_p, C.POINTER(abi.Params)]
        L.gvpm_synth_medium.argtypes = [C.c_void_p, C.POINTER(abi.Medium)]
        L.gvpm_synth_triangles.argtypes = [C.c_void_p, C.POINTER(abi.Triangles)]
        L.gvpm_synth_shoot.restype = C.c_uint64
        L.gvpm_synth_shoot.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(abi.PhotonSoA),
                                       C.POINTER(C.c_uint64)]
        L.gvpm_synth_beams.restype = C.c_uint64
        L.gvpm_synth_beams.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p)]
        L.gvpm_synth_vpm_samples.restype = C.c_uint64
        L.gvpm_synth_vpm_samples.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.gvpm_synth_shoot_beams.restype = C.c_uint64
        L.gvpm_synth_shoot_beams.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(abi.PhotonSoA),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.gvpm_synth_beams_interleaved.restype = C.c_uint64
        L.gvpm_synth_beams_interleaved.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.gvpm_synth_planes.restype = C.c_uint64
        L.gvpm_synth_planes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.gvpm_synth_bsdfs.restype = C.c_uint32
        L.gvpm_synth_stream_check.restype = C.c_uint64
        L.gvpm_synth_stream_check.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
        L.gvpm_synth_bsdfs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.gvpm_synth_sensor.argtypes = [C.c_void_p, C.POINTER(abi.Sensor)]
        L.gvpm_synth_jitter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        _LIB = L
    return _LIB


class SynthScene:
    """Closed-form scene + light-path / camera-beam generators (synth.h)."""

    def __init__(self, name="cbox", width=64, height=64, seed=0x6776706D):
        self._h = lib().gvpm_synth_create(name.encode(), width, height, seed)
        if not self._h:
            raise ValueError(f"unknown synthetic scene {name!r} or bad size")
        self.name, self.width, self.height = name, width, height

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gvpm_synth_destroy(self._h)
            self._h = None

    def params(self):
        p = abi.Params()
        lib().gvpm_synth_params(self._h, C.byref(p))
        return p

    def medium(self):
        m = abi.Medium()
        lib().gvpm_synth_medium(self._h, C.byref(m))
        return m

    def triangles(self):
        """(v0, e1, e2) as (n,3) float32 arrays."""
        t = abi.Triangles()
        lib().gvpm_synth_triangles(self._h, C.byref(t))
        n = t.n
        out = []
        for ptr in (t.v0, t.e1, t.e2):
            buf = (C.c_float * (3 * n)).from_address(ptr)
            out.append(np.array(buf, np.float32).reshape(n, 3).copy())
        return tuple(out)

    def bsdfs(self):
        """the BSDF table of the scene's glossy walls (numpy, abi.BSDF_DTYPE; empty for the Lambertian scenes)"""
        n = lib().gvpm_synth_bsdfs(self._h, None, 0)
        out = np.zeros(n, abi.BSDF_DTYPE)
        if n:
            lib().gvpm_synth_bsdfs(self._h, out.ctypes.data, n)
        return out

    def sensor(self):
        """the scene's pinhole sensor (gvpm_sensor) the compact beam sets are decoded with"""
        s = abi.Sensor()
        assert lib().gvpm_synth_sensor(self._h, C.byref(s)) == 0
        return s

    def jitter(self, iteration, rays):
        """(nsets, 2) float32: the fractional film offsets the base paths of `rays` were sampled at"""
        rays = np.ascontiguousarray(rays)
        n = rays.size // 5
        out = np.zeros((n, 2), np.float32)
        assert lib().gvpm_synth_jitter(self._h, iteration, rays.ctypes.data, n, out.ctypes.data) == 0
        return out

    def devgen_scene(self):
        """The scene as gvpm_devgen_create takes it (arrays owned by this object: keep it alive)."""
        d = abi.DevgenScene()
        rc = lib().gvpm_synth_devgen_scene(self._h, C.byref(d))
        if rc != 0:
            raise RuntimeError(f"gvpm_synth_devgen_scene failed: {rc}")
        return d

    def shoot_photons(self, iteration, capacity):
        """-> (abi.Photons, nb_paths)"""
        soa = abi.PhotonSoA()
        nb = C.c_uint64(0)
        lib().gvpm_synth_shoot(self._h, iteration, capacity, C.byref(soa), C.byref(nb))
        return abi.Photons.from_soa(soa), int(nb.value)

    def shoot_beams(self, iteration, capacity):
        """-> (abi.Photons re-read as photon beams, end_n (n,3) float32, nb_paths)"""
        soa = abi.PhotonSoA()
        nb = C.c_uint64(0)
        ptr = C.c_void_p()
        n = lib().gvpm_synth_shoot_beams(self._h, iteration, capacity, C.byref(soa), C.byref(ptr), C.byref(nb))
        beams = abi.Photons.from_soa(soa)
        end_n = (np.array((C.c_float * (3 * n)).from_address(ptr.value), np.float32).reshape(n, 3).copy()
                 if n else np.zeros((0, 3), np.float32))
        return beams, end_n, int(nb.value)

    def shoot_planes(self, iteration, capacity):
        """-> (beams, end_n, w1 (n,3), len1 (n,), nb_paths): photon planes = beams + second edge"""
        beams, end_n, nb = self.shoot_beams(iteration, capacity)
        p1, p2 = C.c_void_p(), C.c_void_p()
        n = lib().gvpm_synth_planes(self._h, iteration, C.byref(p1), C.byref(p2))
        if n == 0:
            return beams, end_n, np.zeros((0, 3), np.float32), np.zeros(0, np.float32), nb
        w1 = np.array((C.c_float * (3 * n)).from_address(p1.value), np.float32).reshape(n, 3).copy()
        len1 = np.array((C.c_float * n).from_address(p2.value), np.float32).copy()
        return beams, end_n, w1, len1, nb

    def camera_beams(self, iteration, x0=0, y0=0, x1=None, y1=None):
        """-> structured array (n_sets, 5) of gvpm_camera_ray"""
        x1 = self.width if x1 is None else x1
        y1 = self.height if y1 is None else y1
        ptr = C.c_void_p()
        n = lib().gvpm_synth_beams(self._h, iteration, x0, y0, x1, y1, C.byref(ptr))
        if n == 0:
            return np.zeros((0, 5), abi.CAMERA_RAY_DTYPE)
        buf = (C.c_char * (n * 5 * 64)).from_address(ptr.value)
        return np.frombuffer(buf, abi.CAMERA_RAY_DTYPE).reshape(n, 5).copy()

    def camera_beams_interleaved(self, iteration, tile_mod, tile_rem):
        """The beam sets of the 4x4-pixel tiles t with t % tile_mod == tile_rem (image-sharded ranks)."""
        ptr = C.c_void_p()
        n = lib().gvpm_synth_beams_interleaved(self._h, iteration, tile_mod, tile_rem, C.byref(ptr))
        if n == 0:
            return np.zeros((0, 5), abi.CAMERA_RAY_DTYPE)
        buf = (C.c_char * (n * 5 * 64)).from_address(ptr.value)
        return np.frombuffer(buf, abi.CAMERA_RAY_DTYPE).reshape(n, 5).copy()

    def camera_beams_and_vpm_samples(self, iteration, nb_camera_samples, x0=0, y0=0, x1=None, y1=None):
        """-> (rays (n_sets,5), samples (n_sets*nb,) VPM_SAMPLE_DTYPE) for one G-VPM iteration"""
        rays = self.camera_beams(iteration, x0, y0, x1, y1)
        ptr = C.c_void_p()
        n = lib().gvpm_synth_vpm_samples(self._h, iteration, nb_camera_samples, C.byref(ptr))
        if n == 0:
            return rays, np.zeros(0, abi.VPM_SAMPLE_DTYPE)
        buf = (C.c_char * (n * 16)).from_address(ptr.value)
        return rays, np.frombuffer(buf, abi.VPM_SAMPLE_DTYPE).copy()
