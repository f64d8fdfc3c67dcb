"""Load/store the small golden fixtures of tests/golden (inputs + fp64 oracle outputs)."""
import ctypes as C

import numpy as np

from gvpm_amd import abi


class Golden:
    pass


def save(path, p, m, tris, ph, rays, r, it, nb, accum, evaluations, extra=None):
    d = {k: getattr(ph, k) for k in abi.PHOTON_VEC3 + abi.PHOTON_F1 + abi.PHOTON_U1}
    d.update(params=np.frombuffer(bytes(p), np.uint8), medium=np.frombuffer(bytes(m), np.uint8),
             v0=tris[0], e1=tris[1], e2=tris[2], rays=rays.view(np.uint8), radius=np.float64(r),
             it=np.int64(it), nb=np.int64(nb), accum=accum.astype(np.float64), evaluations=np.int64(evaluations))
    for k, v in (extra or {}).items():
        d["x_" + k] = v
    np.savez_compressed(path, **d)


def load(path):
    z = np.load(path)
    g = Golden()
    g.p = abi.Params.from_buffer_copy(z["params"].tobytes())
    # (the fixtures hold the struct as it was saved; the version field names the LIBRARY's ABI, not the data's: version 2
    # changed gvpm_bsdf only, which no fixture holds)
    assert g.p.abi_version in (1, 2, abi.GVPM_ABI_VERSION)
    g.p.abi_version = abi.GVPM_ABI_VERSION
    g.m = abi.Medium.from_buffer_copy(z["medium"].tobytes())
    g.tris = (z["v0"], z["e1"], z["e2"])
    g.ph = abi.Photons.load(z)
    g.rays = z["rays"].view(abi.CAMERA_RAY_DTYPE).reshape(-1, 5)
    g.r, g.it, g.nb = float(z["radius"]), int(z["it"]), int(z["nb"])
    g.accum, g.evaluations = z["accum"], int(z["evaluations"])
    g.extra = {k[2:]: z[k] for k in z.files if k.startswith("x_")}
    return g
