"""The synthetic host's photon shooting against a second implementation that shares no code with it
(tests/indep_lightpaths.py: pure Python from the random-walk conventions of src/libbidir and GPhotonMap::tryAppend)."""
import numpy as np
import pytest

import indep_lightpaths as IL
from gvpm_amd import abi
from gvpm_amd.host import SynthScene


def same_photons(got, ref, rtol=0.0):
    assert np.array_equal(got["flags"], ref.flags) and np.array_equal(got["path_id"], ref.path_id)
    for k in abi.PHOTON_VEC3 + abi.PHOTON_F1:
        a, b = got[k], getattr(ref, k)
        assert a.shape == b.shape, k
        assert np.allclose(a, b, rtol=rtol, atol=rtol * 1e-2), k


@pytest.mark.parametrize("scene,cap,it", [("cbox", 2500, 1), ("cbox_hg", 1500, 3), ("cbox_in", 1500, 2), ("cbox_mirror", 1500, 2),
                                          ("fogroom", 1200, 1), ("laser", 600, 4),
                                          # general position (round 5): one rotation of everything, inner boxes tilted
                                          ("cbox_rot", 1500, 1), ("fogroom_rot", 800, 2), ("laser_rot", 400, 3), ("cbox_mirror_rot", 1000, 2)])
def test_host_generator_equals_the_independent_walk(scene, cap, it):
    sc = SynthScene(scene, 16, 16)
    got, nb = IL.shoot_photons(IL.Scene(sc.devgen_scene()), it, cap)
    ref, nb_ref = sc.shoot_photons(it, cap)
    assert nb == nb_ref and ref.n == cap
    same_photons(got, ref, rtol=1e-6)   # (both double precision with the host's libm: equal to the last float bit)
    kinds = np.bincount(ref.flags & 3, minlength=3)
    assert kinds.min() > 0                # emitter, surface and medium parents all occur


def test_philox_known_answers():
    """Philox4x32-10 (Salmon et al., SC'11), the Random123 known-answer vectors: counter / key all zero, all ones, pi."""
    def block(ctr, key):
        p = IL.Philox(key[0], key[1], ctr[1], ctr[2], ctr[3])
        p.ctr[0] = ctr[0]
        p._refill()
        return p.buf
    assert block([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert block([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert block([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
