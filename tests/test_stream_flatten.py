"""The streaming flattening the device generator uses (StreamPath, gvpm_amd/host/synth_core.h: the path flattened while
it is walked, a ring of four vertices) against flattenPath / flattenBeams over the whole path: the same records, bit
for bit, and the same `counted` flag, on every closed-form scene (host build of the shared header; no GPU)."""
import ctypes as C

import pytest

from gvpm_amd.host import SynthScene, lib


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "laser", "laser_in", "fogroom", "cbox_mirror", "cbox_phong", "cbox_conductor"])
@pytest.mark.parametrize("beams", [0, 1])
def test_streaming_flattening_is_the_array_flattening(scene, beams):
    sc = SynthScene(scene, 64, 48)
    n = C.c_uint64(0)
    differ = lib().gvpm_synth_stream_check(sc._h, 2, 60000, beams, C.byref(n))
    assert differ == 0 and n.value > 5000
