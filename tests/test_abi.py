"""The C-ABI library loads and exports every symbol include/gvpm_hip.h declares; the ctypes
mirror has the C layout.  No compute calls (runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from gvpm_amd import abi, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gvpm_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gvpm_[a-z0-9_]+)\s*\(", text)) - {"gvpm_context"})


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(hip.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gvpm_hip.h but not exported"
    assert sorted(hip.SYMBOLS) == names


def test_abi_version_without_gpu():
    lib = C.CDLL(hip.LIB_PATH)
    assert lib.gvpm_abi_version() == abi.GVPM_ABI_VERSION


def test_struct_layout_matches_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gvpm_hip.h"\nint main(){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(gvpm_params), sizeof(gvpm_medium),'
                   'sizeof(gvpm_triangles), sizeof(gvpm_photon_soa), sizeof(gvpm_camera_ray), sizeof(gvpm_stats),'
                   'offsetof(gvpm_params, alpha), offsetof(gvpm_camera_ray, gop));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)]).split()))
    want = [C.sizeof(abi.Params), C.sizeof(abi.Medium), C.sizeof(abi.Triangles), C.sizeof(abi.PhotonSoA),
            abi.CAMERA_RAY_DTYPE.itemsize, C.sizeof(abi.Stats), abi.Params.alpha.offset,
            abi.CAMERA_RAY_DTYPE.fields["gop"][1]]
    assert got == want


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "h.c"
    src.write_text('#include "gvpm_hip.h"\nint main(void){return GVPM_ABI_VERSION - 1;}')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "h.o")])


def test_create_reports_errors_without_gpu():
    """Parameter validation happens before any device call and never aborts."""
    lib = hip.lib()
    p = abi.Params()
    h = C.c_void_p()
    assert lib.gvpm_create(C.byref(p), 0, C.byref(h)) == abi.GVPM_ERR_INVALID_ARG  # abi_version 0
    assert lib.gvpm_create(None, 0, C.byref(h)) == abi.GVPM_ERR_INVALID_ARG
    from gvpm_amd.host import SynthScene
    p = SynthScene("cbox", 8, 8).params()
    p.vol_technique = abi.GVPM_VOL_BRE2D  # useShiftNull with a 2D kernel: GPMConfig::load raises EError
    assert lib.gvpm_create(C.byref(p), 0, C.byref(h)) == abi.GVPM_ERR_UNSUPPORTED
    assert lib.gvpm_gather(None, 1, 1) == abi.GVPM_ERR_INVALID_ARG
    assert lib.gvpm_last_error(None) == b"null handle"


def test_photon_flag_packing():
    f = abi.pf_make(abi.GVPM_PARENT_MEDIUM, 2, 1, 7, abi.GVPM_BSDF_DIFFUSE_REFLECTION)
    assert f & 3 == 2 and (f >> 2) & 7 == 2 and (f >> 5) & 1 == 1 and (f >> 8) & 0xFF == 7 and f >> 16 == 2
    assert abi.ray_info(1, 2) == 0x201
