"""Static checks of shim/gvpm_hip_bridge.h, which cannot be compiled here (Mitsuba's dependencies are absent): every
field of the ABI records is assigned, every C-ABI call is declared in include/gvpm_hip.h with the right arity, and --
when the reference tree is present -- every Mitsuba / gvpm member the bridge touches exists in the reference headers."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = open(os.path.join(ROOT, "shim", "gvpm_hip_bridge.h")).read()
HDR = open(os.path.join(ROOT, "include", "gvpm_hip.h")).read()
REF = "/root/reference"


def struct_fields(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HDR, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            out.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1].lstrip("*")))
    return out


def test_every_photon_soa_field_is_filled():
    for f in struct_fields("gvpm_photon_soa"):
        if f == "n":
            assert "v.n = flags.size()" in SHIM
            continue
        assert re.search(r"m_soa\.%s\b" % f, SHIM) or re.search(r"v\.%s = %s\.data\(\)" % (f, f), SHIM), f
        assert re.search(r"v\.%s = " % f, SHIM), f


def test_every_camera_ray_and_vpm_sample_field_is_assigned():
    for f in struct_fields("gvpm_camera_ray"):
        assert re.search(r"\b(r|set\[0\])\.%s\b(\[\d\])? = " % f, SHIM), f
    for f in struct_fields("gvpm_vpm_sample"):
        assert re.search(r"\bsm\.%s = " % f, SHIM), f
    for f in struct_fields("gvpm_medium"):
        if f == "reserved":
            continue
        assert re.search(r"\bgm\.%s\b" % f, SHIM), f
    params = [f for f in struct_fields("gvpm_params") if f != "reserved"]
    for f in params:
        assert re.search(r"\bp\.%s = " % f, SHIM), f


def test_every_abi_call_is_declared_with_that_arity():
    decls = {}
    for m in re.finditer(r"^int (gvpm_\w+)\((.*?)\);", HDR, re.S | re.M):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    calls = set(re.findall(r"\b(gvpm_(?!hip|context|params|medium|photon|camera|vpm_sample|triangles|status)\w+)\(", SHIM))
    assert {"gvpm_create", "gvpm_upload_scene", "gvpm_upload_medium", "gvpm_upload_photons", "gvpm_upload_beams",
            "gvpm_upload_planes", "gvpm_upload_camera_beams", "gvpm_upload_vpm_samples", "gvpm_gather",
            "gvpm_download_accum", "gvpm_download_vpm_state", "gvpm_destroy", "gvpm_reset"} <= calls
    for name in calls:
        if name == "gvpm_last_error":
            continue
        assert name in decls, name
        m = re.search(r"\b%s\(" % name, SHIM)
        depth, i, n = 1, m.end(), 1
        while depth:
            c = SHIM[i]
            depth += c == "("
            depth -= c == ")"
            n += c == "," and depth == 1
            i += 1
        assert n == decls[name], (name, n, decls[name])


def test_every_citation_names_a_file_of_the_reference():
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present")
    base = os.path.join(REF, "src", "integrators", "photonmapper")
    for path in set(re.findall(r"\b((?:src|include)/[\w/.\-]+\.(?:cpp|h))", SHIM)):
        if path == "include/gvpm_hip.h":  # ours
            continue
        assert os.path.exists(os.path.join(REF, path)), path
    for path in set(re.findall(r"(?<![\w/])((?:gvpm/|shift/|gvpm_|plane_struct|beams|volume_utils)[\w/.]*\.(?:cpp|h))", SHIM)):
        if path.startswith("gvpm_hip"):  # ours
            continue
        cands = [os.path.join(base, path), os.path.join(base, "gvpm", path), os.path.join(REF, "src", "integrators", path)]
        assert any(os.path.exists(c) for c in cands), path


@pytest.mark.parametrize("header,members", [
    ("src/integrators/photonmapper/gvpm/gvpm_struct.h",
     ["volTechnique", "maxDepth", "minDepth", "useMIS", "useShiftNull", "pathSet", "powerHeuristic", "noMediumShift", "useManifold",
      "debugShift", "lightingInteractionMode", "bsdfInteractionMode", "nbCameraSamples", "alpha", "initialScaleVolume",
      "minCameraDepth", "maxCameraDepth", "stratified", "getWeightBeam", "getWeightVertex", "getVertexInfo", "GOp", "mediumFlux",
      "shiftedMediumFlux", "weightedMediumFlux", "scaleVol", "NVol", "haveSmoke", "struct GPMThreadData", "MemoryPool pool",
      "relaxME", "offsetGenerator"]),
    ("src/integrators/photonmapper/gvpm/shift/operation/shift_ME.h", ["generateShiftPathME", "ShiftME"]),
    ("src/integrators/photonmapper/gvpm/shift/shift_utilities.h", ["struct ShiftRecord", "throughtput"]),
    ("include/mitsuba/bidir/mut_manifold.h", ["getSpecularManifold"]),
    ("include/mitsuba/bidir/manifold.h", ["Float det(const Path &path, int b, int c)"]),
    ("src/integrators/photonmapper/gvpm/gvpm_accel.h", ["struct GPhotonNodeData", "vertexId", "lightPath", "pathID", "operator[]", "size()"]),
    ("src/integrators/photonmapper/gvpm/gvpm_beams.h", ["struct LTPhotonBeam", "edgeID", "pathID", "const Path *path"]),
    ("src/integrators/photonmapper/gvpm/gvpm_plane.h", ["transformBeam"]),
    ("src/integrators/photonmapper/plane_struct.h", ["w1()", "length1()"]),
    ("src/integrators/photonmapper/beams.h", ["getBeams()"]),
    ("src/integrators/photonmapper/gvpm/shift/shift_utilities.h", ["getTypeShift", "getVertexComponentType", "generateOffsetPos"]),
    ("src/integrators/photonmapper/gvpm/shift/shift_cameraPath.h", ["bool generate(", "validVolumeEdge"]),
    ("include/mitsuba/core/pmf.h", ["sampleReuse", "normalize()", "append("]),
    ("include/mitsuba/render/medium.h", ["getSigmaA", "getSigmaS", "getSigmaT", "isHomogeneous"]),
    ("include/mitsuba/bidir/vertex.h", ["getSamplePosition", "getGeometricNormal", "isEmitterSample", "rrWeight", "getMediumSamplingRecord"]),
    ("include/mitsuba/render/shape.h", ["createTriMesh"]),
    ("include/mitsuba/render/trimesh.h", ["getVertexPositions", "getTriangles", "getTriangleCount"]),
    ("include/mitsuba/render/scene.h", ["getShapes()", "getMedia()", "getSensor"]),
    ("include/mitsuba/render/sensor.h", ["class MTS_EXPORT_RENDER PerspectiveCamera", "getXFov", "getAspect", "getWorldTransform", "getFilm"]),
    ("include/mitsuba/render/film.h", ["getCropOffset", "getCropSize", "getSize()"]),
    ("include/mitsuba/core/transform.h", ["getMatrix", "transformAffine"]),
    ("include/mitsuba/render/bsdf.h", ["getSpecularReflectance", "getRoughness", "pdfComponent", "ESpatiallyVarying"]),
    ("include/mitsuba/bidir/vertex.h", ["sampledComponentIndex", "PathVertex *clone(MemoryPool &pool) const"]),
    ("include/mitsuba/bidir/edge.h", ["PathEdge *clone(MemoryPool &pool) const"]),
    ("include/mitsuba/bidir/mempool.h", ["allocVertex"]),
    ("src/integrators/photonmapper/gvpm/gvpm_geoOps.h", ["fastGOp"]),
    ("src/integrators/photonmapper/beams_struct.h", ["getOri()", "getDir()", "getPos("]),
    ("src/bsdfs/microfacet.h", ["MicrofacetDistribution(const Properties &props", "isIsotropic", "getAlphaU", "getSampleVisible",
                                "getType()", "EBeckmann", "EGGX"]),
    ("src/bsdfs/ior.h", ["inline Float lookupIOR(const Properties &props"]),
    ("include/mitsuba/core/cobject.h", ["getProperties"]),
    ("include/mitsuba/core/properties.h", ["hasProperty", "getSpectrum"]),
    ("src/bsdfs/roughconductor.cpp", ["class RoughConductor", "props.getSpectrum(\"eta\", intEta) / extEta"]),
])
def test_members_the_bridge_touches_exist_in_the_reference(header, members):
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present")
    text = open(os.path.join(REF, header)).read()
    for m in members:
        assert m in text, (header, m)
        token = m.split("(")[0].split()[-1]
        assert token in SHIM or m in ("size()", "operator[]", "MemoryPool pool", "struct GPMThreadData", "const Path *path",
                                      "struct GPhotonNodeData", "struct LTPhotonBeam", "normalize()", "append(", "struct ShiftRecord",
                                      "Float det(const Path &path, int b, int c)", "class MTS_EXPORT_RENDER PerspectiveCamera",
                                      "PathVertex *clone(MemoryPool &pool) const", "PathEdge *clone(MemoryPool &pool) const",
                                      "MicrofacetDistribution(const Properties &props", "inline Float lookupIOR(const Properties &props",
                                      "class RoughConductor", "props.getSpectrum(\"eta\", intEta) / extEta"), (header, m)
