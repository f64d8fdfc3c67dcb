// Synthetic closed-form hosts: what Mitsuba's scene + libbidir produce for the
// gvpm integrator, restated for scenes simple enough that no Mitsuba is
// needed on the GPU box (SURVEY 7 step 3, 8d "Synthetic inputs").
//
// The generators emit exactly the inputs of the C ABI (include/gvpm_hip.h):
//  * light paths: Path::randomWalk(EImportance) conventions of
//    src/libbidir/vertex.cpp:35-332, edge.cpp:27-84, path.cpp:471-501, flattened
//    like GPhotonMap::tryAppend (gvpm/gvpm_accel.h:119-199);
//  * camera beams: randomWalkFromPixelToFirstDiffuse long-beam camera paths
//    (gvpm/gvpm_gatherpoint.h:22-170) with the SVertexPDF caches of
//    GatherPoint::generateVertexInfo (gvpm/gvpm_struct.h:523-565) for the base
//    and ShiftGatherPoint::generate/trace (gvpm/shift/shift_cameraPath.h) for
//    the four offset pixels.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gvpm_hip.h"
#include "synth_core.h"
#include "vecmath.h"

namespace gvpm {

struct SynthScene {
  std::string name;
  std::vector<SynthTri> tris;
  std::vector<SynthMat> mats;
  // one square area light (AreaLight, src/emitters/area.cpp)
  V3 lightC, lightU, lightV, lightN, radiance;
  double lightArea;
  // homogeneous medium filling the box [bmin,bmax]
  gvpm_medium medium;
  V3 bmin, bmax;
  // pinhole (perspective) sensor
  V3 camPos;
  V3 camX = V3(1, 0, 0), camY = V3(0, 1, 0), camZ = V3(0, 0, 1);  // camera -> world: right, towards larger rows, BACK (looks along -camZ)
  double tanHalfFovX;
  int width, height;
  uint32_t seed;
  bool cameraInside;  // sensor inside the medium: the medium edge of the camera path is edge 1
  // light-path walk parameters (GPMConfig maxDepth, rrDepth, minDepth)
  int maxDepth, rrDepth, minDepth;
  double cameraSphere;  // world units, already scaled as in gvpm.cpp:162

  // `_rot` scenes: the rotation every point / direction of the scene description goes through (rows of R)
  bool rotated = false;
  V3 rot[3] = {V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1)};
  V3 toWorld(V3 p) const;

  double bsphereRadius() const;
  SceneView view() const;  // what the generators (host or device) read
  void addQuad(V3 a, V3 b, V3 c, V3 d, int mat);  // a,b,c,d counter-clockwise seen from the front
};

bool makeScene(const std::string &name, int width, int height, uint32_t seed, SynthScene &out);

struct PhotonBuffers {
  std::vector<float> pos, wi, flux, parent_pos, parent_n, prefix_w, parent_scat, parent_wi;
  std::vector<float> parent_pdf, edge_pdf, parent_rr, parent_g;
  std::vector<uint32_t> flags, path_id;
  uint64_t n = 0;
  void clear();
  void view(gvpm_photon_soa &out) const;
};

// Shoot light paths until `capacity` photons are stored (gvpm_proc.cpp:278-350).
// Returns the number of light paths shot (m_numShotVolume).
uint64_t shootPhotons(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out);

// Photon beams (G-Beams): one record per medium edge of a light path (LTBeamMap::tryAppendLT,
// gvpm/gvpm_beams.h:54-84), flattened into the photon SoA re-read as documented for
// gvpm_upload_beams; end_n = geometric normal of the beam's end vertex (zero in the medium).
uint64_t shootBeams(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out,
                    std::vector<float> &endN);

// Photon planes from photon beams (LTPhotonPlane::transformBeam): w1 (3 floats) and length1 per beam.
void planesFromBeams(const SynthScene &sc, int iteration, const PhotonBuffers &beams, std::vector<float> &w1,
                     std::vector<float> &len1);

// Camera beam sets (5 rays each) for the pixels [x0,x1) x [y0,y1) of iteration
// `iteration`; pixels whose camera path has no medium edge produce no set.
// selW (optional): per set, the weight of its edge in the G-VPM edge selection (gvpm.cpp:1117-1129).
void cameraBeams(const SynthScene &sc, int iteration, int x0, int y0, int x1, int y1,
                 std::vector<gvpm_camera_ray> &out, int tileMod = 1, int tileRem = 0, std::vector<float> *selW = nullptr);

// G-VPM camera samples for beam sets produced by cameraBeams() (with their selection weights): nbCameraSamples records
// per PIXEL, consecutive (gvpm.cpp:1117-1172: edge selection by sampleReuse over the pixel's medium edges).
void cameraSamplesVPM(const SynthScene &sc, int iteration, const std::vector<gvpm_camera_ray> &rays,
                      const std::vector<float> &selW, int nbCameraSamples, std::vector<gvpm_vpm_sample> &out);

void defaultParams(const SynthScene &sc, gvpm_params &p);

}  // namespace gvpm
