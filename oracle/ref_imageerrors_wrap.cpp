// TEST INFRASTRUCTURE.  C entry point over the REFERENCE's own image-error header (scripts/rgbe/sources/imageerrors.h,
// compiled from where it lies under /root/reference by oracle/Makefile.ref into oracle/_ref/): metric() with the metric
// BASELINE.json's parity line names (ERelMSE, :117-121) and the others of the header.  Pins gvpm_amd/metrics.py.
#include "imageerrors.h"

extern "C" float ref_image_metric(const double *img, const double *ref, int width, int height, int which) {
  return metric(img, ref, nullptr, nullptr, width, height, (EErrorMetric)which);
}
