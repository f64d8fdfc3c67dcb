// Staging of the per-iteration inputs (C ABI, include/gvpm_hip.h): photon maps, photon beams / planes, camera beams and
// G-VPM samples, from pageable or pinned host memory (copy stream, three slots each, prefetch) or borrowed device memory.
// Reference seam: the flattening of GPhotonNodeData + Path at gvpm/gvpm_accel.h:31-59,119-199.
#include "context.h"

extern "C" {

static bool isPinnedHost(const void *ptr) {
  hipPointerAttribute_t attr;
  if (!ptr || hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory is reported as an error: not one of ours
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int gvpm_host_alloc(uint64_t bytes, void **out) {
  if (!out || bytes == 0) return GVPM_ERR_INVALID_ARG;
  return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? GVPM_OK : GVPM_ERR_HIP;
}
int gvpm_host_free(void *p) { return hipHostFree(p) == hipSuccess ? GVPM_OK : GVPM_ERR_HIP; }
int gvpm_host_alloc_photons(uint64_t n, gvpm_photon_soa *view, void **block) {
  if (!view || !block || n == 0 || n > 0x7FFFFFF0ull) return GVPM_ERR_INVALID_ARG;
  if (hipHostMalloc(block, (size_t)n * 30 * 4, hipHostMallocDefault) != hipSuccess) return GVPM_ERR_HIP;
  float *f = (float *)*block;
  const float **v3[8] = {&view->pos, &view->wi, &view->flux, &view->parent_pos, &view->parent_n, &view->prefix_w,
                         &view->parent_scat, &view->parent_wi};
  const float **v1[4] = {&view->parent_pdf, &view->edge_pdf, &view->parent_rr, &view->parent_g};
  for (auto q : v3) { *q = f; f += (size_t)n * 3; }
  for (auto q : v1) { *q = f; f += n; }
  view->flags = (const uint32_t *)f;
  view->path_id = (const uint32_t *)f + n;
  view->n = n;
  return GVPM_OK;
}

static int uploadPhotonsCommon(gvpm_context *h, const gvpm_photon_soa *p, bool fromDevice, bool prefetch = false) {
  if (!p) return fail(h, GVPM_ERR_INVALID_ARG, "null photon soa");
  if (p->n > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many photons");
  const uint32_t n = (uint32_t)p->n;
  if (n) {
    const void *ptrs[] = {p->pos, p->wi, p->flux, p->parent_pos, p->parent_n, p->prefix_w, p->parent_scat,
                          p->parent_wi, p->parent_pdf, p->edge_pdf, p->parent_rr, p->parent_g, p->flags, p->path_id};
    for (const void *q : ptrs)
      if (!q) return fail(h, GVPM_ERR_INVALID_ARG, "null photon array");
  }
  if (fromDevice) {
    if (prefetch) return fail(h, GVPM_ERR_INVALID_ARG, "prefetch takes host buffers");
    h->rawDev = *p;
    h->phWait = false;
    h->photonsOwnedCur = false;
  } else {
    // pinned only if EVERY array is: one pageable array makes the runtime stage that copy itself, and the call must then
    // not return before the copy stream has drained (the header's contract: the library copies during the call)
    bool pinned = n > 0;
    {
      const void *all[14] = {p->pos, p->wi, p->flux, p->parent_pos, p->parent_n, p->prefix_w, p->parent_scat, p->parent_wi,
                             p->parent_pdf, p->edge_pdf, p->parent_rr, p->parent_g, p->flags, p->path_id};
      // (one block in the ABI's order -- gvpm_host_alloc_photons -- is one allocation: its first array speaks for all)
      bool oneBlock = n > 0;
      size_t off = 0;
      for (int k = 0; k < 14 && oneBlock; ++k) {
        oneBlock = (const char *)all[k] == (const char *)all[0] + off * 4;
        off += (size_t)n * (k < 8 ? 3 : 1);
      }
      for (int k = 0; k < (oneBlock ? 1 : 14) && pinned; ++k) pinned = isPinnedHost(all[k]);
    }
    if (prefetch && !pinned) return fail(h, GVPM_ERR_INVALID_ARG, "gvpm_prefetch_photons needs pinned host memory (gvpm_host_alloc*)");
    if (prefetch && h->phPending >= 0) return fail(h, GVPM_ERR_STATE, "a prefetched photon set is already pending");
    int slot = (h->phCur + 1) % 3;
    if (slot == h->phPending) slot = (h->phCur + 2) % 3;
    gvpm_context::PhotonSlot &ps = h->phSlot[slot];
    // the kernels of the gathers that last read this slot (its build; the G-Planes gather) may still be running
    if (ps.read) {
      HIP_TRY(h, hipStreamWaitEvent(h->copyStream, ps.consumed, 0));
      HIP_TRY(h, hipStreamWaitEvent(h->copyStream, ps.consumedB, 0));
    }
    ps.read = false;
    if (ps.raw.cap < (size_t)n * 30 + 8) {
      // regrowing frees the old buffer
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->streamB));
      HIP_TRY(h, ps.raw.ensure((size_t)n * 30 + 8));
    }
    const void *src[14] = {p->pos, p->wi, p->flux, p->parent_pos, p->parent_n, p->prefix_w, p->parent_scat, p->parent_wi,
                           p->parent_pdf, p->edge_pdf, p->parent_rr, p->parent_g, p->flags, p->path_id};
    const void **dst[14] = {(const void **)&ps.dev.pos, (const void **)&ps.dev.wi, (const void **)&ps.dev.flux,
                            (const void **)&ps.dev.parent_pos, (const void **)&ps.dev.parent_n, (const void **)&ps.dev.prefix_w,
                            (const void **)&ps.dev.parent_scat, (const void **)&ps.dev.parent_wi, (const void **)&ps.dev.parent_pdf,
                            (const void **)&ps.dev.edge_pdf, (const void **)&ps.dev.parent_rr, (const void **)&ps.dev.parent_g,
                            (const void **)&ps.dev.flags, (const void **)&ps.dev.path_id};
    // one packed copy when the host arrays are one block in the ABI's order (gvpm_host_alloc_photons), else one each
    bool packed = n > 0;
    size_t off = 0;
    for (int k = 0; k < 14 && packed; ++k) {
      packed = (const char *)src[k] == (const char *)src[0] + off * 4;
      off += (size_t)n * (k < 8 ? 3 : 1);
    }
    off = 0;
    for (int k = 0; k < 14; ++k) {
      const size_t words = (size_t)n * (k < 8 ? 3 : 1);
      *dst[k] = ps.raw.p + off;
      if (n && !packed) HIP_TRY(h, hipMemcpyAsync(ps.raw.p + off, src[k], words * 4, hipMemcpyHostToDevice, h->copyStream));
      off += words;
    }
    if (packed) HIP_TRY(h, hipMemcpyAsync(ps.raw.p, src[0], off * 4, hipMemcpyHostToDevice, h->copyStream));
    ps.dev.n = n;
    HIP_TRY(h, hipEventRecord(ps.copied, h->copyStream));
    // pageable memory: the caller may reuse its buffers when this returns.  Pinned memory (gvpm_host_alloc*): the copy is
    // left in flight; the buffer must stay untouched until the gather that consumes it has returned (G-BRE) or the
    // handle was synchronised
    if (!pinned) HIP_TRY(h, hipStreamSynchronize(h->copyStream));
    if (prefetch) {
      h->phPending = slot;
      return GVPM_OK;
    }
    h->phCur = slot;
    h->rawDev = ps.dev;
    h->phWait = true;
    h->photonsOwnedCur = true;
  }
  h->nph = n;
  h->havePhotons = true;
  h->photonsDirty = true;
  return GVPM_OK;
}

int gvpm_upload_photons(gvpm_context *h, const gvpm_photon_soa *p) {
  CHECK_H(h);
  return uploadPhotonsCommon(h, p, false);
}
int gvpm_upload_photons_dev(gvpm_context *h, const gvpm_photon_soa *p) {
  CHECK_H(h);
  return uploadPhotonsCommon(h, p, true);
}
int gvpm_prefetch_photons(gvpm_context *h, const gvpm_photon_soa *p) {
  CHECK_H(h);
  return uploadPhotonsCommon(h, p, false, true);
}

static int uploadPhotonBeamsCommon(gvpm_context *h, const gvpm_photon_soa *b, const float *end_n, bool fromDevice) {
  if (b && b->n && !end_n) return fail(h, GVPM_ERR_INVALID_ARG, "null end_n");
  if (b && b->n > 0xFFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "too many beams (24-bit beam index)");
  int rc = uploadPhotonsCommon(h, b, fromDevice);
  if (rc != GVPM_OK) return rc;
  if (fromDevice) {
    h->endNDev = end_n;
  } else {
    HIP_TRY(h, h->endNOwned.ensure((size_t)h->nph * 3 + 4));
    if (h->nph) {
      HIP_TRY(h, hipMemcpyAsync(h->endNOwned.p, end_n, (size_t)h->nph * 12, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->endNDev = h->endNOwned.p;
  }
  h->haveBeamsMap = true;
  return GVPM_OK;
}

static int uploadPlanesCommon(gvpm_context *h, const gvpm_photon_soa *b, const float *w1, const float *len1,
                              bool fromDevice) {
  if (b && b->n && (!w1 || !len1)) return fail(h, GVPM_ERR_INVALID_ARG, "null w1 / len1");
  int rc = uploadPhotonsCommon(h, b, fromDevice);
  if (rc != GVPM_OK) return rc;
  if (fromDevice) {
    h->w1Dev = w1;
    h->len1Dev = len1;
  } else {
    HIP_TRY(h, h->w1Owned.ensure((size_t)h->nph * 3 + 4));
    HIP_TRY(h, h->len1Owned.ensure((size_t)h->nph + 4));
    if (h->nph) {
      HIP_TRY(h, hipMemcpyAsync(h->w1Owned.p, w1, (size_t)h->nph * 12, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipMemcpyAsync(h->len1Owned.p, len1, (size_t)h->nph * 4, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->w1Dev = h->w1Owned.p;
    h->len1Dev = h->len1Owned.p;
  }
  h->havePlanes = true;
  return GVPM_OK;
}

int gvpm_upload_planes(gvpm_context *h, const gvpm_photon_soa *beams, const float *w1, const float *len1) {
  CHECK_H(h);
  return uploadPlanesCommon(h, beams, w1, len1, false);
}
int gvpm_upload_planes_dev(gvpm_context *h, const gvpm_photon_soa *beams, const float *w1, const float *len1) {
  CHECK_H(h);
  return uploadPlanesCommon(h, beams, w1, len1, true);
}

int gvpm_upload_beams(gvpm_context *h, const gvpm_photon_soa *beams, const float *end_n) {
  CHECK_H(h);
  return uploadPhotonBeamsCommon(h, beams, end_n, false);
}
int gvpm_upload_beams_dev(gvpm_context *h, const gvpm_photon_soa *beams, const float *end_n) {
  CHECK_H(h);
  return uploadPhotonBeamsCommon(h, beams, end_n, true);
}

static int uploadBeamsCommon(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t nsets, bool fromDevice,
                             bool prefetch = false) {
  if (nsets && !rays) return fail(h, GVPM_ERR_INVALID_ARG, "null camera rays");
  if (nsets > 0x0FFFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "too many beam sets");
  if (fromDevice) {
    if (prefetch) return fail(h, GVPM_ERR_INVALID_ARG, "prefetch takes host buffers");
    h->raysDev = rays;
    h->rayWait = false;
    h->raysOwnedCur = false;
  } else {
    const bool pinned = nsets && isPinnedHost(rays);
    if (prefetch && !pinned) return fail(h, GVPM_ERR_INVALID_ARG, "gvpm_prefetch_camera_beams needs pinned host memory (gvpm_host_alloc)");
    if (prefetch && h->rayPending >= 0) return fail(h, GVPM_ERR_STATE, "a prefetched camera-beam list is already pending");
    int slot = (h->rayCur + 1) % 3;
    if (slot == h->rayPending) slot = (h->rayCur + 2) % 3;
    gvpm_context::RaySlot &rs = h->raySlot[slot];
    // the evaluation kernel of the step that last used this slot may still be reading it
    if (rs.read) HIP_TRY(h, hipStreamWaitEvent(h->copyStream, rs.freed, 0));
    rs.read = false;
    if (rs.rays.cap < (size_t)nsets * 5 + 1) {
      HIP_TRY(h, hipStreamSynchronize(h->stream));  // regrowing frees the old buffer
      HIP_TRY(h, rs.rays.ensure((size_t)nsets * 5 + 1));
    }
    if (nsets)
      HIP_TRY(h, hipMemcpyAsync(rs.rays.p, rays, (size_t)nsets * 5 * sizeof(gvpm_camera_ray), hipMemcpyHostToDevice,
                                h->copyStream));
    rs.nsets = (uint32_t)nsets;
    HIP_TRY(h, hipEventRecord(rs.copied, h->copyStream));
    if (!pinned) HIP_TRY(h, hipStreamSynchronize(h->copyStream));
    if (prefetch) {
      h->rayPending = slot;
      return GVPM_OK;
    }
    h->rayCur = slot;
    h->raysDev = rs.rays.p;
    h->rayWait = true;
    h->raysOwnedCur = true;
  }
  h->nsets = (uint32_t)nsets;
  h->haveBeams = true;
  h->beamsDirty = true;
  return GVPM_OK;
}

int gvpm_upload_camera_beams(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t n_sets) {
  CHECK_H(h);
  return uploadBeamsCommon(h, rays, n_sets, false);
}
int gvpm_upload_camera_beams_dev(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t n_sets) {
  CHECK_H(h);
  return uploadBeamsCommon(h, rays, n_sets, true);
}
int gvpm_prefetch_camera_beams(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t n_sets) {
  CHECK_H(h);
  return uploadBeamsCommon(h, rays, n_sets, false, true);
}

static int uploadSamplesCommon(gvpm_context *h, const gvpm_vpm_sample *smp, uint64_t n, bool fromDevice) {
  if (n && !smp) return fail(h, GVPM_ERR_INVALID_ARG, "null vpm samples");
  if (n > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many vpm samples");
  if (fromDevice) {
    h->samplesDev = smp;
  } else {
    HIP_TRY(h, h->samplesOwned.ensure(n + 1));
    if (n) {
      HIP_TRY(h, hipMemcpyAsync(h->samplesOwned.p, smp, n * sizeof(gvpm_vpm_sample), hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->samplesDev = h->samplesOwned.p;
  }
  h->nsamples = (uint32_t)n;
  h->haveSamples = true;
  return GVPM_OK;
}

int gvpm_upload_vpm_samples(gvpm_context *h, const gvpm_vpm_sample *samples, uint64_t n) {
  CHECK_H(h);
  return uploadSamplesCommon(h, samples, n, false);
}
int gvpm_upload_vpm_samples_dev(gvpm_context *h, const gvpm_vpm_sample *samples, uint64_t n) {
  CHECK_H(h);
  return uploadSamplesCommon(h, samples, n, true);
}

}  // extern "C"
