// Philox4x32-10 counter-based generator (Salmon et al., SC'11) used by the
// synthetic hosts to replace Mitsuba's per-block SFMT `Sampler` streams
// (gvpm/gvpm_gatherpoint.h:204-236), which cannot be reproduced on a GPU
// ("RNG parity").  Every random decision is keyed by
// (seed, stream, iteration, index) so host, oracle and device inputs agree.
#pragma once
#include <cstdint>

#include "vecmath.h"  // GVPM_HD

namespace gvpm {

struct Philox {
  uint32_t key[2];
  uint32_t ctr[4];
  uint32_t out[4];
  int have;

  GVPM_HD Philox(uint32_t seed, uint32_t stream, uint32_t a, uint32_t b, uint32_t c = 0) {
    key[0] = seed;
    key[1] = stream;
    ctr[0] = 0;
    ctr[1] = a;
    ctr[2] = b;
    ctr[3] = c;
    have = 0;
  }

  GVPM_HD static inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    uint64_t p = (uint64_t)a * b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }

  GVPM_HD void refill() {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0, lo0, hi1, lo1;
      mulhilo(0xD2511F53u, c0, hi0, lo0);
      mulhilo(0xCD9E8D57u, c2, hi1, lo1);
      uint32_t n0 = hi1 ^ c1 ^ k0;
      uint32_t n1 = lo1;
      uint32_t n2 = hi0 ^ c3 ^ k1;
      uint32_t n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    ctr[0]++;
    have = 4;
  }

  GVPM_HD uint32_t nextU32() {
    if (have == 0) refill();
    return out[4 - (have--)];
  }
  // uniform in [0,1): 24 random bits, exactly representable in fp32
  GVPM_HD float next1D() { return (float)(nextU32() >> 8) * (1.0f / 16777216.0f); }
};

}  // namespace gvpm
