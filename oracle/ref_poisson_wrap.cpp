// ORACLE -- TEST INFRASTRUCTURE ONLY.
// C entry point around the REFERENCE's own screened-Poisson solver (poisson::Solver, compiled from
// /root/reference/src/integrators/poisson_solver by Makefile.ref into oracle/_ref/): exactly the call
// sequence of gvpm.cpp:631-637 / 671-680 (importImagesMTS, setupBackend, solveIndirect,
// exportImagesMTS).  This file is ours; the reference sources are compiled where they lie.
#include <string.h>

#include <string>

#include "Solver.hpp"

extern "C" int ref_poisson_solve(const char *preset, const char *backend, float alpha, int width, int height, float *dx,
                                 float *dy, float *throughput, float *direct, float *out) {
  poisson::Solver::Params params;
  if (!params.setConfigPreset(preset)) return -1;
  params.alpha = alpha;
  params.backend = backend;  // "Naive" (the defining loops, Backend.cpp) or "OpenMP"
  params.verbose = false;
  params.setLogFunction(poisson::Solver::Params::LogFunction([](const std::string &) {}));
  poisson::Solver solver(params);
  solver.importImagesMTS(dx, dy, throughput, direct, width, height);
  solver.setupBackend();
  solver.solveIndirect();
  solver.exportImagesMTS(out);
  return 0;
}
