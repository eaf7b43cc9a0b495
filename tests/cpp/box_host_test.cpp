// Host-only test harness of the "box" triangular-solve schedule builder (dune-ddm_amd/csrc/trsv_box_host.hpp): builds the streams and
// the shell system for given ILU(0) factors and runs the CPU walk of the device data flow (the nested shell solver is a sequential
// solve here).  Test infrastructure; the product library compiles the same header into libddm_hip.so.
#include "trsv_box_host.hpp"
#include <cstdio>

extern "C" int box_test_build_and_emulate(int64_t n, const int64_t *rp, const int32_t *ci, const double *lu, const int64_t *diag, int nblocks,
                                          const int64_t *block_ptr, const double *d, double *x, int64_t *stats, char *err, int errlen)
{
  box::Schedule S;
  if (!box::build(n, rp, ci, lu, diag, nblocks, block_ptr, S)) {
    std::snprintf(err, errlen, "%s", S.error.c_str());
    return 1;
  }
  const box::Block &B0 = S.blocks[0];
  const int64_t v[10] = {S.stats.box_rows, S.stats.shell_rows, S.stats.stream_bytes, S.stats.ext_products, S.stats.shell_lower_entries,
                         B0.nx, B0.ny, B0.nz, B0.nsteps, (int64_t)S.fci.size()};
  for (int k = 0; k < 10; ++k) stats[k] = v[k];
  const int64_t ns = (int64_t)S.srow.size();
  auto shell_solve = [&](const double *ds, double *xs) { // sequential solve with the shell's own factor, ascending columns
    for (int64_t t = 0; t < ns; ++t) {
      double s = ds[t];
      for (int64_t p = S.frp[(size_t)t]; p < S.fdiag[(size_t)t]; ++p) {
        const double prod = S.fva[(size_t)p] * xs[S.fci[(size_t)p]];
        s -= prod;
      }
      xs[t] = s;
    }
    for (int64_t t = ns - 1; t >= 0; --t) {
      double s = xs[t];
      for (int64_t p = S.fdiag[(size_t)t] + 1; p < S.frp[(size_t)t + 1]; ++p) {
        const double prod = S.fva[(size_t)p] * xs[S.fci[(size_t)p]];
        s -= prod;
      }
      xs[t] = s * S.fva[(size_t)S.fdiag[(size_t)t]];
    }
  };
  std::vector<double> y1((size_t)n, 0.0), y2((size_t)n, 0.0);
  const std::string e = box::emulate(S, n, d, x, shell_solve, y1.data());
  if (!e.empty()) {
    std::snprintf(err, errlen, "%s", e.c_str());
    return 2;
  }
  // the lane-by-lane walk (the kernel's index arithmetic) must give the same bits
  std::vector<double> x2((size_t)n, 0.0);
  box::emulate_lanes(S, n, d, x2.data(), shell_solve, y2.data());
  for (int64_t r = 0; r < n; ++r)
    if (std::memcmp(&y2[(size_t)r], &y1[(size_t)r], 8) != 0) {
      std::snprintf(err, errlen, "lane-by-lane walk: forward sweep differs at row %lld (%.17g vs %.17g)", (long long)r, y2[(size_t)r], y1[(size_t)r]);
      return 3;
    }
  for (int64_t r = 0; r < n; ++r)
    if (std::memcmp(&x2[(size_t)r], &x[r], 8) != 0) {
      std::snprintf(err, errlen, "lane-by-lane walk differs from the row walk at row %lld (%.17g vs %.17g)", (long long)r, x2[(size_t)r], x[r]);
      return 3;
    }
  return 0;
}

// the shell's own factor of a schedule (for tests that feed it to the other engines): sizes first (rp == NULL), then the arrays
extern "C" int box_test_shell_system(int64_t n, const int64_t *rp, const int32_t *ci, const double *lu, const int64_t *diag, int nblocks, const int64_t *block_ptr,
                                     int64_t *sizes, int64_t *frp, int32_t *fci, double *fva, int64_t *fdiag, int64_t *fbp)
{
  box::Schedule S;
  if (!box::build(n, rp, ci, lu, diag, nblocks, block_ptr, S)) return 1;
  sizes[0] = (int64_t)S.srow.size();
  sizes[1] = (int64_t)S.fci.size();
  if (!frp) return 0;
  std::copy(S.frp.begin(), S.frp.end(), frp);
  std::copy(S.fci.begin(), S.fci.end(), fci);
  std::copy(S.fva.begin(), S.fva.end(), fva);
  std::copy(S.fdiag.begin(), S.fdiag.end(), fdiag);
  std::copy(S.fblock_ptr.begin(), S.fblock_ptr.end(), fbp);
  return 0;
}
