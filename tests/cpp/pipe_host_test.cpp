// Host-only test harness of the "pipe" triangular-solve schedule builder (dune-ddm_amd/csrc/trsv_pipe_host.hpp):
// builds the schedule for given ILU(0) factors and runs the CPU emulation of the device data flow.
// Test infrastructure; the product library compiles the same header into libddm_hip.so.
#include "trsv_pipe_host.hpp"
#include <cstdio>
#include <cstdlib>

extern "C" int pipe_test_build_and_emulate(int64_t n, const int64_t *rp, const int32_t *ci, const double *lu, const int64_t *diag,
                                           int nblocks, const int64_t *block_ptr, int delta, int vote, const double *d, double *x,
                                           int64_t *stats, char *err, int errlen)
{
  pipe::Options opt;
  opt.delta = delta;
  opt.vote = vote;
  pipe::Schedule S;
  if (!pipe::build(n, rp, ci, lu, diag, nblocks, block_ptr, opt, S)) {
    std::snprintf(err, errlen, "%s", S.error.c_str());
    return 1;
  }
  const pipe::Stats &st = S.stats;
  const int64_t v[20] = {st.ntasks[0], st.ntasks[1], st.nsteps[0], st.nsteps[1], st.rows, st.entries, st.entries_local, st.entries_self_global,
                         st.entries_remote, st.max_prod, st.max_steps, st.regrouped, st.nchains[0], st.nchains[1], S.geo.W, S.geo.tile_bytes,
                         (int64_t)S.stream.size(), S.nposL, S.nposU, (int64_t)S.tasks.size()};
  for (int k = 0; k < 20; ++k) stats[k] = v[k];
  if (const char *f = std::getenv("PIPE_DEBUG_TASKS")) { // per task: group sweep W nsteps active-rows nprod
    FILE *fp = std::fopen(f, "w");
    for (const pipe::Task &T : S.tasks) {
      int64_t active = 0, wide = 0;
      for (int t = 0; t < T.nsteps; ++t) {
        const int32_t *hdr = reinterpret_cast<const int32_t *>(S.stream.data() + T.tile_off + (int64_t)S.koff[(size_t)T.koff_base + t] * 1024);
        active += hdr[0];
        wide += hdr[2] > pipe::MIN_W;
      }
      std::fprintf(fp, "%d %d %d %d %lld %d %lld\n", T.group, T.sweep, T.W, T.nsteps, (long long)active, T.nprod, (long long)wide);
    }
    std::fclose(fp);
  }
  const std::string e = pipe::emulate(S, n, d, x);
  if (!e.empty()) {
    std::snprintf(err, errlen, "%s", e.c_str());
    return 2;
  }
  return 0;
}
