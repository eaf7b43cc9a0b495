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
  if (const char *e = std::getenv("PIPE_TEST_SPAN")) opt.max_span = std::atoi(e);   // (diagnostic sweeps of the schedule options: tools/pipe_sim.py)
  if (const char *e = std::getenv("PIPE_TEST_PACK")) opt.pack_steps = std::atoi(e);
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
      int64_t active = 0, wide = 0, remote_slots = 0;
      for (int t = 0; t < T.nsteps; ++t) {
        const int32_t *hdr = reinterpret_cast<const int32_t *>(S.stream.data() + T.tile_off + (int64_t)S.koff[(size_t)T.koff_base + t] * 1024);
        active += hdr[0];
        wide += hdr[2] > pipe::MIN_W;
        { // slots of the first MIN_W entries that hold a gathered operand in some lane
          const pipe::Geometry G(hdr[2]);
          const unsigned char *tile = reinterpret_cast<const unsigned char *>(hdr);
          for (int u = 0; u < pipe::MIN_W; ++u) {
            bool any = false;
            for (int l = 0; l < pipe::LANES; ++l) any |= *reinterpret_cast<const int32_t *>(tile + G.idx_off(u, l)) > pipe::RING_Z;
            remote_slots += any;
          }
        }
      }
      std::fprintf(fp, "%d %d %d %d %lld %d %lld %d %lld", T.group, T.sweep, T.W, T.nsteps, (long long)active, T.nprod, (long long)wide, T.start_level, (long long)remote_slots);
      if (std::getenv("PIPE_DEBUG_NEEDS")) { // producers with the steps of each that the first / the middle / the last step of the task requires
        for (int j = 0; j < T.nprod; ++j) {
          std::fprintf(fp, " %d", T.prod[j]);
          for (int t : {0, T.nsteps / 2, T.nsteps - 1}) {
            const uint32_t *hdr = reinterpret_cast<const uint32_t *>(S.stream.data() + T.tile_off + (int64_t)S.koff[(size_t)T.koff_base + t] * 1024);
            std::fprintf(fp, ":%u", (hdr[pipe::HDR_REQ0 + j / 2] >> (16 * (j & 1))) & 0xffffu);
          }
        }
      }
      std::fprintf(fp, "\n");
    }
    std::fclose(fp);
    if (const char *nf = std::getenv("PIPE_DEBUG_NEEDS_BIN")) { // per task: nprod, nsteps, producer ids, then per step the steps required of each producer, the step's W and its active rows
      FILE *fb = std::fopen(nf, "wb");
      for (const pipe::Task &T : S.tasks) {
        std::fwrite(&T.nprod, 4, 1, fb);
        std::fwrite(&T.nsteps, 4, 1, fb);
        std::fwrite(T.prod, 4, (size_t)T.nprod, fb);
        for (int t = 0; t < T.nsteps; ++t) {
          const uint32_t *hdr = reinterpret_cast<const uint32_t *>(S.stream.data() + T.tile_off + (int64_t)S.koff[(size_t)T.koff_base + t] * 1024);
          for (int j = 0; j < T.nprod; ++j) {
            const uint16_t rq = (uint16_t)((hdr[pipe::HDR_REQ0 + j / 2] >> (16 * (j & 1))) & 0xffffu);
            std::fwrite(&rq, 2, 1, fb);
          }
          const uint16_t w = (uint16_t)hdr[2], act = (uint16_t)hdr[0]; // widest row and active rows of the step
          std::fwrite(&w, 2, 1, fb);
          std::fwrite(&act, 2, 1, fb);
        }
      }
      std::fclose(fb);
    }
  }
  const std::string e = pipe::emulate(S, n, d, x);
  if (!e.empty()) {
    std::snprintf(err, errlen, "%s", e.c_str());
    return 2;
  }
  return 0;
}
