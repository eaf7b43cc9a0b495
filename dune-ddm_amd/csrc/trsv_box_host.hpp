// Host side of the "box" triangular-solve engine for ILU(0) factors of matrices whose diagonal blocks START with a structured box:
// rows [0, nx ny nz) of a block numbered lexicographically (x fastest) with the full 27-point pattern inside the box, followed by any
// number of further rows (the overlap shell of a Schwarz subdomain: dune/ddm/overlap_extension.hh appends the rows it adds behind the
// rows the rank owns).  Bit-identical to the sequential back-solve of dune-istl's ILU (every row subtracts its products in ascending
// column order), as the other engines are.
//
// Why a second single-launch engine beside "pipe" (trsv_pipe_host.hpp): the pipe engine is general -- operands by index from an LDS
// ring or by gathers, one L2 round trip inside every step, 12 B per factor entry through an LDS tile ring -- and is bound by that
// latency (890 levels x 1.8 us at 216^3).  In a lexicographic box the dependency structure is known: row (i, j, k) needs
// (i-1, j, k), (i-1..i+1, j-1, k) and the nine neighbours of plane k-1.  One wavefront walks ONE plane: lane = j mod 64, step
// l = i + 2 j.  Inside the plane every operand comes from the lane itself or from its neighbour lane (a 1 KiB LDS ring), so the
// per-step latency is an LDS round trip; plane k runs a few steps behind plane k-1 and reads its results as an ordered, prefetched
// stream (position-ordered scratch [plane][step][lane]), and the factor values are a second ordered stream of 8 B per entry with no
// indices at all.  The L2 round trips are paid once per PLANE (109) instead of once per level (890).
//
// The rows behind the box are a triangular system of their own once the box is known: forward  L_ss x_s = d_s - L_sb y_b,  backward
// U_ss x_s = y_s (a shell row has no upper entry in the box: every box row comes before it).  They are solved by a nested factor object
// with the general engines (pipe).  Box rows have upper entries in the shell: their PRODUCTS are formed by a pre-pass once the shell is
// solved and subtracted last (ascending columns: shell columns are the largest), bit for bit what the sequential solve does.
//
// Order of a solve:  box forward -> shell right-hand side -> shell forward + backward (nested engine) -> products of the box rows'
// shell entries -> box backward.
//
// Stream layout (per block, per sweep, per plane K, per step l; mirrored coordinates I = nx-1-i, J = ny-1-j, K = nz-1-k in the
// backward sweep): active lines J in [jlo(l), jlo(l) + nact(l)), row I = l - 2 J.  Tile = NQ x nact(l) x 2 doubles,
// [q][J - jlo][2]: values 2q, 2q+1 of the row, so that one 16-byte load per lane and q is a contiguous kilobyte per wavefront.
//   lower: 13 factor entries in ascending column order, 1 pad;   upper: the 13 entries, then the stored inverse pivot
// Entries that do not exist at the faces of the box are stored as 0 (the kernel supplies 0 as their operand).
// einfo[K][l][lane] (backward sweep, lane = J mod 64): the row's slice of the product array, first index | count << 32; rows
// without shell entries point at the 32 zeros the product array starts with.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace box {

constexpr int LANES = 64;
constexpr int NV = 14;                 // doubles per row in either stream
constexpr int E_ZEROS = 32;            // leading zeros of the product array
constexpr int MAX_NX = 120;             // a lane serves lines J, J + 64, ...: line J must be finished before J + 64 starts its run-in
constexpr int MAX_EXT = 20;             // shell entries of a box row the kernel reads (10 two-double loads)
constexpr int PREFETCH = 2;             // steps the previous plane's results are requested ahead
constexpr int MAX_STEPS = 1024;         // steps per plane (the kernel keeps the step table in LDS)

struct Block {
  int32_t nx, ny, nz, nsteps;
  int64_t r0;           // first row of the block (rank-local)
  int64_t nb;           // rows of the box
  int64_t nloc;         // rows of the block
  int64_t s0;           // first shell row of the block in the concatenated shell numbering
  int64_t stream_off[2]; // doubles: start of the block's lower / upper stream
  int64_t plane_len[2];  // doubles per plane
  int64_t step_off;      // index of the block's step table (nsteps + 1 entries of StepTab)
  int64_t einfo_off;     // words: the block's einfo [nz][nsteps][64]
  int64_t e_begin;       // doubles: first product of the block
  int64_t xs_off;        // doubles: start of the block's hand-over scratch [nz][nsteps + 1][64]
  int64_t prog_off;      // words: progress words [2][nz]
};
struct StepTab { // per block and step (the same for every plane)
  int32_t jlo, nact;
  int32_t off;   // rows before this step inside a plane (sum of nact of the earlier steps)
  int32_t pad;
};
struct Stats {
  int64_t box_rows = 0, shell_rows = 0, stream_bytes = 0, ext_products = 0, shell_lower_entries = 0;
};
struct Schedule {
  std::vector<Block> blocks;
  std::vector<StepTab> steps;
  std::vector<double> stream;
  // products of the box rows' shell entries: E[pos] = ext_val[pos] * x_shell[ext_col[pos]] (ext_col < 0: 0.0)
  std::vector<double> ext_val;
  std::vector<int32_t> ext_col;
  std::vector<uint64_t> einfo;
  // right-hand side of the shell: d_s - L_sb y_b, one CSR row per shell row (columns: rank-local box rows, ascending)
  std::vector<int64_t> srp;
  std::vector<int32_t> sci;
  std::vector<double> sva;
  std::vector<int32_t> srow;     // rank-local row of every shell row
  // the shell's own factor (pattern + values in the convention of the whole factor: strict lower part, inverse pivot, strict upper part)
  std::vector<int64_t> frp, fdiag, fblock_ptr;
  std::vector<int32_t> fci;
  std::vector<double> fva;
  int64_t xs_len = 0, prog_len = 0;
  Stats stats;
  std::string error;
};

// (dI, dJ, dK) of entry e in MIRRORED coordinates, ascending column order of the true matrix
struct Nb { int dI, dJ, dK; };
inline Nb neighbour(bool upper, int e)
{
  if (!upper) {
    if (e < 9) return {e % 3 - 1, e / 3 - 1, -1};
    if (e < 12) return {e - 10, -1, 0};
    return {-1, 0, 0};
  }
  if (e == 0) return {-1, 0, 0};
  if (e < 4) return {2 - e, -1, 0};                       // true (i-1, i, i+1; j+1)  =  mirrored (I+1, I, I-1; J-1)
  return {1 - (e - 4) % 3, 1 - (e - 4) / 3, -1};          // true plane k+1, (j-1, j, j+1) x (i-1, i, i+1)
}

inline int steps_of(int nx, int ny) { return nx + 2 * (ny - 1); }
inline int jlo_of(int l, int nx) { return l < nx ? 0 : (l - nx + 2) / 2; }                 // smallest J with l - 2J <= nx - 1
inline int jhi_of(int l, int ny) { return std::min(ny - 1, l / 2); }

// Finds the box of one block: rows [r0, r0 + nb) with the exact 27-point pattern.  Returns false when there is none worth the engine.
inline bool detect(const int64_t *rp, const int32_t *ci, const int64_t *diag, int64_t r0, int64_t r1, int &nx, int &ny, int &nz, std::string &why)
{
  const int64_t nloc = r1 - r0;
  if (nloc < 64) { why = "block too small"; return false; }
  // first row: columns 0, 1, nx, nx+1, nx ny, ... (local)
  const int64_t p0 = rp[r0], p1 = rp[r0 + 1];
  if (p1 - p0 < 8 || ci[p0] != r0 || diag[r0] != p0 || ci[p0 + 1] != r0 + 1) { why = "first row is not the corner of a box"; return false; }
  const int64_t cx = ci[p0 + 2] - r0, cxy = ci[p0 + 4] - r0;
  if (cx < 2 || cx > MAX_NX || ci[p0 + 3] - r0 != cx + 1 || cxy < 2 * cx || cxy % cx || ci[p0 + 5] - r0 != cxy + 1 || ci[p0 + 6] - r0 != cxy + cx ||
      ci[p0 + 7] - r0 != cxy + cx + 1) {
    why = "first row does not have the pattern of a box corner (or nx > " + std::to_string(MAX_NX) + ")";
    return false;
  }
  nx = (int)cx;
  ny = (int)(cxy / cx);
  int64_t nzmax = nloc / cxy;
  // verify plane by plane; the box ends in front of the first plane with a row that does not conform
  int k = 0;
  for (; k < nzmax; ++k) {
    const int64_t nbk = (int64_t)(k + 1) * cxy; // rows of the candidate box: columns >= nbk of ITS rows may still be box rows of the next plane
    bool ok = true;
    for (int64_t q = (int64_t)k * cxy; q < nbk && ok; ++q) {
      const int i = (int)(q % nx), j = (int)((q / nx) % ny);
      int64_t p = rp[r0 + q];
      const int64_t pe = rp[r0 + q + 1];
      // lower part: planes k-1 and k
      for (int dk = -1; dk <= 0 && ok; ++dk)
        for (int dj = -1; dj <= 1 && ok; ++dj)
          for (int di = -1; di <= 1 && ok; ++di) {
            if (k + dk < 0 || j + dj < 0 || j + dj >= ny || i + di < 0 || i + di >= nx) continue;
            const int64_t c = q + di + (int64_t)nx * dj + cxy * dk;
            if (c > q) continue;
            if (p >= pe || ci[p] != r0 + c) ok = false;
            else if (c == q && diag[r0 + q] != p) ok = false;
            ++p;
          }
      // upper part inside plane k
      for (int dj = 0; dj <= 1 && ok; ++dj)
        for (int di = -1; di <= 1 && ok; ++di) {
          if (j + dj >= ny || i + di < 0 || i + di >= nx) continue;
          const int64_t c = q + di + (int64_t)nx * dj;
          if (c <= q) continue;
          if (p >= pe || ci[p] != r0 + c) ok = false;
          ++p;
        }
      // what follows is plane k+1 (checked when that plane is accepted: see below) or shell
      if (ok && p < pe && ci[p] < r0 + nbk) ok = false;
    }
    if (!ok) break;
  }
  // rows of the last accepted plane must not point into a plane that was rejected; rows of earlier planes must have exactly the nine
  // neighbours of the next plane
  for (;;) {
    if (k < 2) { why = "fewer than two conforming planes"; return false; }
    const int64_t nb = (int64_t)k * cxy;
    bool ok = true;
    for (int64_t q = 0; q < nb && ok; ++q) {
      const int i = (int)(q % nx), j = (int)((q / nx) % ny), kk = (int)(q / cxy);
      int64_t p = diag[r0 + q] + 1;
      const int64_t pe = rp[r0 + q + 1];
      while (p < pe && ci[p] < r0 + (int64_t)(kk + 1) * cxy) ++p; // in-plane upper part (verified above)
      if (kk + 1 < k) {
        for (int dj = -1; dj <= 1 && ok; ++dj)
          for (int di = -1; di <= 1 && ok; ++di) {
            if (j + dj < 0 || j + dj >= ny || i + di < 0 || i + di >= nx) continue;
            const int64_t c = q + di + (int64_t)nx * dj + cxy;
            if (p >= pe || ci[p] != r0 + c) ok = false;
            ++p;
          }
      }
      if (ok && p < pe && ci[p] < r0 + nb) ok = false;          // everything else: shell
      if (ok && pe - p > MAX_EXT) ok = false;
    }
    if (ok) break;
    --k;
  }
  nz = k;
  if ((int64_t)nz * cxy * 2 < nloc) { why = "the box covers less than half of the block"; return false; }
  return true;
}

// Builds the streams and the shell system for factor values `lu` stored in the pattern (rp, ci), diag[i] = position of the pivot.
inline bool build(int64_t n, const int64_t *rp, const int32_t *ci, const double *lu, const int64_t *diag, int nblocks, const int64_t *block_ptr, Schedule &S)
{
  S = Schedule();
  if (n >= INT32_MAX) { S.error = "too many rows"; return false; }
  S.blocks.resize((size_t)nblocks);
  int64_t soff = 0, eoff = E_ZEROS, stream_len = 0, einfo_len = 0;
  for (int b = 0; b < nblocks; ++b) {
    Block &B = S.blocks[(size_t)b];
    int nx = 0, ny = 0, nz = 0;
    std::string why;
    if (!detect(rp, ci, diag, block_ptr[b], block_ptr[b + 1], nx, ny, nz, why)) {
      S.error = "block " + std::to_string(b) + ": " + why;
      return false;
    }
    B.nx = nx; B.ny = ny; B.nz = nz; B.nsteps = steps_of(nx, ny);
    if (B.nsteps > MAX_STEPS) { S.error = "block " + std::to_string(b) + ": more than " + std::to_string(MAX_STEPS) + " steps per plane"; return false; }
    B.r0 = block_ptr[b];
    B.nb = (int64_t)nx * ny * nz;
    B.nloc = block_ptr[b + 1] - block_ptr[b];
    B.s0 = soff;
    soff += B.nloc - B.nb;
    B.step_off = (int64_t)S.steps.size();
    int32_t off = 0;
    for (int l = 0; l <= B.nsteps; ++l) {
      StepTab T{0, 0, off, 0};
      if (l < B.nsteps) {
        T.jlo = jlo_of(l, nx);
        T.nact = jhi_of(l, ny) - T.jlo + 1;
        if (T.nact > LANES) { S.error = "more than 64 active lines in a step"; return false; }
      }
      S.steps.push_back(T);
      off += T.nact;
    }
    B.plane_len[0] = B.plane_len[1] = (int64_t)nx * ny * NV;
    B.stream_off[0] = stream_len;
    stream_len += B.plane_len[0] * nz;
    B.stream_off[1] = stream_len;
    stream_len += B.plane_len[1] * nz;
    B.xs_off = S.xs_len;
    S.xs_len += (int64_t)nz * (B.nsteps + 1) * LANES;      // one spare slot per plane (stores of steps without rows)
    B.prog_off = S.prog_len;
    S.prog_len += 2 * (int64_t)nz;
    // products: every box row's shell entries, padded to an even count
    B.einfo_off = einfo_len;
    einfo_len += (int64_t)nz * B.nsteps * LANES;
    B.e_begin = eoff;
    for (int64_t q = 0; q < B.nb; ++q) {
      const int64_t r = B.r0 + q;
      int64_t p = rp[r + 1];
      while (p > diag[r] + 1 && ci[p - 1] >= B.r0 + B.nb) --p;
      const int64_t cnt = rp[r + 1] - p;
      eoff += (cnt + 1) / 2 * 2;
    }
  }
  const int64_t nshell = soff;
  S.stream.assign((size_t)stream_len, 0.0);
  S.ext_val.assign((size_t)eoff, 0.0);
  S.ext_col.assign((size_t)eoff, -1);
  S.einfo.assign((size_t)einfo_len + 2, 0);   // (index 0, count 0: the zeros; two spare words: the kernel fetches 16 bytes per lane)
  S.stats.ext_products = eoff;
  S.stats.shell_rows = nshell;
  S.stats.stream_bytes = stream_len * 8;
  // ---- streams (one host thread per block) ----
  {
    std::vector<std::thread> th;
    for (int b = 0; b < nblocks; ++b)
      th.emplace_back([&, b]() {
        const Block &B = S.blocks[(size_t)b];
        const int nx = B.nx, ny = B.ny, nz = B.nz;
        const StepTab *T = S.steps.data() + B.step_off;
        int64_t ecur = B.e_begin;
        // the product slices are numbered in the order the BACKWARD sweep visits the rows (plane K, step l, line J): neighbouring
        // rows of a step are neighbours in the array
        for (int sweep = 0; sweep < 2; ++sweep) {
          const bool upper = sweep == 1;
          const int NQ = NV / 2;
          for (int K = 0; K < nz; ++K) {
            double *plane = S.stream.data() + B.stream_off[sweep] + (int64_t)K * B.plane_len[sweep];
            for (int l = 0; l < B.nsteps; ++l) {
              double *tile = plane + (int64_t)T[l].off * NV;
              const int nact = T[l].nact;
              for (int a = 0; a < nact; ++a) {
                const int J = T[l].jlo + a, I = l - 2 * J;
                const int i = upper ? nx - 1 - I : I, j = upper ? ny - 1 - J : J, k = upper ? nz - 1 - K : K;
                const int64_t q = i + (int64_t)nx * (j + (int64_t)ny * k), r = B.r0 + q;
                double v[NV];
                for (int e = 0; e < NV; ++e) v[e] = 0.0;
                // the entries e = 0 .. 12 are the row's in-box neighbours in ascending column order (verified by detect): they are
                // consecutive in the row, from its first entry (lower part) or from behind the pivot (upper part)
                int64_t p = upper ? diag[r] + 1 : rp[r];
                for (int e = 0; e < 13; ++e) {
                  const Nb d = neighbour(upper, e);
                  const int ii = upper ? i - d.dI : i + d.dI, jj = upper ? j - d.dJ : j + d.dJ, kk = upper ? k - d.dK : k + d.dK;
                  if (ii < 0 || ii >= nx || jj < 0 || jj >= ny || kk < 0 || kk >= nz) continue;
                  v[e] = lu[p++];
                }
                if (upper) {
                  v[13] = lu[diag[r]];
                  int64_t p = rp[r + 1];
                  while (p > diag[r] + 1 && ci[p - 1] >= B.r0 + B.nb) --p;
                  const int64_t cnt = rp[r + 1] - p;
                  for (int64_t m = 0; m < cnt; ++m) {
                    S.ext_val[(size_t)(ecur + m)] = lu[p + m];
                    S.ext_col[(size_t)(ecur + m)] = (int32_t)(B.s0 + (ci[p + m] - (B.r0 + B.nb)));
                  }
                  if (cnt > 0) S.einfo[(size_t)(B.einfo_off + ((int64_t)K * B.nsteps + l) * LANES + J % LANES)] = (uint64_t)(uint32_t)ecur | ((uint64_t)cnt << 32);
                  ecur += (cnt + 1) / 2 * 2;
                }
                for (int qq = 0; qq < NQ; ++qq) {
                  tile[((int64_t)qq * nact + a) * 2] = v[2 * qq];
                  tile[((int64_t)qq * nact + a) * 2 + 1] = v[2 * qq + 1];
                }
              }
            }
          }
        }
      });
    for (auto &t : th) t.join();
  }
  // ---- shell: right-hand side rows and the shell's own factor ----
  S.srp.assign(1, 0);
  S.frp.assign(1, 0);
  S.fblock_ptr.assign(1, 0);
  S.srow.reserve((size_t)nshell);
  for (int b = 0; b < nblocks; ++b) {
    const Block &B = S.blocks[(size_t)b];
    for (int64_t r = B.r0 + B.nb; r < B.r0 + B.nloc; ++r) {
      S.srow.push_back((int32_t)r);
      bool have_diag = false;
      for (int64_t p = rp[r]; p < rp[r + 1]; ++p) {
        const int64_t c = ci[p];
        if (c < B.r0 + B.nb) {
          if (p > diag[r]) { S.error = "shell row with an upper entry in the box"; return false; }
          S.sci.push_back((int32_t)c);
          S.sva.push_back(lu[p]);
        } else {
          if (p == diag[r]) { S.fdiag.push_back((int64_t)S.fci.size()); have_diag = true; }
          S.fci.push_back((int32_t)(B.s0 + (c - (B.r0 + B.nb))));
          S.fva.push_back(lu[p]);
        }
      }
      if (!have_diag) { S.error = "shell row without pivot"; return false; }
      S.srp.push_back((int64_t)S.sci.size());
      S.frp.push_back((int64_t)S.fci.size());
    }
    S.fblock_ptr.push_back(B.s0 + (B.nloc - B.nb));
    S.stats.box_rows += B.nb;
  }
  S.stats.shell_lower_entries = (int64_t)S.sci.size();
  return true;
}

// CPU walk of the streams in the order of the device kernels (planes, steps, lines), operands by geometric neighbour: checks the
// packing and the accumulation order against the sequential solve.  x: n doubles (the solution), d: right-hand side.
// shell_solve(ds, xs): the nested solver (callers pass a sequential solve of the shell factor).
template <class ShellSolve>
inline std::string emulate(const Schedule &S, int64_t n, const double *d, double *x, ShellSolve &&shell_solve, double *y_out = nullptr)
{
  std::vector<double> y((size_t)n, 0.0);
  auto sweep_box = [&](const Block &B, bool upper, const double *rhs, double *out, const std::vector<double> &E) {
    const int nx = B.nx, ny = B.ny, nz = B.nz;
    const StepTab *T = S.steps.data() + B.step_off;
    const int NQ = NV / 2;
    for (int K = 0; K < nz; ++K) {
      const double *plane = S.stream.data() + B.stream_off[upper] + (int64_t)K * B.plane_len[upper];
      for (int l = 0; l < B.nsteps; ++l) {
        const double *tile = plane + (int64_t)T[l].off * NV;
        const int nact = T[l].nact;
        for (int a = 0; a < nact; ++a) {
          const int J = T[l].jlo + a, I = l - 2 * J;
          const int i = upper ? nx - 1 - I : I, j = upper ? ny - 1 - J : J, k = upper ? nz - 1 - K : K;
          const int64_t r = B.r0 + i + (int64_t)nx * (j + (int64_t)ny * k);
          double v[NV];
          for (int qq = 0; qq < NQ; ++qq) {
            v[2 * qq] = tile[((int64_t)qq * nact + a) * 2];
            v[2 * qq + 1] = tile[((int64_t)qq * nact + a) * 2 + 1];
          }
          double s = rhs[r];
          for (int e = 0; e < 13; ++e) {
            const Nb dd = neighbour(upper, e);
            const int ii = upper ? i - dd.dI : i + dd.dI, jj = upper ? j - dd.dJ : j + dd.dJ, kk = upper ? k - dd.dK : k + dd.dK;
            double w = 0.0;
            if (!(ii < 0 || ii >= nx || jj < 0 || jj >= ny || kk < 0 || kk >= nz)) w = out[B.r0 + ii + (int64_t)nx * (jj + (int64_t)ny * kk)];
            const double prod = v[e] * w;
            s -= prod;
          }
          if (upper) {
            const uint64_t info = S.einfo[(size_t)(B.einfo_off + ((int64_t)K * B.nsteps + l) * LANES + J % LANES)];
            const int64_t e0 = (int64_t)(uint32_t)info, cnt = (int64_t)(info >> 32);
            for (int64_t m = 0; m < (cnt + 1) / 2 * 2; ++m) s -= E[(size_t)(e0 + m)];
            s *= v[13];
          }
          out[r] = s;
        }
      }
    }
  };
  std::vector<double> E;
  for (const Block &B : S.blocks) sweep_box(B, false, d, y.data(), E);
  if (y_out) std::copy(y.begin(), y.end(), y_out);
  const int64_t ns = (int64_t)S.srow.size();
  std::vector<double> ds((size_t)ns), xs((size_t)ns);
  for (int64_t t = 0; t < ns; ++t) {
    double s = d[S.srow[(size_t)t]];
    for (int64_t p = S.srp[(size_t)t]; p < S.srp[(size_t)t + 1]; ++p) {
      const double prod = S.sva[(size_t)p] * y[(size_t)S.sci[(size_t)p]];
      s -= prod;
    }
    ds[(size_t)t] = s;
  }
  shell_solve(ds.data(), xs.data());
  E.resize(S.ext_val.size());
  for (size_t p = 0; p < E.size(); ++p) E[p] = S.ext_col[p] < 0 ? 0.0 : S.ext_val[p] * xs[(size_t)S.ext_col[p]];
  for (int64_t t = 0; t < ns; ++t) x[S.srow[(size_t)t]] = xs[(size_t)t];
  for (const Block &B : S.blocks) {
    // the backward sweep reads the forward result of a row and writes its solution in place
    for (int64_t r = B.r0; r < B.r0 + B.nb; ++r) x[r] = y[(size_t)r];
    sweep_box(B, true, x, x, E);
  }
  return std::string();
}

// The same solve, walked LANE BY LANE as the device kernel does it (trsv_box.hpp: lane = J mod 64, register windows for the previous
// plane's lines, two-slot neighbour ring, position-ordered hand-over xs[K][l][lane]; planes one after the other): checks the kernel's
// index arithmetic on the host.
template <class ShellSolve>
inline std::string emulate_lanes(const Schedule &S, int64_t n, const double *d, double *x, ShellSolve &&shell_solve, double *y_out = nullptr)
{
  std::vector<double> xs((size_t)S.xs_len, 0.0), E;
  auto sweep = [&](const Block &B, bool upper, const double *rhs, double *out) {
    const int nx = B.nx, ny = B.ny, nz = B.nz, nsteps = B.nsteps;
    const StepTab *T = S.steps.data() + B.step_off;
    auto line_of = [&](int lane, int s, int &J, int &I) {
      const int t = s + 3 - 2 * lane;
      const int c = t >> 7;
      J = lane + 64 * c;
      I = (t >= 0 && J < ny) ? (t & 127) - 3 : -1000;
    };
    for (int K = 0; K < nz; ++K) {
      const double *plane = S.stream.data() + B.stream_off[upper] + (int64_t)K * B.plane_len[upper];
      double *xsK = xs.data() + B.xs_off + (int64_t)K * (nsteps + 1) * 64;
      const double *xsP = K > 0 ? xsK - (int64_t)(nsteps + 1) * 64 : xsK;
      const int ktrue = upper ? nz - 1 - K : K;
      double wA[64][3] = {}, wB[64][3] = {}, wC[64][3] = {}, u[64][3] = {}, xprev[64] = {}, ring[2][64] = {};
      for (int l = -1; l < nsteps; ++l) {   // step -1: the run-in of line 0 (its column 0 of the previous plane's lines)
        double xnew[64];
        for (int lane = 0; lane < 64; ++lane) {
          int J, I;
          line_of(lane, l, J, I);
          const bool act = I >= 0 && I < nx, colok = I + 1 >= 0 && I + 1 < nx, prev = K > 0 && colok;
          const int l0 = std::min(std::max(l - 1, 0), nsteps - 1), l1 = std::min(l + 1, nsteps - 1), l2 = std::min(l + 3, nsteps - 1);
          const double r0 = xsP[(int64_t)l0 * 64 + ((lane + 63) & 63)], r1 = xsP[(int64_t)l1 * 64 + lane], r2 = xsP[(int64_t)l2 * 64 + ((lane + 1) & 63)];
          wA[lane][0] = wA[lane][1]; wA[lane][1] = wA[lane][2]; wA[lane][2] = (prev && J >= 1) ? r0 : 0.0;
          wB[lane][0] = wB[lane][1]; wB[lane][1] = wB[lane][2]; wB[lane][2] = prev ? r1 : 0.0;
          wC[lane][0] = wC[lane][1]; wC[lane][1] = wC[lane][2]; wC[lane][2] = (prev && J + 1 < ny) ? r2 : 0.0;
          const double nbv = ring[(l + 1) & 1][(lane + 63) & 63];
          u[lane][0] = u[lane][1]; u[lane][1] = u[lane][2]; u[lane][2] = (colok && J >= 1) ? nbv : 0.0;
          if (I <= 0) xprev[lane] = 0.0;
          double t[NV];
          for (int e = 0; e < NV; ++e) t[e] = 0.0;
          double v = 0.0;
          int64_t row = -1;
          if (act) {
            const int a = J - T[l].jlo, nact = T[l].nact;   // (act implies l >= 0)
            const double *tile = plane + (int64_t)T[l].off * NV;
            for (int q = 0; q < NV / 2; ++q) {
              t[2 * q] = tile[((int64_t)q * nact + a) * 2];
              t[2 * q + 1] = tile[((int64_t)q * nact + a) * 2 + 1];
            }
            const int i = upper ? nx - 1 - I : I, j = upper ? ny - 1 - J : J;
            row = B.r0 + i + (int64_t)nx * (j + (int64_t)ny * ktrue);
            v = rhs[row];
          }
          auto sub = [&](double a, double w) { const double p = a * w; v -= p; };
          if (!upper) {
            sub(t[0], wA[lane][0]); sub(t[1], wA[lane][1]); sub(t[2], wA[lane][2]);
            sub(t[3], wB[lane][0]); sub(t[4], wB[lane][1]); sub(t[5], wB[lane][2]);
            sub(t[6], wC[lane][0]); sub(t[7], wC[lane][1]); sub(t[8], wC[lane][2]);
            sub(t[9], u[lane][0]); sub(t[10], u[lane][1]); sub(t[11], u[lane][2]);
            sub(t[12], xprev[lane]);
          } else {
            sub(t[0], xprev[lane]);
            sub(t[1], u[lane][2]); sub(t[2], u[lane][1]); sub(t[3], u[lane][0]);
            sub(t[4], wC[lane][2]); sub(t[5], wC[lane][1]); sub(t[6], wC[lane][0]);
            sub(t[7], wB[lane][2]); sub(t[8], wB[lane][1]); sub(t[9], wB[lane][0]);
            sub(t[10], wA[lane][2]); sub(t[11], wA[lane][1]); sub(t[12], wA[lane][0]);
            if (act) {
              const uint64_t info = S.einfo[(size_t)(B.einfo_off + ((int64_t)K * nsteps + l) * LANES + lane)];
              const uint32_t ptr = (uint32_t)info, cnt = (uint32_t)(info >> 32);
              for (int q = 0; q < MAX_EXT / 2; ++q) {
                const uint32_t at = (uint32_t)(2 * q) < cnt ? ptr + 2 * q : 0u;
                v -= E[at];
                v -= E[at + 1];
              }
            }
            v *= t[13];
          }
          xnew[lane] = act ? v : 0.0;
          if (act) out[row] = xnew[lane];
        }
        for (int lane = 0; lane < 64; ++lane) {
          xprev[lane] = xnew[lane];
          ring[l & 1][lane] = xnew[lane];
          if (l >= 0) xsK[(int64_t)l * 64 + lane] = xnew[lane];
        }
      }
    }
  };
  std::vector<double> y((size_t)n, 0.0);
  for (const Block &B : S.blocks) sweep(B, false, d, y.data());
  if (y_out) std::copy(y.begin(), y.end(), y_out);
  const int64_t ns = (int64_t)S.srow.size();
  std::vector<double> ds((size_t)ns), xsol((size_t)ns);
  for (int64_t t = 0; t < ns; ++t) {
    double s = d[S.srow[(size_t)t]];
    for (int64_t p = S.srp[(size_t)t]; p < S.srp[(size_t)t + 1]; ++p) {
      const double prod = S.sva[(size_t)p] * y[(size_t)S.sci[(size_t)p]];
      s -= prod;
    }
    ds[(size_t)t] = s;
  }
  shell_solve(ds.data(), xsol.data());
  E.resize(S.ext_val.size());
  for (size_t p = 0; p < E.size(); ++p) E[p] = S.ext_col[p] < 0 ? 0.0 : S.ext_val[p] * xsol[(size_t)S.ext_col[p]];
  for (int64_t t = 0; t < ns; ++t) x[S.srow[(size_t)t]] = xsol[(size_t)t];
  for (const Block &B : S.blocks) {
    for (int64_t r = B.r0; r < B.r0 + B.nb; ++r) x[r] = y[(size_t)r];
    sweep(B, true, x, x);
  }
  return std::string();
}

} // namespace box
