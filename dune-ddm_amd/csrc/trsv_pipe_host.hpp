// Host-side schedule builder of the "pipe" triangular-solve engine (ILU(0) back-solve of
// SchwarzPreconditioner::apply, dune/ddm/schwarz.hh:133; dune-istl SeqILU semantics: natural row
// order, unit lower factor, inverse pivots on the diagonal, row sums in ascending column order).
//
// Idea.  The level-barrier engines pay several dependent L2 round trips per dependency level.  Here the
// rows of a subdomain are cut into CHAINS (row -> the dependent one level later that sits nearest to the
// diagonal; on a structured grid a chain is a grid line), 64 chains form a TASK that ONE wavefront walks
// level by level (lane = chain, step = level), so that
//   * a dependency inside the task is read from a small LDS ring of the wave's own recent results,
//   * a dependency on another task is read from global memory; the producer task only has to be AHEAD,
//     not in lock-step: tasks form a DAG, are queued in topological order and publish a monotone
//     "steps stored" word, so the L2 latency hides behind the natural lag of the consumer,
//   * everything static a step needs (factor entries, operand addresses, inverse pivots, progress
//     requirements) is one fixed-size TILE; the tiles of a task are contiguous in memory in processing
//     order, so a loader wave streams them HBM -> LDS far ahead of the compute wave.
//
// This header is plain C++17 (no HIP): it is compiled into libddm_hip.so and, for the CPU tests of the
// schedule logic, into a host-only test library together with emulate() below.
#pragma once
#include "host_vec.hpp"
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#if defined(__HIPCC__)
#define PIPE_HD __host__ __device__
#else
#define PIPE_HD
#endif

namespace pipe {

constexpr int LANES = 64;
constexpr int RING = 16;       // steps of a wave's own results kept in LDS
// Operand encoding (uint32 per factor entry), Z = RING_Z: op < Z: BYTE offset of the operand in the wave's LDS ring
// (RING rows of 64 results at [0, Z), a row of zeros at [Z, Z + 512)); op > Z: BYTE offset of the operand in the sweep's
// result array (position * 8; the first ZERO_POS + LANES positions of each array are reserved and hold 0.0).  The
// kernel loads BOTH sources unconditionally, LDS at min(op, Z) and global at max(op, Z), and ORs the bit patterns:
// one of the two is always +0.0.  Padding entries have op == Z.
constexpr int RING_Z = RING * LANES * 8;
constexpr int RING_BYTES = RING_Z + LANES * 8;
constexpr int ZERO_POS = RING_Z / 8;            // position of the zero that local operands read from global memory
constexpr int FIRST_POS = ZERO_POS + LANES;     // first position of a task
constexpr int MIN_W = 14;                       // tiles hold at least this many entries per row (the kernel's register chunk)
inline int32_t ring_op(int step, int lane) { return (int32_t)(((step % RING) * LANES + lane) * 8); }
constexpr int32_t PAD_OP = RING_Z;
constexpr int MAXPROD = 112;   // producer tasks per task: two 16-bit step counts per header word HDR_REQ0 .. 63
constexpr int MAX_NC = 3;      // the late-operand masks below are provided for up to MAX_NC compute waves per task
constexpr int HDR_REQ0 = 8;
constexpr int MAX_W = 26;      // widest triangular row taken (27-point stencil: all neighbours on one side); the kernel's LDS
                               // tile ring must hold 3 such tiles (21 KiB each) or loaders and compute wave could wait for each other

struct Task { // device-visible descriptor, 512 bytes
  int64_t tile_off;  // byte offset of the first tile in the stream
  int64_t pos_base;  // first position of the task in the position space of its sweep
  int64_t koff_base; // index of the task's nsteps + 1 cumulative tile offsets (KiB, relative to tile_off) in Schedule::koff
  int32_t nsteps, nprod, group, sweep;
  int32_t W, first_kib, second_kib; // widest row of the task; sizes of its first two tiles in KiB
  int32_t start_level, pad[2];      // dependency level of the task's first step (diagnostics)
  int32_t prod[MAXPROD]; // global task ids of the producers
};
static_assert(sizeof(Task) == 512, "Task layout");
struct Group {
  int32_t task0[2], ntask[2]; // [0] forward (L) sweep, [1] backward (U) sweep; tasks in queue order
};

// Tile layout (bytes): [0,256) header int32[64]: [0] active rows, [1] flags, [2] W = widest row of the step,
// [3] size of the task's NEXT tile in KiB (0 behind the last one), [5] of the tile after that, [4] / [6]: bit u set =
// entry u of some lane is a ring operand produced 1 step / 1 or 2 steps earlier (what a compute wave that shares the task
// with 1 / 2 other waves must read AFTER the previous step has signalled),
// word HDR_REQ0 + p/2, 16-bit half p%2: steps of producer p that must be stored; [256,512) int32[64]: BYTE offset of the row's right-hand side -- U: in the forward results (position order), L: in the caller's vector (natural row * 8); [512,1024) double[64]:
// U: inverse pivot; then idx pieces (1 KiB each: lane l holds int32[4] = operands 4q..4q+3), then value pieces
// (1 KiB each: lane l holds double[2] = entries 2q, 2q+1).  Operand encoding: see above; the U tile's "own" word is
// the BYTE offset of the row's forward value.  W is at least MIN_W (narrower rows are padded).  Tiles are sized by the step's own W and packed back to back.
struct Geometry {
  int W = 0, idx_pieces = 0, val_pieces = 0, tile_bytes = 0;
  constexpr Geometry() = default;
  PIPE_HD constexpr explicit Geometry(int w_) : W(w_ > MIN_W ? w_ : MIN_W), idx_pieces((W + 3) / 4), val_pieces((W + 1) / 2), tile_bytes(1024 * (1 + idx_pieces + val_pieces)) {}
  // Piece order inside a tile: header | value pieces 0..6 | operand pieces 0..3 | further value pieces | further operand
  // pieces -- the first MIN_W entries of every row sit at offsets that do not depend on W (one batch of LDS reads).
  static constexpr int FIXED_VAL = MIN_W / 2, FIXED_IDX = (MIN_W + 3) / 4, FIXED_KIB = 1 + FIXED_VAL + FIXED_IDX;
  PIPE_HD int val_piece(int q2) const { return q2 < FIXED_VAL ? 1 + q2 : FIXED_KIB + (q2 - FIXED_VAL); }
  PIPE_HD int idx_piece(int q4) const { return q4 < FIXED_IDX ? 1 + FIXED_VAL + q4 : FIXED_KIB + (val_pieces - FIXED_VAL) + (q4 - FIXED_IDX); }
  PIPE_HD int val_off(int u, int lane) const { return 1024 * val_piece(u >> 1) + lane * 16 + (u & 1) * 8; } // byte offset of entry u of a lane
  PIPE_HD int idx_off(int u, int lane) const { return 1024 * idx_piece(u >> 2) + lane * 16 + (u & 3) * 4; }
};

struct Options {
  int delta = 24;   // chains are queued by (start level / delta, head row): tasks are blocks of neighbouring chains of one band
  int vote = 1;     // 1: lanes are re-used by later chains; 0: one chain per lane
  int pack_steps = 64; // steps of a task of dependency-free rows
  int max_span = 256;  // longest task (steps); measured at 216^3: delta 16/32/48 -> 4.59/4.47/4.41 ms per solve, stream 1.78x/1.89x/1.98x
};

struct Stats {
  int64_t ntasks[2] = {0, 0}, nsteps[2] = {0, 0};
  int64_t rows = 0, entries = 0, entries_local = 0, entries_self_global = 0, entries_remote = 0;
  int64_t max_prod = 0, max_steps = 0, regrouped = 0, nchains[2] = {0, 0};
  int64_t max_levels = 0;           // deepest sweep of any subdomain
  double max_rows_per_level = 0.0;  // largest average parallelism of a sweep (rows / levels)
};

struct Schedule {
  Geometry geo;
  std::vector<Group> groups;
  std::vector<Task> tasks;
  hvec<unsigned char> stream; // (uninitialised on resize; the builder zero-fills it on several threads)
  std::vector<int32_t> koff;  // per task nsteps + 1 cumulative tile offsets in KiB
  std::vector<int32_t> rowL; // per L position: natural row or -1
  std::vector<int32_t> posU; // per natural row: U position
  int64_t nposL = 0, nposU = 0;
  Stats stats;
  std::string error;
};

namespace detail {

struct BlockSweep { // tasks of one block and one sweep, block-local numbering
  struct TaskRec {
    int32_t start0 = 0, nsteps = 0, W = 1;
    std::vector<int32_t> stepW;  // widest row per step
    std::vector<int32_t> rows; // nsteps * LANES block-local rows or -1
    std::vector<int32_t> prod; // producer tasks (local ids, final queue numbering)
  };
  std::vector<TaskRec> tasks;      // in queue (topological) order
  std::vector<int32_t> task_of, step_of, lane_of; // per block-local row
  int64_t nchains = 0;
  bool cyclic = false, too_many_prod = false;
  int dissolve_rounds = 0;
  int32_t nlev = 0; // dependency levels of the sweep
};

// deps(i, f): calls f(j) for every dependency j of block-local row i in ascending column order
//
// 1. level of every row; CHAINS: row i continues the chain of its dependency nearest to the diagonal when that one
//    sits exactly one level below and has no successor yet (structured grid, lexicographic numbering: a grid line).
// 2. rows without dependencies and without successor are packed into tasks of their own (no producers).
// 3. chain graph -> strongly connected components (mutually dependent chains must share a task) -> topological
//    order of the components, smallest (start level / delta, head row) first.
// 4. tasks = consecutive components of that order (=> the task graph is acyclic and the creation order is a valid
//    queue order); lanes are re-used by later chains of the task once a chain has ended.
template <class Deps>
static void build_block_sweep(int64_t nb, bool upper, const Deps &deps, const Options &opt, BlockSweep &B)
{
  std::vector<int32_t> lev(nb, 0), chain(nb, -1), coff(nb, 0), ndep(nb, 0);
  std::vector<uint8_t> claimed(nb, 0), nochain(nb, 0);
  struct Chain {
    int32_t head, start, len;
  };
  std::vector<Chain> chains;
  std::vector<int64_t> cptr, pptr;
  std::vector<int32_t> crow(nb), pidx, freerows, comp;
  std::vector<uint8_t> is_free;
  int64_t nc = 0;
  int32_t ncomp = 0;
  auto sweep_key = [&](int32_t row) { return upper ? (int32_t)(nb - 1 - row) : row; };
  for (;;) {
    chains.clear();
    chains.reserve(nb / 16 + 16);
    std::fill(claimed.begin(), claimed.end(), 0);
    for (int64_t q = 0; q < nb; ++q) {
      const int64_t i = upper ? nb - 1 - q : q;
      int32_t l = 0, nd = 0;
      int64_t nearest = -1;
      deps(i, [&](int64_t j) {
        l = std::max(l, lev[j] + 1);
        ++nd;
        if (!upper || nearest < 0) nearest = j; // L: last entry, U: first entry of the row
      });
      lev[i] = l;
      ndep[i] = nd;
      if (nearest >= 0 && lev[nearest] == l - 1 && !claimed[nearest] && !nochain[i] && !nochain[nearest]) {
        claimed[nearest] = 1;
        chain[i] = chain[nearest];
        coff[i] = coff[nearest] + 1;
        chains[chain[i]].len++;
      } else {
        chain[i] = (int32_t)chains.size();
        coff[i] = 0;
        chains.push_back(Chain{(int32_t)i, l, 1});
      }
    }
    nc = (int64_t)chains.size();
    cptr.assign(nc + 1, 0);
    for (int64_t c = 0; c < nc; ++c) cptr[c + 1] = cptr[c] + chains[c].len;
    for (int64_t i = 0; i < nb; ++i) crow[cptr[chain[i]] + coff[i]] = (int32_t)i;
    // free rows
    freerows.clear();
    is_free.assign(nc, 0);
    for (int64_t c = 0; c < nc; ++c)
      if (chains[c].len == 1 && ndep[chains[c].head] == 0) {
        is_free[c] = 1;
        freerows.push_back(chains[c].head);
      }
    if (upper) std::reverse(freerows.begin(), freerows.end());
    // chain graph (edges producer -> consumer between non-free chains), CSR by consumer = list of producers
    pptr.assign(nc + 1, 0);
    pidx.clear();
    {
      std::vector<int32_t> mark(nc, -1);
      for (int64_t c = 0; c < nc; ++c) {
        pptr[c] = (int64_t)pidx.size();
        if (is_free[c]) continue;
        for (int64_t k = cptr[c]; k < cptr[c + 1]; ++k)
          deps(crow[k], [&](int64_t j) {
            const int32_t cj = chain[j];
            if (cj != c && !is_free[cj] && mark[cj] != (int32_t)c) {
              mark[cj] = (int32_t)c;
              pidx.push_back(cj);
            }
          });
      }
      pptr[nc] = (int64_t)pidx.size();
    }
    // strongly connected components (iterative Tarjan over the producer lists)
    comp.assign(nc, -1);
    ncomp = 0;
    {
      std::vector<int32_t> index(nc, -1), low(nc, 0), stack, callstack;
      std::vector<int64_t> it(nc, 0);
      std::vector<uint8_t> onstack(nc, 0);
      int32_t counter = 0;
      for (int64_t root = 0; root < nc; ++root) {
        if (is_free[root] || index[root] >= 0) continue;
        callstack.push_back((int32_t)root);
        while (!callstack.empty()) {
          const int32_t v = callstack.back();
          if (index[v] < 0) {
            index[v] = low[v] = counter++;
            stack.push_back(v);
            onstack[v] = 1;
            it[v] = pptr[v];
          }
          bool descended = false;
          while (it[v] < pptr[v + 1]) {
            const int32_t w = pidx[it[v]++];
            if (index[w] < 0) {
              callstack.push_back(w);
              descended = true;
              break;
            }
            if (onstack[w]) low[v] = std::min(low[v], index[w]);
          }
          if (descended) continue;
          if (low[v] == index[v]) {
            for (;;) {
              const int32_t w = stack.back();
              stack.pop_back();
              onstack[w] = 0;
              comp[w] = ncomp;
              if (w == v) break;
            }
            ++ncomp;
          }
          callstack.pop_back();
          if (!callstack.empty()) low[callstack.back()] = std::min(low[callstack.back()], low[v]);
        }
      }
    }
    // more mutually dependent chains than a wave has lanes: dissolve those chains into single rows and repeat
    // (single rows alone form the row dependency graph, which is acyclic, so this terminates)
    std::vector<int32_t> csize(ncomp, 0);
    for (int64_t c = 0; c < nc; ++c)
      if (comp[c] >= 0) csize[comp[c]]++;
    bool big = false;
    for (int64_t c = 0; c < nc; ++c)
      if (comp[c] >= 0 && csize[comp[c]] > LANES) {
        big = true;
        for (int64_t k = cptr[c]; k < cptr[c + 1]; ++k) nochain[crow[k]] = 1;
      }
    if (!big) break;
    ++B.dissolve_rounds;
  }
  B.nchains = nc;
  for (int64_t i = 0; i < nb; ++i) B.nlev = std::max(B.nlev, lev[i] + 1);
  // components: members, key, condensation in-degrees
  std::vector<int64_t> mptr(ncomp + 1, 0);
  for (int64_t c = 0; c < nc; ++c)
    if (comp[c] >= 0) mptr[comp[c] + 1]++;
  for (int32_t k = 0; k < ncomp; ++k) {
    if (mptr[k + 1] > LANES) B.cyclic = true; // more mutually dependent chains than a wave has lanes
    mptr[k + 1] += mptr[k];
  }
  if (B.cyclic) return;
  std::vector<int32_t> members(mptr[ncomp]);
  {
    std::vector<int64_t> fill(mptr.begin(), mptr.end() - 1);
    for (int64_t c = 0; c < nc; ++c)
      if (comp[c] >= 0) members[fill[comp[c]]++] = (int32_t)c;
  }
  const int32_t band_w = std::max(1, opt.delta);
  using Key = std::pair<std::pair<int32_t, int32_t>, int32_t>; // ((band, head key), component)
  std::vector<Key> ckey(ncomp);
  for (int32_t k = 0; k < ncomp; ++k) {
    int32_t smin = 0x7fffffff, hmin = 0x7fffffff;
    for (int64_t m = mptr[k]; m < mptr[k + 1]; ++m) {
      smin = std::min(smin, chains[members[m]].start);
      hmin = std::min(hmin, sweep_key(chains[members[m]].head));
    }
    ckey[k] = {{smin / band_w, hmin}, k};
  }
  std::vector<int32_t> indeg(ncomp, 0);
  std::vector<int64_t> sptr(ncomp + 1, 0); // consumers of a component (with multiplicity removed per chain edge, not per component)
  std::vector<int32_t> sidx;
  {
    for (int64_t c = 0; c < nc; ++c)
      if (comp[c] >= 0)
        for (int64_t k = pptr[c]; k < pptr[c + 1]; ++k)
          if (comp[pidx[k]] != comp[c]) sptr[comp[pidx[k]] + 1]++;
    for (int32_t k = 0; k < ncomp; ++k) sptr[k + 1] += sptr[k];
    sidx.resize(sptr[ncomp]);
    std::vector<int64_t> fill(sptr.begin(), sptr.end() - 1);
    for (int64_t c = 0; c < nc; ++c)
      if (comp[c] >= 0)
        for (int64_t k = pptr[c]; k < pptr[c + 1]; ++k)
          if (comp[pidx[k]] != comp[c]) {
            sidx[fill[comp[pidx[k]]]++] = comp[c];
            indeg[comp[c]]++;
          }
  }
  std::priority_queue<Key, std::vector<Key>, std::greater<Key>> ready;
  for (int32_t k = 0; k < ncomp; ++k)
    if (!indeg[k]) ready.push(ckey[k]);
  // tasks = consecutive components of the topological order
  struct Proto {
    int32_t start0 = 0x7fffffff, end = 0;
    std::vector<int32_t> chains, lanes;
  };
  std::vector<Proto> protos;
  int32_t lane_end[LANES];
  int nlanes = 0;
  auto new_task = [&]() {
    protos.emplace_back();
    nlanes = 0;
  };
  auto place = [&](int64_t m0, int64_t m1, bool commit) { // members sorted by start; returns false if they do not fit
    int32_t le[LANES];
    int nl = nlanes;
    std::copy(lane_end, lane_end + nlanes, le);
    Proto *P = commit ? &protos.back() : nullptr;
    for (int64_t m = m0; m < m1; ++m) {
      const Chain &C = chains[members[m]];
      int best = -1;
      for (int l = 0; l < nl; ++l)
        if (le[l] <= C.start && (best < 0 || le[l] > le[best])) best = l;
      if (best < 0) {
        if (nl == LANES) return false;
        best = nl++;
      }
      le[best] = C.start + C.len;
      if (commit) {
        P->chains.push_back(members[m]);
        P->lanes.push_back(best);
        P->start0 = std::min(P->start0, C.start);
        P->end = std::max(P->end, C.start + C.len);
      }
    }
    if (commit) {
      nlanes = nl;
      std::copy(le, le + nl, lane_end);
    }
    return true;
  };
  int64_t emitted = 0;
  while (!ready.empty()) {
    const int32_t k = ready.top().second;
    ready.pop();
    ++emitted;
    std::sort(members.begin() + mptr[k], members.begin() + mptr[k + 1], [&](int32_t a, int32_t b) {
      if (chains[a].start != chains[b].start) return chains[a].start < chains[b].start;
      return sweep_key(chains[a].head) < sweep_key(chains[b].head);
    });
    int32_t smin = 0x7fffffff, emax = 0;
    for (int64_t m = mptr[k]; m < mptr[k + 1]; ++m) {
      smin = std::min(smin, chains[members[m]].start);
      emax = std::max(emax, chains[members[m]].start + chains[members[m]].len);
    }
    bool fits = !protos.empty();
    if (fits) {
      const Proto &P = protos.back();
      if (std::max(P.end, emax) - std::min(P.start0, smin) > opt.max_span && !P.chains.empty()) fits = false;
      if (fits && opt.vote == 0 && (int)P.chains.size() + (int)(mptr[k + 1] - mptr[k]) > LANES) fits = false; // no lane re-use
      if (fits) fits = place(mptr[k], mptr[k + 1], false);
    }
    if (!fits) new_task();
    place(mptr[k], mptr[k + 1], true);
    for (int64_t s = sptr[k]; s < sptr[k + 1]; ++s)
      if (--indeg[sidx[s]] == 0) ready.push(ckey[sidx[s]]);
  }
  if (emitted != ncomp) { // cannot happen: the condensation is acyclic
    B.cyclic = true;
    return;
  }
  // numbering: packs 0..npack-1, then protos in creation order
  const int64_t per_pack = (int64_t)opt.pack_steps * LANES;
  const int64_t npack = ((int64_t)freerows.size() + per_pack - 1) / per_pack;
  const int64_t ntask = npack + (int64_t)protos.size();
  B.task_of.assign(nb, -1);
  B.step_of.assign(nb, -1);
  B.lane_of.assign(nb, -1);
  for (int64_t f = 0; f < (int64_t)freerows.size(); ++f) {
    const int32_t i = freerows[f];
    B.task_of[i] = (int32_t)(f / per_pack);
    B.step_of[i] = (int32_t)((f % per_pack) / LANES);
    B.lane_of[i] = (int32_t)(f % LANES);
  }
  for (int64_t p = 0; p < (int64_t)protos.size(); ++p)
    for (size_t a = 0; a < protos[p].chains.size(); ++a) {
      const int32_t c = protos[p].chains[a];
      for (int32_t k = 0; k < chains[c].len; ++k) {
        const int32_t i = crow[cptr[c] + k];
        B.task_of[i] = (int32_t)(npack + p);
        B.step_of[i] = chains[c].start + k - protos[p].start0;
        B.lane_of[i] = protos[p].lanes[a];
      }
    }
  B.tasks.resize(ntask);
  for (int64_t t = 0; t < ntask; ++t) {
    auto &T = B.tasks[t];
    if (t < npack) {
      const int64_t cnt = std::min<int64_t>(per_pack, (int64_t)freerows.size() - t * per_pack);
      T.start0 = 0;
      T.nsteps = (int32_t)((cnt + LANES - 1) / LANES);
    } else {
      T.start0 = protos[t - npack].start0;
      T.nsteps = protos[t - npack].end - protos[t - npack].start0;
    }
    T.rows.assign((size_t)T.nsteps * LANES, -1);
    T.stepW.assign((size_t)T.nsteps, 1);
  }
  for (int64_t i = 0; i < nb; ++i) {
    const int32_t t = B.task_of[i];
    B.tasks[t].rows[(size_t)B.step_of[i] * LANES + B.lane_of[i]] = (int32_t)i;
    B.tasks[t].W = std::max(B.tasks[t].W, ndep[i]);
    B.tasks[t].stepW[B.step_of[i]] = std::max(B.tasks[t].stepW[B.step_of[i]], ndep[i]);
    deps(i, [&](int64_t j) {
      const int32_t tj = B.task_of[j];
      if (tj > t) B.cyclic = true; // cannot happen (see 4.)
      if (tj != t) B.tasks[t].prod.push_back(tj);
    });
  }
  for (auto &T : B.tasks) {
    std::sort(T.prod.begin(), T.prod.end());
    T.prod.erase(std::unique(T.prod.begin(), T.prod.end()), T.prod.end());
    if ((int)T.prod.size() > MAXPROD) B.too_many_prod = true;
  }
}

} // namespace detail

// Builds the schedule for the block-diagonal ILU(0) factors stored in the pattern (rp, ci) of the matrix:
// lu = factor values (multipliers below, U above, inverse pivots on the diagonal), diag[i] = index of the
// diagonal entry of row i.  Returns false (S.error set) when the engine is not applicable.
inline bool build(int64_t n, const int64_t *rp, const int32_t *ci, const double *lu, const int64_t *diag, int nblocks,
                  const int64_t *block_ptr, Options opt, Schedule &S, int nthreads = 0)
{
  using detail::BlockSweep;
  S = Schedule();
  int W = 1;
  for (int64_t i = 0; i < n; ++i) W = std::max<int>(W, (int)std::max(diag[i] - rp[i], rp[i + 1] - diag[i] - 1));
  if (W > MAX_W) {
    S.error = "triangular rows wider than the tile format";
    return false;
  }
  S.geo = Geometry(W); // the widest tile (sizes the LDS slots)
  if (nthreads <= 0) nthreads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())); // (one process per GPU on a node: bounded pools)
  nthreads = std::min(nthreads, std::max(1, 2 * nblocks));

  std::vector<BlockSweep> BS((size_t)nblocks * 2);
  std::vector<int> regrouped((size_t)nblocks * 2, 0);
  // work items = (block, sweep) pairs: the two sweeps of a block are independent of each other
  auto run_blocks = [&](const std::function<void(int, int)> &f) {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
      th.emplace_back([&, t]() {
        for (int w = t; w < 2 * nblocks; w += nthreads) f(w >> 1, w & 1);
      });
    for (auto &t : th) t.join();
  };
  run_blocks([&](int b, int sweep) {
    const int64_t r0 = block_ptr[b], nb = block_ptr[b + 1] - r0;
    {
      const bool upper = sweep == 1;
      auto deps = [&](int64_t i, auto &&f) {
        const int64_t g = r0 + i;
        if (!upper)
          for (int64_t k = rp[g]; k < diag[g]; ++k) f((int64_t)ci[k] - r0);
        else
          for (int64_t k = diag[g] + 1; k < rp[g + 1]; ++k) f((int64_t)ci[k] - r0);
      };
      BlockSweep &B = BS[(size_t)b * 2 + sweep];
      detail::build_block_sweep(nb, upper, deps, opt, B);
      Options o2 = opt;
      while (B.too_many_prod && !B.cyclic && o2.max_span > 24) { // shorter tasks meet fewer producers
        o2.max_span /= 2;
        B = BlockSweep();
        detail::build_block_sweep(nb, upper, deps, o2, B);
        regrouped[(size_t)b * 2 + sweep] += 1;
      }
    }
  });
  for (int b = 0; b < nblocks; ++b)
    for (int sweep = 0; sweep < 2; ++sweep) {
      const BlockSweep &B = BS[(size_t)b * 2 + sweep];
      if (B.cyclic) {
        S.error = "more than 64 mutually dependent chains";
        return false;
      }
      if (B.too_many_prod) {
        S.error = "a task has too many producer tasks";
        return false;
      }
      S.stats.regrouped += regrouped[(size_t)b * 2 + sweep];
      S.stats.nchains[sweep] += B.nchains;
      S.stats.max_levels = std::max<int64_t>(S.stats.max_levels, B.nlev);
      if (B.nlev > 0) S.stats.max_rows_per_level = std::max(S.stats.max_rows_per_level, (double)(block_ptr[b + 1] - block_ptr[b]) / B.nlev);
    }
  // global numbering: tasks (group-major, L then U), positions per sweep, tile offsets
  S.groups.resize(nblocks);
  std::vector<int64_t> task_base((size_t)nblocks * 2), pos_base((size_t)nblocks * 2);
  int64_t ntask = 0, npos[2] = {FIRST_POS, FIRST_POS}, nbytes = 0, nkoff = 0; // the first positions of each sweep are reserved (zeros)
  for (int b = 0; b < nblocks; ++b)
    for (int sweep = 0; sweep < 2; ++sweep) {
      const BlockSweep &B = BS[(size_t)b * 2 + sweep];
      task_base[(size_t)b * 2 + sweep] = ntask;
      pos_base[(size_t)b * 2 + sweep] = npos[sweep];
      S.groups[b].task0[sweep] = (int32_t)ntask;
      S.groups[b].ntask[sweep] = (int32_t)B.tasks.size();
      ntask += (int64_t)B.tasks.size();
      for (const auto &T : B.tasks) {
        npos[sweep] += (int64_t)T.nsteps * LANES;
        for (int32_t w : T.stepW) nbytes += Geometry(w).tile_bytes;
        nkoff += T.nsteps + 1;
        S.stats.nsteps[sweep] += T.nsteps;
        S.stats.max_steps = std::max<int64_t>(S.stats.max_steps, T.nsteps);
        S.stats.max_prod = std::max<int64_t>(S.stats.max_prod, (int64_t)T.prod.size());
      }
      S.stats.ntasks[sweep] += (int64_t)B.tasks.size();
    }
  if (S.stats.max_steps > 0xffff) {
    S.error = "a task has more steps than the 16-bit progress requirements hold";
    return false;
  }
  if (npos[0] >= (int64_t)0x0fffffff || npos[1] >= (int64_t)0x0fffffff || ntask >= (int64_t)0x7fffffff || n >= (int64_t)0x0fffffff) {
    S.error = "position space exceeds 32-bit operands";
    return false;
  }
  S.nposL = npos[0];
  S.nposU = npos[1];
  S.tasks.resize((size_t)ntask);
  S.stream.resize((size_t)nbytes);
  parallel_zero(S.stream.data(), (size_t)nbytes, (unsigned)nthreads);
  S.koff.assign((size_t)nkoff, 0);
  S.rowL.assign((size_t)npos[0], -1);
  S.posU.assign((size_t)n, 0);
  S.stats.rows = n;
  {
    int64_t off = 0, kb = 0;
    for (int b = 0; b < nblocks; ++b)
      for (int sweep = 0; sweep < 2; ++sweep) {
        const BlockSweep &B = BS[(size_t)b * 2 + sweep];
        int64_t pos = pos_base[(size_t)b * 2 + sweep];
        for (size_t q = 0; q < B.tasks.size(); ++q) {
          Task &T = S.tasks[(size_t)task_base[(size_t)b * 2 + sweep] + q];
          std::memset(&T, 0, sizeof(Task));
          T.tile_off = off;
          T.W = B.tasks[q].W;
          T.start_level = B.tasks[q].start0;
          T.koff_base = kb;
          int32_t k = 0;
          for (int32_t t = 0; t < B.tasks[q].nsteps; ++t) {
            S.koff[(size_t)kb + t] = k;
            k += Geometry(B.tasks[q].stepW[t]).tile_bytes / 1024;
          }
          S.koff[(size_t)kb + B.tasks[q].nsteps] = k;
          T.first_kib = S.koff[(size_t)kb + 1];
          T.second_kib = B.tasks[q].nsteps > 1 ? S.koff[(size_t)kb + 2] - S.koff[(size_t)kb + 1] : 0;
          kb += B.tasks[q].nsteps + 1;
          off += (int64_t)k * 1024;
          T.pos_base = pos;
          T.nsteps = B.tasks[q].nsteps;
          T.nprod = (int32_t)B.tasks[q].prod.size();
          T.group = b;
          T.sweep = sweep;
          for (int p = 0; p < T.nprod; ++p) T.prod[p] = (int32_t)(task_base[(size_t)b * 2 + sweep] + B.tasks[q].prod[p]);
          pos += (int64_t)T.nsteps * LANES;
        }
      }
  }
  // emit the tiles
  std::vector<Stats> bstats((size_t)nblocks * 2);
  run_blocks([&](int b, int sweep) {
    const int64_t r0 = block_ptr[b];
    Stats &st = bstats[(size_t)b * 2 + sweep];
    {
      const bool upper = sweep == 1;
      const BlockSweep &B = BS[(size_t)b * 2 + sweep];
      const BlockSweep &BL = BS[(size_t)b * 2];
      const int64_t tb = task_base[(size_t)b * 2 + sweep];
      auto pos_of = [&](const BlockSweep &X, int64_t tbase, int64_t i) {
        return S.tasks[(size_t)tbase + X.task_of[i]].pos_base + (int64_t)X.step_of[i] * LANES + X.lane_of[i];
      };
      for (size_t q = 0; q < B.tasks.size(); ++q) {
        const auto &R = B.tasks[q];
        const Task &T = S.tasks[(size_t)tb + q];
        std::vector<int32_t> req(MAXPROD, 0);
        for (int32_t t = 0; t < R.nsteps; ++t) {
          uint32_t late1 = 0, late2 = 0;
          const Geometry G(R.stepW[t]);
          unsigned char *tile = S.stream.data() + T.tile_off + (int64_t)S.koff[(size_t)T.koff_base + t] * 1024;
          int32_t *hdr = reinterpret_cast<int32_t *>(tile);
          int32_t *own = reinterpret_cast<int32_t *>(tile + 256);
          double *s0 = reinterpret_cast<double *>(tile + 512);
          auto idx = [&](int u, int l) -> int32_t & { return *reinterpret_cast<int32_t *>(tile + G.idx_off(u, l)); };
          auto val = [&](int u, int l) -> double & { return *reinterpret_cast<double *>(tile + G.val_off(u, l)); };
          int active = 0;
          for (int l = 0; l < LANES; ++l) {
            const int32_t i = R.rows[(size_t)t * LANES + l];
            // padding everywhere first
            for (int u = 0; u < 4 * G.idx_pieces; ++u) idx(u, l) = PAD_OP;
            own[l] = upper ? ZERO_POS * 8 : 0; // padding lanes: a reserved zero of the forward results / any finite entry of the right-hand side (the result lands in a padding position nobody reads)
            if (i < 0) continue;
            ++active;
            const int64_t g = r0 + i;
            const int64_t pos = T.pos_base + (int64_t)t * LANES + l;
            if (!upper) {
              S.rowL[(size_t)pos] = (int32_t)g;
              own[l] = (int32_t)(g * 8); // the row's right-hand side entry, read from the caller's vector in natural order
              s0[l] = 1.0;
            } else {
              S.posU[(size_t)g] = (int32_t)pos;
              own[l] = (int32_t)(pos_of(BL, task_base[(size_t)b * 2], i) * 8);
              s0[l] = lu[diag[g]];
            }
            const int64_t k0 = upper ? diag[g] + 1 : rp[g], k1 = upper ? rp[g + 1] : diag[g];
            int u = 0;
            for (int64_t k = k0; k < k1; ++k, ++u) {
              const int64_t j = (int64_t)ci[k] - r0;
              const int32_t tj = B.task_of[j], sj = B.step_of[j], lj = B.lane_of[j];
              int32_t op;
              ++st.entries;
              if (tj == (int32_t)q && t - sj < RING) {
                op = ring_op(sj, lj);
                ++st.entries_local;
                if (t - sj <= 1) late1 |= 1u << std::min(u, 31);
                if (t - sj <= 2) late2 |= 1u << std::min(u, 31);
              } else {
                op = (int32_t)(pos_of(B, tb, j) * 8);
                if (tj == (int32_t)q) ++st.entries_self_global;
                else {
                  ++st.entries_remote;
                  const int p = (int)(std::lower_bound(R.prod.begin(), R.prod.end(), tj) - R.prod.begin());
                  req[p] = std::max(req[p], sj + 1);
                }
              }
              idx(u, l) = op;
              val(u, l) = lu[k];
            }
          }
          hdr[0] = active;
          hdr[1] = t + 1 == R.nsteps ? 1 : 0;
          hdr[2] = G.W;
          hdr[3] = t + 1 < R.nsteps ? Geometry(R.stepW[t + 1]).tile_bytes / 1024 : 0;
          hdr[5] = t + 2 < R.nsteps ? Geometry(R.stepW[t + 2]).tile_bytes / 1024 : 0;
          hdr[4] = (int32_t)late1;
          hdr[6] = (int32_t)late2;
          for (int p = 0; p < T.nprod; ++p) hdr[HDR_REQ0 + p / 2] |= (int32_t)((uint32_t)req[p] << (16 * (p & 1))); // cumulative: never decreases along the task
        }
      }
    }
  });
  for (const Stats &b : bstats) {
    S.stats.entries += b.entries;
    S.stats.entries_local += b.entries_local;
    S.stats.entries_self_global += b.entries_self_global;
    S.stats.entries_remote += b.entries_remote;
  }
  return true;
}

// CPU emulation of the device kernel's data flow on the tile stream (test infrastructure for the builder):
// tasks in queue order, steps in order, operands from the emulated LDS ring or the position arrays.  Checks the
// progress requirements on the way.  Returns an empty string or a description of the first violation.
inline std::string emulate(const Schedule &S, int64_t n, const double *d, double *x)
{
  std::vector<double> ypos((size_t)S.nposL, 0.0), xpos((size_t)S.nposU, 0.0);
  std::vector<int32_t> done(S.tasks.size(), 0);
  // position -> task lookup per sweep
  std::vector<std::pair<int64_t, int32_t>> base[2];
  for (size_t t = 0; t < S.tasks.size(); ++t) base[S.tasks[t].sweep].push_back({S.tasks[t].pos_base, (int32_t)t});
  for (auto &b : base) std::sort(b.begin(), b.end());
  auto task_at = [&](int sweep, int64_t pos) {
    auto it = std::upper_bound(base[sweep].begin(), base[sweep].end(), std::make_pair(pos, (int32_t)0x7fffffff));
    return (it - 1)->second;
  };
  for (size_t g = 0; g < S.groups.size(); ++g)
    for (int sweep = 0; sweep < 2; ++sweep) {
      const bool upper = sweep == 1;
      const std::vector<double> &src = upper ? xpos : ypos;
      std::vector<double> &dst = upper ? xpos : ypos;
      for (int32_t q = 0; q < S.groups[g].ntask[sweep]; ++q) {
        const int32_t tid = S.groups[g].task0[sweep] + q;
        const Task &T = S.tasks[(size_t)tid];
        if (T.W > S.geo.W) return "tile geometry mismatch";
        std::vector<double> ring((size_t)RING_BYTES / 8, 0.0);
        std::vector<int32_t> prev_req(MAXPROD, 0);
        for (int p = 0; p < T.nprod; ++p)
          if (T.prod[p] >= tid || T.prod[p] < S.groups[g].task0[sweep]) return "producer is not earlier in the queue of its sweep";
        for (int32_t t = 0; t < T.nsteps; ++t) {
          const unsigned char *tile = S.stream.data() + T.tile_off + (int64_t)S.koff[(size_t)T.koff_base + t] * 1024;
          const int32_t *hdr = reinterpret_cast<const int32_t *>(tile);
          const Geometry G(hdr[2]);
          if (hdr[3] != (t + 1 < T.nsteps ? S.koff[(size_t)T.koff_base + t + 2] - S.koff[(size_t)T.koff_base + t + 1] : 0)) return "next-tile size mismatch";
          if (t == 0 && T.first_kib != S.koff[(size_t)T.koff_base + 1]) return "first-tile size mismatch";
          if (t == 0 && T.second_kib != (T.nsteps > 1 ? S.koff[(size_t)T.koff_base + 2] - S.koff[(size_t)T.koff_base + 1] : 0)) return "second-tile size mismatch";
          if (hdr[5] != (t + 2 < T.nsteps ? S.koff[(size_t)T.koff_base + t + 3] - S.koff[(size_t)T.koff_base + t + 2] : 0)) return "second-next-tile size mismatch";
          if (hdr[2] < MIN_W || G.W > std::max(T.W, MIN_W) || S.koff[(size_t)T.koff_base + t + 1] - S.koff[(size_t)T.koff_base + t] != G.tile_bytes / 1024) return "tile size mismatch";
          const int32_t *own = reinterpret_cast<const int32_t *>(tile + 256);
          const double *s0 = reinterpret_cast<const double *>(tile + 512);
          for (int p = 0; p < T.nprod; ++p) {
            const int32_t rq = (int32_t)(((uint32_t)hdr[HDR_REQ0 + p / 2] >> (16 * (p & 1))) & 0xffffu);
            if (rq < prev_req[p]) return "progress requirement decreases";
            if (rq > S.tasks[(size_t)T.prod[p]].nsteps) return "progress requirement beyond the producer's steps";
            prev_req[p] = rq;
          }
          double out[LANES];
          for (int l = 0; l < LANES; ++l) {
            const int64_t pos = T.pos_base + (int64_t)t * LANES + l;
            if (own[l] < 0 || own[l] % 8 || own[l] / 8 >= (upper ? S.nposL : n)) return "right-hand side operand out of range";
            if (!upper && S.rowL[(size_t)pos] >= 0 && own[l] / 8 != S.rowL[(size_t)pos]) return "right-hand side operand is not the row of the position";
            if (!upper && S.rowL[(size_t)pos] < 0 && own[l] != 0) return "padding lane with a right-hand side operand";
            double s = upper ? ypos[(size_t)own[l] / 8] : d[(size_t)own[l] / 8];
            for (int u = 0; u < G.W; ++u) {
              const int32_t op = *reinterpret_cast<const int32_t *>(tile + G.idx_off(u, l));
              const double a = *reinterpret_cast<const double *>(tile + G.val_off(u, l));
              double xv;
              if (op < 0 || op % 8) return "malformed operand";
              if (op <= RING_Z) {
                if (op < RING_Z) { // which step wrote this ring row last?
                  const int row = op / (LANES * 8);
                  int back = (t % RING) - row;
                  if (back <= 0) back += RING;
                  if (back > t) return "ring operand older than the task";
                  if (back <= 1 && !((uint32_t)hdr[4] >> std::min(u, 31) & 1u)) return "late mask (1 step) misses an operand";
                  if (back <= 2 && !((uint32_t)hdr[6] >> std::min(u, 31) & 1u)) return "late mask (2 steps) misses an operand";
                }
                xv = ring[(size_t)op / 8];
              } else {
                const int64_t opos = op / 8;
                if (opos < FIRST_POS || opos >= (upper ? S.nposU : S.nposL)) return "position operand out of range";
                const int32_t pt = task_at(sweep, opos);
                const int32_t ps = (int32_t)((opos - S.tasks[(size_t)pt].pos_base) / LANES);
                if (pt == tid) {
                  if (ps >= t) return "own operand not yet computed";
                } else {
                  int p = 0;
                  while (p < T.nprod && T.prod[p] != pt) ++p;
                  if (p == T.nprod) return "operand of a task that is not a declared producer";
                  if (prev_req[p] < ps + 1) return "progress requirement does not cover an operand";
                  if (done[(size_t)pt] < ps + 1) return "producer has not run";
                }
                xv = src[(size_t)opos];
              }
              s -= a * xv;
            }
            out[l] = s * s0[l];
          }
          for (int l = 0; l < LANES; ++l) {
            ring[(size_t)(t % RING) * LANES + l] = out[l];
            dst[(size_t)(T.pos_base + (int64_t)t * LANES + l)] = out[l];
          }
          done[(size_t)tid] = t + 1;
        }
      }
    }
  for (int64_t i = 0; i < n; ++i) x[i] = xpos[(size_t)S.posU[(size_t)i]];
  return "";
}

} // namespace pipe
