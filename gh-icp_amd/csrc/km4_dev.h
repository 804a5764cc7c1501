// Kuhn-Munkres, fourth-generation kernel (gfx950): the reference's result (src/km.cpp:13-126) WITHOUT stepping through the
// reference's depth-first search wherever its outcome is order independent.  One 256-thread workgroup per problem, all
// solver state in LDS (44 B per row: 4 problems per CU up to n = 930), the CSR of the explicit entries streamed from global
// memory only by the bulk passes of failed phases.  Rules (oracle/km4_model.inc states them sequentially and is fuzzed against
// the reference traversal; R1 = E1-E3 of round 1's solver, whose rules E1-E12 oracle/km_model.inc still states):
//   R2  per row a list of <= 3 columns in LDS, ascending, EXACTLY the row's tight explicit entries (fl(fl(lx+ly) - w) < eps), plus
//       each entry's offset in its CSR row.  An explicit entry can only BECOME tight when its row label drops, i.e. for rows visited
//       by a failed phase: those lists are rebuilt after the relabelling.  It can only STOP being tight when its column label rises,
//       i.e. for columns visited by a failed phase: rows that were not visited but list such a column re-test their entries (one
//       value load through the stored offset) right after the relabelling.  So a listed entry needs no value and no test at use:
//       the flood, the DFS and the S rounds read columns only.  (Round 2 kept (column, weight) pairs as a superset and re-tested at
//       every use: 61.75 B per row, three problems per CU.)  Rows with more than 3 tight entries are flagged (the flood and the
//       DFS scan their CSR row with the test).
//   R3  every phase starts with an order-free flood from the root (wave 0; lists + one sweep of
//       T_L = {y : fl(fl(L+ly[y]) - bg) < eps} for the smallest label met -- T_L is nested in L).  No free column reached:
//       the phase FAILS and the visited sets are the reference's (a failed findpath() visits exactly the reachable set).
//   R4  failed phase: the slack minima of the visited rows' non-tight explicit entries are order free (min is exact) and
//       are pushed by 4 waves streaming the CSR rows (LDS ds_min_u64: every contribution is >= eps > 0, so the bit pattern
//       orders like the value); background entries contribute through the minimum visited label (E4).  The list rebuild
//       after the relabelling pushes the same rows' minima for the NEXT phase of the root (same labels), so a row is streamed
//       once per failed phase.  Minima pushed for columns that end up visited are never read provided visited columns stay
//       visited in later phases of the root; that is CHECKED after every failed phase, and a violation (only possible when
//       an edge sits within an ulp of eps) sends the problem to the literal single-lane solver at the end of this file.
//   R5  augmenting phase: good = rows that reach a free column in the tight graph.  findpath() of a row that is not good
//       fails whatever has been visited and visits only rows that are not good, so the reference's DFS may skip columns
//       whose owner lies outside any superset S of good without changing its path.  S = fixed point of "(lx - bg) < eps
//       (background-tight to a free column: free columns were never relabelled, their ly is 0 and no ly is negative), or a tight
//       explicit entry in a column of S" (256 threads, two rows each per pass, ~5 rounds).  Listed rows test their list.  A
//       flagged row with 4..6 tight entries when its list was built keeps their COLUMNS (entries 4..6 in the offset slots, which
//       a flagged row never uses): entries only leave the tight set between two rebuilds of a row (R2), so "one of these hints
//       lies in S" is NECESSARY for the row to be good, and the generic round tests it like a list of six.  Rows with more than
//       six test their tight columns, gathered once per phase into a pool (k4_pool_build; ~7 rows per phase on the cfg2
//       matrices).  Until the second half of round 3 every flagged row was simply a member of S, and 28 % of the DFS iterations
//       were pops out of flagged rows that lead nowhere (profiles/r03_km4_second_half.txt: 58 / 104 / 44 -> 49 / 85 / 37 ms).
//       Wave 0 then runs the reference's DFS restricted to S: E7 pointer, E9 march (oracle/km_model.inc), all in LDS; what is left of
//       the search is essentially the augmenting path itself.
// (device code; included by km4.hip -- the stand-alone solve kernel -- and by loop.hip -- the persistent pair loop)
#pragma once
#include "ctx.h"
#include "devmath.h"
#include "km_prob.h"

#include <climits>

namespace {


constexpr int K4_CAP = 3;
constexpr int K4_OVER = 255;  // tln: 0..3 = listed entries; above = flagged (more than 3 tight entries when the list was built):
                              // 4..6 = so many, and their COLUMNS are kept as hints (rule R5); 255 = more than 6
constexpr int K4_HINT = 2 * K4_CAP;
constexpr int K4_T = 256;
constexpr int K4_NONE = 0xFFFF;
constexpr double K4_INF = 1000.0;  // km.cpp:42

typedef __attribute__((address_space(1))) const int* k4_gint;
typedef __attribute__((address_space(1))) const double* k4_gf64;
typedef __attribute__((address_space(1))) const unsigned* k4_gu32;

enum { SH_QT = 0, SH_FREE, SH_RES, SH_QTF, SH_HAZ, SH_CH0, SH_CH1, SH_BAD, SH_NF, SH_UNCERT, SH_LROW, SH_CH2, SH_DFS, SH_NUM = 16 };

struct K4 {
  double *lx, *ly, *slack, *red;
  unsigned *visx, *visy, *prevy, *pushed, *good, *goody, *freey, *ovf;
  int* sh;
  unsigned short *match, *stx, *sty, *tlc, *tlo;  // tlc / tlo: listed columns and their offsets in the CSR row (K4_CAP per row)
  unsigned char* tln;
  int n, nw;
  double bg, eps;
  k4_gu32 rptr;
  k4_gint cols;
  k4_gf64 vals;
};

__device__ inline bool k4_bit(const unsigned* b, int i) { return (b[i >> 5] >> (i & 31)) & 1u; }
__device__ inline bool k4_flagged(int tn) { return tn > K4_CAP; }
__device__ inline int k4_cnt(int tn) { return tn > K4_CAP ? 0 : tn; }  // listed entries (0 for flagged rows)

// Bulk pass over rows list[0..count), all 4 waves: REBUILD writes the rows' lists (R2), PUSH sends the slack minima of their
// non-tight entries (R4), ONLY_UNPUSHED skips rows whose minima are already in slack.  16 lanes per row, so a wave instruction works
// on four rows and every lane has four entries (64 per row) in flight per round: the pass is bound by the latency of the CSR
// loads (L2), not by their volume, and 16 rows per workgroup in flight is what hides it.
// SEED (rule R3', with the rebuild pass of a failed phase): a tight entry into a column the failed phase had visited certifies that
// column when it is the column's TREE EDGE (sty[col] = the row the flood claimed it from; certificates collect in goody, which is idle
// between two augmenting phases); a tight entry into any other column is new: the column is claimed for the next phase's flood on the
// spot (visited bit, tree edge, owner appended to the queue behind the `count` rows this pass reads).
template <bool REBUILD, bool PUSH, bool ONLY_UNPUSHED, bool SEED = false>
__device__ inline void k4_bulk(const K4& s, const unsigned short* list, int count, int wave, int lane) {
  const int grp = lane >> 4, lig = lane & 15;
  unsigned long long* sl = reinterpret_cast<unsigned long long*>(s.slack);
  for (int base = 0; base < count; base += 16) {
    const int i = base + wave * 4 + grp;
    int x = -1;
    if (i < count) {
      x = list[i];
      if (ONLY_UNPUSHED && k4_bit(s.pushed, x)) x = -1;
    }
    unsigned cb = 0, ce = 0;
    double lxr = 0.0;
    if (x >= 0) { cb = s.rptr[x]; ce = s.rptr[x + 1]; lxr = s.lx[x]; }
    int cnt = 0;
    for (unsigned off = 0; __ballot(cb + off < ce); off += 64) {
      int col[4];
      double val[4];
      unsigned c[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        c[j] = cb + off + 16u * j + lig;
        col[j] = 0; val[j] = 0.0;  // a lane without a row, or with an empty row, loads nothing (found by tests/hipsim: cols[cb] may lie past the CSR)
        if (ce > cb) { const unsigned cc = min(c[j], ce - 1u); col[j] = s.cols[cc]; val[j] = s.vals[cc]; }
      }
      double lyv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) lyv[j] = s.ly[col[j]];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool in = c[j] < ce;
        const double d = (lxr + lyv[j]) - val[j];
        const bool td = d < s.eps;
        if (PUSH && in && !td) atomicMin(&sl[col[j]], (unsigned long long)__double_as_longlong(d));
        if (SEED && in && td) {
          const unsigned bit = 1u << (col[j] & 31);
          if (s.prevy[col[j] >> 5] & bit) {
            if (s.sty[col[j]] == x) atomicOr(&s.goody[col[j] >> 5], bit);
          } else if (!(atomicOr(&s.visy[col[j] >> 5], bit) & bit)) {
            s.sty[col[j]] = (unsigned short)x;
            const int m = s.match[col[j]];
            if (m == K4_NONE) s.sh[SH_FREE] = 1;
            else { s.stx[atomicAdd(&s.sh[SH_QT], 1)] = (unsigned short)m; atomicOr(&s.visx[m >> 5], 1u << (m & 31)); }
          }
        }
        if (REBUILD) {
          const unsigned gb = (unsigned)(__ballot(in && td) >> (grp * 16)) & 0xffffu;
          if (in && td) {
            const int rk = cnt + __popc(gb & ((1u << lig) - 1u));
            if (rk < K4_CAP) { s.tlc[x * K4_CAP + rk] = (unsigned short)col[j]; s.tlo[x * K4_CAP + rk] = (unsigned short)(c[j] - cb); }
          }
          // entries 4..6 of a row that turns out flagged: their columns over the offsets, which a flagged row never uses.  A separate
          // store AFTER the one above: entries 1..3 come earlier in the row, so in program order the hints land last.  The two stores to one
          // slot come from different lanes; the wave barrier (no instruction: a fence for the scheduler) states the order the lockstep wave
          // has anyway (found by tests/hipsim with HIPSIM_ORDER=reverse: without it the offset of entry 1 could land on the hint of entry 4)
          __builtin_amdgcn_wave_barrier();
          if (in && td) {
            const int rk = cnt + __popc(gb & ((1u << lig) - 1u));
            if (rk >= K4_CAP && rk < K4_HINT) s.tlo[x * K4_CAP + rk - K4_CAP] = (unsigned short)col[j];
          }
          cnt += __popc(gb);
        }
      }
    }
    if (REBUILD && lig == 0 && x >= 0) {
      const int tn = cnt > K4_HINT ? K4_OVER : cnt;
      const bool over = tn == K4_OVER, was = s.tln[x] == K4_OVER;
      s.tln[x] = (unsigned char)tn;
      if (over != was) {
        if (over) atomicOr(&s.ovf[x >> 5], 1u << (x & 31));
        else atomicAnd(&s.ovf[x >> 5], ~(1u << (x & 31)));
      }
    }
  }
}

// R2, second half: after the relabelling of a failed phase the labels of the VISITED columns have risen, so a listed entry of a row
// that was NOT visited may have left the tight set.  Such rows re-test their entries (value through the stored CSR offset) and
// compact their list; visited rows are rebuilt from their CSR row by the bulk pass.  Few rows qualify (the visited sets are small),
// so the pass is an LDS sweep plus, rarely, one round of value loads.
__device__ inline void k4_revalidate(const K4& s, int tid) {
  // visited rows / columns of the failed phase through their copies `pushed` / `prevy`: the next phase clears visx / visy, the copies
  // stay until the next failed phase, so this pass needs no barrier of its own (the one that opens the next phase orders its writes)
  for (int base = tid; base < s.n; base += 4 * K4_T) {
    int tn[4], lc[4][K4_CAP];
    unsigned px[4], pv[4][K4_CAP];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // every load of the common case issued before the first use
      const int ic = min(base + k * K4_T, s.n - 1);
      tn[k] = s.tln[ic];
      px[k] = s.pushed[ic >> 5];
#pragma unroll
      for (int e = 0; e < K4_CAP; e++) lc[k][e] = s.tlc[ic * K4_CAP + e];
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int e = 0; e < K4_CAP; e++) pv[k][e] = s.prevy[lc[k][e] >> 5];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = base + k * K4_T;
      if (i >= s.n || tn[k] == 0 || k4_flagged(tn[k]) || ((px[k] >> (i & 31)) & 1u)) continue;
      bool hit = false;
#pragma unroll
      for (int e = 0; e < K4_CAP; e++) hit |= (int)(e < tn[k]) & (int)((pv[k][e] >> (lc[k][e] & 31)) & 1u);
      if (!hit) continue;
      const double lxv = s.lx[i];
      const unsigned rb = s.rptr[i];
      int keep = 0;
      for (int e = 0; e < tn[k]; e++) {
        const int col = s.tlc[i * K4_CAP + e];
        const unsigned short o = s.tlo[i * K4_CAP + e];
        const double w = s.vals[rb + o];
        if (((lxv + s.ly[col]) - w) < s.eps) { s.tlc[i * K4_CAP + keep] = (unsigned short)col; s.tlo[i * K4_CAP + keep] = o; keep++; }
      }
      s.tln[i] = (unsigned char)keep;
    }
  }
}

// R5, flagged rows without hints (more than K4_HINT tight entries): S should hold such a row only if the row can be good -- one of its
// tight entries lies in a column of S, or it is background-tight to a free column -- not unconditionally (round 2 and the first half
// of round 3 did that for every flagged row, and the DFS then descended into flagged rows that lead nowhere: 28 % of its iterations were
// the pops that followed, profiles/r03_km_dfs_kinds.txt).  Pooling EVERY flagged row cost more in the S rounds than it saved in the DFS
// (profiles/r03_km4_second_half.txt, v1); with the hints only the few rows beyond six entries come here.
// The S rounds cannot afford a CSR scan per flagged row per round, so the rows' tight columns are gathered ONCE per augmenting
// phase into a pool (region r = `region` u16 per flagged row: count, then the columns, ascending) next to the list of flagged rows.  Both live
// in the SLACK array's storage, which is dead from the start of an augmenting phase to the next root (8 n bytes: the row -> column map of
// the matching, the flagged rows, the pool: n + n + 2 n u16).  Until round 6 they lived in the flood's queue and the DFS's column stack
// (stx / sty), which the lazy S of rule R5' needs intact: S is now computed in the MIDDLE of the search.  A row with more tight entries than
// its region holds joins S unconditionally, as before (any superset of good is valid, R5).  16 lanes per row, as in k4_bulk.
__device__ inline void k4_pool_build(const K4& s, const unsigned short* __restrict__ flist, unsigned short* __restrict__ pool, int nf, int region, int wave, int lane) {
  const int grp = lane >> 4, lig = lane & 15, capf = region - 1;
  for (int base = 0; base < nf; base += 16) {
    const int i = base + wave * 4 + grp;
    int x = -1;
    if (i < nf) x = flist[i];
    unsigned cb = 0, ce = 0;
    double lxr = 0.0;
    if (x >= 0) { cb = s.rptr[x]; ce = s.rptr[x + 1]; lxr = s.lx[x]; }
    int cnt = 0;
    for (unsigned off = 0; __ballot(cb + off < ce); off += 64) {
      int col[4];
      double val[4];
      unsigned c[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        c[j] = cb + off + 16u * j + lig;
        col[j] = 0; val[j] = 0.0;
        if (ce > cb) { const unsigned cc = min(c[j], ce - 1u); col[j] = s.cols[cc]; val[j] = s.vals[cc]; }
      }
      double lyv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) lyv[j] = s.ly[col[j]];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool td = (int)(c[j] < ce) & (int)(((lxr + lyv[j]) - val[j]) < s.eps);
        const unsigned gb = (unsigned)(__ballot(td) >> (grp * 16)) & 0xffffu;
        if (td) {
          const int rk = cnt + __popc(gb & ((1u << lig) - 1u));
          if (rk < capf) pool[i * region + 1 + rk] = (unsigned short)col[j];
        }
        cnt += __popc(gb);
      }
    }
    if (lig == 0 && x >= 0) {
      if (cnt > capf) { atomicOr(&s.good[x >> 5], 1u << (x & 31)); cnt = 0; }
      pool[i * region] = (unsigned short)cnt;
    }
  }
}

// Minimum over the wave, in every lane.  Four DPP steps (quad_perm [1,0,3,2] and [2,3,0,1], row_ror:4, row_ror:8) leave the minimum of
// each row of 16 in all its lanes, v_readlane fetches the four rows.  (The __shfl_xor butterfly this replaces is twelve dependent
// ds_bpermute_b32, ~1000 cycles for a wave that runs alone; a flood level or an S round is only a few thousand.)
#define K4_DPP_MIN(V, CTRL)                                                                                   \
  V = fmin(V, __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(V), CTRL, 0xf, 0xf, false),     \
                               __builtin_amdgcn_update_dpp(0, __double2loint(V), CTRL, 0xf, 0xf, false)))
__device__ inline double k4_wave_min(double v) {
  K4_DPP_MIN(v, 0xb1);
  K4_DPP_MIN(v, 0x4e);
  K4_DPP_MIN(v, 0x124);
  K4_DPP_MIN(v, 0x128);
  const int hi = __double2hiint(v), lo = __double2loint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return fmin(fmin(r0, r1), fmin(r2, r3));
}
#undef K4_DPP_MIN

// ---- R3: the flood (wave 0), level by level.  Returns true when a free column is reachable; the visited rows are stx[0 .. *qt_out).
// A column is claimed by the returning ds_or on its visited bit (two rows reaching it in the same instruction are serialised by
// the LDS); every matched row owns exactly one column, so the claim of a column is also the one enqueue of its owner, and the
// queue tail lives in a register (ballot ranks, no LDS counter).  The row a column is claimed FROM is its tree edge (sty[col], the
// DFS's column stack is idle until the augmenting phase): rule R3' certifies the tree after the next relabelling instead of flooding
// from the root again.  The flood starts at queue position qh0 with qt0 rows queued and lflood0 the smallest label already swept.
#define K4_CLAIM(WANT, COL, OWNER, PARENT)                                                  \
  do {                                                                                      \
    unsigned old_ = 0u;                                                                     \
    const unsigned bit_ = 1u << ((COL) & 31);                                               \
    if (WANT) old_ = atomicOr(&s.visy[(COL) >> 5], bit_);                                   \
    const bool fresh_ = (WANT) && !(old_ & bit_);                                           \
    if (fresh_) s.sty[COL] = (unsigned short)(PARENT);                                      \
    free_l |= fresh_ && (OWNER) == K4_NONE;                                                 \
    const bool enq_ = fresh_ && (OWNER) != K4_NONE;                                         \
    const unsigned long long eb_ = __ballot(enq_);                                          \
    if (enq_) {                                                                             \
      s.stx[qt + __popcll(eb_ & ((1ull << lane) - 1ull))] = (unsigned short)(OWNER);        \
      atomicOr(&s.visx[(OWNER) >> 5], 1u << ((OWNER) & 31));                                \
    }                                                                                       \
    qt += __popcll(eb_);                                                                    \
  } while (0)

template <bool PROF>
__device__ inline bool k4_flood(const K4& s, int qh0, int qt0, double lflood0, int lane, int* qt_out, long long* pc) {
  const int n = s.n;
  int qh = qh0, qt = qt0;
  bool free_l = false;
  double lflood = lflood0;
  while (qh < qt) {
    const int qe = qt;
    double lcand = INFINITY;
    int xcand = 0;
    const long long tl0 = PROF ? (long long)__builtin_readcyclecounter() : 0;
    if (PROF) pc[0]++;
    for (int base = qh; base < qe; base += 64) {
      const int i = base + lane;
      const bool act = i < qe;
      const int xr = s.stx[min(i, qe - 1)];
      const double lxr = s.lx[xr];
      const int tn = s.tln[xr];
      int lc[K4_CAP], mc[K4_CAP];
      unsigned vw[K4_CAP];
#pragma unroll
      for (int k = 0; k < K4_CAP; k++) lc[k] = s.tlc[xr * K4_CAP + k];
#pragma unroll
      for (int k = 0; k < K4_CAP; k++) { vw[k] = s.visy[lc[k] >> 5]; mc[k] = s.match[lc[k]]; }
      const int t = act ? k4_cnt(tn) : 0;
#pragma unroll
      for (int k = 0; k < K4_CAP; k++) {  // listed entries are tight (R2): unvisited is all that is asked
        const bool want = (int)(k < t) & (int)(((vw[k] >> (lc[k] & 31)) & 1u) == 0u);
        K4_CLAIM(want, lc[k], mc[k], xr);
      }
      if (act && (lxr - s.bg) < s.eps && lxr < lflood && lxr < lcand) { lcand = lxr; xcand = xr; }
      unsigned long long ob = __ballot(act && k4_flagged(tn));
      if (PROF) pc[1] += __popcll(ob);
      while (ob) {  // flagged rows: every tight entry of the CSR row
        const int l = (int)__ffsll((long long)ob) - 1;
        ob &= ob - 1ull;
        const int xo = __builtin_amdgcn_readlane(xr, l);
        const double lxo = s.lx[xo];
        const unsigned cb = s.rptr[xo], ce = s.rptr[xo + 1];
        for (unsigned c0 = cb; c0 < ce; c0 += 64) {
          const unsigned c = c0 + lane, cc = min(c, ce - 1u);
          const int col = s.cols[cc];
          const double val = s.vals[cc];
          const int m = s.match[col];
          const bool want = (int)(c < ce) & (int)(((lxo + s.ly[col]) - val) < s.eps) & (int)!k4_bit(s.visy, col);
          K4_CLAIM(want, col, m, xo);
        }
      }
    }
    const long long tl1 = PROF ? (long long)__builtin_readcyclecounter() : 0;
    if (PROF) pc[2] += tl1 - tl0;
    if (__ballot(lcand < lflood)) {  // a label below every one swept so far (rare: a reduction per level was a third of a level's time);
      const double lmin = k4_wave_min(lcand);  // T_L of the smallest label contains T_L of every larger one
      const int xpar = __builtin_amdgcn_readlane(xcand, (int)__ffsll((long long)__ballot(lcand == lmin)) - 1);  // a row that has it
      lcand = lmin;
      lflood = lcand;
      if (PROF) pc[3]++;
      for (int y0 = 0; y0 < n; y0 += 256) {  // four independent windows per round
        bool c[4];
        int m[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int y = y0 + k * 64 + lane, yc = min(y, n - 1);
          const unsigned vw = s.visy[yc >> 5];
          const double lv = s.ly[yc];
          m[k] = s.match[yc];
          c[k] = (int)(y < n) & (int)(((vw >> (yc & 31)) & 1u) == 0u) & (int)(((lcand + lv) - s.bg) < s.eps);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (y0 + k * 64 >= n) break;
          const int y = min(y0 + k * 64 + lane, n - 1);
          K4_CLAIM(c[k], y, m[k], xpar);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (PROF) pc[4] += (long long)__builtin_readcyclecounter() - tl1;
    qh = qe;
    if (__ballot(free_l)) { *qt_out = qt; return true; }
  }
  *qt_out = qt;
  return false;
}
#undef K4_CLAIM

// ---- R5: the reference's DFS restricted to S (wave 0).  Returns false only on an internal error.
// One iteration == one findpath() activation or resumption (km.cpp:13-37) and costs two dependent LDS round trips: (1) the
// row record (label, listed columns), (2) everything the verdict needs -- the visited / S words and the owner of the <= 3 listed
// columns, and for a 64-column window of background candidates at the E7 pointer of the row's label: ly, visited / S words, owner.
// The search's registers, kept by the caller across the S computation of rule R5' (lazy S).
struct K4Dfs {
  int sp, x, ystart, cp, cnext;
  double ck;  // E7 cache, one label per lane
};
// Returns 1: augmented; 0: internal error; 2 (only with lazy): the search is about to take its first step back -- D holds its state, the
// stacks stx / sty its frames; the caller computes S, cuts the stack back to its deepest good frame (k4_dfs_unwind) and calls again with
// lazy = false.  A fresh search: D.sp = 0, D.x = root, D.ystart = 0, D.ck = NaN (never matches), D.cp = D.cnext = 0, stx[0] = root.
template <bool PROF>
__device__ inline int k4_dfs(const K4& s, K4Dfs& D, const bool lazy, int lane, long long* q_iter, long long* q_act, long long* pd) {
  const int n = s.n;
  const double bg = s.bg, eps = s.eps;
  int sp = D.sp, x = D.x, ystart = D.ystart;
  double ck = D.ck;
  int cp = D.cp, cnext = D.cnext;
  const int lk = min(lane, K4_CAP - 1);
  for (;;) {
    if (PROF) { ++*q_iter; ++*q_act; }
    const long long t_it0 = PROF ? (long long)__builtin_readcyclecounter() : 0;
    int it_type = 0;  // PROF: 0 flagged row, 1 listed-only row, 2 background-tight row with a list, 3 march, 4 ended in a pop
    // ---- round trip 1: the row record
    // (round 6, measured and dropped: asking for the row's CSR extent HERE, for every kind of row, so that a flagged row's two dependent
    // global round trips -- extent, then entries: 1.8-3.3 k cycles per flagged-row step -- become one.  The load has to be volatile or the
    // compiler sinks it back into the flagged branch, and a volatile load is waited for at the loop's back edge: every step then pays a
    // global latency.  47.7 / 72.0 / 38.0 / 342.4 / 57.6 -> 51.6 / 75.0 / 43.0 / 346.7 / 65.7 ms on the five real matrices, mean solve of
    // the bench 44.3 -> 48.3 ms: profiles/r06_km_variants_call3.txt)
    const double lxv = s.lx[x];
    const int tn = s.tln[x];
    const int lc = s.tlc[x * K4_CAP + lk];
    const int ncnt = k4_cnt(tn);
    const bool bgt = (lxv - bg) < eps;
    if (PROF) it_type = k4_flagged(tn) ? 0 : (!bgt ? 1 : (tn != 0 ? 2 : 3));
    int slot = 0, p = n;
    if (bgt) {  // E7: one scan pointer per distinct label value; everything below ystart is dead for this label as well
      const unsigned long long hit = __ballot(ck == lxv);
      p = ystart;
      if (hit) {
        slot = (int)__ffsll((long long)hit) - 1;
        p = max(p, __builtin_amdgcn_readlane(cp, slot));
      } else {
        slot = cnext;
        cnext = (cnext + 1) & 63;
        if (lane == slot) ck = lxv;
      }
    }
    // ---- round trip 2: listed entries and the first window, issued together
    const unsigned vwL = s.visy[lc >> 5], gwL = s.goody[lc >> 5];
    const int mL = s.match[lc];
    int yw = p + lane, ywc = min(yw, n - 1);
    unsigned vwW = s.visy[ywc >> 5], gwW = s.goody[ywc >> 5];
    double lyW = s.ly[ywc];
    int mW = s.match[ywc];
    int best = INT_MAX, mbest = K4_NONE;
    if (k4_flagged(tn)) {  // flagged row: lowest tight unvisited good column of the CSR row
      const unsigned cb = s.rptr[x], ce = s.rptr[x + 1];
      for (unsigned c0 = cb; c0 < ce; c0 += 64) {
        const unsigned c = c0 + lane, cc = min(c, ce - 1u);
        const int col = s.cols[cc];
        const double val = s.vals[cc];
        const bool t = (int)(c < ce) & (int)(((lxv + s.ly[col]) - val) < eps) & (int)(col >= ystart) & (int)!k4_bit(s.visy, col) & (int)k4_bit(s.goody, col);
        const unsigned long long b = __ballot(t);
        if (b) { best = __builtin_amdgcn_readlane(col, (int)__ffsll((long long)b) - 1); mbest = s.match[best]; break; }
      }
    } else {
      const bool t = (int)(lane < ncnt) & (int)(lc >= ystart) & (int)((((~vwL & gwL) >> (lc & 31)) & 1u) != 0u);  // listed = tight (R2)
      const unsigned long long b = __ballot(t);
      if (b) {
        const int l = (int)__ffsll((long long)b) - 1;
        best = __builtin_amdgcn_readlane(lc, l);
        mbest = __builtin_amdgcn_readlane(mL, l);
      }
    }
    bool augment = false;
    if (bgt) {
      const int lim = min(n, best);
      bool cand = (int)(yw < lim) & (int)((((~vwW & gwW) >> (ywc & 31)) & 1u) != 0u) & (int)(((lxv + lyW) - bg) < eps);
      unsigned long long bw = __ballot(cand);
      while (!bw && p + 64 < lim) {  // windows without a candidate
        p += 64;
        yw = p + lane; ywc = min(yw, n - 1);
        vwW = s.visy[ywc >> 5]; gwW = s.goody[ywc >> 5]; lyW = s.ly[ywc]; mW = s.match[ywc];
        cand = (int)(yw < lim) & (int)((((~vwW & gwW) >> (ywc & 31)) & 1u) != 0u) & (int)(((lxv + lyW) - bg) < eps);
        bw = __ballot(cand);
      }
      if (tn != 0) {
        if (bw) {
          const int l = (int)__ffsll((long long)bw) - 1;
          best = p + l;
          mbest = __builtin_amdgcn_readlane(mW, l);
          if (lane == slot) cp = best;
        } else if (lane == slot) cp = max(p, lim);
      } else {
        // ---- E9 march (oracle/km_model.inc): a chain of rows without tight explicit entries that share the label L picks the members of
        // T_L in column order, up to 64 activations per window (lim == n here: the row has no listed entry)
        int outcome = bw ? 3 : 0;
        if (!bw) p = n;  // the skip loop above has covered every column below n
        while (outcome == 3) {
          bool cont = false;
          if (cand && mW != K4_NONE) cont = (int)(s.tln[mW] == 0) & (int)(s.lx[mW] == lxv);
          const unsigned long long stop = __ballot(cand && !cont);
          const int jstar = stop ? (int)__ffsll((long long)stop) - 1 : 63;
          const unsigned long long R = bw & (~0ull >> (63 - jstar));  // the picks of this window, in column order
          const unsigned long long below = R & ((1ull << lane) - 1ull);
          const int rank = __popcll(below);
          const int pm = __shfl(mW, below ? 63 - __clzll((long long)below) : 0, 64);
          if ((R >> lane) & 1ull) {
            atomicOr(&s.visy[yw >> 5], 1u << (yw & 31));
            s.sty[sp + rank] = (unsigned short)yw;
            if (rank > 0) s.stx[sp + rank] = (unsigned short)pm;  // the row that picked this column (frame sp holds x)
          }
          const int k = __popcll(R), lastlane = 63 - __clzll((long long)R);
          const int mlast = __builtin_amdgcn_readlane(mW, lastlane);
          if (PROF) *q_act += k - 1;
          sp += k - 1;  // frame of the row that made the last pick
          p += lastlane + 1;
          if (mlast == K4_NONE) { outcome = 1; break; }
          sp++;
          if (lane == 0) { s.stx[sp] = (unsigned short)mlast; s.sty[sp] = (unsigned short)K4_NONE; }
          x = mlast; ystart = 0;
          if (stop) { outcome = 2; break; }  // the owner of the last pick is not part of the chain
          if (PROF) ++*q_act;                // ... it is: its activation continues the march
          bw = 0;
          while (!bw && p < n) {
            yw = p + lane; ywc = min(yw, n - 1);
            vwW = s.visy[ywc >> 5]; gwW = s.goody[ywc >> 5]; lyW = s.ly[ywc]; mW = s.match[ywc];
            cand = (int)(yw < n) & (int)((((~vwW & gwW) >> (ywc & 31)) & 1u) != 0u) & (int)(((lxv + lyW) - bg) < eps);
            bw = __ballot(cand);
            if (!bw) p += 64;
          }
          if (!bw) outcome = 0;
        }
        if (lane == slot) cp = min(p, n);
        __builtin_amdgcn_wave_barrier();
        if (outcome == 1) { augment = true; }
        else if (outcome == 2) { if (PROF) { pd[it_type]++; pd[5 + it_type] += (long long)__builtin_readcyclecounter() - t_it0; } continue; }
        // outcome 0: x (possibly a row reached by the march: its frame is sp) has no candidate left -> pop below
      }
    }
    if (PROF && (augment || best != INT_MAX)) { pd[it_type]++; pd[5 + it_type] += (long long)__builtin_readcyclecounter() - t_it0; }
    if (augment) break;
    if (best != INT_MAX) {
      if (lane == 0) { atomicOr(&s.visy[best >> 5], 1u << (best & 31)); s.sty[sp] = (unsigned short)best; }
      if (mbest == K4_NONE) break;
      sp++;
      if (lane == 0) { s.stx[sp] = (unsigned short)mbest; s.sty[sp] = (unsigned short)K4_NONE; }
      x = mbest; ystart = 0;
    } else {
      if (lazy) {  // R5': the first step back -- S is due.  The row x (frame sp) has no candidate left under "every column passes"
        D.sp = sp; D.x = x; D.ystart = ystart; D.ck = ck; D.cp = cp; D.cnext = cnext;
        __builtin_amdgcn_wave_barrier();
        return 2;
      }
      sp--;
      if (sp < 0) return 0;
      x = s.stx[sp]; ystart = (int)s.sty[sp] + 1;
      if (PROF) { pd[4]++; pd[9] += (long long)__builtin_readcyclecounter() - t_it0; }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // augment: match[y] = x on every level of the recursion (km.cpp:26-29); the last column is no longer free
  __builtin_amdgcn_wave_barrier();
  const int ylast = s.sty[sp];
  for (int f = lane; f <= sp; f += 64) s.match[s.sty[f]] = s.stx[f];
  if (lane == 0) atomicAnd(&s.freey[ylast >> 5], ~(1u << (ylast & 31)));
  return 1;
}

// R5', after S has been computed: the stack back to its deepest frame whose row is in S.  Every frame above a row that is not good holds
// a row that is not good either (a row that reaches a good row is good), and S is a superset of good whose complement holds no good row:
// frames are cut from the first row outside S on.  The search then goes on after the column that frame was waiting on; if every frame's
// row is in S the same row is looked at again, now under S.  Returns false when the root itself is outside S (impossible: the flood
// reached a free column from it).
__device__ inline bool k4_dfs_unwind(const K4& s, K4Dfs& D, int lane) {
  int keep = D.sp + 1;
  for (int base = 0; base <= D.sp; base += 64) {
    const int f = base + lane;
    const bool out = f <= D.sp && !k4_bit(s.good, s.stx[min(f, D.sp)]);
    const unsigned long long b = __ballot(out);
    if (b) { keep = base + (int)__ffsll((long long)b) - 1; break; }
  }
  if (keep == 0) return false;
  if (keep <= D.sp) {
    D.sp = keep - 1;
    D.x = s.stx[D.sp];
    D.ystart = (int)s.sty[D.sp] + 1;
  }
  return true;
}

// ---- literal solver (hazard fallback of R4): lane 0 of wave 0 runs the reference line by line on background + CSR.
// A matrix that gets here has an edge within an ulp of eps after a relabelling; the reference itself usually does not
// terminate on such input (its delta becomes 0), hence the step budget (the O(n^3) bound of a legitimate instance).  Slow by design,
// never on the hot path.
__device__ inline int k4_literal(const K4& s, const Km2Problem& P) {
  const int n = s.n;
  const double bg = s.bg, eps = s.eps;
  unsigned short* curs = s.tlc;  // per frame: CSR cursor relative to the row start (n entries of the 3n available)
  for (int i = 0; i < n; i++) { s.lx[i] = P.lx_init[i]; s.ly[i] = 0.0; s.match[i] = (unsigned short)K4_NONE; }
  // bound of the reference itself: <= n descents per phase, <= n failed phases per root, n roots (the budget only stops non-finite input)
  long long budget = 2ll * n * n * n + 4096;
  for (int root = 0; root < n; root++) {
    for (int j = 0; j < n; j++) s.slack[j] = K4_INF;
    for (;;) {
      for (int w = 0; w < s.nw; w++) { s.visx[w] = 0u; s.visy[w] = 0u; }
      int sp = 0;
      s.stx[0] = (unsigned short)root; s.sty[0] = 0; curs[0] = 0;
      s.visx[root >> 5] |= 1u << (root & 31);
      bool ok = false;
      while (sp >= 0) {
        const int x = s.stx[sp];
        const double lxv = s.lx[x];
        const unsigned rb = s.rptr[x], re = s.rptr[x + 1];
        unsigned c = rb + curs[sp];
        int y = s.sty[sp];
        bool descended = false;
        for (; y < n; ++y) {
          double wv = bg;
          if (c < re && s.cols[c] == y) { wv = s.vals[c]; ++c; }
          if (k4_bit(s.visy, y)) continue;
          const double t = lxv + s.ly[y] - wv;
          if (t < eps) {
            s.visy[y >> 5] |= 1u << (y & 31);
            const int m = s.match[y];
            s.sty[sp] = (unsigned short)y;  // the column this frame is waiting on
            curs[sp] = (unsigned short)(c - rb);
            if (m == K4_NONE) { ok = true; break; }
            if (--budget < 0) return 5;
            sp++;
            s.stx[sp] = (unsigned short)m; s.sty[sp] = 0; curs[sp] = 0;
            s.visx[m >> 5] |= 1u << (m & 31);
            descended = true;
            break;
          } else
            s.slack[y] = fmin(t, s.slack[y]);
        }
        if (ok) break;
        if (descended) continue;
        sp--;  // findpath(x) returns false: the caller resumes after the column it was waiting on
        if (sp >= 0) s.sty[sp] = (unsigned short)(s.sty[sp] + 1);
      }
      if (ok) {
        for (int f = 0; f <= sp; f++) s.match[s.sty[f]] = s.stx[f];
        break;
      }
      double delta = K4_INF;
      for (int j = 0; j < n; j++)
        if (!k4_bit(s.visy, j)) delta = fmin(delta, s.slack[j]);
      for (int i = 0; i < n; i++)
        if (k4_bit(s.visx, i)) s.lx[i] -= delta;
      for (int i = 0; i < n; i++) {
        if (k4_bit(s.visy, i)) s.ly[i] += delta;
        else s.slack[i] -= delta;
      }
      if (--budget < 0) return 5;
    }
  }
  return 0;
}

// One solve by the calling 256-thread workgroup: `smem` = lds_bytes of LDS (gh_km4_lds_bytes(P.n) at least), every thread of the
// workgroup must call.  Shared by the stand-alone kernel k_km4 (km4.hip) and the persistent pair loop (loop.hip).
template <bool PROF>
__device__ inline void k4_solve_block(const Km2Problem& P, int flags, char* smem, int lds_bytes, unsigned long long* __restrict__ lstat) {
  const unsigned long long t_wall0 = lstat ? __builtin_amdgcn_s_memrealtime() : 0ull;  // 100 MHz, common to all CUs
  const int n = P.n, nw = (n + 31) / 32, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  K4 s;
  s.n = n; s.nw = nw; s.bg = P.bg; s.eps = P.eps;
  s.rptr = (k4_gu32)P.row_ptr; s.cols = (k4_gint)P.cols; s.vals = (k4_gf64)P.vals;
  s.lx = (double*)smem;
  s.ly = s.lx + n;
  s.slack = s.ly + n;
  s.red = s.slack + n;  // 16 doubles
  s.visx = (unsigned*)(s.red + 16);
  s.visy = s.visx + nw; s.prevy = s.visy + nw; s.pushed = s.prevy + nw; s.good = s.pushed + nw; s.goody = s.good + nw;
  s.freey = s.goody + nw; s.ovf = s.freey + nw;
  s.sh = (int*)(s.ovf + nw);
  s.match = (unsigned short*)(s.sh + SH_NUM);
  s.stx = s.match + n;
  s.sty = s.stx + n + 2;
  s.tlc = s.sty + n + 2;
  s.tlo = s.tlc + (size_t)n * K4_CAP;
  s.tln = (unsigned char*)(s.tlo + (size_t)n * K4_CAP);
  if ((long long)(reinterpret_cast<char*>(s.tln + n) - smem) > (long long)lds_bytes) {  // the launch gave this workgroup less LDS than the
    if (tid == 0 && P.status) *P.status = 6;                                            // problem needs: refuse loudly, touch nothing
    for (int i = tid; i < n; i += K4_T) P.match_out[i] = -1;
    return;
  }
  const double bg = s.bg, eps = s.eps;
  // (the flood and the DFS are one wave's work; letting co-resident workgroups use different waves for it -- block id >> 8 & 3 -- was
  // measured in round 3: 49.04 against 49.08 ms per solve at four problems per CU, no effect, removed)
  constexpr int sw = 0;

  long long c_flood = 0, c_fail = 0, c_pull = 0, c_dfs = 0, q_phase = 0, q_fail = 0, q_rounds = 0, q_iter = 0, q_act = 0, q_frows = 0, q_prows = 0;
  long long pcf[5] = {0, 0, 0, 0, 0};
  long long pd[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = PROF ? (long long)__builtin_readcyclecounter() : 0;

  for (int i = tid; i < n; i += K4_T) { s.lx[i] = P.lx_init[i]; s.ly[i] = 0.0; s.match[i] = (unsigned short)K4_NONE; s.tln[i] = 0; }
  for (int i = tid; i < n * K4_CAP; i += K4_T) { s.tlc[i] = 0; s.tlo[i] = 0; }  // list slots are read unconditionally: keep them valid
  for (int w = tid; w < nw; w += K4_T) {
    unsigned all = ~0u;
    if (w == nw - 1 && (n & 31)) all = (1u << (n & 31)) - 1u;
    s.freey[w] = all; s.ovf[w] = 0u;
  }
  if (tid < SH_NUM) s.sh[tid] = 0;
  __syncthreads();
  // initial lists: every row once (R2); stx doubles as the list of all rows
  for (int i = tid; i < n; i += K4_T) s.stx[i] = (unsigned short)i;
  __syncthreads();
  k4_bulk<true, false, false>(s, s.stx, n, wave, lane);
  __syncthreads();

  int bad = 0;
  bool hazard = false;
  for (int root = 0; root < n && !bad && !hazard; ++root) {
    for (int i = tid; i < n; i += K4_T) s.slack[i] = K4_INF;
    for (int w = tid; w < nw; w += K4_T) s.pushed[w] = 0u;
    bool have_prev = false;
    // phase 0 floods from the root; a later phase expands the certified sets of the failed phase before it (R3')
    for (int w = tid; w < nw; w += K4_T) { s.visx[w] = 0u; s.visy[w] = 0u; }
    __syncthreads();
    if (wave == sw) {
      if (lane == 0) { s.stx[0] = (unsigned short)root; s.visx[root >> 5] = 1u << (root & 31); }
      __builtin_amdgcn_wave_barrier();
      int fq = 0;
      const long long tf0 = PROF ? (long long)__builtin_readcyclecounter() : 0;
      const bool fr = k4_flood<PROF>(s, 0, 1, INFINITY, lane, &fq, pcf);
      if (PROF) c_flood += (long long)__builtin_readcyclecounter() - tf0;  // (round-4 advisor: steps[7] stayed 0 after the flood moved out of the phase loop)
      if (lane == 0) { s.sh[SH_RES] = fr ? 1 : 0; s.sh[SH_QTF] = fq; }
    }
    for (int phase = 0;; ++phase) {
      if (PROF) q_phase++;
      __syncthreads();
      const bool free_found = s.sh[SH_RES] != 0;
      const int qt = s.sh[SH_QTF];
      const long long t1 = PROF ? (long long)__builtin_readcyclecounter() : 0;
      if (!free_found) {
        // ---- R4: failed phase
        if (PROF) { q_fail++; q_frows += qt; }
        k4_bulk<false, true, true>(s, s.stx, qt, wave, lane);  // rows that are new in this phase of the root
        double lm = INFINITY;
        for (int i = tid; i < qt; i += K4_T) lm = fmin(lm, s.lx[s.stx[i]]);
        lm = k4_wave_min(lm);
        if (lane == 0) s.red[wave] = lm;
        __syncthreads();
        const double lxmin = fmin(fmin(s.red[0], s.red[1]), fmin(s.red[2], s.red[3]));
        double dl = K4_INF;
        for (int y = tid; y < n; y += K4_T)
          if (!k4_bit(s.visy, y)) {
            const double s2 = fmin(s.slack[y], (lxmin + s.ly[y]) - bg);  // E4
            s.slack[y] = s2;
            dl = fmin(dl, s2);
          }
        dl = k4_wave_min(dl);
        if (lane == 0) s.red[4 + wave] = dl;
        for (int w = tid; w < nw; w += K4_T) {
          if (have_prev && (s.prevy[w] & ~s.visy[w])) s.sh[SH_HAZ] = 1;  // a visited column dropped out (R4; only after a flood from the root)
          s.prevy[w] = s.visy[w];
        }
        __syncthreads();
        dl = fmin(fmin(s.red[4], s.red[5]), fmin(s.red[6], s.red[7]));
        if (s.sh[SH_HAZ] || ((flags & 4) && phase == 0 && root == n / 2)) { hazard = true; break; }
        for (int i = tid; i < n; i += K4_T) {  // km.cpp:86-97
          if (k4_bit(s.visx, i)) s.lx[i] -= dl;
          if (k4_bit(s.visy, i)) s.ly[i] += dl;
          else s.slack[i] -= dl;
        }
        for (int w = tid; w < nw; w += K4_T) { s.pushed[w] = s.visx[w]; s.goody[w] = 0u; }  // goody: the certificates of this relabelling
        if (tid == 0) { s.sh[SH_QT] = qt; s.sh[SH_FREE] = 0; s.sh[SH_UNCERT] = 0; s.sh[SH_LROW] = 0; }
        __syncthreads();
        k4_revalidate(s, tid);                                        // rows that were not visited: entries in visited columns may have left
        k4_bulk<true, true, false, true>(s, s.stx, qt, wave, lane);  // visited rows: lists under the new labels, the minima of the next phase,
                                                                      // certificates of their tree edges, claims of their newly tight columns
        double lb = INFINITY;  // the smallest label of a visited row that can be background-tight at all
        for (int i = tid; i < qt; i += K4_T) {
          const double l = s.lx[s.stx[i]];
          if ((l - bg) < eps) lb = fmin(lb, l);
        }
        lb = k4_wave_min(lb);
        if (lane == 0) s.red[8 + wave] = lb;
        __syncthreads();
        const double lnew = fmin(fmin(s.red[8], s.red[9]), fmin(s.red[10], s.red[11]));
        for (int i = tid; i < qt; i += K4_T)
          if (s.lx[s.stx[i]] == lnew) s.sh[SH_LROW] = s.stx[i];  // any row that has it
        __syncthreads();
        const int lrow = s.sh[SH_LROW];
        for (int y = tid; y < n; y += K4_T) {
          const unsigned bit = 1u << (y & 31);
          if (s.prevy[y >> 5] & bit) {  // an old column: its tree edge has to be tight under the new labels (explicit: seen by the pass above)
            if (!(s.goody[y >> 5] & bit) && !(((s.lx[s.sty[y]] + s.ly[y]) - bg) < eps)) s.sh[SH_UNCERT] = 1;
          } else if (!(s.visy[y >> 5] & bit) && ((lnew + s.ly[y]) - bg) < eps) {  // T_L of the smallest label (INF: never)
            if (!(atomicOr(&s.visy[y >> 5], bit) & bit)) {
              s.sty[y] = (unsigned short)lrow;
              const int m = s.match[y];
              if (m == K4_NONE) s.sh[SH_FREE] = 1;
              else { s.stx[atomicAdd(&s.sh[SH_QT], 1)] = (unsigned short)m; atomicOr(&s.visx[m >> 5], 1u << (m & 31)); }
            }
          }
        }
        __syncthreads();
        const bool certified = s.sh[SH_UNCERT] == 0;
        if (!certified) {  // an ulp moved a tree edge across eps: flood from the root, rule R4's check decides
          __syncthreads();
          for (int w = tid; w < nw; w += K4_T) { s.visx[w] = 0u; s.visy[w] = 0u; }
          __syncthreads();
        }
        long long c_fl_in = 0;  // PROF: the re-flood of this failed phase (counted as flood time, not as failed-phase passes)
        if (wave == sw) {
          int fq = s.sh[SH_QT];
          bool fr = s.sh[SH_FREE] != 0;
          const long long tf0 = PROF ? (long long)__builtin_readcyclecounter() : 0;
          if (!certified) {
            if (lane == 0) { s.stx[0] = (unsigned short)root; s.visx[root >> 5] = 1u << (root & 31); }
            __builtin_amdgcn_wave_barrier();
            fr = k4_flood<PROF>(s, 0, 1, INFINITY, lane, &fq, pcf);
          } else if (!fr) {
            fr = k4_flood<PROF>(s, qt, fq, lnew, lane, &fq, pcf);
          }
          __builtin_amdgcn_wave_barrier();
          if (PROF) { c_fl_in = (long long)__builtin_readcyclecounter() - tf0; c_flood += c_fl_in; }
          if (lane == 0) { s.sh[SH_RES] = fr ? 1 : 0; s.sh[SH_QTF] = fq; }
        }
        have_prev = true;
        if (PROF) { q_prows += qt; c_fail += (long long)__builtin_readcyclecounter() - t1 - c_fl_in; }
        if (phase > 4 * n + 16) { bad = 2; break; }  // only reachable with non-finite weights
        continue;
      }
      // ---- R5 / R5': augmenting phase.  The search starts WITHOUT S (every column passes); S is computed when it first steps back.
      for (int w = tid; w < nw; w += K4_T) { s.visx[w] = 0u; s.visy[w] = 0u; s.goody[w] = ~0u; }
      __syncthreads();
      K4Dfs dfs;
      dfs.sp = 0; dfs.x = root; dfs.ystart = 0; dfs.cp = 0; dfs.cnext = 0;
      dfs.ck = __longlong_as_double(0x7ff8000000000000ll);  // NaN never matches
      if (wave == sw) {
        if (lane == 0) { s.stx[0] = (unsigned short)root; s.sty[0] = (unsigned short)K4_NONE; }
        __builtin_amdgcn_wave_barrier();
        const int r = k4_dfs<PROF>(s, dfs, true, lane, &q_iter, &q_act, pd);
        if (lane == 0) s.sh[SH_DFS] = r;
      }
      __syncthreads();
      const long long t1b = PROF ? (long long)__builtin_readcyclecounter() : 0;
      if (PROF) c_dfs += t1b - t1;
      long long t2 = t1b;
      if (s.sh[SH_DFS] == 2) {
      // ---- S by pull rounds (the search's stacks stx / sty stay as they are: everything S needs lives in the slack array's storage)
      for (int w = tid; w < nw; w += K4_T) { s.good[w] = 0u; s.goody[w] = s.freey[w]; }
      if (tid == 0) { s.sh[SH_CH0] = 0; s.sh[SH_CH1] = 0; s.sh[SH_CH2] = 0; s.sh[SH_NF] = 0; }
      // row -> column map of the matching, the list of flagged rows without hints and the pool of their tight columns: an augmenting phase
      // is the LAST phase of its root and the next root starts by resetting slack, so its 8 n bytes are dead from here to the end of the phase
      unsigned short* mx = reinterpret_cast<unsigned short*>(s.slack);
      unsigned short* flist = mx + n;
      unsigned short* pool = mx + 2 * (size_t)n;
      if (wave == 0) {  // the flagged rows without hints, ascending (without the pool they would all be members of S: measured, v4)
        int cnt = 0;
        for (int base = 0; base < n; base += 64) {
          const int x = base + lane;
          const bool f = x < n && k4_bit(s.ovf, x);
          const unsigned long long b = __ballot(f);
          if (f) flist[cnt + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)x;
          cnt += __popcll(b);
        }
        if (lane == 0) s.sh[SH_NF] = cnt;
      }
      for (int i = tid; i < n; i += K4_T) mx[i] = (unsigned short)K4_NONE;
      __syncthreads();
      for (int y = tid; y < n; y += K4_T) {
        const int m = s.match[y];
        if (m != K4_NONE) mx[m] = (unsigned short)y;
      }
      __syncthreads();
      const int nf = s.sh[SH_NF];
      const int region = nf > 0 ? min(64, (n + 2) / nf) : 0;  // u16 of the pool per flagged row: count + columns (region 1: every flagged row stays in S)
      if (nf > 0) {
        k4_pool_build(s, flist, pool, nf, region, wave, lane);
        __syncthreads();
      }
      for (int round = 0;; round++) {
        if (PROF) q_rounds++;
        // "something changed" flag of round r in slot r % 3: one barrier per round is enough -- the slot reset here (r + 1) is neither the one
        // this round sets nor the one a thread that is late out of the previous round may still be reading (r - 1)
        const int chslot[3] = {SH_CH0, SH_CH1, SH_CH2};
        if (tid == 0) s.sh[chslot[(round + 1) % 3]] = 0;
        bool ch = false;
        for (int base = tid; base < n; base += 2 * K4_T) {  // rows (two per thread and pass: the hints doubled the registers a row needs): background-tight to the best column of S, or a listed entry in S
          int xx[2], tn[2], lc[2][K4_HINT];
          unsigned gdw[2];
          double lxv[2];
#pragma unroll
          for (int k = 0; k < 2; k++) {
            xx[k] = base + k * K4_T;
            const int xc = min(xx[k], n - 1);
            gdw[k] = s.good[xc >> 5]; lxv[k] = s.lx[xc]; tn[k] = s.tln[xc];
#pragma unroll
            for (int e = 0; e < K4_CAP; e++) { lc[k][e] = s.tlc[xc * K4_CAP + e]; lc[k][K4_CAP + e] = s.tlo[xc * K4_CAP + e]; }
          }
          unsigned gyw[2][K4_HINT];
#pragma unroll
          for (int k = 0; k < 2; k++)
#pragma unroll
            for (int e = 0; e < K4_HINT; e++) gyw[k][e] = s.goody[lc[k][e] >> 5];  // (a listed row's slots 4..6 hold CSR offsets: < n, any word will do)
#pragma unroll
          for (int k = 0; k < 2; k++) {
            if (xx[k] >= n) continue;
            if ((gdw[k] >> (xx[k] & 31)) & 1u) {  // already in S (an earlier round, or the pool build): its column has to be there as well
              const int yo = mx[xx[k]];
              if (yo != K4_NONE && !k4_bit(s.goody, yo)) { atomicOr(&s.goody[yo >> 5], 1u << (yo & 31)); ch = true; }
              continue;
            }
            bool g = (lxv[k] - bg) < eps;  // background-tight to a free column: those are in S and their label is still 0, the smallest there is
            // listed entries are tight (R2); the hints of a flagged row are the columns that WERE tight when its list was built -- entries
            // only leave between two rebuilds of a row, so "a hint in S" is necessary for the row to be good (R5)
            const int t = tn[k] <= K4_HINT ? tn[k] : 0;
#pragma unroll
            for (int e = 0; e < K4_HINT; e++) g |= (int)(e < t) & (int)((gyw[k][e] >> (lc[k][e] & 31)) & 1u);
            if (g) {
              atomicOr(&s.good[xx[k] >> 5], 1u << (xx[k] & 31));
              const int yo = mx[xx[k]];
              if (yo != K4_NONE) atomicOr(&s.goody[yo >> 5], 1u << (yo & 31));
              ch = true;
            }
          }
        }
        for (int fb = 0; fb < nf; fb += 16) {  // flagged rows: a pooled tight column in S (16 lanes per row)
          const int i = fb + wave * 4 + (lane >> 4), lig = lane & 15;
          int x = -1, cnt = 0;
          if (i < nf) { x = flist[i]; cnt = pool[i * region]; }
          bool hit = false;
          if (x >= 0 && !k4_bit(s.good, x))
            for (int e = lig; e < cnt; e += 16) hit |= k4_bit(s.goody, pool[i * region + 1 + e]);
          const unsigned gb = (unsigned)(__ballot(hit) >> (lane & 48)) & 0xffffu;
          if (gb && lig == 0) {
            atomicOr(&s.good[x >> 5], 1u << (x & 31));
            const int yo = mx[x];
            if (yo != K4_NONE) atomicOr(&s.goody[yo >> 5], 1u << (yo & 31));
            ch = true;
          }
        }
        if (ch) s.sh[chslot[round % 3]] = 1;
        __syncthreads();
        if (!s.sh[chslot[round % 3]]) break;
        if (round >= 40) {  // give up pruning for this phase: any superset of good is valid (R5)
          for (int w = tid; w < nw; w += K4_T) { s.goody[w] = ~0u; s.good[w] = ~0u; }
          break;
        }
      }
      __syncthreads();
      t2 = PROF ? (long long)__builtin_readcyclecounter() : 0;
      if (PROF) c_pull += t2 - t1b;
      // ... then the search goes on under S, from its deepest good frame (wave 0)
      if (wave == sw) {
        int r = 0;
        if (k4_dfs_unwind(s, dfs, lane)) r = k4_dfs<PROF>(s, dfs, false, lane, &q_iter, &q_act, pd);
        if (lane == 0) s.sh[SH_DFS] = r;
      }
      __syncthreads();
      if (PROF) c_dfs += (long long)__builtin_readcyclecounter() - t2;
      }
      if (s.sh[SH_DFS] != 1 && tid == 0) s.sh[SH_BAD] = 3;
      __syncthreads();
      if (s.sh[SH_BAD]) bad = s.sh[SH_BAD];
      break;
    }
  }
  __syncthreads();
  if (hazard) {
    if (tid == 0) s.sh[SH_BAD] = k4_literal(s, P);
    __syncthreads();
    bad = s.sh[SH_BAD];
  }
  for (int i = tid; i < n; i += K4_T) P.match_out[i] = s.match[i] == K4_NONE ? -1 : (int)s.match[i];
  if (tid == 0 && lstat) {  // launch record: first start, last end, sum and maximum of the solve times, solves
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), dt = t1 - t_wall0;
    atomicMax(&lstat[0], (1ull << 62) - t_wall0);
    atomicMax(&lstat[1], t1);
    atomicAdd(&lstat[2], dt);
    atomicMax(&lstat[3], dt);
    atomicAdd(&lstat[4], 1ull);
  }
  if (tid == 0) {
    if (bad && P.status) *P.status = bad;
    // diagnostics: solves that went through the literal fallback of rule R4, counted above the status bits (a persistent slot's pair keeps
    // one status word for all its iterations; the host reads the count with the batch's results -- ghicp_ctx_loop_hazards)
    if (hazard && P.status) atomicAdd(P.status, 0x10000);
    if (P.steps) {
      P.steps[0] = q_act;
      if (PROF) {
        P.steps[1] = q_phase; P.steps[2] = q_fail; P.steps[3] = q_rounds; P.steps[4] = q_iter; P.steps[5] = q_frows; P.steps[6] = q_prows;
        P.steps[7] = c_flood; P.steps[8] = c_fail; P.steps[9] = c_pull; P.steps[10] = c_dfs;
        P.steps[11] = (long long)__builtin_readcyclecounter() - t_begin; P.steps[12] = hazard ? 1 : 0;
        for (int k = 0; k < 5; k++) P.steps[13 + k] = pcf[k];
        for (int k = 0; k < 10; k++) P.steps[18 + k] = pd[k];
      }
    }
  }
}


}  // namespace
