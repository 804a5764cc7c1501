// Kuhn-Munkres, second-generation kernel (gfx950): the reference's exact traversal (src/km.cpp:13-126)
// on a SPARSE + BACKGROUND representation of the weight matrix, all solver state in LDS, one wave per
// problem (wave-synchronous: no barriers on the dependency chain), many problems per launch.
//
// GH-ICP's KM matrix (src/ghicp_reg.cpp:348-365) is -penalty everywhere except where CD < penalty.
// Rows are therefore stored as "background bg = -penalty + sorted explicit entries (col, w > bg)".
// What makes the emulation exact (all by monotonicity of IEEE-754 rounding; ly >= 0 always):
//   E1  for a background entry the reference evaluates d = fl(fl(lx+ly[y]) - bg): same expression here,
//       bg comes from a register instead of memory.
//   E2  d >= fl(lx - bg) for every background entry, so a row whose fl(lx-bg) >= eps has no tight
//       background edge and only its explicit entries are looked at.
//   E3  for an explicit entry w >= bg: fl(t - w) <= fl(t - bg), so "background-tight" implies tight;
//       the first tight unvisited y of a row is min(first background-tight y, first tight explicit y).
//   E4  in a FAILED phase every visited row ends up scanning all its columns, so for an unvisited y
//       slack[y] = min(old, min_x d(x,y)); over background entries the minimum is attained by the visited
//       row with the smallest lx (monotone), hence ONE O(n) pass with lxmin replaces |visx| row sweeps;
//       explicit entries update slack individually.  In a SUCCESSFUL phase slack is dead (reset per root).
//   E5  the DFS itself (explicit stack, lowest tight unvisited y first) decides the augmenting path: it is
//       emulated step by step, one wave-parallel row probe per findpath() call.
#include "ctx.h"
#include "devmath.h"

#include <climits>
#include <cstdlib>

#include "km_prob.h"

size_t gh_km2_lds_bytes(int n) { return (size_t)n * 26 + 16 + 2 * (size_t)((n + 31) / 32) * 4 + 64; }

namespace {

constexpr double KM_INF2 = 1000.0;  // km.cpp:42

__device__ inline bool bit_get(const unsigned* b, int i) { return (b[i >> 5] >> (i & 31)) & 1u; }

// lowest y >= y0 whose bit is clear, or n
__device__ inline int first_clear(const unsigned* __restrict__ bits, int y0, int n, int nw, int lane) {
  const int w0 = y0 >> 5;
  for (int wb = w0; wb < nw; wb += 64) {
    const int w = wb + lane;
    unsigned inv = 0u;
    if (w < nw) {
      inv = ~bits[w];
      if (w == w0) inv &= ~0u << (y0 & 31);
      if (w == nw - 1 && (n & 31)) inv &= (1u << (n & 31)) - 1u;
    }
    const unsigned long long b = __ballot(inv != 0u);
    if (b) {
      const int l = (int)__ffsll((long long)b) - 1;
      const unsigned iv = (unsigned)__builtin_amdgcn_readlane((int)inv, l);
      return (wb + l) * 32 + (__ffs((int)iv) - 1);
    }
  }
  return n;
}


// First unvisited background-tight column in [start, limit), 64 columns per round.  A lone wave issues one instruction
// every ~4 cycles, so the scan is written for the fewest instructions on the common path (a hit in the first block, which
// the E7 pointer makes the rule): the two LDS reads are unconditional (clamped index) and issued together, the verdict is
// formed with bitwise operators, one ballot decides.  (Measured alternatives, all exact, all slower on the cfg2 matrices:
// 4 x 64 columns per round with the eight reads issued together; a 16 x 64 full sweep; short-circuit evaluation.)
__device__ inline int bg_scan(const unsigned* __restrict__ visy, const double* __restrict__ ly, double lxv, double bg, double eps, int start,
                              int limit, int lane, long long* groups = nullptr) {
  for (int y0 = start; y0 < limit; y0 += 64) {
    if (groups) ++*groups;
    const int y = y0 + lane, yc = min(y, limit - 1);
    const unsigned vw = visy[yc >> 5];
    const double lv = ly[yc];
    const unsigned t = (unsigned)(y < limit) & (unsigned)(((vw >> (yc & 31)) & 1u) == 0u) & (unsigned)(((lxv + lv) - bg) < eps);
    const unsigned long long b = __ballot(t != 0u);
    if (b) return y0 + (int)__ffsll((long long)b) - 1;
  }
  return INT_MAX;
}

typedef __attribute__((address_space(1))) const int* gc_int;
typedef __attribute__((address_space(1))) const double* gc_f64;
typedef __attribute__((address_space(1))) unsigned long long* g_u64;

// LDS per problem: lx, ly (f64), rptr (u32), match / stack x / stack y (u16), two bitmaps = 26.25 B per row.
// slack lives in global memory (L2): explicit entries hit it with fire-and-forget 64-bit atomic minima (d >= eps > 0,
// so the IEEE bit pattern orders like the value), the end of a failed phase reads it with L1-bypassing loads.
template <bool PROF, bool SWEEP>
__global__ __launch_bounds__(64) void k_km2(const Km2Problem* __restrict__ probs, int flags) {
  const Km2Problem P = probs[blockIdx.x];
  if (P.n <= 0 || (P.done && *P.done)) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = P.n, nw = (n + 31) / 32, lane = threadIdx.x;
  double* lx = (double*)smem;
  double* ly = lx + n;
  unsigned* rptr = (unsigned*)(ly + n);
  unsigned* visx = rptr + ((n + 2) & ~1);
  unsigned* visy = visx + nw;
  unsigned short* match = (unsigned short*)(visy + nw);
  unsigned short* stx = match + n;
  unsigned short* sty = stx + n;
  const gc_int cols = (gc_int)P.cols;
  const gc_f64 vals = (gc_f64)P.vals;
  const g_u64 slack = (g_u64)P.slack;
  const double bg = P.bg, eps = P.eps;
  const int NONE = 0xFFFF;
  const unsigned long long INF_BITS = (unsigned long long)__double_as_longlong(KM_INF2);

  for (int i = lane; i < n; i += 64) { lx[i] = P.lx_init[i]; ly[i] = 0.0; match[i] = (unsigned short)NONE; rptr[i] = P.row_ptr[i]; }
  if (lane == 0) rptr[n] = P.row_ptr[n];
  __syncthreads();

  long long nsteps = 0;
  // profiling (flags & 2, GHICP_KM_STATS): step mix and cycle split of the DFS
  constexpr bool prof = PROF;  // GHICP_KM_STATS: step mix and cycle split of the DFS (compiled out otherwise)
  long long q_bg = 0, q_grp = 0, q_hit = 0, q_succ = 0, q_fail = 0, q_failph = 0, q_csr = 0, c_csr = 0, c_bg = 0, c_move = 0, c_relabel = 0, q_push = 0;
  int bad = 0;
  for (int root = 0; root < n && !bad; ++root) {
    for (int i = lane; i < n; i += 64) slack[i] = INF_BITS;
    __threadfence_block();
    for (int phase = 0;; ++phase) {
      for (int i = lane; i < nw; i += 64) { visx[i] = 0u; visy[i] = 0u; }
      __syncthreads();
      if (lane == 0) { stx[0] = (unsigned short)root; sty[0] = (unsigned short)NONE; visx[root >> 5] = 1u << (root & 31); }
      __syncthreads();
      int sp = 0, x = root, ystart = 0;
      double lxmin = lx[root];
      bool ok = false;
      double ck = __longlong_as_double(0x7ff8000000000000ll);  // E7 cache, one entry per lane: NaN never matches
      int cp = 0, cnext = 0;
      const long long steps0 = nsteps;
      // ---- E10 (template parameter SWEEP, selected by GHICP_KM_SWEEP=1; prototyped and fuzzed in oracle/km_model.inc): an order-free sweep decides the
      // fate of the phase before any DFS.  A failed findpath() visits exactly the set reachable from the root in the tight
      // graph, in whatever order (E1); so the wave floods that set breadth-first -- 64 entries / 64 columns per iteration, the
      // queue of rows in the (still unused) stack array -- and when no free column turns up, its visited bits, slack minima
      // and minimum visited label ARE the failed phase: straight to the relabelling.  When a free column turns up the sweep
      // is abandoned (its slack minima are dead: the phase augments and the next root starts from fresh slack) and the
      // order-dependent DFS below runs as before.  On mid-run matrices 60 % of all activations belong to failed phases.
      bool swept_failed = false;
      if constexpr (SWEEP) {
        int qh = 0, qt = 1;  // stx[0] == root
        bool free_found = false;
        double sk = __longlong_as_double(0x7ff8000000000000ll);  // labels whose background-tight set has been flooded (one per lane)
        int snext = 0;
        while (qh < qt && !free_found) {
          const int xr = (int)stx[qh];
          qh++;
          const double lxr = lx[xr];
          lxmin = fmin(lxmin, lxr);
          const unsigned cb = rptr[xr], ce = rptr[xr + 1];
          for (unsigned c0 = cb; c0 < ce && !free_found; c0 += 64) {
            const unsigned c = c0 + lane, cc = min(c, ce - 1u);
            const int col = cols[cc];
            const double wv = vals[cc];
            const double d = (lxr + ly[col]) - wv;
            const bool in = c < ce, td = d < eps;
            if (in & !td) __hip_atomic_fetch_min(&slack[col], (unsigned long long)__double_as_longlong(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool fresh = in & td & !bit_get(visy, col);
            int m = NONE;
            if (fresh) m = match[col];
            if (__ballot(fresh && m == NONE)) { free_found = true; break; }
            const unsigned long long fb = __ballot(fresh);
            if (fresh) {  // a matched column is entered once, so its owner is enqueued once
              atomicOr(&visy[col >> 5], 1u << (col & 31));
              atomicOr(&visx[m >> 5], 1u << (m & 31));
              stx[qt + __popcll(fb & ((1ull << lane) - 1ull))] = (unsigned short)m;
            }
            qt += __popcll(fb);
            __builtin_amdgcn_wave_barrier();
          }
          if (!free_found && (lxr - bg) < eps && !__ballot(sk == lxr)) {  // flood T_L once per label: nothing of it is left afterwards
            for (int y0 = 0; y0 < n && !free_found; y0 += 64) {
              const int y = y0 + lane, yc = min(y, n - 1);
              const bool fresh = (y < n) & !bit_get(visy, yc) & (((lxr + ly[yc]) - bg) < eps);
              int m = NONE;
              if (fresh) m = match[y];
              if (__ballot(fresh && m == NONE)) { free_found = true; break; }
              const unsigned long long fb = __ballot(fresh);
              if (fresh) {
                atomicOr(&visy[y >> 5], 1u << (y & 31));
                atomicOr(&visx[m >> 5], 1u << (m & 31));
                stx[qt + __popcll(fb & ((1ull << lane) - 1ull))] = (unsigned short)m;
              }
              qt += __popcll(fb);
              __builtin_amdgcn_wave_barrier();
            }
            if (lane == snext) sk = lxr;
            snext = (snext + 1) & 63;
          }
        }
        if (!free_found) swept_failed = true;
        else {  // restore the start state of the phase for the DFS
          for (int i = lane; i < nw; i += 64) { visx[i] = 0u; visy[i] = 0u; }
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) { stx[0] = (unsigned short)root; sty[0] = (unsigned short)NONE; visx[root >> 5] = 1u << (root & 31); }
          __builtin_amdgcn_wave_barrier();
          lxmin = lx[root];
        }
      }
      // E8: the first 64 explicit entries of a row live in registers (one entry per lane), for the row being scanned
      // (r*) and for the frame below it (p*).  A push loads the child's row while the stack is being updated, a pop takes
      // the frame below from registers and starts loading the one below that: the L2 round trip of the CSR row is no
      // longer on the critical path of the step.
      unsigned rb, re, pb = 0, pe = 0;
      int rcol = 0, pcol = 0;
      double rval = 0.0, pval = 0.0;
      int par_sp = -1, par_x = 0, par_ys = 0;
#define KM2_LOAD_ROW(XR, B, E, C, V)                                   \
  do {                                                                 \
    B = rptr[XR]; E = rptr[(XR) + 1];                                  \
    if (E > B) { const unsigned cc_ = min(B + (unsigned)lane, E - 1u); C = cols[cc_]; V = vals[cc_]; } \
  } while (0)
      KM2_LOAD_ROW(x, rb, re, rcol, rval);
      for (; !swept_failed;) {  // one iteration == one findpath() activation or resumption (km.cpp:13-37)
        nsteps++;
        const long long tq0 = prof ? (long long)__builtin_readcyclecounter() : 0;
        const double lxv = lx[x];
        lxmin = fmin(lxmin, lxv);
        int best = INT_MAX;
        // ---- E9: marching through a chain of label-sharing rows without explicit entries, up to 64 findpath() activations
        // per wave iteration.  Such a row x (label L, background-tight) picks the lowest unvisited column of
        // T_L = {y : fl(fl(L + ly[y]) - bg) < eps}; if that column's owner c is again a row without explicit entries and
        // with lx[c] == L, findpath(c) starts from column 0, finds every lower member of T_L visited and picks the NEXT
        // unvisited member -- and so on down the chain until a column is free (augment) or its owner is a different kind
        // of row (generic path).  The 64 lanes test a window of candidate columns and their owners at once; everything up
        // to the first stopping candidate is retired exactly as the recursion would: visy/visx bits, one stack frame per
        // pick.  Rows of this kind have no explicit entries, hence no slack traffic, and L is already in lxmin.
        if (re == rb && (lxv - bg) < eps && !(flags & 1)) {
          const unsigned long long hit = __ballot(ck == lxv);
          int slot, p = ystart;  // everything below ystart was rejected by this row's earlier scans: a valid E7 pointer too
          if (hit) {
            slot = (int)__ffsll((long long)hit) - 1;
            p = max(p, __builtin_amdgcn_readlane(cp, slot));
          } else {
            slot = cnext;
            cnext = (cnext + 1) & 63;
            if (lane == slot) ck = lxv;
          }
          int outcome = 0;  // 0: x has no unvisited background-tight column left (falls through to the pop below)
          while (p < n) {
            const int y = p + lane, yc = min(y, n - 1);
            const unsigned vw = visy[yc >> 5];
            const double lv = ly[yc];
            const bool cand = (y < n) & (((vw >> (yc & 31)) & 1u) == 0u) & (((lxv + lv) - bg) < eps);
            const unsigned long long b = __ballot(cand);
            if (!b) { p += 64; continue; }
            int m = NONE;
            bool cont = false;
            if (cand) {
              m = match[y];
              if (m != NONE) cont = (rptr[m + 1] == rptr[m]) & (lx[m] == lxv);
            }
            const unsigned long long stop = __ballot(cand && !cont);
            const int jstar = stop ? (int)__ffsll((long long)stop) - 1 : 63;
            const unsigned long long R = b & (~0ull >> (63 - jstar));  // the picks of this window, in column order
            const unsigned long long below = R & ((1ull << lane) - 1ull);
            const int rank = __popcll(below);
            const int pm = __shfl(m, below ? 63 - __clzll((long long)below) : 0, 64);
            if ((R >> lane) & 1ull) {
              atomicOr(&visy[y >> 5], 1u << (y & 31));
              sty[sp + rank] = (unsigned short)y;
              if (rank > 0) stx[sp + rank] = (unsigned short)pm;  // the row that picked this column (frame sp holds x)
              if (m != NONE) atomicOr(&visx[m >> 5], 1u << (m & 31));
            }
            const int k = __popcll(R), lastlane = 63 - __clzll((long long)R);
            const int mlast = __builtin_amdgcn_readlane(m, lastlane);
            nsteps += k - 1;  // the activation of x itself was counted above
            sp += k - 1;      // frame of the row that made the last pick
            p += lastlane + 1;
            if (mlast == NONE) { outcome = 1; break; }
            sp++;
            if (lane == 0) { stx[sp] = (unsigned short)mlast; sty[sp] = (unsigned short)NONE; }
            x = mlast; ystart = 0;
            par_sp = -1;
            if (stop) { outcome = 2; break; }  // the owner of the last pick is not part of the chain
            nsteps++;                          // ... it is: its activation continues the march
          }
          if (lane == slot) cp = p;
          __builtin_amdgcn_wave_barrier();
          if (outcome == 1) {  // augment (km.cpp:26-29)
            for (int f = lane; f <= sp; f += 64) match[sty[f]] = stx[f];
            ok = true;
            break;
          }
          if (outcome == 2) {
            KM2_LOAD_ROW(x, rb, re, rcol, rval);
            continue;
          }
          // outcome 0: fall through with best == INT_MAX -> pop (x may be a row reached by the march: its frame is sp)
          re = rb;
        }
        // ---- explicit entries of row x (E3); non-tight ones feed slack (km.cpp:33).  Loads are unconditional (clamped
        // entry index) and the verdict uses bitwise operators so that the two LDS gathers are issued together.
        if (re > rb) {
          if (prof) q_csr++;
          const bool in = rb + (unsigned)lane < re;
          const double lyc = ly[rcol];
          const unsigned vwc = visy[rcol >> 5];
          const double d = (lxv + lyc) - rval;
          const bool td = d < eps;
          const bool tight = (int)in & (int)td & (int)(rcol >= ystart) & (int)(((vwc >> (rcol & 31)) & 1u) == 0u);
          // a resumed frame (ystart > 0) already sent this block's slack minima when it was first activated
          if (ystart == 0 && (in & !td)) __hip_atomic_fetch_min(&slack[rcol], (unsigned long long)__double_as_longlong(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long b0 = __ballot(tight);
          if (b0) best = __builtin_amdgcn_readlane(rcol, (int)__ffsll((long long)b0) - 1);
          else {
            for (unsigned c0 = rb + 64u; c0 < re; c0 += 64) {  // rows with more than 64 explicit entries
              if (prof) q_csr++;
              const unsigned c = c0 + lane, cc = min(c, re - 1u);
              const int colc = cols[cc];
              const double wv = vals[cc];
              const double ly2 = ly[colc];
              const unsigned vw2 = visy[colc >> 5];
              const double d2 = (lxv + ly2) - wv;
              const bool in2 = c < re, td2 = d2 < eps;
              const bool tight2 = (int)in2 & (int)td2 & (int)(colc >= ystart) & (int)(((vw2 >> (colc & 31)) & 1u) == 0u);
              if (in2 & !td2) __hip_atomic_fetch_min(&slack[colc], (unsigned long long)__double_as_longlong(d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const unsigned long long b2 = __ballot(tight2);
              if (b2) { best = __builtin_amdgcn_readlane(colc, (int)__ffsll((long long)b2) - 1); break; }
            }
          }
        }
        const long long tq1 = prof ? (long long)__builtin_readcyclecounter() : 0;
        // ---- background entries (E1, E2).  E7: whether a column is background-tight for a row depends on the row only
        // through the VALUE of lx, ly is fixed within a phase and visited columns stay visited, so "every column below p
        // is visited or not background-tight for this lx" survives until the phase ends.  One such p per distinct lx
        // value lives in a lane (key ck, pointer cp): rows that share a label -- whole chains of them do, and each one
        // re-scans when its child returns -- resume where the last scan for that label stopped.
        if ((lxv - bg) < eps && (re > rb || (flags & 1))) {  // rows without explicit entries were handled by the march
          const unsigned long long hit = __ballot(ck == lxv);
          int slot, p0 = 0;
          if (hit) {
            slot = (int)__ffsll((long long)hit) - 1;
            p0 = __builtin_amdgcn_readlane(cp, slot);
          } else {
            slot = cnext;
            cnext = (cnext + 1) & 63;
            if (lane == slot) { ck = lxv; cp = 0; }
          }
          const int lim = min(n, best);
          if (prof) { q_bg++; q_hit += hit ? 1 : 0; }
          const int yb = bg_scan(visy, ly, lxv, bg, eps, max(ystart, p0), lim, lane, prof ? &q_grp : nullptr);
          if (ystart <= p0 && lane == slot) cp = (yb != INT_MAX) ? yb : max(p0, lim);
          best = min(best, yb);
        }
        const long long tq2 = prof ? (long long)__builtin_readcyclecounter() : 0;
        if (best != INT_MAX) {
          if (prof) q_push++;
          const int ystar = best;
          const int m = match[ystar];
          if (m == NONE) {  // augment: match[y] = x on every level of the recursion (km.cpp:26-29)
            if (lane == 0) sty[sp] = (unsigned short)ystar;
            __builtin_amdgcn_wave_barrier();
            for (int f = lane; f <= sp; f += 64) match[sty[f]] = stx[f];
            ok = true;
            break;
          }
          // descend: this frame moves to the p* registers, the child's row is requested before the bookkeeping
          pb = rb; pe = re; pcol = rcol; pval = rval;
          par_sp = sp; par_x = x; par_ys = ystar + 1;
          KM2_LOAD_ROW(m, rb, re, rcol, rval);
          if (lane == 0) {
            atomicOr(&visy[ystar >> 5], 1u << (ystar & 31));  // ds_or without a return value: no read-modify-write round trip
            sty[sp] = (unsigned short)ystar;
            stx[sp + 1] = (unsigned short)m;
            sty[sp + 1] = (unsigned short)NONE;
            atomicOr(&visx[m >> 5], 1u << (m & 31));
          }
          sp++;
          x = m; ystart = 0;
        } else {
          sp--;
          if (sp < 0) break;
          if (sp == par_sp) {
            x = par_x; ystart = par_ys;
            rb = pb; re = pe; rcol = pcol; rval = pval;
          } else {
            x = stx[sp]; ystart = (int)sty[sp] + 1;
            KM2_LOAD_ROW(x, rb, re, rcol, rval);
          }
          par_sp = -1;
          if (sp > 0) {  // request the frame below: if this one fails too, its data is already here
            par_sp = sp - 1;
            par_x = stx[sp - 1];
            par_ys = (int)sty[sp - 1] + 1;
            KM2_LOAD_ROW(par_x, pb, pe, pcol, pval);
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (prof) { const long long tq3 = (long long)__builtin_readcyclecounter(); c_csr += tq1 - tq0; c_bg += tq2 - tq1; c_move += tq3 - tq2; }
      }
#undef KM2_LOAD_ROW
      __syncthreads();
      if (ok) { if (prof) q_succ += nsteps - steps0; break; }
      if (prof) { q_fail += nsteps - steps0; q_failph++; }
      const long long tr0 = prof ? (long long)__builtin_readcyclecounter() : 0;
      // ---- failed phase: deferred background slack (E4) + relabel (km.cpp:80-98)
      __threadfence_block();  // every slack atomic of this phase has reached L2
      double dl = KM_INF2;
      for (int y = lane; y < n; y += 64) {
        if (!bit_get(visy, y)) {
          const double cur = __longlong_as_double((long long)__hip_atomic_load(&slack[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          const double s2 = fmin(cur, (lxmin + ly[y]) - bg);
          dl = fmin(dl, s2);
          __hip_atomic_store(&slack[y], (unsigned long long)__double_as_longlong(s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dl = fmin(dl, __shfl_xor(dl, o, 64));
      __threadfence_block();
      for (int i = lane; i < n; i += 64) {
        if (bit_get(visx, i)) lx[i] -= dl;
        if (bit_get(visy, i)) ly[i] += dl;
        else {
          const double cur = __longlong_as_double((long long)__hip_atomic_load(&slack[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          __hip_atomic_store(&slack[i], (unsigned long long)__double_as_longlong(cur - dl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __threadfence_block();
      __syncthreads();
      if (prof) c_relabel += (long long)__builtin_readcyclecounter() - tr0;
      if (phase > 4 * n + 16) { bad = 2; break; }  // only reachable with non-finite weights
    }
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) P.match_out[i] = match[i] == NONE ? -1 : (int)match[i];
  if (lane == 0) {
    if (bad && P.status) *P.status = bad;
    if (P.steps) {
      P.steps[0] = nsteps;
      if (prof) {
        P.steps[1] = q_succ; P.steps[2] = q_fail; P.steps[3] = q_failph; P.steps[4] = q_bg; P.steps[5] = q_grp; P.steps[6] = q_hit; P.steps[7] = q_csr;
        P.steps[8] = c_csr; P.steps[9] = c_bg; P.steps[10] = c_move; P.steps[11] = c_relabel; P.steps[12] = q_push;
      }
    }
  }
}

}  // namespace

// =====================================================================================================
// Third-generation solver: same traversal, but the DFS never touches global memory.  Every row keeps in LDS a
// small list (<= 2 entries, ascending column) that is a SUPERSET of its currently tight explicit entries:
//   L1  an explicit entry (x,y) that is not tight can only become tight when lx[x] drops, i.e. after a failed
//       phase that visited x -- exactly the rows whose lists are rebuilt at the end of that phase;
//   L2  listed entries are re-tested with the fresh expression fl(fl(lx+ly[y]) - w) < eps at every use, so a
//       stale member is harmless;
//   L3  (E4 again) slack contributions of explicit entries are order independent within a failed phase and dead
//       in a successful one, so they are applied once, at the end of a failed phase, for the visited rows.
// A row with more than 2 tight explicit entries is flagged and scanned from its CSR row (global) instead.
// LDS: 51.25 B per row/column -> n <= ~3100 in 160 KB; 16-bit match/stack entries (n <= 65534).
constexpr int TL_CAP = 2;

size_t gh_km3_lds_bytes(int n) {
  return (size_t)n * (24 + 16 + 2 * 3 + 2 * TL_CAP) + (size_t)((n + 3) / 4) * 4 + 2 * (size_t)((n + 31) / 32) * 4 + 128;
}

namespace {

__global__ __launch_bounds__(64) void k_km3(const Km2Problem* __restrict__ probs) {
  const Km2Problem P = probs[blockIdx.x];
  if (P.n <= 0 || (P.done && *P.done)) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = P.n, nw = (n + 31) / 32, lane = threadIdx.x;
  double* lx = (double*)smem;
  double* ly = lx + n;
  double* slack = ly + n;
  double* tlv = slack + n;                                   // [n][TL_CAP]
  unsigned* visx = (unsigned*)(tlv + (size_t)n * TL_CAP);
  unsigned* visy = visx + nw;
  unsigned short* match = (unsigned short*)(visy + nw);
  unsigned short* stx = match + n;
  unsigned short* sty = stx + n;
  unsigned short* tlc = sty + n;                             // [n][TL_CAP]
  unsigned char* tln = (unsigned char*)(tlc + (size_t)n * TL_CAP);
  const double bg = P.bg, eps = P.eps;
  const int NONE = 0xFFFF;

  for (int i = lane; i < n; i += 64) { lx[i] = P.lx_init[i]; ly[i] = 0.0; match[i] = (unsigned short)NONE; }
  __syncthreads();

  // (re)build the tight list of row x under the current labels; wave-uniform x
  auto build_list = [&](int x) {
    const unsigned cb = P.row_ptr[x], ce = P.row_ptr[x + 1];
    const double lxv = lx[x];
    int cnt = 0;
    for (unsigned c0 = cb; c0 < ce; c0 += 64) {
      const unsigned c = c0 + lane;
      bool t = false;
      int col = 0;
      double val = 0;
      if (c < ce) { col = P.cols[c]; val = P.vals[c]; t = ((lxv + ly[col]) - val) < eps; }
      const unsigned long long b = __ballot(t);
      if (t) {
        const int r = cnt + __popcll(b & ((1ull << lane) - 1ull));
        if (r < TL_CAP) { tlc[x * TL_CAP + r] = (unsigned short)col; tlv[x * TL_CAP + r] = val; }
      }
      cnt += __popcll(b);
    }
    if (lane == 0) tln[x] = (unsigned char)(cnt > TL_CAP ? (0x80 | TL_CAP) : cnt);
  };
  for (int x = 0; x < n; x++) build_list(x);
  __syncthreads();

  long long nsteps = 0, n_over = 0, n_flat = 0, n_failph = 0, n_failrows = 0, cyc_dfs = 0, cyc_fail = 0, cyc_init = 0, cycA = 0, cycB = 0, cycC = 0, cycD = 0, cycE = 0;
  const long long t_start = __builtin_readcyclecounter();
  int bad = 0;
  for (int root = 0; root < n && !bad; ++root) {
    for (int i = lane; i < n; i += 64) slack[i] = KM_INF2;
    for (int phase = 0;; ++phase) {
      const long long t0 = __builtin_readcyclecounter();
      for (int i = lane; i < nw; i += 64) { visx[i] = 0u; visy[i] = 0u; }
      __syncthreads();
      if (lane == 0) { stx[0] = (unsigned short)root; sty[0] = (unsigned short)NONE; visx[root >> 5] = 1u << (root & 31); }
      __syncthreads();
      int sp = 0, x = root, ystart = 0;
      int par_sp = -1, par_x = 0, par_ys = 0;  // register copy of the frame we just descended from
      double lxmin = lx[root];
      bool ok = false;
      for (;;) {  // one iteration == one findpath() activation or resumption (km.cpp:13-37)
        nsteps++;
        const long long tA0 = __builtin_readcyclecounter();
        // ---- LDS round trip A: the row record and the visited bitmap (independent addresses, issued together)
        const double lxv = lx[x];
        const int tn = tln[x];
        int colL = 0xFFFF;
        double valL = 0.0;
        if (lane < TL_CAP) { colL = tlc[x * TL_CAP + lane]; valL = tlv[x * TL_CAP + lane]; }
        lxmin = fmin(lxmin, lxv);
        const long long tA1 = __builtin_readcyclecounter(); cycA += tA1 - tA0;
        // ---- LDS round trip B: every gather the decision needs
        const bool act = lane < (tn & 0x7f);
        double lyc = 0.0, lyu = 0.0;
        unsigned visw = ~0u;
        int mL = NONE, mU = NONE;
        if (act) { lyc = ly[colL]; visw = visy[colL >> 5]; mL = match[colL]; }
        const bool pen_tight = (lxv - bg) < eps;  // E2
        (void)lyu; (void)mU;
        const long long tB1 = __builtin_readcyclecounter(); cycB += tB1 - tA1;
        // ---- decide
        const bool tl = act && colL >= ystart && !((visw >> (colL & 31)) & 1u) && ((lxv + lyc) - valL) < eps;
        const unsigned long long bl = __ballot(tl);
        int best = INT_MAX, mbest = NONE;
        if (bl) {
          const int l = (int)__ffsll((long long)bl) - 1;
          best = __builtin_amdgcn_readlane(colL, l);
          mbest = __builtin_amdgcn_readlane(mL, l);
        } else if (tn & 0x80) {  // more tight explicit entries than the list holds: scan the CSR row beyond it
          n_over++;
          const int from = max(ystart, __builtin_amdgcn_readlane(colL, TL_CAP - 1) + 1);
          const unsigned cb = P.row_ptr[x], ce = P.row_ptr[x + 1];
          for (unsigned c0 = cb; c0 < ce; c0 += 64) {
            const unsigned c = c0 + lane;
            int col = INT_MAX;
            bool tight = false;
            if (c < ce) {
              col = P.cols[c];
              tight = col >= from && !bit_get(visy, col) && ((lxv + ly[col]) - P.vals[c]) < eps;
            }
            const unsigned long long b = __ballot(tight);
            if (b) { best = __builtin_amdgcn_readlane(col, (int)__ffsll((long long)b) - 1); mbest = match[best]; break; }
          }
        }
        const long long tC0 = __builtin_readcyclecounter();
        cycE += tC0 - tB1;
        if (pen_tight) {  // background entries (E1-E3): lowest unvisited background-tight column below `best`
          n_flat++;
          const int yb = bg_scan(visy, ly, lxv, bg, eps, ystart, min(n, best), lane);
          if (yb < best) { best = yb; mbest = match[yb]; }
        }
        const long long tC1 = __builtin_readcyclecounter(); cycC += tC1 - tB1;
        if (best != INT_MAX) {
          const int ystar = best;
          if (lane == 0) { visy[ystar >> 5] |= 1u << (ystar & 31); sty[sp] = (unsigned short)ystar; }
          if (mbest == NONE) {  // augment: match[y] = x on every level of the recursion (km.cpp:26-29)
            __builtin_amdgcn_wave_barrier();
            for (int f = lane; f <= sp; f += 64) match[sty[f]] = stx[f];
            ok = true;
            break;
          }
          par_sp = sp; par_x = x; par_ys = ystar + 1;
          sp++;
          if (lane == 0) { stx[sp] = (unsigned short)mbest; sty[sp] = NONE; visx[mbest >> 5] |= 1u << (mbest & 31); }
          x = mbest; ystart = 0;
        } else {
          sp--;
          if (sp < 0) break;
          if (sp == par_sp) { x = par_x; ystart = par_ys; par_sp = -1; }
          else { x = stx[sp]; ystart = (int)sty[sp] + 1; }
        }
        cycD += __builtin_readcyclecounter() - tC1;
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      const long long t1 = __builtin_readcyclecounter();
      cyc_dfs += t1 - t0;
      if (ok) break;
      n_failph++;
      // ---- failed phase.  (1) explicit entries of every visited row feed slack (L3)
      for (int w = 0; w < nw; w++) {
        unsigned bits = visx[w];
        while (bits) {
          const int xr = w * 32 + (__ffs((int)bits) - 1);
          bits &= bits - 1u;
          n_failrows++;
          const unsigned cb = P.row_ptr[xr], ce = P.row_ptr[xr + 1];
          const double lxr = lx[xr];
          for (unsigned c = cb + lane; c < ce; c += 64) {
            const int col = P.cols[c];
            const double d = (lxr + ly[col]) - P.vals[c];
            if (!(d < eps)) slack[col] = fmin(slack[col], d);
          }
        }
      }
      __syncthreads();
      // (2) deferred background slack (E4) + delta, (3) relabel (km.cpp:80-98)
      double dl = KM_INF2;
      for (int y = lane; y < n; y += 64)
        if (!bit_get(visy, y)) {
          const double s2 = fmin(slack[y], (lxmin + ly[y]) - bg);
          slack[y] = s2;
          dl = fmin(dl, s2);
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dl = fmin(dl, __shfl_xor(dl, o, 64));
      for (int i = lane; i < n; i += 64) {
        if (bit_get(visx, i)) lx[i] -= dl;
        if (bit_get(visy, i)) ly[i] += dl;
        else slack[i] -= dl;
      }
      __syncthreads();
      // (4) lists of the visited rows under the new labels (L1)
      for (int w = 0; w < nw; w++) {
        unsigned bits = visx[w];
        while (bits) {
          const int xr = w * 32 + (__ffs((int)bits) - 1);
          bits &= bits - 1u;
          build_list(xr);
        }
      }
      __syncthreads();
      cyc_fail += __builtin_readcyclecounter() - t1;
      if (phase > 4 * n + 16) { bad = 2; break; }  // only reachable with non-finite weights
    }
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) P.match_out[i] = match[i] == NONE ? -1 : (int)match[i];
  if (lane == 0) {
    if (bad && P.status) *P.status = bad;
    if (P.steps) {
      P.steps[0] = nsteps; P.steps[1] = n_over; P.steps[2] = n_flat; P.steps[3] = n_failph; P.steps[4] = n_failrows;
      P.steps[5] = cyc_dfs; P.steps[6] = cyc_fail; P.steps[7] = __builtin_readcyclecounter() - t_start; (void)cyc_init;
      P.steps[8] = cycA; P.steps[9] = cycB; P.steps[10] = cycC; P.steps[11] = cycD; P.steps[12] = cycE;
    }
  }
}

}  // namespace

namespace {

// ---- dense matrix -> background + CSR (for the public ghicp_km_solve entry point)
__global__ __launch_bounds__(256) void k_dense_min(const double* __restrict__ w, size_t total, unsigned long long* __restrict__ out) {
  // order-preserving key of a double (ascending)
  double m = INFINITY;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) m = fmin(m, w[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) {
    unsigned long long b = (unsigned long long)__double_as_longlong(m);
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    atomicMin(out, b);
  }
}

__global__ __launch_bounds__(256) void k_dense_rows(const double* __restrict__ w, int n, const unsigned long long* __restrict__ minkey,
                                                    unsigned* __restrict__ cnt, double* __restrict__ lx, const unsigned* __restrict__ row_ptr,
                                                    int* __restrict__ cols, double* __restrict__ vals, double* __restrict__ bg_out, int fill) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  unsigned long long k = *minkey;
  k = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  const double bg = __longlong_as_double((long long)k);
  if (row == 0 && lane == 0) *bg_out = bg;
  const double* r = w + (size_t)row * n;
  unsigned c = 0;
  double mx = r[0];
  const unsigned base = fill ? row_ptr[row] : 0u;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    double v = bg;
    if (j < n) { v = r[j]; mx = fmax(mx, v); }
    const bool e = (j < n) && (v != bg);
    const unsigned long long b = __ballot(e);
    if (fill && e) {
      const unsigned off = c + __popcll(b & ((1ull << lane) - 1ull));
      cols[base + off] = j;
      vals[base + off] = v;
    }
    c += __popcll(b);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if (lane == 0 && !fill) { cnt[row] = c; lx[row] = mx; }
}

}  // namespace

// exclusive scan of n row counts -> row_ptr[0..n] (single block; n <= ~10^5)
__global__ __launch_bounds__(1024) void k_gh_scan_rows(const unsigned* __restrict__ cnt, int n, unsigned* __restrict__ row_ptr) {
  __shared__ int sc[17];
  int carry = 0;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n ? (int)cnt[i] : 0;
    int tot;
    const int ex = gh_block_excl_scan(v, sc, &tot);
    if (i < n) row_ptr[i] = (unsigned)(carry + ex);
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) row_ptr[n] = (unsigned)carry;
}

// launches one k_km2 block per problem; descriptors already on device
int gh_km2_launch(ghicp_ctx* ctx, const Km2Problem* d_probs, int nprob, int n_max) {
  // default: the flood-first solver (km4.hip) whenever its state fits LDS; GHICP_KM_V2=1 keeps this file's DFS emulation
  if (gh_km4_fits(n_max) && !getenv("GHICP_KM_V2")) return gh_km4_launch(ctx, d_probs, nprob, n_max);
  {  // per device, and cheap: set before every launch instead of behind a process-wide flag
    const size_t want = 160 * 1024;
    GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km2<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
    GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km2<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
    GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km2<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
    GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
  }
  // v2 and v3 solve at the same speed (the DFS is instruction-issue bound, not memory bound: profiles/r01_km_step_counters.txt);
  // v2 needs less LDS per problem, so more problems are resident per CU -> default. GHICP_KM_V3=1 selects the list kernel.
  const bool v3 = n_max <= 65534 && gh_km3_lds_bytes(n_max) <= 160 * 1024 - 256 && getenv("GHICP_KM_V3") != nullptr;
  size_t lds = v3 ? gh_km3_lds_bytes(n_max) : gh_km2_lds_bytes(n_max);
  // GHICP_KM_SLOTS=s: never more than s solves resident per CU (the request is padded to 160 KB / s), which leaves LDS and
  // wave slots to kernels of other streams -- e.g. the front ends of the next batch -- while a solve launch fills the chip.
  if (const char* e = getenv("GHICP_KM_SLOTS")) {
    const int slots = atoi(e);
    if (slots >= 1 && slots <= 32) lds = std::max(lds, (size_t)(160 * 1024) / (size_t)(slots + 1) + 1024);
  }
  hipEvent_t kt = ctx->kt_begin(KT_KM_SOLVE);
  if (v3) hipLaunchKernelGGL(k_km3, dim3(nprob), dim3(64), lds, ctx->stream, d_probs);
  const int kflags = getenv("GHICP_KM_NOMARCH") ? 1 : 0;
  if (v3) {}
  else if (getenv("GHICP_KM_STATS")) hipLaunchKernelGGL((k_km2<true, false>), dim3(nprob), dim3(64), lds, ctx->stream, d_probs, kflags);
  else if (getenv("GHICP_KM_SWEEP")) hipLaunchKernelGGL((k_km2<false, true>), dim3(nprob), dim3(64), lds, ctx->stream, d_probs, kflags);  // E10, experimental
  else hipLaunchKernelGGL((k_km2<false, false>), dim3(nprob), dim3(64), lds, ctx->stream, d_probs, kflags);
  ctx->kt_end(KT_KM_SOLVE, kt);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

bool gh_km2_fits(int n) { return n <= 65534 && gh_km2_lds_bytes(n) <= 160 * 1024 - 256; }  // (covers gh_km4_fits)

// dense front door: background = the matrix minimum, everything above it explicit
int gh_km2_solve_dense(ghicp_ctx* ctx, const double* w, int n, double eps, int32_t* match, int* status_dev) {
  hipStream_t s = ctx->stream;
  unsigned long long* mk;
  unsigned *cnt, *rptr;
  double *lx, *vals, *bgd;
  int* cols;
  Km2Problem* dp;
  GH_TRY(ctx->reserve(B_KM_MISC, 64, (int**)&mk));  // [0..1] min key, [2] status, [4..] bg
  GH_TRY(ctx->reserve(B_KM_LX, (size_t)n + 1, &lx));
  char* scr;
  GH_TRY(ctx->reserve(B_LOOP_KMSCR, (size_t)(n + 2) * 8 + 256 + sizeof(Km2Problem), &scr));
  cnt = (unsigned*)scr;
  rptr = cnt + n + 1;
  dp = (Km2Problem*)(scr + (size_t)(n + 2) * 8 + 64);
  GH_TRY(ctx->reserve(B_LOOP_SP, (size_t)n * n + 1, &cols));
  GH_TRY(ctx->reserve(B_LOOP_KMW, (size_t)n * n + 1, &vals));
  bgd = (double*)((char*)mk + 32);
  GH_HIP(hipMemsetAsync(mk, 0xff, 8, s));
  hipLaunchKernelGGL(k_dense_min, dim3(min(cdiv((long long)n * n, 256), 1024)), dim3(256), 0, s, w, (size_t)n * n, mk);
  hipLaunchKernelGGL(k_dense_rows, dim3(cdiv(n, 4)), dim3(256), 0, s, w, n, mk, cnt, lx, (const unsigned*)nullptr, cols, vals, bgd, 0);
  hipLaunchKernelGGL(k_gh_scan_rows, dim3(1), dim3(1024), 0, s, cnt, n, rptr);
  hipLaunchKernelGGL(k_dense_rows, dim3(cdiv(n, 4)), dim3(256), 0, s, w, n, mk, cnt, lx, rptr, cols, vals, bgd, 1);
  double bg_h = 0;
  GH_HIP(hipMemcpyAsync(&bg_h, bgd, sizeof(double), hipMemcpyDeviceToHost, s));
  GH_HIP(hipStreamSynchronize(s));
  Km2Problem hp;
  memset(&hp, 0, sizeof(hp));
  hp.n = n; hp.bg = bg_h; hp.eps = eps; hp.row_ptr = rptr; hp.cols = cols; hp.vals = vals; hp.lx_init = lx; hp.match_out = match;
  hp.status = status_dev;
  GH_TRY(ctx->reserve(B_KM_SLACK, (size_t)n + 2, &hp.slack));
  long long* dstats = nullptr;
  if (getenv("GHICP_KM_STATS")) {
    GH_TRY(ctx->reserve(B_P_PATTERN, 32, &dstats));
    GH_HIP(hipMemsetAsync(dstats, 0, 24 * sizeof(long long), s));
    hp.steps = dstats;
  }
  GH_HIP(hipMemcpyAsync(dp, &hp, sizeof(hp), hipMemcpyHostToDevice, s));
  GH_TRY(gh_km2_launch(ctx, dp, 1, n));
  if (dstats) {
    long long h[24];
    GH_HIP(hipMemcpyAsync(h, dstats, sizeof(h), hipMemcpyDeviceToHost, s));
    GH_HIP(hipStreamSynchronize(s));
    if (gh_km4_fits(n) && !getenv("GHICP_KM_V2"))
      fprintf(stderr, "[km4 stats] n=%d activations=%lld phases=%lld failed=%lld pull_rounds=%lld dfs_iterations=%lld flood_rows(failed)=%lld rebuilt_rows=%lld | cycles: flood=%lld failed=%lld pull=%lld dfs=%lld total=%lld | hazard=%lld | flood: levels=%lld flagged_rows=%lld cyc_rows=%lld sweeps=%lld cyc_sweeps=%lld\n",
              n, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], h[16], h[17]);
    else if (getenv("GHICP_KM_V3"))
      fprintf(stderr, "[km stats] n=%d steps=%lld overflow=%lld flat=%lld failph=%lld failrows=%lld cyc_dfs=%lld cyc_fail=%lld cyc_total=%lld | A=%lld B=%lld C=%lld (list+overflow %lld) D=%lld\n", n, h[0], h[1],
              h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[12], h[11]);
    else
      fprintf(stderr, "[km2 stats] n=%d steps=%lld (successful phases %lld, failed phases %lld in %lld phases) push=%lld | bg scans=%lld groups=%lld cache hits=%lld | csr blocks=%lld | cycles: csr=%lld bg=%lld move=%lld relabel=%lld\n",
              n, h[0], h[1], h[2], h[3], h[12], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
  }
  return GHICP_OK;
}
