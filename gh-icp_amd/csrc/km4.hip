// Kuhn-Munkres, fourth-generation kernel (gfx950): the reference's result (src/km.cpp:13-126) WITHOUT stepping through the
// reference's depth-first search wherever its outcome is order independent.  One 256-thread workgroup per problem, all
// solver state in LDS (61.75 B per row), the CSR of the explicit entries streamed from global memory only by the bulk
// passes of failed phases.  Rules (oracle/km4_model.inc states them sequentially and is fuzzed against the reference
// traversal; R1 = E1-E3 of km2.hip):
//   R2  per row a list of <= 3 (column, weight) pairs in LDS, ascending column, a SUPERSET of the row's tight explicit entries;
//       members are re-tested with fl(fl(lx+ly) - w) < eps at every use.  An explicit entry can only become tight when its
//       row label drops, i.e. for rows visited by a failed phase: those lists are rebuilt after the relabelling.  Rows with
//       more than 3 tight entries are flagged (scanned from the CSR row, never pruned).
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
//       whose owner lies outside any superset S of good without changing its path.  S = fixed point of "background-tight to
//       the good column of smallest ly, or a listed tight entry in a good column" (256 threads, one row each per round,
//       ~4 rounds), flagged rows included.  Wave 0 then runs the reference's DFS restricted to S: E7 pointer, E9 march
//       (km2.hip), all in LDS; what is left of the search is essentially the augmenting path itself.
#include "ctx.h"
#include "devmath.h"
#include "km_prob.h"

#include <climits>
#include <cstdlib>

namespace {

constexpr int K4_CAP = 3;
constexpr int K4_OVER = 255;
constexpr int K4_T = 256;
constexpr int K4_NONE = 0xFFFF;
constexpr double K4_INF = 1000.0;  // km.cpp:42

typedef __attribute__((address_space(1))) const int* k4_gint;
typedef __attribute__((address_space(1))) const double* k4_gf64;
typedef __attribute__((address_space(1))) const unsigned* k4_gu32;

enum { SH_QT = 0, SH_FREE, SH_RES, SH_QTF, SH_HAZ, SH_CH0, SH_CH1, SH_BAD, SH_NUM = 16 };

struct K4 {
  double *lx, *ly, *slack, *tlv, *red;
  unsigned *visx, *visy, *prevy, *pushed, *good, *goody, *freey, *ovf;
  int* sh;
  unsigned short *match, *stx, *sty, *tlc;
  unsigned char* tln;
  int n, nw;
  double bg, eps;
  k4_gu32 rptr;
  k4_gint cols;
  k4_gf64 vals;
};

__device__ inline bool k4_bit(const unsigned* b, int i) { return (b[i >> 5] >> (i & 31)) & 1u; }

// R3: mark column y visited; a free column ends the flood, a matched one enqueues its owner (each row owns one column)
__device__ inline void k4_visit(const K4& s, int y) {
  const unsigned bit = 1u << (y & 31);
  if (atomicOr(&s.visy[y >> 5], bit) & bit) return;
  const int m = s.match[y];
  if (m == K4_NONE) { s.sh[SH_FREE] = 1; return; }
  const unsigned b2 = 1u << (m & 31);
  if (!(atomicOr(&s.visx[m >> 5], b2) & b2)) s.stx[atomicAdd(&s.sh[SH_QT], 1)] = (unsigned short)m;
}

// One wave streams the CSR row x: REBUILD writes the row's list (R2), PUSH sends the slack minima of the non-tight entries (R4).
// (cb, ce) = the row's CSR range, (c0col, c0val) = its first 64 entries, already requested by the caller (software pipeline).
template <bool REBUILD, bool PUSH>
__device__ inline void k4_scan_row(const K4& s, int x, unsigned cb, unsigned ce, int c0col, double c0val, int lane) {
  const double lxr = s.lx[x];
  int cnt = 0;
  unsigned long long* sl = reinterpret_cast<unsigned long long*>(s.slack);
  auto block = [&](unsigned c, int col, double val) {
    const bool in = c < ce;
    const double d = (lxr + s.ly[col]) - val;
    const bool td = d < s.eps;
    if (PUSH && in && !td) atomicMin(&sl[col], (unsigned long long)__double_as_longlong(d));
    if (REBUILD) {
      const unsigned long long tb = __ballot(in && td);
      if (in && td) {
        const int r = cnt + __popcll(tb & ((1ull << lane) - 1ull));
        if (r < K4_CAP) { s.tlc[x * K4_CAP + r] = (unsigned short)col; s.tlv[x * K4_CAP + r] = val; }
      }
      cnt += __popcll(tb);
    }
  };
  if (ce > cb) block(cb + lane, c0col, c0val);
  for (unsigned c0 = cb + 64u; c0 < ce; c0 += 128u) {  // two blocks per round, both requested before either is used
    const unsigned ca = c0 + lane, cbb = c0 + 64u + lane;
    const unsigned cca = min(ca, ce - 1u), ccb = min(cbb, ce - 1u);
    const int col_a = s.cols[cca], col_b = s.cols[ccb];
    const double val_a = s.vals[cca], val_b = s.vals[ccb];
    block(ca, col_a, val_a);
    if (c0 + 64u < ce) block(cbb, col_b, val_b);
  }
  if (REBUILD && lane == 0) {
    const bool over = cnt > K4_CAP;
    const bool was = s.tln[x] == K4_OVER;
    s.tln[x] = (unsigned char)(over ? K4_OVER : cnt);
    if (over != was) {
      if (over) atomicOr(&s.ovf[x >> 5], 1u << (x & 31));
      else atomicAnd(&s.ovf[x >> 5], ~(1u << (x & 31)));
    }
  }
}

// Bulk pass over rows list[0..count) (all 4 waves, a row per wave at a time); ONLY_UNPUSHED skips rows whose minima are in slack.
// Each wave first fetches the CSR ranges of up to 64 of its rows in one go, then walks them with the next row's first block
// in flight while the current row is processed.
template <bool REBUILD, bool PUSH, bool ONLY_UNPUSHED>
__device__ inline void k4_bulk(const K4& s, const unsigned short* list, int count, int wave, int lane) {
  for (int base = 0; base < count; base += 4 * 64) {
    const int i = base + lane * 4 + wave;
    int x = -1;
    if (i < count) {
      x = list[i];
      if (ONLY_UNPUSHED && k4_bit(s.pushed, x)) x = -1;
    }
    unsigned cb = 0, ce = 0;
    if (x >= 0) { cb = s.rptr[x]; ce = s.rptr[x + 1]; }
    unsigned long long todo = __ballot(x >= 0);
    if (!todo) continue;
    int l = (int)__ffsll((long long)todo) - 1;
    unsigned ncb = __builtin_amdgcn_readlane(cb, l), nce = __builtin_amdgcn_readlane(ce, l);
    int nx = __builtin_amdgcn_readlane(x, l);
    int ncol = 0;
    double nval = 0.0;
    if (nce > ncb) { const unsigned cc = min(ncb + (unsigned)lane, nce - 1u); ncol = s.cols[cc]; nval = s.vals[cc]; }
    while (todo) {
      todo &= todo - 1ull;
      const int cx = nx, ccol = ncol;
      const unsigned ccb = ncb, cce = nce;
      const double cval = nval;
      if (todo) {  // request the next row's first block before working on this one
        l = (int)__ffsll((long long)todo) - 1;
        ncb = __builtin_amdgcn_readlane(cb, l); nce = __builtin_amdgcn_readlane(ce, l); nx = __builtin_amdgcn_readlane(x, l);
        if (nce > ncb) { const unsigned cc = min(ncb + (unsigned)lane, nce - 1u); ncol = s.cols[cc]; nval = s.vals[cc]; }
      }
      k4_scan_row<REBUILD, PUSH>(s, cx, ccb, cce, ccol, cval, lane);
    }
  }
}

__device__ inline double k4_wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}

// First unvisited good background-tight column in [start, limit) (E7 scan of km2.hip with the R5 filter)
__device__ inline int k4_bg_scan(const K4& s, double lxv, int start, int limit, int lane) {
  for (int y0 = start; y0 < limit; y0 += 64) {
    const int y = y0 + lane, yc = min(y, limit - 1);
    const unsigned vw = s.visy[yc >> 5], gw = s.goody[yc >> 5];
    const double lv = s.ly[yc];
    const unsigned t = (unsigned)(y < limit) & (((~vw & gw) >> (yc & 31)) & 1u) & (unsigned)(((lxv + lv) - s.bg) < s.eps);
    const unsigned long long b = __ballot(t != 0u);
    if (b) return y0 + (int)__ffsll((long long)b) - 1;
  }
  return INT_MAX;
}

// ---- R3: the flood (wave 0).  Returns true when a free column is reachable; the visited rows are stx[0 .. sh[SH_QT]).
__device__ inline bool k4_flood(const K4& s, int root, int lane) {
  const int n = s.n;
  if (lane == 0) { s.stx[0] = (unsigned short)root; s.visx[root >> 5] = 1u << (root & 31); s.sh[SH_QT] = 1; s.sh[SH_FREE] = 0; }
  __builtin_amdgcn_wave_barrier();
  int qh = 0, qt = 1;
  double lflood = INFINITY;
  while (qh < qt) {
    double lcand = INFINITY;
    for (int base = qh; base < qt; base += 64) {
      const int i = base + lane;
      const bool act = i < qt;
      int xr = 0, tn = 0;
      double lxr = INFINITY;
      if (act) { xr = s.stx[i]; lxr = s.lx[xr]; tn = s.tln[xr]; }
      const bool over = act && tn == K4_OVER;
      if (act && !over) {
        for (int k = 0; k < K4_CAP; k++)
          if (k < tn) {
            const int col = s.tlc[xr * K4_CAP + k];
            if (((lxr + s.ly[col]) - s.tlv[xr * K4_CAP + k]) < s.eps && !k4_bit(s.visy, col)) k4_visit(s, col);
          }
      }
      if (act && (lxr - s.bg) < s.eps) lcand = fmin(lcand, lxr);
      unsigned long long ob = __ballot(over);
      while (ob) {  // flagged rows: every tight entry of the CSR row
        const int l = (int)__ffsll((long long)ob) - 1;
        ob &= ob - 1ull;
        const int xo = __builtin_amdgcn_readlane(xr, l);
        const double lxo = s.lx[xo];
        const unsigned cb = s.rptr[xo], ce = s.rptr[xo + 1];
        for (unsigned c = cb + lane; c < ce; c += 64) {
          const int col = s.cols[c];
          if (((lxo + s.ly[col]) - s.vals[c]) < s.eps && !k4_bit(s.visy, col)) k4_visit(s, col);
        }
      }
    }
    lcand = k4_wave_min(lcand);
    if (lcand < lflood) {  // T_L of the smallest label so far contains T_L of every larger one
      lflood = lcand;
      for (int y0 = 0; y0 < n; y0 += 64) {
        const int y = y0 + lane;
        if (y < n && !k4_bit(s.visy, y) && ((lcand + s.ly[y]) - s.bg) < s.eps) k4_visit(s, y);
      }
    }
    __builtin_amdgcn_wave_barrier();
    qh = qt;
    qt = s.sh[SH_QT];
    if (s.sh[SH_FREE]) return true;
  }
  return false;
}

// ---- R5: the reference's DFS restricted to S (wave 0).  Returns false only on an internal error.
template <bool PROF>
__device__ inline bool k4_dfs(const K4& s, int root, int lane, long long* q_iter, long long* q_act) {
  const int n = s.n;
  const double bg = s.bg, eps = s.eps;
  if (lane == 0) { s.stx[0] = (unsigned short)root; s.sty[0] = (unsigned short)K4_NONE; }
  __builtin_amdgcn_wave_barrier();
  int sp = 0, x = root, ystart = 0;
  double ck = __longlong_as_double(0x7ff8000000000000ll);  // E7 cache, one label per lane: NaN never matches
  int cp = 0, cnext = 0;
  for (;;) {  // one iteration == one findpath() activation or resumption (km.cpp:13-37)
    if (PROF) { ++*q_iter; ++*q_act; }
    const double lxv = s.lx[x];
    const int tn = s.tln[x];
    const bool bgt = (lxv - bg) < eps;
    int best = INT_MAX;
    bool exhausted = false;
    if (tn == 0 && bgt) {
      // ---- E9 march (km2.hip): a chain of rows without tight explicit entries that share the label L picks the members of
      // T_L in column order, up to 64 activations per window; candidates are filtered by S (R5).
      const unsigned long long hit = __ballot(ck == lxv);
      int slot, p = ystart;
      if (hit) {
        slot = (int)__ffsll((long long)hit) - 1;
        p = max(p, __builtin_amdgcn_readlane(cp, slot));
      } else {
        slot = cnext;
        cnext = (cnext + 1) & 63;
        if (lane == slot) ck = lxv;
      }
      int outcome = 0;
      while (p < n) {
        const int y = p + lane, yc = min(y, n - 1);
        const unsigned vw = s.visy[yc >> 5], gw = s.goody[yc >> 5];
        const double lv = s.ly[yc];
        const bool cand = (y < n) & ((((~vw & gw) >> (yc & 31)) & 1u) != 0u) & (((lxv + lv) - bg) < eps);
        const unsigned long long b = __ballot(cand);
        if (!b) { p += 64; continue; }
        int m = K4_NONE;
        bool cont = false;
        if (cand) {
          m = s.match[y];
          if (m != K4_NONE) cont = (s.tln[m] == 0) & (s.lx[m] == lxv);
        }
        const unsigned long long stop = __ballot(cand && !cont);
        const int jstar = stop ? (int)__ffsll((long long)stop) - 1 : 63;
        const unsigned long long R = b & (~0ull >> (63 - jstar));  // the picks of this window, in column order
        const unsigned long long below = R & ((1ull << lane) - 1ull);
        const int rank = __popcll(below);
        const int pm = __shfl(m, below ? 63 - __clzll((long long)below) : 0, 64);
        if ((R >> lane) & 1ull) {
          atomicOr(&s.visy[y >> 5], 1u << (y & 31));
          s.sty[sp + rank] = (unsigned short)y;
          if (rank > 0) s.stx[sp + rank] = (unsigned short)pm;  // the row that picked this column (frame sp holds x)
        }
        const int k = __popcll(R), lastlane = 63 - __clzll((long long)R);
        const int mlast = __builtin_amdgcn_readlane(m, lastlane);
        if (PROF) *q_act += k - 1;
        sp += k - 1;  // frame of the row that made the last pick
        p += lastlane + 1;
        if (mlast == K4_NONE) { outcome = 1; break; }
        sp++;
        if (lane == 0) { s.stx[sp] = (unsigned short)mlast; s.sty[sp] = (unsigned short)K4_NONE; }
        x = mlast; ystart = 0;
        if (stop) { outcome = 2; break; }  // the owner of the last pick is not part of the chain
        if (PROF) ++*q_act;                // ... it is: its activation continues the march
      }
      if (lane == slot) cp = p;
      __builtin_amdgcn_wave_barrier();
      if (outcome == 1) break;
      if (outcome == 2) continue;
      exhausted = true;  // x (possibly a row reached by the march: its frame is sp) has no candidate left
    }
    if (!exhausted) {
      if (tn == K4_OVER) {  // flagged row: lowest tight unvisited good column of the CSR row
        const unsigned cb = s.rptr[x], ce = s.rptr[x + 1];
        for (unsigned c0 = cb; c0 < ce; c0 += 64) {
          const unsigned c = c0 + lane, cc = min(c, ce - 1u);
          const int col = s.cols[cc];
          const double val = s.vals[cc];
          const bool t = (c < ce) & (((lxv + s.ly[col]) - val) < eps) & (col >= ystart) & !k4_bit(s.visy, col) & k4_bit(s.goody, col);
          const unsigned long long b = __ballot(t);
          if (b) { best = __builtin_amdgcn_readlane(col, (int)__ffsll((long long)b) - 1); break; }
        }
      } else if (tn > 0) {
        int col = 0;
        bool t = false;
        if (lane < tn) {
          col = s.tlc[x * K4_CAP + lane];
          const double val = s.tlv[x * K4_CAP + lane];
          t = (((lxv + s.ly[col]) - val) < eps) & (col >= ystart) & !k4_bit(s.visy, col) & k4_bit(s.goody, col);
        }
        const unsigned long long b = __ballot(t);
        if (b) best = __builtin_amdgcn_readlane(col, (int)__ffsll((long long)b) - 1);
      }
      if (bgt) {  // E7: one scan pointer per distinct label value
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
        const int yb = k4_bg_scan(s, lxv, max(ystart, p0), lim, lane);
        if (ystart <= p0 && lane == slot) cp = (yb != INT_MAX) ? yb : max(p0, lim);
        best = min(best, yb);
      }
    }
    if (best != INT_MAX) {
      const int m = s.match[best];
      if (lane == 0) { atomicOr(&s.visy[best >> 5], 1u << (best & 31)); s.sty[sp] = (unsigned short)best; }
      if (m == K4_NONE) { __builtin_amdgcn_wave_barrier(); break; }
      sp++;
      if (lane == 0) { s.stx[sp] = (unsigned short)m; s.sty[sp] = (unsigned short)K4_NONE; }
      x = m; ystart = 0;
    } else {
      sp--;
      if (sp < 0) return false;
      x = s.stx[sp]; ystart = (int)s.sty[sp] + 1;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // augment: match[y] = x on every level of the recursion (km.cpp:26-29); the last column is no longer free
  __builtin_amdgcn_wave_barrier();
  const int ylast = s.sty[sp];
  for (int f = lane; f <= sp; f += 64) s.match[s.sty[f]] = s.stx[f];
  if (lane == 0) atomicAnd(&s.freey[ylast >> 5], ~(1u << (ylast & 31)));
  return true;
}

// ---- literal solver (hazard fallback of R4): lane 0 of wave 0 runs the reference line by line on background + CSR.
// A matrix that gets here has an edge within an ulp of eps after a relabelling; the reference itself usually does not
// terminate on such input (its delta becomes 0), hence the step budget.  Slow by design, never on the hot path.
__device__ inline int k4_literal(const K4& s, const Km2Problem& P) {
  const int n = s.n;
  const double bg = s.bg, eps = s.eps;
  unsigned short* curs = s.tlc;  // per frame: CSR cursor relative to the row start (n entries of the 3n available)
  for (int i = 0; i < n; i++) { s.lx[i] = P.lx_init[i]; s.ly[i] = 0.0; s.match[i] = (unsigned short)K4_NONE; }
  long long budget = 64ll * n * n + 4096;
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

template <bool PROF>
__global__ __launch_bounds__(K4_T) void k_km4(const Km2Problem* __restrict__ probs, int flags) {
  const Km2Problem P = probs[blockIdx.x];
  if (P.n <= 0 || (P.done && *P.done)) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = P.n, nw = (n + 31) / 32, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  K4 s;
  s.n = n; s.nw = nw; s.bg = P.bg; s.eps = P.eps;
  s.rptr = (k4_gu32)P.row_ptr; s.cols = (k4_gint)P.cols; s.vals = (k4_gf64)P.vals;
  s.lx = (double*)smem;
  s.ly = s.lx + n;
  s.slack = s.ly + n;
  s.tlv = s.slack + n;
  s.red = s.tlv + (size_t)n * K4_CAP;  // 16 doubles
  s.visx = (unsigned*)(s.red + 16);
  s.visy = s.visx + nw; s.prevy = s.visy + nw; s.pushed = s.prevy + nw; s.good = s.pushed + nw; s.goody = s.good + nw;
  s.freey = s.goody + nw; s.ovf = s.freey + nw;
  s.sh = (int*)(s.ovf + nw);
  s.match = (unsigned short*)(s.sh + SH_NUM);
  s.stx = s.match + n;
  s.sty = s.stx + n + 2;
  s.tlc = s.sty + n + 2;
  s.tln = (unsigned char*)(s.tlc + (size_t)n * K4_CAP);
  const double bg = s.bg, eps = s.eps;

  long long c_flood = 0, c_fail = 0, c_pull = 0, c_dfs = 0, q_phase = 0, q_fail = 0, q_rounds = 0, q_iter = 0, q_act = 0, q_frows = 0, q_prows = 0;
  const long long t_begin = PROF ? (long long)__builtin_readcyclecounter() : 0;

  for (int i = tid; i < n; i += K4_T) { s.lx[i] = P.lx_init[i]; s.ly[i] = 0.0; s.match[i] = (unsigned short)K4_NONE; s.tln[i] = 0; }
  for (int w = tid; w < nw; w += K4_T) {
    unsigned all = ~0u;
    if (w == nw - 1 && (n & 31)) all = (1u << (n & 31)) - 1u;
    s.freey[w] = all; s.ovf[w] = 0u;
  }
  if (tid < SH_NUM) s.sh[tid] = 0;
  __syncthreads();
  // initial lists: every row once (R2)
  for (int base = 0; base < n; base += 4 * 64) {
    const int i = base + lane * 4 + wave;
    unsigned cb = 0, ce = 0;
    if (i < n) { cb = s.rptr[i]; ce = s.rptr[i + 1]; }
    for (int l = 0; l < 64; l++) {
      const int x = base + l * 4 + wave;
      if (x >= n) break;
      const unsigned rcb = __builtin_amdgcn_readlane(cb, l), rce = __builtin_amdgcn_readlane(ce, l);
      int col = 0;
      double val = 0.0;
      if (rce > rcb) { const unsigned cc = min(rcb + (unsigned)lane, rce - 1u); col = s.cols[cc]; val = s.vals[cc]; }
      k4_scan_row<true, false>(s, x, rcb, rce, col, val, lane);
    }
  }
  __syncthreads();

  int bad = 0;
  bool hazard = false;
  for (int root = 0; root < n && !bad && !hazard; ++root) {
    for (int i = tid; i < n; i += K4_T) s.slack[i] = K4_INF;
    for (int w = tid; w < nw; w += K4_T) s.pushed[w] = 0u;
    bool have_prev = false;
    for (int phase = 0;; ++phase) {
      if (PROF) q_phase++;
      for (int w = tid; w < nw; w += K4_T) { s.visx[w] = 0u; s.visy[w] = 0u; }
      __syncthreads();
      const long long t0 = PROF ? (long long)__builtin_readcyclecounter() : 0;
      if (wave == 0) {
        const bool fr = k4_flood(s, root, lane);
        if (lane == 0) { s.sh[SH_RES] = fr ? 1 : 0; s.sh[SH_QTF] = s.sh[SH_QT]; }
      }
      __syncthreads();
      const bool free_found = s.sh[SH_RES] != 0;
      const int qt = s.sh[SH_QTF];
      const long long t1 = PROF ? (long long)__builtin_readcyclecounter() : 0;
      if (PROF) c_flood += t1 - t0;
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
          if (have_prev && (s.prevy[w] & ~s.visy[w])) s.sh[SH_HAZ] = 1;  // a visited column dropped out (R4)
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
        for (int w = tid; w < nw; w += K4_T) s.pushed[w] = s.visx[w];
        __syncthreads();
        k4_bulk<true, true, false>(s, s.stx, qt, wave, lane);  // lists under the new labels + the minima of the next phase
        have_prev = true;
        if (PROF) { q_prows += qt; c_fail += (long long)__builtin_readcyclecounter() - t1; }
        if (phase > 4 * n + 16) { bad = 2; break; }  // only reachable with non-finite weights
        continue;
      }
      // ---- R5: augmenting phase.  S by pull rounds ...
      for (int w = tid; w < nw; w += K4_T) { s.visx[w] = 0u; s.visy[w] = 0u; s.good[w] = s.ovf[w]; s.goody[w] = s.freey[w]; }
      if (tid == 0) { s.sh[SH_CH0] = 0; s.sh[SH_CH1] = 0; }
      __syncthreads();
      for (int round = 0;; round++) {
        if (PROF) q_rounds++;
        double gm = INFINITY;
        for (int y = tid; y < n; y += K4_T) {
          bool g = k4_bit(s.goody, y);
          if (!g) {
            const int m = s.match[y];
            if (m != K4_NONE && k4_bit(s.good, m)) { atomicOr(&s.goody[y >> 5], 1u << (y & 31)); g = true; }
          }
          if (g) gm = fmin(gm, s.ly[y]);
        }
        gm = k4_wave_min(gm);
        if (lane == 0) s.red[8 + wave] = gm;
        __syncthreads();
        if (tid == 0) s.sh[SH_CH0 + ((round + 1) & 1)] = 0;
        const double gmin = fmin(fmin(s.red[8], s.red[9]), fmin(s.red[10], s.red[11]));
        bool ch = false;
        for (int x = tid; x < n; x += K4_T) {
          if (k4_bit(s.good, x)) continue;
          const double lxv = s.lx[x];
          bool g = ((lxv - bg) < eps) & (((lxv + gmin) - bg) < eps);
          if (!g) {
            const int tn = s.tln[x];
            for (int k = 0; k < K4_CAP; k++)
              if (k < tn) {
                const int col = s.tlc[x * K4_CAP + k];
                g |= (((lxv + s.ly[col]) - s.tlv[x * K4_CAP + k]) < eps) & k4_bit(s.goody, col);
              }
          }
          if (g) { atomicOr(&s.good[x >> 5], 1u << (x & 31)); ch = true; }
        }
        if (ch) s.sh[SH_CH0 + (round & 1)] = 1;
        __syncthreads();
        if (!s.sh[SH_CH0 + (round & 1)]) break;
        if (round >= 40) {  // give up pruning for this phase: any superset of good is valid (R5)
          for (int w = tid; w < nw; w += K4_T) s.goody[w] = ~0u;
          break;
        }
      }
      __syncthreads();
      const long long t2 = PROF ? (long long)__builtin_readcyclecounter() : 0;
      if (PROF) c_pull += t2 - t1;
      // ... then the DFS (wave 0)
      if (wave == 0) {
        const bool ok = k4_dfs<PROF>(s, root, lane, &q_iter, &q_act);
        if (!ok && lane == 0) s.sh[SH_BAD] = 3;
      }
      __syncthreads();
      if (PROF) c_dfs += (long long)__builtin_readcyclecounter() - t2;
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
  if (tid == 0) {
    if (bad && P.status) *P.status = bad;
    if (P.steps) {
      P.steps[0] = q_act;
      if (PROF) {
        P.steps[1] = q_phase; P.steps[2] = q_fail; P.steps[3] = q_rounds; P.steps[4] = q_iter; P.steps[5] = q_frows; P.steps[6] = q_prows;
        P.steps[7] = c_flood; P.steps[8] = c_fail; P.steps[9] = c_pull; P.steps[10] = c_dfs;
        P.steps[11] = (long long)__builtin_readcyclecounter() - t_begin; P.steps[12] = hazard ? 1 : 0;
      }
    }
  }
}

}  // namespace

size_t gh_km4_lds_bytes(int n) {
  const size_t nw = (size_t)(n + 31) / 32;
  return (size_t)n * (3 + K4_CAP) * 8 + 16 * 8 + 8 * nw * 4 + SH_NUM * 4 + ((size_t)n * (1 + K4_CAP) + 2 * ((size_t)n + 2)) * 2 + (size_t)n + 64;
}

bool gh_km4_fits(int n) { return n <= 65534 && gh_km4_lds_bytes(n) <= 160 * 1024 - 256; }

int gh_km4_launch(ghicp_ctx* ctx, const Km2Problem* d_probs, int nprob, int n_max) {
  const size_t lds = gh_km4_lds_bytes(n_max);
  const size_t want = 160 * 1024;
  // per device and thread safe: the attribute is cheap to set, so it is simply set before every launch
  GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km4<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
  GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km4<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
  const int kflags = getenv("GHICP_KM_FORCE_HAZARD") ? 4 : 0;  // test hook: sends one phase through the hazard fallback
  hipEvent_t kt = ctx->kt_begin(KT_KM_SOLVE);
  if (getenv("GHICP_KM_STATS")) hipLaunchKernelGGL((k_km4<true>), dim3(nprob), dim3(K4_T), lds, ctx->stream, d_probs, kflags);
  else hipLaunchKernelGGL((k_km4<false>), dim3(nprob), dim3(K4_T), lds, ctx->stream, d_probs, kflags);
  ctx->kt_end(KT_KM_SOLVE, kt);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}
