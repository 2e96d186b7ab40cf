// asmc_resample.hip — cdf scans (exact sequential-order and fast), PCG64 uniforms, upper-bound
// search, row gather, validity compaction.
//
// Replaces (reference mj-will/aspire):
//   src/aspire/samples.py:1277-1278  w = exp(lw - LSE); idx = rng.choice(N, n, True, p=w)
//     == numpy Generator.choice: cdf = cumsum(w); cdf /= cdf[-1]; u = random(n);
//        idx = cdf.searchsorted(u, side="right")                      
//   src/aspire/samples.py:1279-1287  x[idx], log_likelihood[idx], log_prior[idx], log_q[idx]
//   src/aspire/samplers/mcmc.py:88-108  finite-mask filter of draw_initial_samples
//
// Exact cdf.  numpy's cumsum is a strictly sequential fp64 accumulation s_k = fl(s_{k-1} + w_k).
// While s stays inside one binade [2^e, 2^(e+1)) every such add is an INTEGER add on the grid
// q = 2^(e-52):  s_k/q = s_{k-1}/q + RNE(w_k/q), ties-to-even depending only on the parity of
// s_{k-1}/q.  Each element is therefore a 2-state transducer (a0, a1) = increment when the
// running integer is even / odd; transducers compose associatively, so a block-wide scan
// reproduces the sequential rounding bit-for-bit.  When the running sum would leave the binade
// (S + floor(w/q) >= 2^53) the first such element is added with a genuine fp64 add and the scan
// restarts behind it in the new binade.
#include <stdlib.h>

#include "asmc_common.h"

// =============================================================================================
// exact cdf
// =============================================================================================
struct TD {
    long long a0, a1;
};
#define TD_SAT (1LL << 60)
#define TD_BIG (1LL << 54)
#define TWO53 9007199254740992.0
#define TWO53_LL (1LL << 53)

__device__ __forceinline__ TD td_compose(TD f, TD g) {  // apply f, then g
    TD h;
    h.a0 = f.a0 + ((f.a0 & 1) ? g.a1 : g.a0);
    h.a1 = f.a1 + ((f.a1 & 1) ? g.a0 : g.a1);
    h.a0 = h.a0 > TD_SAT ? TD_SAT : h.a0;
    h.a1 = h.a1 > TD_SAT ? TD_SAT : h.a1;
    return h;
}

// element transducer for weight w >= 0 on the grid 2^(e-52); nf = floor(w/q) (BIG if >= 2^53)
__device__ __forceinline__ void td_of(double w, int e, long long& a0, long long& a1, long long& nf) {
    const double x = ldexp(w, 52 - e);
    const bool big = !(x < TWO53);
    const long long n = big ? TD_BIG : (long long)x;  // x >= 0: truncation == floor
    const double f = big ? 0.0 : x - (double)n;
    nf = n;
    const long long up = (f > 0.5) ? 1 : 0;
    const bool tie = (f == 0.5);
    a0 = n + (tie ? (n & 1) : up);        // running integer even: tie rounds to even
    a1 = n + (tie ? ((n + 1) & 1) : up);  // running integer odd
}

// ---- block-level ordered scan of transducers --------------------------------------------------
// 256 threads x 8 elements.  (512 x 4 was measured in round 5: nothing at 1M - the passes are chains of barriers and LDS round
// trips, not instruction issue - and the transducer pass of 8M particles went from 45 to 75 us.)
#define XT_THREADS ASMC_BLOCK
#define XT_E (ASMC_SCAN_TILE / XT_THREADS)  // 8 elements per thread, one 2048-element tile per block pass

// ---- wave-level ordered scan of transducers on the DPP network (round 5) ---------------------------------------------
// A lane's (a0, a1) moves as four 32-bit DPP moves; a lane without a source reads 0 = the identity transducer, so no step needs a
// lane test.  row_shr 1 / 2 / 4 / 8 scan the rows of 16, row_bcast15 hands rows 1 and 3 the totals of rows 0 and 2, row_bcast31
// hands rows 2 and 3 the total of the first two.  (Before: six rounds of four ds_bpermute, ~130 cycles of LDS latency each, in
// kernels that run one wave per SIMD.  Composition is associative - exact integers - so the network does not change a bit.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_ll(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)v & 0xFFFFFFFFull), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)v >> 32), CTRL, ROW_MASK, 0xF, false);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138

__device__ __forceinline__ TD td_wave_scan(TD inc) {  // inclusive, lane order
#define TD_SCAN_STEP(CTRL, MASK)                                          \
    {                                                                     \
        const TD t_ = {dpp_ll<CTRL, MASK>(inc.a0), dpp_ll<CTRL, MASK>(inc.a1)}; \
        inc = td_compose(t_, inc);                                        \
    }
    TD_SCAN_STEP(DPP_ROW_SHR(1), 0xF)
    TD_SCAN_STEP(DPP_ROW_SHR(2), 0xF)
    TD_SCAN_STEP(DPP_ROW_SHR(4), 0xF)
    TD_SCAN_STEP(DPP_ROW_SHR(8), 0xF)
    TD_SCAN_STEP(DPP_ROW_BCAST15, 0xA)
    TD_SCAN_STEP(DPP_ROW_BCAST31, 0xC)
#undef TD_SCAN_STEP
    return inc;
}
__device__ __forceinline__ TD td_wave_shr1(TD inc) {  // the lane below's value, identity in lane 0
    return TD{dpp_ll<DPP_WAVE_SHR1, 0xF>(inc.a0), dpp_ll<DPP_WAVE_SHR1, 0xF>(inc.a1)};
}

// running sum s > 0  <->  (binade e, integer S on the grid 2^(e-52)): the mantissa with its implicit bit - the same values as
// binade_of(s) and (long long)ldexp(s, 52 - e), without the library sequences (the chain's walk is a lone wave: ~10 cycles an instruction)
__device__ __forceinline__ void sum_split(double s, int& e, long long& S) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(s);
    const int ex = (int)((b >> 52) & 0x7FF);
    const long long m = (long long)(b & 0xFFFFFFFFFFFFFull);
    e = ex ? ex - 1023 : -1022;
    S = ex ? (m | (1LL << 52)) : m;
}
// ldexp((double)S, e - 52) for 2^52 <= S <= 2^53 (any S < 2^53 at e = -1022): S's bit 52 is the exponent field's last bit
__device__ __forceinline__ double sum_join(long long S, int e) {
    return __longlong_as_double(S + ((long long)(e + 1022) << 52));
}

// in: `mine` = composition of this thread's elements.  out: excl = composition of all lower threads,
// total = composition of the whole block (valid in every thread).  sh: [XT_THREADS/64 + 1] TDs.
__device__ __forceinline__ void td_block_scan(TD mine, TD& excl, TD& total, TD* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TD inc = td_wave_scan(mine);
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    TD wave_prefix = {0, 0};
    TD tot = {0, 0};
#pragma unroll
    for (int v = 0; v < XT_THREADS / 64; v++) {
        if (v < wave) wave_prefix = td_compose(wave_prefix, sh[v]);
        tot = td_compose(tot, sh[v]);
    }
    excl = td_compose(wave_prefix, td_wave_shr1(inc));
    total = tot;
    __syncthreads();  // sh may be reused by the caller
}

__device__ __forceinline__ long long readlane_ll(long long v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFLL), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), lane);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ int binade_of(double s) {
    int e = (s > 0.0) ? ilogb(s) : -1022;
    return e < -1022 ? -1022 : e;
}

// ---- a tile with SEVERAL binade crossings in O(1) per crossing (round 5) ------------------------------------------
// Tile 0 (the running sum grows from nothing: a crossing at elements ~1, 2, 4, ... 1024) and every tile pass C flags 0
// used to take one block-wide pass PER CROSSING in exact_tile's loop (2.7 - 3 us each: 21.7 us for tile 0 at 1M, the
// critical path of the transducer kernel).  Here the whole tile is handled the way passes C - E handle the population:
//   predict   an approximate prefix P (parallel rounding) gives every element the binade of the sum it is added to
//             (be = the binade AFTER its predecessor, handed on literally, so a run of elements between two predicted
//             crossings sits on ONE grid by construction) and marks the elements whose add leaves it;
//   scan      transducers of the non-crossing elements on their grids, ordered scan SEGMENTED at the crossing elements
//             (per element: the composition since the last crossing);
//   verify    one wave walks the crossings with the EXACT sum: the segment in front of crossing k maps S -> S_A < 2^53
//             (nothing inside it leaves the binade), S_A + floor(w_c / q) >= 2^53 (the crossing element does), the
//             genuine fp64 add lands in the predicted next binade.  ~35 instructions per crossing;
//   write     element i of a verified segment: S_in + composition-up-to-i, on the segment's grid.
// Whatever fails verification (or more than XF_KMAX predicted crossings) is left to the loop below, which resumes
// behind the last verified crossing.  Predictions are hints; only verified integers reach the cdf: the same bits.
#define XF_KMAX 64
struct XFast {
    int c_pos[XF_KMAX], c_e[XF_KMAX], c_ea[XF_KMAX];
    double c_w[XF_KMAX], cross_s[XF_KMAX];
    long long c_nf[XF_KMAX], c_T0[XF_KMAX], c_T1[XF_KMAX];
    long long seg_S[XF_KMAX + 1];
    int seg_e[XF_KMAX + 1];
    long long T_last0, T_last1;
    double wsum[XT_THREADS / 64];
    int wcnt[XT_THREADS / 64], wf[XT_THREADS / 64];
    TD wv[XT_THREADS / 64];
    int last_ba[XT_THREADS];
    int k_ok;
};

__device__ __forceinline__ int binade_bits(double p) {  // binade_of for p >= 0 (exponent field; 0 and subnormals: -1022)
    const int ex = (int)(((unsigned long long)__double_as_longlong(p) >> 52) & 0x7FF);
    return ex ? ex - 1023 : -1022;
}

// Leaves *sh_pos = first element not written, *sh_s = exact running sum in front of it (thread 0; the caller synchronises).
__device__ void exact_tile_fast(double* __restrict__ cdf, int64_t lo, int64_t hi, double s0, const double (&wv)[XT_E],
                                XFast& F, double* sh_s, long long* sh_pos) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = lo + (int64_t)tid * XT_E;
    // approximate prefixes
    double tsum = 0.0;
#pragma unroll
    for (int j = 0; j < XT_E; j++) tsum += wv[j];
    double inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) F.wsum[wave] = inc;
    __syncthreads();
    double p = s0;
    for (int v = 0; v < wave; v++) p += F.wsum[v];
    p += inc - tsum;
    int ba[XT_E];
#pragma unroll
    for (int j = 0; j < XT_E; j++) {
        p += wv[j];
        ba[j] = binade_bits(p);
    }
    F.last_ba[tid] = ba[XT_E - 1];
    __syncthreads();
    int be[XT_E];
    bool cross[XT_E];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < XT_E; j++) {
        be[j] = j ? ba[j - 1] : (tid ? F.last_ba[tid - 1] : binade_bits(s0));
        cross[j] = ba[j] != be[j];
        cnt += cross[j] ? 1 : 0;
    }
    int cinc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(cinc, o, 64);
        if (lane >= o) cinc += v;
    }
    // transducers, the thread's aggregate since its last crossing
    long long ta0[XT_E], ta1[XT_E], nfc[XT_E];
    TD agg = {0, 0};
    int f = 0;
#pragma unroll
    for (int j = 0; j < XT_E; j++) {
        td_of(wv[j], be[j], ta0[j], ta1[j], nfc[j]);
        if (cross[j]) {
            agg = TD{0, 0};
            f = 1;
        } else {
            agg = td_compose(agg, TD{ta0[j], ta1[j]});
        }
    }
    // segmented inclusive scan over the wave (a lane that has met a crossing keeps its value)
    TD iv = agg;
    int fl = f;
#define XF_SEG_STEP(CTRL, MASK)                                               \
    {                                                                         \
        const TD l_ = {dpp_ll<CTRL, MASK>(iv.a0), dpp_ll<CTRL, MASK>(iv.a1)}; \
        const int lf_ = __builtin_amdgcn_update_dpp(0, fl, CTRL, MASK, 0xF, false); \
        const TD c_ = td_compose(l_, iv);                                     \
        iv = fl ? iv : c_;                                                    \
        fl |= lf_;                                                            \
    }
    XF_SEG_STEP(DPP_ROW_SHR(1), 0xF)
    XF_SEG_STEP(DPP_ROW_SHR(2), 0xF)
    XF_SEG_STEP(DPP_ROW_SHR(4), 0xF)
    XF_SEG_STEP(DPP_ROW_SHR(8), 0xF)
    XF_SEG_STEP(DPP_ROW_BCAST15, 0xA)
    XF_SEG_STEP(DPP_ROW_BCAST31, 0xC)
#undef XF_SEG_STEP
    const TD ev = td_wave_shr1(iv);  // what lies in front of this thread inside the wave (identity in lane 0)
    const int ef = __builtin_amdgcn_update_dpp(0, fl, DPP_WAVE_SHR1, 0xF, 0xF, false);
    if (lane == 63) F.wv[wave] = iv, F.wf[wave] = fl, F.wcnt[wave] = cinc;
    __syncthreads();
    TD pv = {0, 0};
    int k = cinc - cnt, K = 0;
#pragma unroll
    for (int v = 0; v < XT_THREADS / 64; v++) {
        if (v < wave) {
            pv = F.wf[v] ? F.wv[v] : td_compose(pv, F.wv[v]);
            k += F.wcnt[v];
        }
        K += F.wcnt[v];
    }
    if (K > XF_KMAX) {  // uniform: not this way
        if (tid == 0) *sh_s = s0, *sh_pos = lo;
        return;
    }
    TD run = ef ? ev : td_compose(pv, ev);
    const int k0 = k;
    long long in0[XT_E], in1[XT_E];
    int seg[XT_E];
#pragma unroll
    for (int j = 0; j < XT_E; j++) {
        if (cross[j]) {
            F.c_pos[k] = tid * XT_E + j;
            F.c_e[k] = be[j];
            F.c_ea[k] = ba[j];
            F.c_w[k] = wv[j];
            F.c_nf[k] = nfc[j];
            F.c_T0[k] = run.a0;
            F.c_T1[k] = run.a1;
            run = TD{0, 0};
            seg[j] = k;  // (its own index among the crossings)
            k++;
        } else {
            run = td_compose(run, TD{ta0[j], ta1[j]});
            seg[j] = k;  // the segment behind crossing k - 1
        }
        in0[j] = run.a0;
        in1[j] = run.a1;
    }
    if (tid == XT_THREADS - 1) F.T_last0 = run.a0, F.T_last1 = run.a1;
    __syncthreads();
    if (tid < 64) {  // the walk: every lane the same scalars
        double s = s0, s_end = s0;
        int kk = 0;
        for (; kk < K; kk++) {
            int e_cur;
            long long S;
            sum_split(s, e_cur, S);
            if (e_cur != F.c_e[kk]) break;
            const long long S_A = S + ((S & 1) ? F.c_T1[kk] : F.c_T0[kk]);
            if (!(S_A < TWO53_LL && S_A + F.c_nf[kk] >= TWO53_LL)) break;  // nothing in front of c leaves the binade, c does
            const double s_new = sum_join(S_A, e_cur) + F.c_w[kk];
            if (binade_bits(s_new) != F.c_ea[kk]) break;  // ... into the binade the next segment was built on
            if (lane == 0) F.seg_S[kk] = S, F.seg_e[kk] = e_cur, F.cross_s[kk] = s_new;
            s = s_new;
        }
        if (kk == K) {  // the stretch behind the last crossing
            int e_cur;
            long long S;
            sum_split(s, e_cur, S);
            const long long S_end = S + ((S & 1) ? F.T_last1 : F.T_last0);
            if (S_end < TWO53_LL) {
                if (lane == 0) F.seg_S[K] = S, F.seg_e[K] = e_cur;
                s_end = sum_join(S_end, e_cur);
                kk = K + 1;
            }
        }
        if (lane == 0) {
            F.k_ok = kk;
            *sh_s = kk == K + 1 ? s_end : s;
            *sh_pos = kk == K + 1 ? hi : (kk == 0 ? lo : lo + F.c_pos[kk - 1] + 1);
        }
    }
    __syncthreads();
    const int k_ok = F.k_ok;
#pragma unroll
    for (int j = 0; j < XT_E; j++) {
        if (base + j >= hi || seg[j] >= k_ok) continue;
        if (cross[j]) {
            cdf[base + j] = F.cross_s[seg[j]];
        } else {
            const long long S_in = F.seg_S[seg[j]];
            cdf[base + j] = sum_join(S_in + ((S_in & 1) ? in1[j] : in0[j]), F.seg_e[seg[j]]);
        }
    }
    (void)k0;
}

// Exact sequential-order cumulative sum of ONE tile w[lo, hi) (hi - lo <= 2048) starting from the exact
// running sum s0, by the whole block.  Each thread owns XT_E fixed elements (loaded once); every pass scans the
// still-open elements on the grid of the current binade and restarts behind the first element whose add
// leaves the binade (that add is a genuine fp64 add).  Returns the exact running sum at `hi`.
// try_fast: several crossings expected (tile 0, tiles flagged 0): exact_tile_fast first, the loop for what it leaves.
__device__ double exact_tile(const double* __restrict__ w, double* __restrict__ cdf, int64_t lo, int64_t hi,
                             double s0, TD* sh_td, double* sh_s, long long* sh_pos, long long* sh_cross,
                             bool try_fast = false) {
    const int tid = threadIdx.x;
    const int64_t base = lo + (int64_t)tid * XT_E;
    double wv[XT_E];
#pragma unroll
    for (int j = 0; j < XT_E; j++) wv[j] = (base + j < hi) ? w[base + j] : 0.0;
    __shared__ XFast sh_fast;
    if (try_fast) {  // uniform
        exact_tile_fast(cdf, lo, hi, s0, wv, sh_fast, sh_s, sh_pos);
    } else if (tid == 0) {
        *sh_s = s0;
        *sh_pos = lo;
    }
    __syncthreads();
    // the very first elements change binade at almost every add: plain sequential adds by one thread, on values
    // staged through LDS (a dependent chain of global loads would cost ~0.5 us per element)
    __shared__ double sh_head[64];
    const bool head = lo == 0 && *sh_pos == 0;  // (uniform; with try_fast only if nothing at all was verified)
    __syncthreads();
    if (head && tid < 64) sh_head[tid] = (tid < hi) ? w[tid] : 0.0;
    __syncthreads();
    if (head && tid < 64) {
        // every lane runs the same chain (LDS broadcast reads); lane j keeps the j-th partial sum and writes it
        const int64_t lim = hi < 64 ? hi : 64;
        double s = s0, mine_s = 0.0;
        for (int64_t pos = 0; pos < lim; pos++) {
            s = s + sh_head[pos];
            if (pos == tid) mine_s = s;
        }
        if (tid < lim) cdf[tid] = mine_s;
        if (tid == 0) {
            *sh_s = s;
            *sh_pos = lim;
        }
    }
    __syncthreads();
    while (true) {
        const long long pos = *sh_pos;
        const double s = *sh_s;
        if (pos >= hi) break;
        __syncthreads();  // everyone has read the state
        if (tid == 0) *sh_cross = INT64_MAX;
        const int e = binade_of(s);
        const long long S0 = (long long)ldexp(s, 52 - e);
        long long ta0[XT_E], ta1[XT_E], nf[XT_E];
        TD mine = {0, 0};
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            const int64_t i = base + j;
            td_of((i >= pos && i < hi) ? wv[j] : 0.0, e, ta0[j], ta1[j], nf[j]);  // closed elements: identity
            mine = td_compose(mine, TD{ta0[j], ta1[j]});
        }
        TD excl, total;
        td_block_scan(mine, excl, total, sh_td);  // contains __syncthreads: the sh_cross reset is visible after it
        long long S = S0 + ((S0 & 1) ? excl.a1 : excl.a0);
        const long long S_in = S;
        long long Sout[XT_E];
        int cross_j = XT_E;
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            const int64_t i = base + j;
            if (cross_j == XT_E) {
                if (i >= pos && i < hi && S + nf[j] >= TWO53_LL) {
                    cross_j = j;
                } else {
                    S += (S & 1) ? ta1[j] : ta0[j];
                }
            }
            Sout[j] = S;
        }
        if (cross_j < XT_E) atomicMin(sh_cross, (long long)(base + cross_j));
        __syncthreads();
        const long long c = *sh_cross;  // first crossing index (global) or INT64_MAX
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            const int64_t i = base + j;
            if (i >= pos && i < hi && i < c) cdf[i] = ldexp((double)Sout[j], e - 52);
        }
        if (c < hi) {
#pragma unroll
            for (int j = 0; j < XT_E; j++) {
                if (c == base + j) {  // owner of the crossing element: genuine fp64 add, restart behind it
                    const long long Sprev = (j == 0) ? S_in : Sout[j > 0 ? j - 1 : 0];
                    const double s_new = ldexp((double)Sprev, e - 52) + wv[j];
                    cdf[c] = s_new;
                    *sh_s = s_new;
                    *sh_pos = c + 1;
                }
            }
        } else {
            if (tid == XT_THREADS - 1) {  // no crossing left: the block total is the state after the tile
                const long long Send = S0 + ((S0 & 1) ? total.a1 : total.a0);
                *sh_s = ldexp((double)Send, e - 52);
                *sh_pos = hi;
            }
        }
        __syncthreads();
    }
    return *sh_s;
}

// ---- multi-tile exact cdf ---------------------------------------------------------------------------
// Pass C: per tile, from the APPROXIMATE incoming prefix (fast scan) guess the binade e of the running sum and
// where (if anywhere) it leaves that binade inside the tile:
//   flag 1  no crossing predicted: ONE transducer (a0, a1) on the grid of e for the whole tile
//   flag 2  one crossing predicted at element c: transducer A = elements before c on the grid of e, transducer
//           B = elements after c on the grid of e+1; element c itself is added with a genuine fp64 add
//   flag 0  anything else (empty prefix, two crossings): the chain processes the tile element-wise
// Predictions are only hints: pass D verifies every one of them against the exact running sum.
// tile_info (long long, 4 per tile): {a0, a1, e, flag};  tile_split (4 per tile): {b0, b1, c - tile start, nf_c}
__device__ __forceinline__ void k_exact_tile_td_body(int64_t n, const double* __restrict__ w,
                                                             const double lo, int64_t n_tiles,
                                                             long long* __restrict__ tile_info,
                                                             long long* __restrict__ tile_split,
                                                             double* __restrict__ tile_s2,
                                                             long long* __restrict__ rec_pk = nullptr) {
    __shared__ TD sh_td[XT_THREADS / 64 + 1];
    __shared__ double sh_ws[XT_THREADS / 64];
    __shared__ int sh_c;
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = binade_of(lo);
    const double b1 = ldexp(1.0, e + 1);
    const int64_t base = t * ASMC_SCAN_TILE + (int64_t)tid * XT_E;
    double wv[XT_E];
#pragma unroll
    for (int j = 0; j < XT_E; j++) wv[j] = (base + j < n) ? w[base + j] : 0.0;
    // approximate inclusive prefix of every element (parallel-order rounding is fine for a prediction)
    double tsum = 0.0;
#pragma unroll
    for (int j = 0; j < XT_E; j++) tsum += wv[j];
    double inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) sh_ws[wave] = inc;
    if (tid == 0) sh_c = ASMC_SCAN_TILE;
    __syncthreads();
    double pre = lo + (inc - tsum);
    for (int v = 0; v < wave; v++) pre += sh_ws[v];
    double tile_end = lo;
    for (int v = 0; v < XT_THREADS / 64; v++) tile_end += sh_ws[v];
    {
        double p = pre;
        int cj = -1;
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            p += wv[j];
            if (cj < 0 && p >= b1 && base + j < n) cj = j;
        }
        if (cj >= 0) atomicMin(&sh_c, tid * XT_E + cj);
    }
    __syncthreads();
    const int c = sh_c;  // predicted crossing element (tile-local) or ASMC_SCAN_TILE
    int flag = 1;
    if (!(lo > 0.0)) flag = 0;
    if (c < ASMC_SCAN_TILE) flag = (lo > 0.0 && tile_end < 2.0 * b1) ? 2 : 0;
    TD mineA = {0, 0}, mineB = {0, 0};
    long long nf_c = 0;
    if (flag != 0) {
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            const int k = tid * XT_E + j;
            long long a0, a1, nf;
            if (k < c) {
                td_of(wv[j], e, a0, a1, nf);
                mineA = td_compose(mineA, TD{a0, a1});
            } else if (k == c) {
                td_of(wv[j], e, a0, a1, nf);
                nf_c = nf;
            } else {
                td_of(wv[j], e + 1, a0, a1, nf);
                mineB = td_compose(mineB, TD{a0, a1});
            }
        }
    }
    TD excl, totalA, totalB = {0, 0};
    td_block_scan(mineA, excl, totalA, sh_td);
    if (flag == 2) td_block_scan(mineB, excl, totalB, sh_td);  // uniform per block
    if (rec_pk) {  // sharded form: one packed record per tile {a0, a1, e, flag, b0, b1, c, nf_c, bits(w_c)}
        long long* r = rec_pk + ASMC_CDF_REC * t;
        if (tid == 0) {
            r[0] = totalA.a0, r[1] = totalA.a1, r[2] = e, r[3] = flag;
            r[4] = totalB.a0, r[5] = totalB.a1, r[6] = c;
            if (flag != 2) r[7] = 0, r[8] = 0;
        }
        if (flag == 2 && tid == c / XT_E) {
            double wc = 0.0;
#pragma unroll
            for (int j = 0; j < XT_E; j++)
                if (j == c % XT_E) wc = wv[j];
            r[7] = nf_c;
            r[8] = __double_as_longlong(wc);
        }
        return;
    }
    if (tid == 0) {
        tile_info[4 * t + 0] = totalA.a0;
        tile_info[4 * t + 1] = totalA.a1;
        tile_info[4 * t + 2] = e;
        tile_info[4 * t + 3] = flag;
        tile_split[4 * t + 0] = totalB.a0;
        tile_split[4 * t + 1] = totalB.a1;
        tile_split[4 * t + 2] = c;
    }
    if (flag == 2 && tid == c / XT_E) {
        tile_split[4 * t + 3] = nf_c;
        double wc = 0.0;
#pragma unroll
        for (int j = 0; j < XT_E; j++)
            if (j == c % XT_E) wc = wv[j];
        tile_s2[t] = wc;  // the chain reads it from here and replaces it by the sum behind element c
    }
}

// launch shim: the approximate total is still on the device (no host round trip)
__global__ __launch_bounds__(XT_THREADS) void k_exact_tile_td_launch(int64_t n, const double* __restrict__ w,
                                                                    const double* __restrict__ approx_prefix,
                                                                    const double* __restrict__ approx_total,
                                                                    int64_t n_tiles, long long* __restrict__ tile_info,
                                                                    long long* __restrict__ tile_split,
                                                                    double* __restrict__ tile_s2, double* __restrict__ cdf,
                                                                    double carry_in, double* __restrict__ tile_s,
                                                                    int first_exact, long long* __restrict__ rec_pk) {
    if (blockIdx.x == 0 && first_exact) {
        // Tile 0 enters with an EXACTLY known running sum (the carry), and it is the one tile that leaves several
        // binades in a row (the sum grows from nothing): it is scanned element-wise right here, in parallel with the
        // other tiles' transducers, instead of in the sequential chain.  tile_s[0] = carry, tile_s2[0] = exact sum
        // behind the tile; flag 3 tells the chain to start behind it and the write pass to leave it alone.
        __shared__ TD sh_td0[XT_THREADS / 64 + 1];
        __shared__ double sh_s0;
        __shared__ long long sh_pos0, sh_cross0;
        const int64_t hi = ASMC_SCAN_TILE < n ? ASMC_SCAN_TILE : n;
        const double s_out = exact_tile(w, cdf, 0, hi, carry_in, sh_td0, &sh_s0, &sh_pos0, &sh_cross0, true);
        if (threadIdx.x == 0) {
            if (rec_pk) {
                rec_pk[0] = rec_pk[1] = rec_pk[2] = 0, rec_pk[3] = 3;
                rec_pk[4] = rec_pk[5] = rec_pk[6] = rec_pk[7] = 0;
                rec_pk[8] = __double_as_longlong(s_out);
            } else {
                tile_info[0] = 0, tile_info[1] = 0, tile_info[2] = 0, tile_info[3] = 3;
                tile_split[0] = tile_split[1] = tile_split[2] = tile_split[3] = 0;
                tile_s[0] = carry_in;
                tile_s2[0] = s_out;
            }
        }
        return;
    }
    k_exact_tile_td_body(n, w, approx_prefix[blockIdx.x], n_tiles, tile_info, tile_split, tile_s2, rec_pk);
}

// The sharded step's form: the tiles' sums arrive with the weights (k_weights_map_shard) and the incoming sum is a device-side
// result; a block adds up what lies in front of its tile itself (hints for the binade guess only: any order will do) -
// k_tile_sum and k_scan_tiles are not launched
__global__ __launch_bounds__(XT_THREADS) void k_exact_tile_td_shard(int64_t n, const double* __restrict__ w,
                                                                   const double* __restrict__ tile_sums,
                                                                   const double* __restrict__ carry_dev, int64_t n_tiles,
                                                                   double* __restrict__ cdf, int first_exact,
                                                                   long long* __restrict__ rec_pk) {
    if (blockIdx.x == 0 && first_exact) {
        __shared__ TD sh_td0[XT_THREADS / 64 + 1];
        __shared__ double sh_s0;
        __shared__ long long sh_pos0, sh_cross0;
        const int64_t hi = ASMC_SCAN_TILE < n ? ASMC_SCAN_TILE : n;
        const double s_out = exact_tile(w, cdf, 0, hi, 0.0, sh_td0, &sh_s0, &sh_pos0, &sh_cross0, true);
        if (threadIdx.x == 0) {
            rec_pk[0] = rec_pk[1] = rec_pk[2] = 0, rec_pk[3] = 3;
            rec_pk[4] = rec_pk[5] = rec_pk[6] = rec_pk[7] = 0;
            rec_pk[8] = __double_as_longlong(s_out);
        }
        return;
    }
    __shared__ double s_pre[XT_THREADS / 64];
    double acc = 0.0;
    for (int64_t u = threadIdx.x; u < (int64_t)blockIdx.x; u += XT_THREADS) acc += tile_sums[u];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_pre[threadIdx.x >> 6] = acc;
    __syncthreads();
    double pre = first_exact ? 0.0 : *carry_dev;
    for (int k = 0; k < XT_THREADS / 64; k++) pre += s_pre[k];
    k_exact_tile_td_body(n, w, pre, n_tiles, (long long*)nullptr, (long long*)nullptr, (double*)nullptr, rec_pk);
}

// Pass D: one block chains the EXACT running sum through the tiles.  Flag-1 tiles cost O(1) (verify the binade
// guess against the exact incoming sum, apply the tile transducer); flag-2 tiles cost O(1) as well (apply A, verify
// that the add of element c is the one that leaves the binade, do that add in fp64, apply B on the next grid); any
// tile that fails a check or is flagged 0 is processed element-wise by exact_tile.  tile_s[t] = exact sum entering
// tile t, tile_s2[t] = exact sum right after element c; tile_info[4t+3] stays 1 / 2 if pass E must still write the
// tile and becomes 0 if it has been written here.
//
// Round 5: the scans left the sequential chain.  A run of flag-1 tiles of one predicted binade (a PIECE) acts on the
// running sum through the composition of its transducers whatever the sum is, so the ordered scans INSIDE the pieces do
// not depend on the chain: the four waves compute them for a whole chunk of 512 tiles at once (segmented DPP scans, 64
// tiles each), and the chain itself - wave 0 alone, no workgroup barrier - only verifies a piece's binade against the
// exact incoming sum, adds the prefixes (one ballot for the 2^53 check), and steps over the flag-2 tiles: ~30
// instructions per piece instead of one 64-lane scan behind every crossing (1M particles: 15 batches x 1.12 us +
// 8 x 0.6 us of a lone wave -> the figure in profiles/README.md).  Same integers in the same association-free algebra:
// the same bits.
__device__ __forceinline__ void exact_chain_body(int64_t n, const double* __restrict__ w, double* __restrict__ cdf,
                                                 int64_t n_tiles, long long* __restrict__ tile_info,
                                                 const long long* __restrict__ tile_split, double* __restrict__ tile_s,
                                                 double* __restrict__ tile_s2, double* __restrict__ total_out) {
    __shared__ TD sh_td[XT_THREADS / 64 + 1];
    __shared__ double sh_s, sh_walk_s;
    __shared__ long long sh_pos, sh_cross, sh_walk_t;
    __shared__ int sh_need;
    // tile records of the current chunk of CHAIN_CHUNK tiles, staged once by the whole block: the walk below then
    // reads LDS instead of paying a global-load latency in every round of the sequential chain
    constexpr int CHAIN_CHUNK = 512;
    constexpr int GROUPS = CHAIN_CHUNK / 64;
    __shared__ __attribute__((aligned(16))) long long sh_info[CHAIN_CHUNK * 4];
    __shared__ __attribute__((aligned(16))) long long sh_split[CHAIN_CHUNK * 4];
    __shared__ double sh_wc[CHAIN_CHUNK];  // weight of the predicted crossing element (pass C leaves it in tile_s2)
    __shared__ long long sh_inc[CHAIN_CHUNK * 2];  // inclusive composition of the tile's piece up to and including the tile
    __shared__ unsigned long long sh_headm[GROUPS], sh_okm[GROUPS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t t = 1;            // tile 0 has been scanned exactly by pass C
    double s = tile_s2[0];    // exact running sum behind tile 0
    while (t < n_tiles) {
        __syncthreads();
        const int64_t chunk0 = t;  // (re)stage: the tiles from t on
        const int64_t n_in = n_tiles - chunk0 < CHAIN_CHUNK ? n_tiles - chunk0 : CHAIN_CHUNK;
        {   // every load of the chunk in flight at once (16-byte halves of the 32-byte records): one memory latency
            static_assert(CHAIN_CHUNK % XT_THREADS == 0, "whole tiles per thread");
            constexpr int PER = CHAIN_CHUNK / XT_THREADS;
            typedef unsigned long long __attribute__((ext_vector_type(2))) rec_half;
            const rec_half* gi = reinterpret_cast<const rec_half*>(tile_info + 4 * chunk0);
            const rec_half* gs = reinterpret_cast<const rec_half*>(tile_split + 4 * chunk0);
            rec_half vi[2 * PER], vs[2 * PER];
            double wc[PER];
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int k = (int)threadIdx.x + q * XT_THREADS;
                const bool in = k < n_in;
                const rec_half z = {0ull, 0ull};
                vi[2 * q] = in ? gi[2 * k] : z;
                vi[2 * q + 1] = in ? gi[2 * k + 1] : z;
                vs[2 * q] = in ? gs[2 * k] : z;
                vs[2 * q + 1] = in ? gs[2 * k + 1] : z;
                wc[q] = in ? tile_s2[chunk0 + k] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int k = (int)threadIdx.x + q * XT_THREADS;
                reinterpret_cast<rec_half*>(sh_info)[2 * k] = vi[2 * q];
                reinterpret_cast<rec_half*>(sh_info)[2 * k + 1] = vi[2 * q + 1];
                reinterpret_cast<rec_half*>(sh_split)[2 * k] = vs[2 * q];
                reinterpret_cast<rec_half*>(sh_split)[2 * k + 1] = vs[2 * q + 1];
                sh_wc[k] = wc[q];
            }
        }
        __syncthreads();
        // the pieces' scans, 64 tiles per wave and pass: a piece starts at a flag-1 tile whose predecessor (in the group) is not
        // flag 1 or sits in another predicted binade; every other tile is a piece of its own (identity)
        for (int g = wave; g < GROUPS; g += XT_THREADS / 64) {
            const int k = g * 64 + lane;
            long long a0 = 0, a1 = 0;
            int ee = 0;
            bool ok = false;
            if (k < n_in) {
                const long long* rec = sh_info + 4 * k;
                ok = rec[3] == 1;
                ee = (int)rec[2];
                // (clamped to 2^54: still "past 2^53" for the chain's check, and 64 of them cannot overflow an int64, so the
                // scan below composes without the saturation)
                a0 = ok ? (rec[0] < TD_BIG ? rec[0] : TD_BIG) : 0;
                a1 = ok ? (rec[1] < TD_BIG ? rec[1] : TD_BIG) : 0;
            }
            const int e_prev = __builtin_amdgcn_update_dpp(0, ee, DPP_WAVE_SHR1, 0xF, 0xF, false);
            const int ok_prev = __builtin_amdgcn_update_dpp(0, (int)ok, DPP_WAVE_SHR1, 0xF, 0xF, false);
            int f = (!ok || lane == 0 || !ok_prev || e_prev != ee) ? 1 : 0;  // piece start (or a tile outside the pieces)
            const unsigned long long headm = __ballot(f != 0), okm = __ballot(ok);
            TD inc = {a0, a1};
            // segmented inclusive scan: a lane that has met its piece's start keeps its value
#define SEG_STEP(CTRL, MASK)                                                    \
    {                                                                           \
        const TD l_ = {dpp_ll<CTRL, MASK>(inc.a0), dpp_ll<CTRL, MASK>(inc.a1)}; \
        const int lf_ = __builtin_amdgcn_update_dpp(0, f, CTRL, MASK, 0xF, false); \
        const TD c_ = {l_.a0 + ((l_.a0 & 1) ? inc.a1 : inc.a0), l_.a1 + ((l_.a1 & 1) ? inc.a0 : inc.a1)}; \
        inc = f ? inc : c_;                                                     \
        f |= lf_;                                                               \
    }
            SEG_STEP(DPP_ROW_SHR(1), 0xF)
            SEG_STEP(DPP_ROW_SHR(2), 0xF)
            SEG_STEP(DPP_ROW_SHR(4), 0xF)
            SEG_STEP(DPP_ROW_SHR(8), 0xF)
            SEG_STEP(DPP_ROW_BCAST15, 0xA)
            SEG_STEP(DPP_ROW_BCAST31, 0xC)
#undef SEG_STEP
            sh_inc[2 * k] = inc.a0;
            sh_inc[2 * k + 1] = inc.a1;
            if (lane == 0) sh_headm[g] = headm, sh_okm[g] = okm;
        }
        __syncthreads();
        // the chain: wave 0 alone, every lane the same scalars (same-address LDS reads, readlane)
        if (threadIdx.x < 64) {
            int need = 0;
            for (int g = 0; g < GROUPS && !need; g++) {
                const int64_t base_t = chunk0 + (int64_t)g * 64;
                if (base_t >= n_tiles) break;
                const int k = g * 64 + lane;
                const int64_t my_t = base_t + lane;
                const long long inc0 = sh_inc[2 * k], inc1 = sh_inc[2 * k + 1];
                const int ee = (int)sh_info[4 * k + 2];
                const unsigned long long headm = (unsigned long long)readlane_ll((long long)sh_headm[g], 0),
                                         okm = (unsigned long long)readlane_ll((long long)sh_okm[g], 0);  // scalars
                const bool head = (headm >> lane) & 1;
                TD ex = td_wave_shr1(TD{inc0, inc1});  // composition of the piece in front of this tile
                if (head) ex = TD{0, 0};
                int pos = 0;
                t = base_t;
                while (pos < 64 && base_t + pos < n_tiles) {
                    t = base_t + pos;
                    if (!((okm >> pos) & 1)) {  // a tile outside the pieces
                        const int kk = g * 64 + pos;
                        if (sh_info[4 * kk + 3] == 2) {  // predicted single crossing: O(1) with verification
                            const long long* rec = sh_info + 4 * kk;
                            const int e = (int)rec[2];
                            const long long* sp = sh_split + 4 * kk;
                            const long long b0 = sp[0], b1 = sp[1], nf_c = sp[3];
                            bool done = false;
                            int e_s;
                            long long S1;
                            sum_split(s, e_s, S1);
                            if (s > 0.0 && e_s == e) {
                                const long long S_A = S1 + ((S1 & 1) ? rec[1] : rec[0]);
                                if (S_A < TWO53_LL && S_A + nf_c >= TWO53_LL) {  // no add before c leaves the binade, c's does
                                    const double s_new = sum_join(S_A, e) + sh_wc[kk];  // (2^52 <= S_A < 2^53, or e = -1022)
                                    int e2;
                                    long long S2;
                                    sum_split(s_new, e2, S2);
                                    if (e2 == e + 1) {
                                        const long long S_B = S2 + ((S2 & 1) ? b1 : b0);
                                        if (S_B < TWO53_LL) {  // and none after it does
                                            if (lane == 0) {
                                                tile_s[t] = s;
                                                tile_s2[t] = s_new;
                                            }
                                            s = sum_join(S_B, e + 1);
                                            done = true;
                                        }
                                    }
                                }
                            }
                            if (done) {
                                pos++;
                                t = base_t + pos;
                                continue;
                            }
                        }
                        need = 1;  // the whole block, element-wise
                        break;
                    }
                    // the piece [pos, end)
                    const unsigned long long rest = pos < 63 ? headm >> (pos + 1) : 0ull;
                    int end = rest ? pos + 1 + (int)__builtin_ctzll(rest) : 64;
                    const int e_p = __builtin_amdgcn_readlane(ee, pos);
                    int e_cur;
                    long long S;
                    sum_split(s, e_cur, S);
                    if (!(s > 0.0) || e_cur != e_p) {  // the binade guess fails at the piece's first tile
                        need = 1;
                        break;
                    }
                    const long long S_out = S + ((S & 1) ? inc1 : inc0);
                    const long long my_S = S + ((S & 1) ? ex.a1 : ex.a0);
                    const bool in_piece = lane >= pos && lane < end;
                    const unsigned long long ovf = __ballot(in_piece && S_out >= TWO53_LL);
                    if (ovf) end = (int)__builtin_ctzll(ovf);  // that tile would push S past 2^53
                    if (lane >= pos && lane < end) tile_s[my_t] = sum_join(my_S, e_cur);
                    if (end > pos) s = sum_join(readlane_ll(S_out, end - 1), e_cur);
                    pos = end;
                    t = base_t + pos;
                    if (ovf) {
                        need = 1;
                        break;
                    }
                }
            }
            if (lane == 0) {
                sh_walk_t = t;
                sh_walk_s = s;
                sh_need = need;
            }
        }
        __syncthreads();
        t = sh_walk_t;
        s = sh_walk_s;
        const bool need = sh_need != 0;
        __syncthreads();
        if (!need) continue;  // the chunk is used up (or the chain is done)
        const int64_t lo = t * ASMC_SCAN_TILE;
        const int64_t hi = (lo + ASMC_SCAN_TILE < n) ? lo + ASMC_SCAN_TILE : n;
        if (threadIdx.x == 0) {
            tile_info[4 * t + 3] = 0;
            tile_s[t] = s;
        }
        // (a tile pass C itself flagged 0 expects several crossings; one that failed a verification has one, somewhere else)
        s = exact_tile(w, cdf, lo, hi, s, sh_td, &sh_s, &sh_pos, &sh_cross, sh_info[4 * (t - chunk0) + 3] == 0);
        __syncthreads();
        t++;
    }
    if (threadIdx.x == 0) *total_out = s;
}

__global__ __launch_bounds__(XT_THREADS) void k_exact_chain(int64_t n, const double* __restrict__ w,
                                                           double* __restrict__ cdf, double carry_in,
                                                           int64_t n_tiles, long long* __restrict__ tile_info,
                                                           const long long* __restrict__ tile_split,
                                                           double* __restrict__ tile_s, double* __restrict__ tile_s2,
                                                           double* __restrict__ total_out) {
    exact_chain_body(n, w, cdf, n_tiles, tile_info, tile_split, tile_s, tile_s2, total_out);
}


// ---- sharded exact cdf (one process per GPU; the reference has no distributed mode) -------------------------------
// numpy's cumsum over the GLOBAL weight vector is one sequential chain through every rank's shard.  Passes A-C above
// need only an APPROXIMATE incoming sum, so every rank runs them on its own shard at once and emits one packed record
// per tile; the records of all ranks are all-gathered (72 B per 2048 particles) and every rank then walks the SAME
// chain over all of them - the exact sum entering each of its own tiles and the global total come out bit-identical
// on every rank without a rank waiting for another rank's scan.
// A tile whose record fails verification needs its elements.  That is the rule, not the exception, at the very end of
// the population: normalised weights sum to 1 up to the accumulated rounding of the sequential sum, so whether the last
// adds cross 2^0 cannot be predicted from an approximate prefix.  The rank that OWNS such a tile scans it element-wise
// right here (exact_tile) and PUBLISHES {tile, exact sum behind it}; a rank that meets a failing tile it does not own
// stops.  The callers all-gather the ranks' states (ASMC_CDF_STATE doubles each) and run the kernel once more: it resumes
// where it stopped and passes foreign failing tiles through the published sums.  Two rounds settle every case in which
// the owners of failing tiles were not themselves blocked in front of them (in particular the tail case); anything else
// leaves `complete` at 0 and the caller falls back to a replicated scan over the all-gathered weights.
#define CDF_PUB_MAX 16
// state row (ASMC_CDF_STATE doubles): [0] next tile, [1] exact sum entering it, [2] complete, [3] entries, [4 + 2k] tile,
// [5 + 2k] exact sum behind that tile
__global__ __launch_bounds__(XT_THREADS) void k_exact_chain_pk(int64_t n_tiles, const long long* __restrict__ pk,
                                                              double* __restrict__ tile_s, double* __restrict__ tile_s2,
                                                              double* __restrict__ done_here,
                                                              const double* __restrict__ w, double* __restrict__ cdf,
                                                              int64_t n_local, int64_t tile0, int64_t n_tiles_local,
                                                              const double* __restrict__ states_all, int world, int rank,
                                                              double* __restrict__ state_out) {
    __shared__ TD sh_td[XT_THREADS / 64 + 1];
    __shared__ double sh_s, sh_walk_s;
    __shared__ long long sh_pos, sh_cross, sh_walk_t;
    __shared__ int sh_split_ok;
    constexpr int CHAIN_CHUNK = 512;
    __shared__ long long sh_rec[CHAIN_CHUNK * ASMC_CDF_REC];
    int64_t chunk0 = -CHAIN_CHUNK;
    int64_t t = 0;
    double s = 0.0;
    int n_pub = 0;
    bool blocked = false;
    if (!states_all) {  // first round: nothing has been scanned element-wise, nothing published (one block: no memset launches)
        for (int64_t i = threadIdx.x; i < n_tiles; i += XT_THREADS) done_here[i] = 0.0;
        if (threadIdx.x < 2 * CDF_PUB_MAX) state_out[4 + threadIdx.x] = 0.0;
        __syncthreads();
    }
    if (states_all) {  // second round: resume from this rank's own stop point
        const double* mine = states_all + (int64_t)rank * ASMC_CDF_STATE;
        t = (int64_t)mine[0];
        s = mine[1];
        n_pub = (int)mine[3];
        if (threadIdx.x < 2 * CDF_PUB_MAX) state_out[4 + threadIdx.x] = mine[4 + threadIdx.x];  // keep what was published
    } else if (pk[3] == 3) {  // the first rank's first tile was scanned element-wise from an exactly known sum (0)
        s = __longlong_as_double(pk[8]);
        if (threadIdx.x == 0) tile_s[0] = 0.0, tile_s2[0] = s;
        t = 1;
    }
    while (t < n_tiles) {
        if (t + 64 > chunk0 + CHAIN_CHUNK || t < chunk0) {
            __syncthreads();
            chunk0 = t;
            const int64_t cnt = (n_tiles - chunk0 < CHAIN_CHUNK ? n_tiles - chunk0 : CHAIN_CHUNK) * ASMC_CDF_REC;
            for (int64_t i = threadIdx.x; i < cnt; i += XT_THREADS) sh_rec[i] = pk[ASMC_CDF_REC * chunk0 + i];
            __syncthreads();
        }
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const int64_t my_t = t + lane;
            long long a0 = 0, a1 = 0, ee = 0, sf = 0;
            if (my_t < n_tiles) {
                const long long* rec = sh_rec + ASMC_CDF_REC * (my_t - chunk0);
                a0 = rec[0], a1 = rec[1], ee = rec[2], sf = rec[3];
            }
            const int e_cur = binade_of(s);
            const long long S = (s > 0.0) ? (long long)ldexp(s, 52 - e_cur) : -1;
            const bool ok = (sf == 1) && (S >= 0) && ((int)ee == e_cur);
            const unsigned long long bad_mask = ~__ballot(ok);
            int stop = bad_mask ? (int)__builtin_ctzll(bad_mask) : 64;
            TD inc = {ok ? a0 : 0, ok ? a1 : 0};
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                TD t2;
                t2.a0 = __shfl_up(inc.a0, o, 64);
                t2.a1 = __shfl_up(inc.a1, o, 64);
                if (lane >= o) inc = td_compose(t2, inc);
            }
            const long long S_out = S + ((S & 1) ? inc.a1 : inc.a0);
            long long ex0 = __shfl_up(inc.a0, 1, 64), ex1 = __shfl_up(inc.a1, 1, 64);
            if (lane == 0) ex0 = ex1 = 0;
            const long long my_S = S + ((S & 1) ? ex1 : ex0);
            const unsigned long long ovf_mask = __ballot(S_out >= TWO53_LL);
            const int first_ovf = ovf_mask ? (int)__builtin_ctzll(ovf_mask) : 64;
            stop = stop < first_ovf ? stop : first_ovf;
            const double my_s = ldexp((double)my_S, e_cur - 52);
            const long long S_end = stop > 0 ? __shfl(S_out, stop - 1, 64) : S;
            const double ss = (stop > 0) ? ldexp((double)S_end, e_cur - 52) : s;
            if (lane < stop) tile_s[my_t] = my_s;
            if (lane == 0) {
                sh_walk_t = t + stop;
                sh_walk_s = ss;
            }
        }
        __syncthreads();
        const bool advanced_full = (sh_walk_t == t + 64);
        t = sh_walk_t;
        s = sh_walk_s;
        __syncthreads();
        if (advanced_full) continue;
        if (t >= n_tiles) break;
        const long long* rec = sh_rec + ASMC_CDF_REC * (t - chunk0);
        if (threadIdx.x == 0) {
            int ok = 0;
            if (rec[3] == 2) {  // predicted single crossing: O(1) with verification (as in k_exact_chain)
                const int e = (int)rec[2];
                const long long b0 = rec[4], b1 = rec[5], nf_c = rec[7];
                if (s > 0.0 && binade_of(s) == e) {
                    const long long S = (long long)ldexp(s, 52 - e);
                    const long long S_A = S + ((S & 1) ? rec[1] : rec[0]);
                    if (S_A < TWO53_LL && S_A + nf_c >= TWO53_LL) {
                        const double s_new = ldexp((double)S_A, e - 52) + __longlong_as_double(rec[8]);
                        if (binade_of(s_new) == e + 1) {
                            const long long S2 = (long long)ldexp(s_new, 52 - (e + 1));
                            const long long S_B = S2 + ((S2 & 1) ? b1 : b0);
                            if (S_B < TWO53_LL) {
                                tile_s[t] = s;
                                tile_s2[t] = s_new;
                                sh_walk_s = ldexp((double)S_B, e + 1 - 52);
                                ok = 1;
                            }
                        }
                    }
                }
            }
            if (!ok && states_all) {  // a sum some rank published for this tile in the first round?
                for (int q = 0; q < world && !ok; q++) {
                    const double* row = states_all + (int64_t)q * ASMC_CDF_STATE;
                    const int cnt = (int)row[3];
                    for (int k = 0; k < cnt; k++)
                        if ((int64_t)row[4 + 2 * k] == t) {
                            if (q != rank) tile_s[t] = s;  // (own tiles were written when they were scanned)
                            sh_walk_s = row[5 + 2 * k];
                            ok = 2;
                            break;
                        }
                }
            }
            sh_split_ok = ok;
        }
        __syncthreads();
        const int how = sh_split_ok;
        if (how) s = sh_walk_s;
        __syncthreads();
        if (how) {
            t++;
            continue;
        }
        if (t < tile0 || t >= tile0 + n_tiles_local) {  // needs the tile's elements, and they are on another rank
            blocked = true;
            break;
        }
        // own tile: element-wise, exact; the write pass leaves it alone (done_here) and only divides it by the total
        const int64_t lo = (t - tile0) * ASMC_SCAN_TILE;
        const int64_t hi = (lo + ASMC_SCAN_TILE < n_local) ? lo + ASMC_SCAN_TILE : n_local;
        if (threadIdx.x == 0) {
            tile_s[t] = s;
            done_here[t] = 1.0;
        }
        s = exact_tile(w, cdf, lo, hi, s, sh_td, &sh_s, &sh_pos, &sh_cross);
        __syncthreads();
        if (threadIdx.x == 0 && n_pub < CDF_PUB_MAX) {
            state_out[4 + 2 * n_pub] = (double)t;
            state_out[5 + 2 * n_pub] = s;
        }
        if (n_pub < CDF_PUB_MAX) n_pub++;
        t++;
    }
    if (threadIdx.x == 0) {
        state_out[0] = (double)t;
        state_out[1] = s;
        state_out[2] = (!blocked && t >= n_tiles) ? 1.0 : 0.0;
        state_out[3] = (double)n_pub;
    }
}

// out[4] = {fail, global total, (exact sum entering this shard) / total, cdf of this shard's last element}: the last two
// are the shard's slice [lo, hi) of the normalised global cdf, formed with the same division the write pass applies
__global__ void k_shard_edges(const double* __restrict__ state, const double* __restrict__ s_in,
                              const double* __restrict__ cdf_last, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = state[2] != 0.0 ? 0.0 : 1.0;
        out[1] = state[1];
        out[2] = *s_in / state[1];
        out[3] = *cdf_last;
    }
}

// ---- ordered range selection: out = u[(lo <= u) & (u < hi)] in index order (count / scan / scatter) ------------------
__global__ __launch_bounds__(ASMC_BLOCK) void k_range_count(int64_t n, const double* __restrict__ u,
                                                           const double* __restrict__ lohi,
                                                           long long* __restrict__ tiles) {
    const double lo = lohi[0], hi = lohi[1];
    const int64_t base = (int64_t)blockIdx.x * ASMC_SCAN_TILE + (int64_t)threadIdx.x * (ASMC_SCAN_TILE / ASMC_BLOCK);
    long long c = 0;
#pragma unroll
    for (int j = 0; j < ASMC_SCAN_TILE / ASMC_BLOCK; j++)
        if (base + j < n) {
            const double v = u[base + j];
            c += (v >= lo && v < hi) ? 1 : 0;
        }
    __shared__ long long s_p[ASMC_BLOCK / 64];
    c = wave_sum_ll(c);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tiles[blockIdx.x] = s_p[0] + s_p[1] + s_p[2] + s_p[3];
}

// Guide table of the resampling search, filled by the pass that writes the normalised cdf (asmc_importance_step):
// G[b] = #{k : cdf[k] <= b / NB}, b = 0..NB (k_guide_build's definition, the same table).  Element k owns the buckets
// b with cdf[k-1] <= b / NB < cdf[k], i.e. [B(cdf[k-1]), B(cdf[k])) where B(v) = the smallest b with b / NB >= v
// (B of the value in front of element 0 is 0); the last element also owns [B(cdf[n-1]), NB] -> n.
__device__ __forceinline__ long long guide_first_bucket(double v, double fnb, long long nb) {
    long long b = (long long)(v * fnb);
    b = b < 0 ? 0 : (b > nb ? nb : b);
    // settle on the very expression the search evaluates
    while (b > 0 && (double)(b - 1) / fnb >= v) b--;
    while (b <= nb && (double)b / fnb < v) b++;
    return b;
}

#define GUIDE_QUEUE 512
#define GUIDE_LONG 32  // runs of more buckets than this are filled by the whole block
// c[j]: this thread's XT_E consecutive normalised cdf values (elements base + j < n), prev0: the value in front of the
// tile's first element
__device__ __forceinline__ void guide_fill_tile(int64_t n, int64_t base, const double (&c)[XT_E], double prev0,
                                                int64_t nb, unsigned int* __restrict__ guide) {
    __shared__ double sh_last[XT_THREADS];
    __shared__ long long sh_q[GUIDE_QUEUE][2];
    __shared__ unsigned int sh_qk[GUIDE_QUEUE];
    __shared__ int sh_nq;
    const int tid = threadIdx.x;
    const double fnb = (double)nb;
    double last = prev0;
#pragma unroll
    for (int j = 0; j < XT_E; j++)
        if (base + j < n) last = c[j];
    sh_last[tid] = last;  // a thread without elements passes the value in front of it on
    if (tid == 0) sh_nq = 0;
    __syncthreads();
    if (base >= n) {
        // no elements: nothing to own (threads behind the end of the population)
    } else {
        double prev = prev0;
        if (tid > 0) prev = sh_last[tid - 1];
        auto emit = [&](long long b_lo, long long b_hi, unsigned int k) {
            if (b_hi - b_lo > GUIDE_LONG) {
                const int slot = atomicAdd(&sh_nq, 1);
                if (slot < GUIDE_QUEUE) {
                    sh_q[slot][0] = b_lo, sh_q[slot][1] = b_hi, sh_qk[slot] = k;
                    return;
                }
            }
            for (long long b = b_lo; b < b_hi; b++) guide[b] = k;
        };
        long long b_lo = (base == 0) ? 0 : guide_first_bucket(prev, fnb, nb);
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            if (base + j >= n) break;
            const long long b_hi = guide_first_bucket(c[j], fnb, nb);
            emit(b_lo, b_hi, (unsigned int)(base + j));
            // the buckets at and behind cdf[n-1] count all n elements
            if (base + j == n - 1) emit(b_hi, nb + 1, (unsigned int)n);
            b_lo = b_hi;
        }
    }
    __syncthreads();
    const int nq = sh_nq < GUIDE_QUEUE ? sh_nq : GUIDE_QUEUE;
    for (int e = 0; e < nq; e++) {
        const long long b_lo = sh_q[e][0], b_hi = sh_q[e][1];
        const unsigned int k = sh_qk[e];
        for (long long b = b_lo + tid; b < b_hi; b += XT_THREADS) guide[b] = k;
    }
}

// Pass E: write the flag-1 / flag-2 tiles from their exact incoming sums (integer scans on the tile's grids).
// guide != NULL (with norm_ptr): also fill the search's guide table for this tile's stretch of [0, 1].
__global__ __launch_bounds__(XT_THREADS) void k_exact_tile_write(int64_t n, const double* __restrict__ w,
                                                                double* __restrict__ cdf,
                                                                const long long* __restrict__ tile_info,
                                                                const long long* __restrict__ tile_split,
                                                                const double* __restrict__ tile_s,
                                                                const double* __restrict__ tile_s2,
                                                                const double* __restrict__ norm_ptr,
                                                                const long long* __restrict__ rec_pk,
                                                                const double* __restrict__ done_here,
                                                                unsigned int* __restrict__ guide, int64_t nb) {
    __shared__ TD sh_td[XT_THREADS / 64 + 1];
    const int64_t t = blockIdx.x;
    const long long flag = (done_here && done_here[t] != 0.0) ? 0
                           : (rec_pk ? rec_pk[ASMC_CDF_REC * t + 3] : tile_info[4 * t + 3]);
    // norm_ptr != NULL: the caller wants cdf / cdf[-1] (numpy's `cdf /= cdf[-1]`); the total is known by now, so
    // the division rides on this pass (tiles the chain wrote element-wise are divided in place)
    const double norm = norm_ptr ? *norm_ptr : 1.0;
    const int64_t base = t * ASMC_SCAN_TILE + (int64_t)threadIdx.x * XT_E;
    double cv[XT_E];  // the values this thread leaves in cdf[base ..]
    if (flag != 1 && flag != 2) {  // uniform per block
        if (norm_ptr) {
#pragma unroll
            for (int j = 0; j < XT_E; j++) {
                cv[j] = 0.0;
                if (base + j < n) cdf[base + j] = cv[j] = cdf[base + j] / norm;
            }
            if (guide) guide_fill_tile(n, base, cv, tile_s[t] / norm, nb, guide);
        }
        return;
    }
    const int e = (int)(rec_pk ? rec_pk[ASMC_CDF_REC * t + 2] : tile_info[4 * t + 2]);
    const int c = flag == 2 ? (int)(rec_pk ? rec_pk[ASMC_CDF_REC * t + 6] : tile_split[4 * t + 2]) : ASMC_SCAN_TILE;
    double wv[XT_E];
#pragma unroll
    for (int j = 0; j < XT_E; j++) wv[j] = (base + j < n) ? w[base + j] : 0.0, cv[j] = 0.0;
    {   // elements before c (all of them for flag 1) on the grid of e
        const long long S0 = (long long)ldexp(tile_s[t], 52 - e);
        long long ta0[XT_E], ta1[XT_E];
        TD mine = {0, 0};
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            long long nf;
            td_of(threadIdx.x * XT_E + j < c ? wv[j] : 0.0, e, ta0[j], ta1[j], nf);
            mine = td_compose(mine, TD{ta0[j], ta1[j]});
        }
        TD excl, total;
        td_block_scan(mine, excl, total, sh_td);
        long long S = S0 + ((S0 & 1) ? excl.a1 : excl.a0);
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            S += (S & 1) ? ta1[j] : ta0[j];
            if (base + j < n && threadIdx.x * XT_E + j < c) cdf[base + j] = cv[j] = ldexp((double)S, e - 52) / norm;
        }
    }
    if (flag == 2) {  // element c, then the elements behind it on the grid of e + 1
        const double s_new = tile_s2[t];
        const long long S0 = (long long)ldexp(s_new, 52 - (e + 1));
        long long ta0[XT_E], ta1[XT_E];
        TD mine = {0, 0};
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            long long nf;
            td_of(threadIdx.x * XT_E + j > c ? wv[j] : 0.0, e + 1, ta0[j], ta1[j], nf);
            mine = td_compose(mine, TD{ta0[j], ta1[j]});
        }
        TD excl, total;
        td_block_scan(mine, excl, total, sh_td);
        long long S = S0 + ((S0 & 1) ? excl.a1 : excl.a0);
#pragma unroll
        for (int j = 0; j < XT_E; j++) {
            const int k = threadIdx.x * XT_E + j;
            S += (S & 1) ? ta1[j] : ta0[j];
            if (base + j < n && k > c) cdf[base + j] = cv[j] = ldexp((double)S, e + 1 - 52) / norm;
            if (k == c) cdf[base + j] = cv[j] = s_new / norm;
        }
    }
    if (guide && norm_ptr) guide_fill_tile(n, base, cv, tile_s[t] / norm, nb, guide);
}


// k_scan_tiles folded into the transducer pass: a block adds up the tile sums in front of its tile itself (they are a
// hint for the binade guess only, any summation order will do)
__device__ __forceinline__ void exact_tile_td_scan_block(int64_t n, const double* __restrict__ w,
                                                         const double* __restrict__ tile_sums, int64_t n_tiles,
                                                         long long* __restrict__ tile_info, long long* __restrict__ tile_split,
                                                         double* __restrict__ tile_s2, double* __restrict__ cdf,
                                                         double* __restrict__ tile_s) {
    if (blockIdx.x == 0) {
        __shared__ TD sh_td0[XT_THREADS / 64 + 1];
        __shared__ double sh_s0;
        __shared__ long long sh_pos0, sh_cross0;
        const int64_t hi = ASMC_SCAN_TILE < n ? ASMC_SCAN_TILE : n;
        const double s_out = exact_tile(w, cdf, 0, hi, 0.0, sh_td0, &sh_s0, &sh_pos0, &sh_cross0, true);
        if (threadIdx.x == 0) {
            tile_info[0] = 0, tile_info[1] = 0, tile_info[2] = 0, tile_info[3] = 3;
            tile_split[0] = tile_split[1] = tile_split[2] = tile_split[3] = 0;
            tile_s[0] = 0.0;
            tile_s2[0] = s_out;
        }
        return;
    }
    __shared__ double s_pre[XT_THREADS / 64];
    double acc = 0.0;
    for (int64_t u = threadIdx.x; u < (int64_t)blockIdx.x; u += XT_THREADS) acc += tile_sums[u];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_pre[threadIdx.x >> 6] = acc;
    __syncthreads();
    double pre = 0.0;
    for (int k = 0; k < XT_THREADS / 64; k++) pre += s_pre[k];
    k_exact_tile_td_body(n, w, pre, n_tiles, tile_info, tile_split, tile_s2);
}

__global__ __launch_bounds__(XT_THREADS) void k_exact_tile_td_scan(int64_t n, const double* __restrict__ w,
                                                                  const double* __restrict__ tile_sums, int64_t n_tiles,
                                                                  long long* __restrict__ tile_info,
                                                                  long long* __restrict__ tile_split,
                                                                  double* __restrict__ tile_s2, double* __restrict__ cdf,
                                                                  double* __restrict__ tile_s) {
    exact_tile_td_scan_block(n, w, tile_sums, n_tiles, tile_info, tile_split, tile_s2, cdf, tile_s);
}

// (Passes C and D as ONE launch - the chain as an extra block behind pass C's, waiting on an arrival counter behind agent-scope
// releases - was measured in round 5: 45.9 us against 17.7 + 24.1 us for the two launches.  489 release fences and the chain's
// 50 KB of LDS in every block cost more than the launch boundary.  Not kept.)

// =============================================================================================
// fast cdf: reduce-then-scan (fixed order => deterministic)
// =============================================================================================
#define SC_E (ASMC_SCAN_TILE / ASMC_BLOCK)

__global__ __launch_bounds__(ASMC_BLOCK) void k_tile_sum(int64_t n, const double* __restrict__ w,
                                                        double* __restrict__ tiles) {
    const int64_t base = (int64_t)blockIdx.x * ASMC_SCAN_TILE + (int64_t)threadIdx.x * SC_E;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < SC_E; j++)
        if (base + j < n) acc += w[base + j];
    __shared__ double s_p[ASMC_BLOCK / 64];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tiles[blockIdx.x] = ((s_p[0] + s_p[1]) + s_p[2]) + s_p[3];
}

// exclusive scan of tile sums by one block, sequential over chunks of 1024; tiles[] overwritten
// with the exclusive prefix (+carry); total written to total_out.
__global__ __launch_bounds__(1024) void k_scan_tiles(int64_t n_tiles, double* __restrict__ tiles,
                                                    double carry_in, double* __restrict__ total_out,
                                                    const double* __restrict__ carry_dev) {
    __shared__ double s_wave[16];
    __shared__ double s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = carry_dev ? *carry_dev : carry_in;  // (sharded step: the incoming sum is a device-side result)
    __syncthreads();
    for (int64_t start = 0; start < n_tiles; start += 1024) {
        const int64_t i = start + tid;
        double v = i < n_tiles ? tiles[i] : 0.0;
        double inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            double t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        double wp = s_carry;
        for (int k = 0; k < wave; k++) wp += s_wave[k];
        if (i < n_tiles) tiles[i] = wp + (inc - v);
        __syncthreads();
        if (tid == 1023) s_carry = wp + inc;
        __syncthreads();
    }
    if (tid == 0) *total_out = s_carry;
}

__global__ __launch_bounds__(ASMC_BLOCK) void k_tile_scan(int64_t n, const double* __restrict__ w,
                                                         const double* __restrict__ tiles,
                                                         double* __restrict__ cdf,
                                                         const double* __restrict__ norm_ptr) {
    const double norm = norm_ptr ? *norm_ptr : 1.0;
    const int64_t base = (int64_t)blockIdx.x * ASMC_SCAN_TILE + (int64_t)threadIdx.x * SC_E;
    double v[SC_E];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < SC_E; j++) {
        v[j] = (base + j < n) ? w[base + j] : 0.0;
        acc += v[j];
        v[j] = acc;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = acc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __shared__ double s_wave[ASMC_BLOCK / 64];
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    double off = tiles[blockIdx.x];
    for (int k = 0; k < wave; k++) off += s_wave[k];
    off += inc - acc;
#pragma unroll
    for (int j = 0; j < SC_E; j++)
        if (base + j < n) cdf[base + j] = (off + v[j]) / norm;
}

// divisor read from the device (the total the preceding cdf pass left there): no host round trip
__global__ __launch_bounds__(ASMC_BLOCK) void k_divide_dev(int64_t n, double* __restrict__ cdf,
                                                          const double* __restrict__ last_ptr) {
    const double last = *last_ptr;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) cdf[i] = cdf[i] / last;
}

__global__ __launch_bounds__(ASMC_BLOCK) void k_divide(int64_t n, double* __restrict__ cdf, double last) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride)
        cdf[i] = cdf[i] / last;
}

// =============================================================================================
// PCG64 (XSL-RR 128/64) uniforms — numpy.random.PCG64 stream, parallel through LCG jump-ahead
// =============================================================================================
struct U128 {
    unsigned long long lo, hi;
};
__host__ __device__ __forceinline__ U128 u128_mul(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo * b.lo;
#if defined(__HIP_DEVICE_COMPILE__)
    r.hi = __umul64hi(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
#else
    r.hi = (unsigned long long)(((unsigned __int128)a.lo * b.lo) >> 64) + a.lo * b.hi + a.hi * b.lo;
#endif
    return r;
}
__host__ __device__ __forceinline__ U128 u128_add(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ULL : 0ULL);
    return r;
}

#define PCG_THREADS_LOG2 16
#define PCG_THREADS (1 << PCG_THREADS_LOG2)

// tab[i] = {A_lo, A_hi, C_lo, C_hi} for a jump of 2^i steps
__global__ __launch_bounds__(ASMC_BLOCK) void k_pcg64_uniforms(const unsigned long long* __restrict__ tab,
                                                              U128 state0, unsigned long long offset,
                                                              int64_t n, double* __restrict__ u) {
    const unsigned long long j = (unsigned long long)blockIdx.x * ASMC_BLOCK + threadIdx.x;
    if ((int64_t)j >= n) return;
    // post-step state of draw j: advance(state0, offset + j + 1)
    unsigned long long delta = offset + j + 1ULL;
    U128 st = state0;
    for (int b = 0; b < 64 && delta; b++, delta >>= 1) {
        if (delta & 1ULL) {
            U128 A = {tab[4 * b], tab[4 * b + 1]}, C = {tab[4 * b + 2], tab[4 * b + 3]};
            st = u128_add(u128_mul(st, A), C);
        }
    }
    const U128 AT = {tab[4 * PCG_THREADS_LOG2], tab[4 * PCG_THREADS_LOG2 + 1]};
    const U128 CT = {tab[4 * PCG_THREADS_LOG2 + 2], tab[4 * PCG_THREADS_LOG2 + 3]};
    for (int64_t i = (int64_t)j; i < n; i += PCG_THREADS) {
        const unsigned long long x = st.hi ^ st.lo;
        const unsigned rot = (unsigned)(st.hi >> 58);
        const unsigned long long out = (x >> rot) | (x << ((64u - rot) & 63u));
        u[i] = (double)(out >> 11) * (1.0 / 9007199254740992.0);
        st = u128_add(u128_mul(st, AT), CT);
    }
}

// Sharded owner-layout resampling: all n_total draws of the stream are generated on every rank; a wave keeps those
// inside this rank's cdf slice [lo, hi) in its own segment of `stage` (no atomics: the order of the kept draws is
// (wave, iteration, lane), a fixed function of the draw index), mapped to the local cdf's coordinate.
#define SEL_LOG2 18  // log2(ASMC_SELECT_THREADS): 4096 waves keep every SIMD busy (65536 threads: 17 us per 2M draws)
#define SEL_WAVES (ASMC_SELECT_THREADS / 64)
__global__ __launch_bounds__(ASMC_BLOCK) void k_pcg64_select(const unsigned long long* __restrict__ tab, U128 state0,
                                                            int64_t n, double lo, double hi, int64_t iters,
                                                            double* __restrict__ stage,
                                                            long long* __restrict__ counts) {
    const unsigned long long j = (unsigned long long)blockIdx.x * ASMC_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)(j >> 6);
    unsigned long long delta = j + 1ULL;
    U128 st = state0;
    for (int b = 0; b < 64 && delta; b++, delta >>= 1) {
        if (delta & 1ULL) {
            U128 A = {tab[4 * b], tab[4 * b + 1]}, C = {tab[4 * b + 2], tab[4 * b + 3]};
            st = u128_add(u128_mul(st, A), C);
        }
    }
    const U128 AT = {tab[4 * SEL_LOG2], tab[4 * SEL_LOG2 + 1]};
    const U128 CT = {tab[4 * SEL_LOG2 + 2], tab[4 * SEL_LOG2 + 3]};
    const double inv_w = 1.0 / (hi - lo);
    double* seg = stage + wave * 64 * iters;
    long long pos = 0;
    for (int64_t k = 0; k < iters; k++) {
        const int64_t i = k * ASMC_SELECT_THREADS + (int64_t)j;
        const unsigned long long x = st.hi ^ st.lo;
        const unsigned rot = (unsigned)(st.hi >> 58);
        const unsigned long long out = (x >> rot) | (x << ((64u - rot) & 63u));
        const double u = (double)(out >> 11) * (1.0 / 9007199254740992.0);
        const bool keep = i < n && u >= lo && u < hi;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            double q = (u - lo) * inv_w;
            if (q >= 1.0) q = 0x1.fffffffffffffp-1;
            seg[pos + __popcll(mask & ((1ULL << lane) - 1ULL))] = q;
        }
        pos += __popcll(mask);
        st = u128_add(u128_mul(st, AT), CT);
    }
    if (lane == 0) counts[wave] = pos;
}

// exclusive scan of the SEL_WAVES wave counts; offsets behind the counts, the total behind the offsets
__global__ __launch_bounds__(1024) void k_select_scan(long long* __restrict__ counts) {
    constexpr int PER = SEL_WAVES / 1024;  // consecutive counts per thread
    __shared__ long long s[1024];
    const int t = threadIdx.x;
    long long loc[PER], tot = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        loc[q] = counts[t * PER + q];
        tot += loc[q];
    }
    s[t] = tot;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long v = t >= o ? s[t - o] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    long long run = s[t] - tot;  // exclusive prefix of this thread's first count
#pragma unroll
    for (int q = 0; q < PER; q++) {
        counts[SEL_WAVES + t * PER + q] = run;
        run += loc[q];
    }
    if (t == 1023) counts[2 * SEL_WAVES] = s[t];
}

__global__ __launch_bounds__(64) void k_select_compact(int64_t iters, const double* __restrict__ stage,
                                                       const long long* __restrict__ counts,
                                                       double* __restrict__ q) {
    const int64_t wave = blockIdx.x;
    const long long cnt = counts[wave], off = counts[SEL_WAVES + wave];
    const double* seg = stage + wave * 64 * iters;
    for (long long t = threadIdx.x; t < cnt; t += 64) q[off + t] = seg[t];
}

__global__ __launch_bounds__(ASMC_BLOCK) void k_systematic(int64_t n_out, int64_t j0, int64_t n_total,
                                                          double u0, const double* __restrict__ v,
                                                          double* __restrict__ u) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; j < n_out; j += stride) {
        const double off = v ? v[j] : u0;
        u[j] = ((double)(j0 + j) + off) / (double)n_total;
    }
}

// =============================================================================================
// search: idx[j] = #{k : cdf[k] <= u[j]}   (searchsorted side="right")
// =============================================================================================
__global__ __launch_bounds__(ASMC_BLOCK) void k_search(int64_t n, const double* __restrict__ cdf,
                                                      int64_t n_out, const double* __restrict__ u,
                                                      int64_t* __restrict__ idx) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; j < n_out; j += stride) {
        const double key = u[j];
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (cdf[mid] <= key)
                lo = mid + 1;
            else
                hi = mid;
        }
        idx[j] = lo;
    }
}

// Guided search.  A guide table G[b] = #{k : cdf[k] <= b / NB}, b = 0..NB, is built from NB + 1 binary searches with
// SORTED queries (neighbouring threads walk the same cache lines, so the build is cheap), and every lookup then
// only searches the window [G[b], G[b+1]] of its bucket b = floor(u NB): about two random cache lines per output
// instead of the ~6 cold levels of a full binary search over the 8 MB cdf.  The result is the same upper bound.
__global__ __launch_bounds__(ASMC_BLOCK) void k_guide_build(int64_t n, const double* __restrict__ cdf, int64_t nb,
                                                           unsigned int* __restrict__ guide) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t b = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; b <= nb; b += stride) {
        const double key = (double)b / (double)nb;
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (cdf[mid] <= key)
                lo = mid + 1;
            else
                hi = mid;
        }
        guide[b] = (unsigned int)lo;
    }
}

__global__ __launch_bounds__(ASMC_BLOCK) void k_search_guided(int64_t n, const double* __restrict__ cdf, int64_t nb,
                                                             const unsigned int* __restrict__ guide, int64_t n_out,
                                                             const double* __restrict__ u, int64_t* __restrict__ idx) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    const double fnb = (double)nb;
    for (int64_t j = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; j < n_out; j += stride) {
        const double key = u[j];
        int64_t lo = 0, hi = n;
        if (key >= 0.0 && key < 1.0) {
            int64_t b = (int64_t)(key * fnb);
            b = b > nb - 1 ? nb - 1 : b;
            // make b / NB <= key < (b + 1) / NB hold for the very expressions the table was built with
            while (b > 0 && (double)b / fnb > key) b--;
            while (b < nb - 1 && (double)(b + 1) / fnb <= key) b++;
            lo = guide[b];
            hi = guide[b + 1];
        }
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (cdf[mid] <= key)
                lo = mid + 1;
            else
                hi = mid;
        }
        idx[j] = lo;
    }
}

// k_pcg64_uniforms + k_search_guided in one pass (asmc_importance_step): 2^SP_LOG2 threads, thread j takes the draws
// j, j + T, j + 2T, ... of the stream (one jump-ahead per thread, then strides of T), four at a time so that four
// independent lookups are in flight per lane; the uniforms never touch memory.
#define SP_LOG2 18
__global__ __launch_bounds__(ASMC_BLOCK) void k_search_pcg(const unsigned long long* __restrict__ tab, U128 state0,
                                                          int tlog2, int64_t n, const double* __restrict__ cdf,
                                                          int64_t nb, const unsigned int* __restrict__ guide,
                                                          int64_t n_out, int64_t* __restrict__ idx) {
    const int64_t T = (int64_t)1 << tlog2;
    const unsigned long long j = (unsigned long long)blockIdx.x * ASMC_BLOCK + threadIdx.x;
    if ((int64_t)j >= T || (int64_t)j >= n_out) return;
    unsigned long long delta = j + 1ULL;  // post-step state of draw j: advance(state0, j + 1)
    U128 st = state0;
    for (int b = 0; b < 64 && delta; b++, delta >>= 1) {
        if (delta & 1ULL) {
            U128 A = {tab[4 * b], tab[4 * b + 1]}, C = {tab[4 * b + 2], tab[4 * b + 3]};
            st = u128_add(u128_mul(st, A), C);
        }
    }
    const U128 AT = {tab[4 * tlog2], tab[4 * tlog2 + 1]};
    const U128 CT = {tab[4 * tlog2 + 2], tab[4 * tlog2 + 3]};
    const double fnb = (double)nb;
    for (int64_t i0 = (int64_t)j; i0 < n_out; i0 += 4 * T) {
        double key[4];
        int64_t lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            lo[q] = 0, hi[q] = 0, key[q] = 0.0;
            if (i0 + q * T < n_out) {
                const unsigned long long x = st.hi ^ st.lo;
                const unsigned rot = (unsigned)(st.hi >> 58);
                const unsigned long long out = (x >> rot) | (x << ((64u - rot) & 63u));
                key[q] = (double)(out >> 11) * (1.0 / 9007199254740992.0);
                st = u128_add(u128_mul(st, AT), CT);
                hi[q] = n;
                if (guide) {
                    int64_t b = (int64_t)(key[q] * fnb);
                    b = b > nb - 1 ? nb - 1 : b;
                    // make b / NB <= key < (b + 1) / NB hold for the very expressions the table was built with
                    while (b > 0 && (double)b / fnb > key[q]) b--;
                    while (b < nb - 1 && (double)(b + 1) / fnb <= key[q]) b++;
                    lo[q] = guide[b];
                    hi[q] = guide[b + 1];
                }
            }
        }
        bool any = true;
        while (any) {
            any = false;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (lo[q] < hi[q]) {
                    const int64_t mid = lo[q] + ((hi[q] - lo[q]) >> 1);
                    if (cdf[mid] <= key[q])
                        lo[q] = mid + 1;
                    else
                        hi[q] = mid;
                }
                any |= lo[q] < hi[q];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (i0 + q * T < n_out) idx[i0 + q * T] = lo[q] < n ? lo[q] : n - 1;  // (== lo: u < 1 = cdf[n-1]; a garbage cdf must not index past the rows)
    }
}

// =============================================================================================
// gather rows
// =============================================================================================
// 16-byte chunks: chunk c -> (row = c / cpr, col = c % cpr); lanes of a wave cover whole rows, so
// every random row is fetched as full 64-B+ bursts and the output is written fully coalesced.
// power-of-two rows (cpr = 2^SH 16-byte chunks): shifts instead of 64-bit divisions, four chunks in flight per lane,
// non-temporal stores (the output is not re-read by this kernel and should not displace the cdf / input rows in L2)
// The three log-probabilities of the gathered rows: one output per lane, coalesced stores.  A random 8-byte read costs
// a whole sector request, and at 1M x 32 fp64 the three of them per draw cost 40 % of the gather; with `rec` (the
// source's (ll, lp, lq, 0) records, k_pack_records) a draw reads one 32-byte sector instead.
struct alignas(32) LogRec {
    double ll, lp, lq, pad;
};
__global__ __launch_bounds__(ASMC_BLOCK) void k_pack_records(int64_t n, const double* __restrict__ ll, const double* __restrict__ lp,
                                                            const double* __restrict__ lq, LogRec* __restrict__ rec) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) rec[i] = LogRec{ll[i], lp[i], lq[i], 0.0};
}
__device__ __forceinline__ void gather_scalars(int64_t n_out, const int64_t* __restrict__ idx, const LogRec* __restrict__ rec,
                                               const double* __restrict__ ll_in, const double* __restrict__ lp_in,
                                               const double* __restrict__ lq_in, double* __restrict__ ll_out,
                                               double* __restrict__ lp_out, double* __restrict__ lq_out) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; j < n_out; j += stride) {
        const int64_t s = idx[j];
        double a, b, c;
        if (rec) {
            const LogRec r = rec[s];
            a = r.ll, b = r.lp, c = r.lq;
        } else {
            a = ll_in[s], b = lp_in[s], c = lq_in[s];
        }
        __builtin_nontemporal_store(a, &ll_out[j]);
        __builtin_nontemporal_store(b, &lp_out[j]);
        __builtin_nontemporal_store(c, &lq_out[j]);
    }
}

// CS: the rows are fp64 - the column sums of the gathered rows ride along (a lane always handles the same 16-byte piece of a
// row: the grid stride is a multiple of the pieces per row), one partial row of d sums per block in cs_partials; the
// reference fit behind the gather (asmc_mean_gram_enqueue) then needs no column-sum pass over the rows (70 us at 1M x 32).
template <int SH, bool CS>
__global__ __launch_bounds__(ASMC_BLOCK) void k_gather16_pow2(int64_t n_out, const int64_t* __restrict__ idx,
                                                             const LogRec* __restrict__ rec,
                                                             const uint4* __restrict__ x_in, uint4* __restrict__ x_out,
                                                             const double* __restrict__ ll_in,
                                                             const double* __restrict__ lp_in,
                                                             const double* __restrict__ lq_in,
                                                             double* __restrict__ ll_out, double* __restrict__ lp_out,
                                                             double* __restrict__ lq_out, double* __restrict__ cs_partials) {
    double cs0 = 0.0, cs1 = 0.0;
#ifdef GATHER_U
    constexpr int U = GATHER_U;
#else
    constexpr int U = 4;
#endif
    const int64_t total = n_out << SH;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t c0 = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; c0 < total; c0 += U * stride) {
        int64_t src[U];
        uint4 v[U];
#pragma unroll
        for (int q = 0; q < U; q++) {
            const int64_t c = c0 + q * stride;
            src[q] = c < total ? idx[c >> SH] : 0;
        }
#pragma unroll
        for (int q = 0; q < U; q++) {
            const int64_t c = c0 + q * stride;
            if (c < total) {
#ifdef GATHER_NTLOAD
                const uint4* sp4 = &x_in[(src[q] << SH) + (c & ((1 << SH) - 1))];
                v[q].x = __builtin_nontemporal_load(&sp4->x), v[q].y = __builtin_nontemporal_load(&sp4->y);
                v[q].z = __builtin_nontemporal_load(&sp4->z), v[q].w = __builtin_nontemporal_load(&sp4->w);
#else
                v[q] = x_in[(src[q] << SH) + (c & ((1 << SH) - 1))];
#endif
            }
        }
#pragma unroll
        for (int q = 0; q < U; q++) {
            const int64_t c = c0 + q * stride;
            if (c < total) {
                __builtin_nontemporal_store(v[q].x, &x_out[c].x);
                __builtin_nontemporal_store(v[q].y, &x_out[c].y);
                __builtin_nontemporal_store(v[q].z, &x_out[c].z);
                __builtin_nontemporal_store(v[q].w, &x_out[c].w);
                if (CS) {
                    cs0 += __longlong_as_double((long long)(((unsigned long long)v[q].y << 32) | v[q].x));
                    cs1 += __longlong_as_double((long long)(((unsigned long long)v[q].w << 32) | v[q].z));
                }
            }
        }
    }
    if (CS) {
        constexpr int CPR = 1 << SH;
        __shared__ double s_cs[ASMC_BLOCK / 64][2 * CPR];
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) {
            cs0 += __shfl_xor(cs0, o, 64);
            cs1 += __shfl_xor(cs1, o, 64);
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane < CPR) s_cs[wave][2 * lane] = cs0, s_cs[wave][2 * lane + 1] = cs1;
        __syncthreads();
        if (threadIdx.x < 2 * CPR) {
            double t = 0.0;
            for (int w = 0; w < ASMC_BLOCK / 64; w++) t += s_cs[w][threadIdx.x];
            cs_partials[(size_t)blockIdx.x * (2 * CPR) + threadIdx.x] = t;
        }
    }
    gather_scalars(n_out, idx, rec, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out);
}

__global__ __launch_bounds__(ASMC_BLOCK) void k_gather16(int64_t n_out, const int64_t* __restrict__ idx,
                                                        const LogRec* __restrict__ rec, int cpr,
                                                        const uint4* __restrict__ x_in,
                                                        uint4* __restrict__ x_out,
                                                        const double* __restrict__ ll_in,
                                                        const double* __restrict__ lp_in,
                                                        const double* __restrict__ lq_in,
                                                        double* __restrict__ ll_out,
                                                        double* __restrict__ lp_out,
                                                        double* __restrict__ lq_out) {
    const int64_t total = n_out * cpr;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    int64_t c = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x;
    // two chunks in flight per lane
    for (; c + stride < total; c += 2 * stride) {
        const int64_t c1 = c + stride;
        const int64_t r0 = c / cpr, r1 = c1 / cpr;
        const int k0 = (int)(c - r0 * cpr), k1 = (int)(c1 - r1 * cpr);
        const int64_t s0 = idx[r0], s1 = idx[r1];
        const uint4 v0 = x_in[s0 * cpr + k0];
        const uint4 v1 = x_in[s1 * cpr + k1];
        x_out[c] = v0;
        x_out[c1] = v1;
    }
    for (; c < total; c += stride) {
        const int64_t r0 = c / cpr;
        const int k0 = (int)(c - r0 * cpr);
        const int64_t s0 = idx[r0];
        x_out[c] = x_in[s0 * cpr + k0];
    }
    gather_scalars(n_out, idx, rec, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out);
}

// generic element-wise fallback (row bytes not a multiple of 16)
template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_gather_elem(int64_t n_out, const int64_t* __restrict__ idx,
                                                           const LogRec* __restrict__ rec, int d,
                                                           const T* __restrict__ x_in,
                                                           T* __restrict__ x_out,
                                                           const double* __restrict__ ll_in,
                                                           const double* __restrict__ lp_in,
                                                           const double* __restrict__ lq_in,
                                                           double* __restrict__ ll_out,
                                                           double* __restrict__ lp_out,
                                                           double* __restrict__ lq_out) {
    const int64_t total = n_out * d;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t c = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; c < total; c += stride) {
        const int64_t r = c / d;
        const int k = (int)(c - r * d);
        const int64_t s = idx[r];
        x_out[c] = x_in[s * d + k];
    }
    gather_scalars(n_out, idx, rec, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out);
}

// =============================================================================================
// validity compaction (finite log_prior & log_likelihood), order preserving
// =============================================================================================
__device__ __forceinline__ bool row_valid(double ll, double lp) { return isfinite(ll) && isfinite(lp); }

__global__ __launch_bounds__(ASMC_BLOCK) void k_valid_count(int64_t n, const double* __restrict__ ll,
                                                           const double* __restrict__ lp,
                                                           long long* __restrict__ tiles) {
    const int64_t base = (int64_t)blockIdx.x * ASMC_SCAN_TILE + (int64_t)threadIdx.x * SC_E;
    long long c = 0;
#pragma unroll
    for (int j = 0; j < SC_E; j++)
        if (base + j < n && row_valid(ll[base + j], lp[base + j])) c++;
    __shared__ long long s_p[ASMC_BLOCK / 64];
    c = wave_sum_ll(c);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tiles[blockIdx.x] = s_p[0] + s_p[1] + s_p[2] + s_p[3];
}

__global__ __launch_bounds__(64) void k_scan_tiles_ll(int64_t n_tiles, long long* __restrict__ tiles,
                                                     long long* __restrict__ total_out) {
    // single wave, sequential chunks of 64 (n_tiles is small)
    long long carry = 0;
    const int lane = threadIdx.x;
    for (int64_t start = 0; start < n_tiles; start += 64) {
        const int64_t i = start + lane;
        long long v = i < n_tiles ? tiles[i] : 0;
        long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            long long t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (i < n_tiles) tiles[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) *total_out = carry;
}

template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_compact_scatter(
    int64_t n, int d, const T* __restrict__ x, const double* __restrict__ ll, const double* __restrict__ lp,
    const double* __restrict__ lq, const long long* __restrict__ tiles, T* __restrict__ x_out,
    double* __restrict__ ll_out, double* __restrict__ lp_out, double* __restrict__ lq_out) {
    const int64_t base = (int64_t)blockIdx.x * ASMC_SCAN_TILE + (int64_t)threadIdx.x * SC_E;
    bool ok[SC_E];
    long long c = 0;
#pragma unroll
    for (int j = 0; j < SC_E; j++) {
        ok[j] = (base + j < n) && row_valid(ll[base + j], lp[base + j]);
        c += ok[j] ? 1 : 0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        long long t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __shared__ long long s_wave[ASMC_BLOCK / 64];
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    long long dst = tiles[blockIdx.x] + (inc - c);
    for (int k = 0; k < wave; k++) dst += s_wave[k];
#pragma unroll
    for (int j = 0; j < SC_E; j++) {
        if (ok[j]) {
            const int64_t src = base + j;
            for (int k = 0; k < d; k++) x_out[dst * d + k] = x[src * d + k];
            ll_out[dst] = ll[src];
            lp_out[dst] = lp[src];
            lq_out[dst] = lq[src];
            dst++;
        }
    }
}


// k_shard_edges + k_range_count in one launch (the sharded step's chain of launches): every block forms the slice's edges
// itself, with k_shard_edges' own division, and block 0 leaves them in edges_out[4]
__global__ __launch_bounds__(ASMC_BLOCK) void k_range_count_shard(int64_t n, const double* __restrict__ u,
                                                                 const double* __restrict__ state,
                                                                 const double* __restrict__ s_in,
                                                                 const double* __restrict__ cdf_last,
                                                                 long long* __restrict__ tiles,
                                                                 double* __restrict__ edges_out) {
    const double lo = *s_in / state[1], hi = *cdf_last;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        edges_out[0] = state[2] != 0.0 ? 0.0 : 1.0;
        edges_out[1] = state[1];
        edges_out[2] = lo;
        edges_out[3] = hi;
    }
    const int64_t base = (int64_t)blockIdx.x * ASMC_SCAN_TILE + (int64_t)threadIdx.x * (ASMC_SCAN_TILE / ASMC_BLOCK);
    long long c = 0;
#pragma unroll
    for (int j = 0; j < ASMC_SCAN_TILE / ASMC_BLOCK; j++)
        if (base + j < n) {
            const double v = u[base + j];
            c += (v >= lo && v < hi) ? 1 : 0;
        }
    __shared__ long long s_p[ASMC_BLOCK / 64];
    c = wave_sum_ll(c);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tiles[blockIdx.x] = s_p[0] + s_p[1] + s_p[2] + s_p[3];
}

// SELF_SCAN: `tiles` holds the tiles' COUNTS (k_range_count's output as it is) and a block adds up the counts in front of its
// tile itself (integers: any order); the last block leaves info_out = {kept, failure flag of the slice (edges[0])} - the
// k_scan_tiles_ll and k_range_info launches of the step-by-step form
template <bool SELF_SCAN>
__global__ __launch_bounds__(ASMC_BLOCK) void k_range_scatter(int64_t n, const double* __restrict__ u,
                                                             const double* __restrict__ lohi,
                                                             const long long* __restrict__ tiles,
                                                             double* __restrict__ out, long long* __restrict__ info_out) {
    const double lo = lohi[0], hi = lohi[1];
    __shared__ long long s_front;
    if (SELF_SCAN) {
        long long f = 0;
        for (int64_t k = threadIdx.x; k < (int64_t)blockIdx.x; k += ASMC_BLOCK) f += tiles[k];
        f = wave_sum_ll(f);
        __shared__ long long s_f[ASMC_BLOCK / 64];
        if ((threadIdx.x & 63) == 0) s_f[threadIdx.x >> 6] = f;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long tot = 0;
            for (int k = 0; k < ASMC_BLOCK / 64; k++) tot += s_f[k];
            s_front = tot;
            if (blockIdx.x == gridDim.x - 1) {
                info_out[0] = tot + tiles[blockIdx.x];
                info_out[1] = (long long)llrint(lohi[-2]);  // edges[0]
            }
        }
        __syncthreads();
    }
    const int64_t base = (int64_t)blockIdx.x * ASMC_SCAN_TILE + (int64_t)threadIdx.x * SC_E;
    double v[SC_E];
    bool ok[SC_E];
    long long c = 0;
#pragma unroll
    for (int j = 0; j < SC_E; j++) {
        v[j] = (base + j < n) ? u[base + j] : -1.0;
        ok[j] = (base + j < n) && v[j] >= lo && v[j] < hi;
        c += ok[j] ? 1 : 0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        long long t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __shared__ long long s_wave[ASMC_BLOCK / 64];
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    long long dst = (SELF_SCAN ? s_front : tiles[blockIdx.x]) + (inc - c);
    for (int k = 0; k < wave; k++) dst += s_wave[k];
#pragma unroll
    for (int j = 0; j < SC_E; j++)
        if (ok[j]) out[dst++] = v[j];
}

// =============================================================================================
// host entry points
// =============================================================================================
extern "C" {

int asmc_cdf(asmc_ctx* ctx, int64_t n, const double* w, double* cdf, int mode, double carry_in,
             double* total_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && w && cdf, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    hipStream_t st = as_stream(stream);
    double* d_total = ctx->d_small + 1024;
    const double* d_norm = (mode & ASMC_CDF_NORMALIZE) ? d_total : nullptr;
    mode &= ~ASMC_CDF_NORMALIZE;
    if (mode == ASMC_CDF_EXACT) {
        // A+B: approximate (parallel-order) tile prefixes; C: per-tile transducers; D: exact chain; E: write
        const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
        double* d_tile_s = ctx->d_tiles + ctx->n_tiles_max * 2;
        double* d_tile_s2 = ctx->d_tiles + ctx->n_tiles_max * 3;
        long long* d_split = ctx->d_tiles_i + ctx->n_tiles_max * 4;
        double* d_approx_total = ctx->d_small + 1025;
        ASMC_LAUNCH(ctx, st, "k_tile_sum", k_tile_sum, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, w, ctx->d_tiles);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_scan_tiles", k_scan_tiles, dim3(1), dim3(1024), 0, st, n_tiles, ctx->d_tiles, carry_in, d_approx_total, (const double*)nullptr);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_exact_tile_td_launch", k_exact_tile_td_launch, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n, w,
                           (const double*)ctx->d_tiles, (const double*)d_approx_total, n_tiles, ctx->d_tiles_i, d_split,
                           d_tile_s2, cdf, carry_in, d_tile_s, 1, (long long*)nullptr);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_exact_chain", k_exact_chain, dim3(1), dim3(XT_THREADS), 0, st, n, w, cdf, carry_in, n_tiles,
                           ctx->d_tiles_i, (const long long*)d_split, d_tile_s, d_tile_s2, d_total);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_exact_tile_write", k_exact_tile_write, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n, w, cdf,
                           (const long long*)ctx->d_tiles_i, (const long long*)d_split, (const double*)d_tile_s,
                           (const double*)d_tile_s2, d_norm, (const long long*)nullptr, (const double*)nullptr,
                           (unsigned int*)nullptr, (int64_t)0);
        ASMC_LAUNCH_CHECK();
    } else if (mode == ASMC_CDF_FAST) {
        const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
        ASMC_LAUNCH(ctx, st, "k_tile_sum", k_tile_sum, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, w, ctx->d_tiles);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_scan_tiles", k_scan_tiles, dim3(1), dim3(1024), 0, st, n_tiles, ctx->d_tiles, carry_in, d_total, (const double*)nullptr);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_tile_scan", k_tile_scan, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, w,
                           (const double*)ctx->d_tiles, cdf, d_norm);
        ASMC_LAUNCH_CHECK();
    } else {
        asmc_set_error("asmc_cdf: unknown mode %d", mode);
        return ASMC_ERR_ARG;
    }
    if (total_host) {
        ASMC_HIP(hipMemcpyAsync(ctx->h_pinned, d_total, sizeof(double), hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipStreamSynchronize(st));
        *total_host = ctx->h_pinned[0];
    }
    return ASMC_OK;
}

int64_t asmc_cdf_shard_tiles(int64_t n) { return n <= 0 ? 0 : (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE; }

static int cdf_shard_records_impl(asmc_ctx* ctx, int64_t n, const double* w, double* cdf, double approx_carry,
                                  const double* approx_carry_dev, int first_rank, int64_t* rec_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && w && cdf && rec_dev, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    ASMC_REQUIRE(approx_carry >= 0.0, "approx_carry must be non-negative");
    hipStream_t st = as_stream(stream);
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    double* d_approx_total = ctx->d_small + 1025;
    ASMC_LAUNCH(ctx, st, "k_tile_sum", k_tile_sum, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, w, ctx->d_tiles);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_scan_tiles", k_scan_tiles, dim3(1), dim3(1024), 0, st, n_tiles, ctx->d_tiles,
                first_rank ? 0.0 : approx_carry, d_approx_total, first_rank ? (const double*)nullptr : approx_carry_dev);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_exact_tile_td_launch", k_exact_tile_td_launch, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n, w,
                (const double*)ctx->d_tiles, (const double*)d_approx_total, n_tiles, (long long*)nullptr, (long long*)nullptr,
                (double*)nullptr, cdf, 0.0, (double*)nullptr, first_rank ? 1 : 0, (long long*)rec_dev);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_cdf_shard_records(asmc_ctx* ctx, int64_t n, const double* w, double* cdf, double approx_carry, int first_rank,
                           int64_t* rec_dev, asmc_stream stream) {
    return cdf_shard_records_impl(ctx, n, w, cdf, approx_carry, nullptr, first_rank, rec_dev, stream);
}

int asmc_cdf_shard_records_dev(asmc_ctx* ctx, int64_t n, const double* w, double* cdf, const double* approx_carry_dev,
                               const double* tile_sums_dev, int first_rank, int64_t* rec_dev, asmc_stream stream) {
    ASMC_REQUIRE(approx_carry_dev != nullptr, "null pointer");
    if (!tile_sums_dev) return cdf_shard_records_impl(ctx, n, w, cdf, 0.0, approx_carry_dev, first_rank, rec_dev, stream);
    ASMC_REQUIRE(ctx && w && cdf && rec_dev, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    hipStream_t st = as_stream(stream);
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    ASMC_LAUNCH(ctx, st, "k_exact_tile_td_shard", k_exact_tile_td_shard, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n, w,
                tile_sums_dev, approx_carry_dev, n_tiles, cdf, first_rank ? 1 : 0, (long long*)rec_dev);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_cdf_shard_chain(asmc_ctx* ctx, int64_t n, const double* w, double* cdf, const int64_t* recs_all_dev,
                         int64_t n_tiles_total, int64_t tile0, double* work_dev, const double* states_all_dev, int world,
                         int rank, double* state_out_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && w && cdf && recs_all_dev && work_dev && state_out_dev, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    ASMC_REQUIRE(tile0 >= 0 && tile0 + n_tiles <= n_tiles_total, "this shard's tiles do not fit the global tile list");
    ASMC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    hipStream_t st = as_stream(stream);
    ASMC_LAUNCH(ctx, st, "k_exact_chain_pk", k_exact_chain_pk, dim3(1), dim3(XT_THREADS), 0, st, n_tiles_total,
                (const long long*)recs_all_dev, work_dev, work_dev + n_tiles_total, work_dev + 2 * n_tiles_total, w, cdf, n,
                tile0, n_tiles, states_all_dev, world, rank, state_out_dev);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_cdf_shard_finish(asmc_ctx* ctx, int64_t n, const double* w, double* cdf, const int64_t* recs_all_dev,
                          int64_t n_tiles_total, int64_t tile0, double* work_dev, const double* state_dev, double* out_dev,
                          asmc_stream stream) {
    ASMC_REQUIRE(ctx && w && cdf && recs_all_dev && work_dev && state_dev && out_dev, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    ASMC_REQUIRE(tile0 >= 0 && tile0 + n_tiles <= n_tiles_total, "this shard's tiles do not fit the global tile list");
    hipStream_t st = as_stream(stream);
    double* tile_s = work_dev;
    double* tile_s2 = work_dev + n_tiles_total;
    const double* done_here = work_dev + 2 * n_tiles_total;
    ASMC_LAUNCH(ctx, st, "k_exact_tile_write", k_exact_tile_write, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n, w, cdf,
                (const long long*)nullptr, (const long long*)nullptr, (const double*)(tile_s + tile0),
                (const double*)(tile_s2 + tile0), state_dev + 1, (const long long*)recs_all_dev + ASMC_CDF_REC * tile0,
                done_here + tile0, (unsigned int*)nullptr, (int64_t)0);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_shard_edges", k_shard_edges, dim3(1), dim3(64), 0, st, state_dev, (const double*)(tile_s + tile0),
                (const double*)(cdf + n - 1), out_dev);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// asmc_cdf_shard_finish + asmc_select_range_dev in three launches instead of six (write pass; edges + count; scan + scatter +
// info): the form the one-chain sharded step uses.  Same results, bit for bit.
int asmc_cdf_shard_finish_select(asmc_ctx* ctx, int64_t n, const double* w, double* cdf, const int64_t* recs_all_dev,
                                 int64_t n_tiles_total, int64_t tile0, double* work_dev, const double* state_dev,
                                 int64_t n_u, const double* u_dev, double* edges_out_dev, double* out_dev, int64_t* info_dev,
                                 asmc_stream stream) {
    ASMC_REQUIRE(ctx && w && cdf && recs_all_dev && work_dev && state_dev && u_dev && edges_out_dev && out_dev && info_dev,
                 "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max && n_u > 0 && n_u <= ctx->n_max, "n out of range for this ctx");
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    ASMC_REQUIRE(tile0 >= 0 && tile0 + n_tiles <= n_tiles_total, "this shard's tiles do not fit the global tile list");
    hipStream_t st = as_stream(stream);
    double* tile_s = work_dev;
    double* tile_s2 = work_dev + n_tiles_total;
    const double* done_here = work_dev + 2 * n_tiles_total;
    ASMC_LAUNCH(ctx, st, "k_exact_tile_write", k_exact_tile_write, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n, w, cdf,
                (const long long*)nullptr, (const long long*)nullptr, (const double*)(tile_s + tile0),
                (const double*)(tile_s2 + tile0), state_dev + 1, (const long long*)recs_all_dev + ASMC_CDF_REC * tile0,
                done_here + tile0, (unsigned int*)nullptr, (int64_t)0);
    ASMC_LAUNCH_CHECK();
    const int64_t u_tiles = (n_u + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    ASMC_LAUNCH(ctx, st, "k_range_count_shard", k_range_count_shard, dim3((unsigned)u_tiles), dim3(ASMC_BLOCK), 0, st, n_u, u_dev,
                state_dev, (const double*)(tile_s + tile0), (const double*)(cdf + n - 1), ctx->d_tiles_i, edges_out_dev);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_range_scatter<scan>", k_range_scatter<true>, dim3((unsigned)u_tiles), dim3(ASMC_BLOCK), 0, st, n_u,
                u_dev, (const double*)(edges_out_dev + 2), (const long long*)ctx->d_tiles_i, out_dev,
                reinterpret_cast<long long*>(info_dev));
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_select_range(asmc_ctx* ctx, int64_t n, const double* u_dev, const double* lohi_dev, double* out_dev,
                      int64_t* count_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && u_dev && lohi_dev && out_dev && count_host, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    hipStream_t st = as_stream(stream);
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    long long* d_total = ctx->d_tiles_i + ctx->n_tiles_max * 8;
    ASMC_LAUNCH(ctx, st, "k_range_count", k_range_count, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, u_dev, lohi_dev,
                ctx->d_tiles_i);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_scan_tiles_ll", k_scan_tiles_ll, dim3(1), dim3(64), 0, st, n_tiles, ctx->d_tiles_i, d_total);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_range_scatter", k_range_scatter<false>, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, u_dev, lohi_dev,
                (const long long*)ctx->d_tiles_i, out_dev, (long long*)nullptr);
    ASMC_LAUNCH_CHECK();
    long long* h = reinterpret_cast<long long*>(ctx->h_pinned);
    ASMC_HIP(hipMemcpyAsync(h, d_total, sizeof(long long), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    *count_host = (int64_t)h[0];
    return ASMC_OK;
}

__global__ void k_range_info(const long long* __restrict__ total, const double* __restrict__ edges, long long* __restrict__ info) {
    info[0] = total[0];
    info[1] = (long long)llrint(edges[0]);
}

// asmc_select_range without the synchronisation: the count and the slice's failure flag (edges_dev[0], asmc_cdf_shard_finish)
// go to info_dev[0..1] (int64) - owner-layout resampling all-gathers them over the ranks and reads everything back at once
int asmc_select_range_dev(asmc_ctx* ctx, int64_t n, const double* u_dev, const double* edges_dev, double* out_dev,
                          int64_t* info_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && u_dev && edges_dev && out_dev && info_dev, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    hipStream_t st = as_stream(stream);
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    long long* d_total = ctx->d_tiles_i + ctx->n_tiles_max * 8;
    const double* lohi_dev = edges_dev + 2;
    ASMC_LAUNCH(ctx, st, "k_range_count", k_range_count, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, u_dev, lohi_dev,
                ctx->d_tiles_i);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_scan_tiles_ll", k_scan_tiles_ll, dim3(1), dim3(64), 0, st, n_tiles, ctx->d_tiles_i, d_total);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_range_scatter", k_range_scatter<false>, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, u_dev, lohi_dev,
                (const long long*)ctx->d_tiles_i, out_dev, (long long*)nullptr);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_range_info", k_range_info, dim3(1), dim3(1), 0, st, (const long long*)d_total, edges_dev,
                reinterpret_cast<long long*>(info_dev));
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_cdf_normalize(asmc_ctx* ctx, int64_t n, double* cdf, double last, asmc_stream stream) {
    ASMC_REQUIRE(ctx && cdf, "null pointer");
    ASMC_REQUIRE(n > 0, "n must be positive");
    const int grid = grid_for(n, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS);
    ASMC_LAUNCH(ctx, as_stream(stream), "k_divide", k_divide, dim3(grid), dim3(ASMC_BLOCK), 0, as_stream(stream), n, cdf, last);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_cdf_normalize_last(asmc_ctx* ctx, int64_t n, double* cdf, asmc_stream stream) {
    ASMC_REQUIRE(ctx && cdf, "null pointer");
    ASMC_REQUIRE(n > 0, "n must be positive");
    const int grid = grid_for(n, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS);
    ASMC_LAUNCH(ctx, as_stream(stream), "k_divide", k_divide_dev, dim3(grid), dim3(ASMC_BLOCK), 0, as_stream(stream), n, cdf,
                (const double*)(ctx->d_small + 1024));
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

static int pcg_prepare_impl(asmc_ctx* ctx, const uint64_t state_host[4], hipStream_t st);
static inline int pcg_prepare(asmc_ctx* ctx, const uint64_t state_host[4], hipStream_t st) {
    return pcg_prepare_impl(ctx, state_host, st);
}

int asmc_pcg64_uniforms(asmc_ctx* ctx, const uint64_t state_host[4], uint64_t offset, int64_t n,
                        double* u, asmc_stream stream) {
    ASMC_REQUIRE(ctx && state_host && u, "null pointer");
    ASMC_REQUIRE(n > 0, "n must be positive");
    hipStream_t st = as_stream(stream);
    int rc = pcg_prepare(ctx, state_host, st);
    if (rc) return rc;
    U128 s0 = {state_host[1], state_host[0]};
    const int64_t threads = n < PCG_THREADS ? n : PCG_THREADS;
    const int grid = (int)((threads + ASMC_BLOCK - 1) / ASMC_BLOCK);
    ASMC_LAUNCH(ctx, st, "k_pcg64_uniforms", k_pcg64_uniforms, dim3(grid), dim3(ASMC_BLOCK), 0, st,
                       (const unsigned long long*)ctx->d_pcgtab, s0, (unsigned long long)offset, n, u);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int64_t asmc_pcg64_select_stage_len(int64_t n_total) {
    if (n_total <= 0) return 0;
    return ((n_total + ASMC_SELECT_THREADS - 1) / ASMC_SELECT_THREADS) * ASMC_SELECT_THREADS;
}

int asmc_pcg64_select(asmc_ctx* ctx, const uint64_t state_host[4], int64_t n_total, double lo, double hi,
                      double* stage, int64_t* count_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && state_host && stage && count_host, "null pointer");
    ASMC_REQUIRE(n_total > 0, "n_total must be positive");
    ASMC_REQUIRE(lo >= 0.0 && hi > lo && hi <= 1.0, "need 0 <= lo < hi <= 1");
    static_assert((1 << SEL_LOG2) == ASMC_SELECT_THREADS && SEL_WAVES % 1024 == 0, "select kernel geometry");
    hipStream_t st = as_stream(stream);
    int rc = pcg_prepare(ctx, state_host, st);
    if (rc) return rc;
    U128 s0 = {state_host[1], state_host[0]};
    const int64_t iters = (n_total + ASMC_SELECT_THREADS - 1) / ASMC_SELECT_THREADS;
    ASMC_LAUNCH(ctx, st, "k_pcg64_select", k_pcg64_select, dim3(ASMC_SELECT_THREADS / ASMC_BLOCK), dim3(ASMC_BLOCK), 0, st,
                (const unsigned long long*)ctx->d_pcgtab, s0, n_total, lo, hi, iters, stage, ctx->d_select);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_select_scan", k_select_scan, dim3(1), dim3(1024), 0, st, ctx->d_select);
    ASMC_LAUNCH_CHECK();
    long long* h = reinterpret_cast<long long*>(ctx->h_pinned);
    ASMC_HIP(hipMemcpyAsync(h, ctx->d_select + 2 * SEL_WAVES, sizeof(long long), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    *count_host = (int64_t)h[0];
    return ASMC_OK;
}

int asmc_pcg64_select_compact(asmc_ctx* ctx, int64_t n_total, const double* stage, double* q, asmc_stream stream) {
    ASMC_REQUIRE(ctx && stage && q, "null pointer");
    ASMC_REQUIRE(n_total > 0, "n_total must be positive");
    hipStream_t st = as_stream(stream);
    const int64_t iters = (n_total + ASMC_SELECT_THREADS - 1) / ASMC_SELECT_THREADS;
    ASMC_LAUNCH(ctx, st, "k_select_compact", k_select_compact, dim3(SEL_WAVES), dim3(64), 0, st, iters, stage,
                (const long long*)ctx->d_select, q);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_cdf_total_dev(asmc_ctx* ctx, double* out_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && out_dev, "null pointer");
    ASMC_HIP(hipMemcpyAsync(out_dev, ctx->d_small + 1024, sizeof(double), hipMemcpyDeviceToDevice, as_stream(stream)));
    return ASMC_OK;
}

// jump table for this stream's increment: A_0 = MULT, C_0 = inc; A_{i+1} = A_i^2, C_{i+1} = (A_i+1) C_i
// (rebuilt only when the increment changes, i.e. when a different Generator is passed)
static int pcg_prepare_impl(asmc_ctx* ctx, const uint64_t state_host[4], hipStream_t st) {
    if (!ctx->pcg_tab_valid || ctx->pcg_inc[0] != state_host[2] || ctx->pcg_inc[1] != state_host[3]) {
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(ctx->h_pinned) + 4096;
        ASMC_HIP(hipStreamSynchronize(st));  // pinned staging may still be in use by a previous call
        U128 A = {0x4385DF649FCCF645ULL, 0x2360ED051FC65DA4ULL};
        U128 C = {state_host[3], state_host[2]};
        for (int i = 0; i < 64; i++) {
            tab[4 * i] = A.lo;
            tab[4 * i + 1] = A.hi;
            tab[4 * i + 2] = C.lo;
            tab[4 * i + 3] = C.hi;
            U128 A1 = u128_add(A, U128{1ULL, 0ULL});
            C = u128_mul(A1, C);
            A = u128_mul(A, A);
        }
        ASMC_HIP(hipMemcpyAsync(ctx->d_pcgtab, tab, sizeof(unsigned long long) * 256, hipMemcpyHostToDevice, st));
        ASMC_HIP(hipStreamSynchronize(st));
        ctx->pcg_inc[0] = state_host[2];
        ctx->pcg_inc[1] = state_host[3];
        ctx->pcg_tab_valid = 1;
    }
    return ASMC_OK;
}

int asmc_systematic_uniforms(asmc_ctx* ctx, int64_t n_out, int64_t j0, int64_t n_total, double u0,
                             const double* v, double* u, asmc_stream stream) {
    ASMC_REQUIRE(ctx && u, "null pointer");
    ASMC_REQUIRE(n_out > 0 && n_total > 0, "bad sizes");
    const int grid = grid_for(n_out, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS);
    ASMC_LAUNCH(ctx, as_stream(stream), "k_systematic", k_systematic, dim3(grid), dim3(ASMC_BLOCK), 0, as_stream(stream), n_out, j0, n_total,
                       u0, v, u);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_search(asmc_ctx* ctx, int64_t n, const double* cdf, int64_t n_out, const double* u,
                int64_t* idx, asmc_stream stream) {
    ASMC_REQUIRE(ctx && cdf && u && idx, "null pointer");
    ASMC_REQUIRE(n > 0 && n_out > 0, "bad sizes");
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(n_out, ASMC_BLOCK, ASMC_MAX_BLOCKS * 4);
    // guide table: one bucket per four cdf entries; worth building when the cdf no longer sits in L2 and there are
    // enough lookups to pay for the nb + 1 (cache-friendly) searches of the build
    static const bool no_guide = getenv("ASMC_SEARCH_PLAIN") != nullptr;
    const int64_t nb = n / 4;
    if (!no_guide && n >= (1 << 17) && n <= ctx->n_max && n < (1LL << 32) && n_out >= n / 8) {
        ASMC_LAUNCH(ctx, st, "k_guide_build", k_guide_build, dim3(grid_for(nb + 1, ASMC_BLOCK, ASMC_MAX_BLOCKS * 4)), dim3(ASMC_BLOCK), 0, st,
                    n, cdf, nb, ctx->d_guide);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_search", k_search_guided, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, cdf, nb,
                    (const unsigned int*)ctx->d_guide, n_out, u, idx);
    } else {
        ASMC_LAUNCH(ctx, st, "k_search", k_search, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, cdf, n_out, u, idx);
    }
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// The sharded step's one synchronisation AND what follows it, without the host interpreter in between: waits for the chain
// (asmc_shard_step_result), and when the numbers read back are those of a finished step - search converged, no NaN weights,
// no slice that needs the replicated scan, every rank's offspring count in (0, cap]: conditions every rank evaluates on the same
// gathered values - puts this rank's search (asmc_search over its kept draws) and gather (asmc_gather into the caller's
// cap-row buffers; the records packed on the way are claimed by rec_token) onto the stream right away.  *launched = 1 then, and
// out_host[13 + 2 world + 2 rank] is the number of rows written.  The caller still takes resample_owner's decisions (shares,
// totals) from out_host and drops the rows if they say so.
int asmc_shard_step_finish(asmc_ctx* ctx, const double* res_dev, int world, int rank, int64_t n_local, const double* cdf_dev,
                           const double* kept_dev, int64_t cap, int64_t* idx_dev, int d, int x_dtype, const void* x_in,
                           void* x_out, const double* ll_in, const double* lp_in, const double* lq_in, double* ll_out,
                           double* lp_out, double* lq_out, int64_t rec_token, double* out_host, int* launched,
                           asmc_stream stream) {
    ASMC_REQUIRE(ctx && res_dev && cdf_dev && kept_dev && idx_dev && x_in && x_out && ll_in && lp_in && lq_in && ll_out && lp_out &&
                     lq_out && out_host && launched,
                 "null pointer");
    ASMC_REQUIRE(world >= 1 && rank >= 0 && rank < world && n_local > 0 && cap > 0 && d > 0, "bad sizes");
    *launched = 0;
    int rc = asmc_shard_step_result(ctx, res_dev, world, out_host, stream);
    if (rc) return rc;
    const double* info = out_host + 13 + 2 * world;  // (kept, fail) per rank, as doubles
    bool usable = out_host[2] != 0.0 && out_host[5] == 0.0 && out_host[9] != 0.0;
    for (int r = 0; r < world && usable; r++) usable = info[2 * r + 1] == 0.0 && info[2 * r] > 0.0 && info[2 * r] <= (double)cap;
    if (!usable) return ASMC_OK;
    const int64_t cnt = (int64_t)info[2 * rank];
    rc = asmc_search(ctx, n_local, cdf_dev, cnt, kept_dev, idx_dev, stream);
    if (rc) return rc;
    (void)asmc_rec_claim(ctx, rec_token, n_local, ll_in, lp_in, lq_in);
    rc = asmc_gather(ctx, n_local, cnt, idx_dev, d, x_dtype, x_in, x_out, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out, stream);
    if (rc) return rc;
    *launched = 1;
    return ASMC_OK;
}

int asmc_importance_step(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq, double beta0,
                         double target_eff, double tol, const uint64_t rng_state[4], int64_t n_out, double* w_scratch,
                         double* cdf_scratch, int64_t* idx_out, asmc_stream stream) {
    ASMC_REQUIRE(ctx && ll && lp && lq && rng_state && w_scratch && cdf_scratch && idx_out, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max && n < (1LL << 32) && n_out > 0, "bad sizes");
    ASMC_REQUIRE(tol > 0.0 && beta0 >= 0.0 && beta0 < 1.0, "bad beta0 / tolerance");
    hipStream_t st = as_stream(stream);
    // jump table of the generator's increment (host work + one upload, only when the increment changes)
    int rc = pcg_prepare(ctx, rng_state, st);
    if (rc) return rc;
    // 1: beta search, evidence moments, normalised weights, tile sums (one persistent launch)
    // (the kernel also leaves the (ll, lp, lq, 0) records of asmc_gather in ctx->d_rec: tagged at the end of this call)
    static const bool no_rec = getenv("ASMC_IS_NO_RECORDS") != nullptr;
    rc = asmc_is_weights_launch(ctx, n, ll, lp, lq, beta0, target_eff, tol, w_scratch, ctx->d_tiles, no_rec ? nullptr : ctx->d_rec, st);
    if (rc) return rc;
    if (!no_rec) ctx->rec_token++;  // (d_rec rewritten)
    // 2-4: numpy's sequential cumsum / cdf[-1] (passes C, D, E of asmc_cdf; the tile prefixes ride on pass C while
    // there are few tiles), pass E also filling the search's guide table
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    double* d_total = ctx->d_small + 1024;
    double* d_tile_s = ctx->d_tiles + ctx->n_tiles_max * 2;
    double* d_tile_s2 = ctx->d_tiles + ctx->n_tiles_max * 3;
    long long* d_split = ctx->d_tiles_i + ctx->n_tiles_max * 4;
    if (n_tiles <= 1024) {
        ASMC_LAUNCH(ctx, st, "k_exact_tile_td_scan", k_exact_tile_td_scan, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n,
                    (const double*)w_scratch, (const double*)ctx->d_tiles, n_tiles, ctx->d_tiles_i, d_split, d_tile_s2,
                    cdf_scratch, d_tile_s);
    } else {
        ASMC_LAUNCH(ctx, st, "k_scan_tiles", k_scan_tiles, dim3(1), dim3(1024), 0, st, n_tiles, ctx->d_tiles, 0.0, ctx->d_small + 1025, (const double*)nullptr);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_exact_tile_td_launch", k_exact_tile_td_launch, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n,
                    (const double*)w_scratch, (const double*)ctx->d_tiles, (const double*)(ctx->d_small + 1025), n_tiles,
                    ctx->d_tiles_i, d_split, d_tile_s2, cdf_scratch, 0.0, d_tile_s, 1, (long long*)nullptr);
    }
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_exact_chain", k_exact_chain, dim3(1), dim3(XT_THREADS), 0, st, n, (const double*)w_scratch, cdf_scratch,
                0.0, n_tiles, ctx->d_tiles_i, (const long long*)d_split, d_tile_s, d_tile_s2, d_total);
    ASMC_LAUNCH_CHECK();
    static const bool no_guide = getenv("ASMC_SEARCH_PLAIN") != nullptr;
    // one bucket per cdf entry (round 5; n / 4 before): a lookup is then the guide's pair and 1 - 2 dependent cdf reads instead
    // of ~3 - the search is bound by its random requests (fetching whole 64-byte windows instead of bisecting doubled its time) -
    // 34.6 -> 26.2 us at 1M, 430 -> 281 us at 8M, for + 1.3 / + 7 us of table fill in pass E.  2 n buckets gain nothing more at 1M.
    const int64_t nb = n;
    const bool guided = !no_guide && n >= (1 << 17) && n_out >= n / 8;
    ASMC_LAUNCH(ctx, st, "k_exact_tile_write", k_exact_tile_write, dim3((unsigned)n_tiles), dim3(XT_THREADS), 0, st, n,
                (const double*)w_scratch, cdf_scratch, (const long long*)ctx->d_tiles_i, (const long long*)d_split,
                (const double*)d_tile_s, (const double*)d_tile_s2, (const double*)d_total, (const long long*)nullptr,
                (const double*)nullptr, guided ? ctx->d_guide : (unsigned int*)nullptr, nb);
    ASMC_LAUNCH_CHECK();
    // 5: rng.random(n_out) and searchsorted(cdf, u, side="right") in one pass
    int tlog2 = SP_LOG2;
    while (tlog2 > 6 && ((int64_t)1 << (tlog2 - 1)) >= n_out) tlog2--;
    const int64_t threads = (int64_t)1 << tlog2;
    U128 s0 = {rng_state[1], rng_state[0]};
    ASMC_LAUNCH(ctx, st, "k_search_pcg", k_search_pcg, dim3((unsigned)((threads + ASMC_BLOCK - 1) / ASMC_BLOCK)), dim3(ASMC_BLOCK), 0, st,
                (const unsigned long long*)ctx->d_pcgtab, s0, tlog2, n, (const double*)cdf_scratch, nb,
                guided ? (const unsigned int*)ctx->d_guide : (const unsigned int*)nullptr, n_out, idx_out);
    ASMC_LAUNCH_CHECK();
    if (!no_rec) ctx->rec_src[0] = ll, ctx->rec_src[1] = lp, ctx->rec_src[2] = lq, ctx->rec_n = n;  // (any later launch clears it)
    return ASMC_OK;
}

// The (ll, lp, lq, 0) records that asmc_normalized_weights_shard packed on its way (pack_records = 1) become the next asmc_gather's
// records if `token` (asmc_rec_token right after that call) is still d_rec's generation, i.e. no other pass has rewritten d_rec
// since, and the arrays are the ones packed.  Returns 1 when claimed, 0 otherwise (the gather then packs for itself).  Call it
// directly in front of asmc_gather: any launch in between drops the claim again.
int64_t asmc_rec_token(asmc_ctx* ctx) { return ctx ? (int64_t)ctx->rec_token : -1; }
int asmc_rec_claim(asmc_ctx* ctx, int64_t token, int64_t n, const double* ll, const double* lp, const double* lq) {
    if (!ctx || token <= 0 || (uint64_t)token != ctx->rec_token || ctx->rec_hold_token != (uint64_t)token || ctx->rec_hold_n != n ||
        ctx->rec_hold_src[0] != ll || ctx->rec_hold_src[1] != lp || ctx->rec_hold_src[2] != lq)
        return 0;
    ctx->rec_src[0] = ll, ctx->rec_src[1] = lp, ctx->rec_src[2] = lq, ctx->rec_n = n;
    return 1;
}

int asmc_gather(asmc_ctx* ctx, int64_t n_in, int64_t n_out, const int64_t* idx, int d, int x_dtype, const void* x_in,
                void* x_out, const double* ll_in, const double* lp_in, const double* lq_in,
                double* ll_out, double* lp_out, double* lq_out, asmc_stream stream) {
    ASMC_REQUIRE(ctx && idx && x_in && x_out && ll_in && lp_in && lq_in && ll_out && lp_out && lq_out,
                 "null pointer");
    ASMC_REQUIRE(n_in > 0 && n_out > 0 && d > 0, "bad sizes");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    hipStream_t st = as_stream(stream);
    const LogRec* rec = nullptr;
    if (ctx->rec_n == n_in && ctx->rec_src[0] == ll_in && ctx->rec_src[1] == lp_in && ctx->rec_src[2] == lq_in) {
        rec = reinterpret_cast<const LogRec*>(ctx->d_rec);  // packed by the importance step that selected these draws
    } else if (n_in <= ctx->n_max && n_out >= n_in / 4 && n_in >= (1 << 16)) {  // enough draws to pay for the packing pass
        LogRec* r = reinterpret_cast<LogRec*>(ctx->d_rec);
        ASMC_LAUNCH(ctx, st, "k_pack_records", k_pack_records, dim3(grid_for(n_in, ASMC_BLOCK * 2, ASMC_MAX_BLOCKS * 2)), dim3(ASMC_BLOCK), 0, st,
                    n_in, ll_in, lp_in, lq_in, r);
        ASMC_LAUNCH_CHECK();
        ctx->rec_token++;  // (d_rec rewritten)
        rec = r;
    }
    const size_t elem = x_dtype == ASMC_F64 ? 8 : 4;
    const size_t rowbytes = elem * (size_t)d;
    const bool vec_ok = (rowbytes % 16 == 0) && (((uintptr_t)x_in | (uintptr_t)x_out) % 16 == 0);
    if (vec_ok) {
        const int cpr = (int)(rowbytes / 16);
        const int grid = grid_for(n_out * cpr, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
        static const bool plain_gather = getenv("ASMC_GATHER_PLAIN") != nullptr;
        bool cs_done = false;
        // fp64 rows of a width the moment kernels take: the column sums of the gathered rows ride along (tagged below)
        const bool cs = x_dtype == ASMC_F64 && (d == 32 || d == 64 || d == 16) && (size_t)grid * d <= ctx->gram_cap &&
                        !getenv("ASMC_GATHER_NO_COLSUM");
#define ASMC_GATHER_POW2(SHV)                                                                                       \
    if (!plain_gather && cpr == (1 << SHV)) {                                                                      \
        if (cs)                                                                                                    \
            ASMC_LAUNCH(ctx, st, "k_gather16", (k_gather16_pow2<SHV, true>), dim3(grid), dim3(ASMC_BLOCK), 0, st, n_out, idx, \
                        rec, (const uint4*)x_in, (uint4*)x_out, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out, ctx->d_gram);  \
        else                                                                                                       \
            ASMC_LAUNCH(ctx, st, "k_gather16", (k_gather16_pow2<SHV, false>), dim3(grid), dim3(ASMC_BLOCK), 0, st, n_out, idx, \
                        rec, (const uint4*)x_in, (uint4*)x_out, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out, (double*)nullptr); \
        cs_done = cs;                                                                                              \
    } else
        ASMC_GATHER_POW2(4)
        ASMC_GATHER_POW2(3)
        ASMC_GATHER_POW2(5)
        {
    ASMC_LAUNCH(ctx, st, "k_gather16", k_gather16, dim3(grid), dim3(ASMC_BLOCK), 0, st, n_out, idx, rec, cpr,
                           (const uint4*)x_in, (uint4*)x_out, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out);
        }
#undef ASMC_GATHER_POW2
        // (any later launch drops the tag: the partials live in the moment kernels' scratch)
        if (cs_done) ctx->cs_x = x_out, ctx->cs_n = n_out, ctx->cs_d = d, ctx->cs_grid = grid;
    } else if (x_dtype == ASMC_F64) {
        const int grid = grid_for(n_out * d, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
        ASMC_LAUNCH(ctx, st, "k_gather_elem<double>", k_gather_elem<double>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n_out, idx, rec, d,
                           (const double*)x_in, (double*)x_out, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out);
    } else {
        const int grid = grid_for(n_out * d, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
        ASMC_LAUNCH(ctx, st, "k_gather_elem<float>", k_gather_elem<float>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n_out, idx, rec, d,
                           (const float*)x_in, (float*)x_out, ll_in, lp_in, lq_in, ll_out, lp_out, lq_out);
    }
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_compact_valid(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const double* ll,
                       const double* lp, const double* lq, void* x_out, double* ll_out, double* lp_out,
                       double* lq_out, int64_t* n_valid_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && ll && lp && lq && x_out && ll_out && lp_out && lq_out && n_valid_host,
                 "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max && d > 0, "bad sizes");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    hipStream_t st = as_stream(stream);
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    long long* d_total = ctx->d_tiles_i + ctx->n_tiles_max * 8;
    ASMC_LAUNCH(ctx, st, "k_valid_count", k_valid_count, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, ll, lp, ctx->d_tiles_i);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_scan_tiles_ll", k_scan_tiles_ll, dim3(1), dim3(64), 0, st, n_tiles, ctx->d_tiles_i, d_total);
    ASMC_LAUNCH_CHECK();
    long long* h = reinterpret_cast<long long*>(ctx->h_pinned);
    ASMC_HIP(hipMemcpyAsync(h, d_total, sizeof(long long), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    *n_valid_host = (int64_t)h[0];
    if (h[0] == (long long)n) return ASMC_OK;  // already compact (the usual case): nothing is copied, the caller keeps its inputs
    if (x_dtype == ASMC_F64)
        ASMC_LAUNCH(ctx, st, "k_compact_scatter<double>", k_compact_scatter<double>, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, d,
                           (const double*)x, ll, lp, lq, (const long long*)ctx->d_tiles_i, (double*)x_out,
                           ll_out, lp_out, lq_out);
    else
        ASMC_LAUNCH(ctx, st, "k_compact_scatter<float>", k_compact_scatter<float>, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, d,
                           (const float*)x, ll, lp, lq, (const long long*)ctx->d_tiles_i, (float*)x_out,
                           ll_out, lp_out, lq_out);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

}  // extern "C"
