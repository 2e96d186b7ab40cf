// Decision half of the adaptive-temperature search (reference: src/aspire/samplers/smc/base.py:167-186) - what closes a
// round of 16 candidate temperatures and chooses the next 16.  Plain C++ (host and device): the HIP kernels call it from
// one lane behind their reductions (asmc_weights.hip: k_bis_sums' last block, k_bis_decide, k_is_weights), and
// tests/tools/bisect_plan_check.cpp drives it on the CPU against the sequential loop.
//
// The reference bisects [beta0, 1] until the bracket is no wider than the tolerance and returns its lower end: with a
// non-increasing ESS(beta) (d log ESS / d beta = 2 (E_t[Delta] - E_2t[Delta]) <= 0) that is the largest node K / 2^L of the
// dyadic grid whose ESS/N still reaches the target, L = the first level whose cells are within the tolerance.  Every
// dyadic node has ONE float value whatever path reaches it (the midpoint of its two neighbours one level up, in the
// reference's expression 0.5 * (hi + lo)): bis_node_beta.  So the search may visit the grid in any order, as long as it
// ends on the adjacent pair (K, K + 1) with ESS(K) >= target > ESS(K + 1) and reports node K's float:
//   * round 0 is the four-level tree of [beta0, 1] plus beta = 1 (smc/base.py:170-175), sixteen equally spaced nodes;
//   * later rounds evaluate sixteen equally spaced nodes K_first + j S (the kernels need equal spacing: one geometric
//     progression per particle) - either the plain k-ary step (S = bracket / 16: four levels per round, what rounds 1-2 of
//     this repository did throughout), or a WINDOW around the root predicted by inverse interpolation of log ESS through
//     the nodes next to the bracket, with seven nodes of margin >= four times the difference between the last two
//     interpolation orders;
//   * a window that misses the root narrows the bracket from one side only; the search then stays with plain steps, so it
//     never needs more than one round above the plain search; smooth populations finish in 3 rounds instead of 5
//     (tolerance 1e-6) or 7 (1e-8);
//   * the stopping level follows the reference's own test on the floats of the final bracket AND of its parent (the loop
//     would have stopped there): a tolerance within rounding of a cell width moves the level by one, which is followed.
// Node indices are kept in units of the finest level LU = max(L, 4) as integers in doubles (exact below 2^53).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define BIS_HD __host__ __device__ inline
#else
#define BIS_HD inline
#endif

// state record (doubles); [0..15] and [32..39] as in rounds 1-2, [16..31] are this file's
enum {
    BIS_BMIN = 0, BIS_BMAX = 1, BIS_DONE = 2, BIS_TARGET = 3, BIS_TOL = 4, BIS_LOGN = 5, BIS_ROUNDS = 6, BIS_BETA0 = 7,
    BIS_N = 8, BIS_EFF_ONE = 9, BIS_M_ONE = 10, BIS_TRIP_M = 11, BIS_TRIP_S1 = 12, BIS_TRIP_S2 = 13, BIS_TRIP_OK = 14,
    BIS_NAN = 15,
    BIS_LU = 16,      // level of the index unit
    BIS_LFIN = 17,    // level at which the reference's loop stops
    BIS_KLO = 18, BIS_KHI = 19,        // bracket: ESS(K_lo) >= target > ESS(K_hi)
    BIS_KFIRST = 20, BIS_STRIDE = 21,  // the grid the NEXT round evaluates: nodes K_first + j stride, j = 0 .. 15
    BIS_YLO = 22, BIS_YHI = 23,        // log(ESS/N) - log(target) at the bracket's ends
    BIS_MODE = 24,    // 1: plain steps only (a window has missed)
    BIS_WINDOW = 25,  // 1: the grid of [20], [21] is a prediction window
    BIS_WINDOWS = 26, BIS_MISSES = 27,  // statistics: windows evaluated, windows that missed
    BIS_S1_ONE = 32, BIS_S2_ONE = 33,
    BIS_C1 = 34, BIS_C2 = 35, BIS_M1 = 36, BIS_H = 37, BIS_DMAX = 38,  // the next grid for the sums kernels
    BIS_NEWPACK = 39
};

// sorted candidate j (ascending beta) -> column pair of the partial records (the heap order of the first round's tree)
BIS_HD int bis_col_of_sorted(int j) { return (int)((0xFE6D2C5B0A491837ULL >> (4 * j)) & 15ULL); }

// The float the reference's loop holds for the dyadic node K / 2^LU of [beta0, 1]
BIS_HD double bis_node_beta(long long K, int LU, double beta0) {
    if (K <= 0) return beta0;
    if (K >= (1LL << LU)) return 1.0;
    const int tz = __builtin_ctzll((unsigned long long)K);
    const int lev = LU - tz;
    const long long k = K >> tz;  // odd, at level lev
    double lo = beta0, hi = 1.0, mid = 0.5 * (hi + lo);  // the reference's expression (smc/base.py:178)
    for (int b = lev - 1; b >= 1; b--) {
        if ((k >> b) & 1)
            lo = mid;
        else
            hi = mid;
        mid = 0.5 * (hi + lo);
    }
    return mid;
}

BIS_HD long long bis_pow2_ceil(double v) {
    long long p = 1;
    while ((double)p < v && p < (1LL << 58)) p <<= 1;
    return p;
}

// Root of y = log(ESS/N) - log(target) in index units by inverse (Neville) interpolation through the bracket's ends and the
// evaluated nodes next to them (jl / jr: the grid node below K_lo / above K_hi, -1 when there is none);
// *E = |last order - the one before|.  false: too few usable points / not monotone.
BIS_HD bool bis_predict(long long K_lo, double y_lo, long long K_hi, double y_hi, long long K_first, long long stride, int jl,
                        int jr, const double* y, double* Kr, double* E) {
    double xs[4], ys[4];
    if (!(y_lo >= 0.0) || !(y_hi < 0.0) || !(y_lo < 1e300) || !(y_hi > -1e300)) return false;
    xs[0] = (double)K_lo, ys[0] = y_lo, xs[1] = (double)K_hi, ys[1] = y_hi;
    int np = 2;
    if (jl >= 0) {
        const double v = y[jl];
        if (v > y_lo && v < 1e300) xs[np] = (double)(K_first + jl * stride), ys[np] = v, np++;
    }
    if (jr >= 0) {
        const double v = y[jr];
        if (v < y_hi && v > -1e300) xs[np] = (double)(K_first + jr * stride), ys[np] = v, np++;
    }
    if (np < 3) return false;
    const double P01 = (-ys[1] * xs[0] + ys[0] * xs[1]) / (ys[0] - ys[1]);
    const double P12 = (-ys[2] * xs[1] + ys[1] * xs[2]) / (ys[1] - ys[2]);
    const double P012 = (-ys[2] * P01 + ys[0] * P12) / (ys[0] - ys[2]);
    double best = P012, prev = P01;
    if (np == 4) {
        const double P23 = (-ys[3] * xs[2] + ys[2] * xs[3]) / (ys[2] - ys[3]);
        const double P123 = (-ys[3] * P12 + ys[1] * P23) / (ys[1] - ys[3]);
        best = (-ys[3] * P012 + ys[0] * P123) / (ys[0] - ys[3]);
        prev = P012;
    }
    if (!(best >= (double)K_lo) || !(best <= (double)K_hi)) return false;
    *Kr = best;
    *E = fabs(best - prev);
    return true;
}

// Closes a round.  In: the record `st` (read once, written once: a caller on a GPU hands in a register copy), and for the
// sixteen candidates in ascending order their floats `beta`, log-sum-exp shifts `m`, ESS/N `eff`,
// y = log(ESS/N) - log(target) and the 32 column sums `S` (S1, S2 of sorted candidate j in columns 2 c, 2 c + 1 with
// c = bis_col_of_sorted(j)).  `first`: the candidates were the four-level tree of [beta0, 1] and beta = 1.
// Out: st[BIS_DONE], the bracket, (m, S1, S2) at its lower end, the next grid ([20], [21], [34..38]) when not done.
BIS_HD void bis_plan(double* st, bool first, const double* beta, const double* m, const double* eff, const double* y,
                     const double* S) {
    const double target = st[BIS_TARGET], tol = st[BIS_TOL], beta0 = st[BIS_BETA0], m_one = st[BIS_M_ONE];
    st[BIS_ROUNDS] += 1.0;
    st[BIS_NEWPACK] = 0.0;
    int LU, Lf, mode, window;
    long long K_lo, K_hi, K_first, stride;
    double y_lo, y_hi, bmin, bmax;
    bool trip_ok;
    if (first) {
        // depth of the reference's loop: the first level whose cells are within the tolerance (an estimate from the ratio;
        // the loop's own test on the floats of the final bracket and of its parent settles it below)
        Lf = 0;
        if (1.0 - beta0 > tol) {
            int e;
            const double mant = frexp((1.0 - beta0) / tol, &e);
            Lf = mant == 0.5 ? e - 1 : e;
            if (Lf < 1) Lf = 1;
            if (Lf > 50) Lf = 50;
        }
        LU = Lf < 4 ? 4 : Lf;
        K_lo = 0, K_hi = 1LL << LU;
        K_first = stride = 1LL << (LU - 4);
        y_lo = -log(target), y_hi = y[15], bmin = beta0, bmax = 1.0, window = 0, trip_ok = false;
        mode = st[BIS_MODE] != 0.0;  // a caller may start in plain mode (ablation: ASMC_BISECT_PLAIN)
        st[BIS_WINDOWS] = st[BIS_MISSES] = 0.0;
        st[BIS_EFF_ONE] = eff[15];
        st[BIS_S1_ONE] = S[30], st[BIS_S2_ONE] = S[31];
        st[BIS_BMIN] = beta0, st[BIS_BMAX] = 1.0;
        if (eff[15] >= target) {  // smc/base.py:174-175: beta* = 1
            st[BIS_BMIN] = 1.0;
            st[BIS_TRIP_M] = m_one, st[BIS_TRIP_S1] = S[30], st[BIS_TRIP_S2] = S[31], st[BIS_TRIP_OK] = 1.0;
            st[BIS_DONE] = 1.0;
            return;
        }
        if (Lf == 0) {  // 1 - beta0 <= tolerance: the loop body never runs
            st[BIS_DONE] = 1.0;
            return;
        }
    } else {
        LU = (int)st[BIS_LU], Lf = (int)st[BIS_LFIN], mode = (int)st[BIS_MODE], window = (int)st[BIS_WINDOW];
        K_lo = (long long)st[BIS_KLO], K_hi = (long long)st[BIS_KHI];
        K_first = (long long)st[BIS_KFIRST], stride = (long long)st[BIS_STRIDE];
        y_lo = st[BIS_YLO], y_hi = st[BIS_YHI], bmin = st[BIS_BMIN], bmax = st[BIS_BMAX];
        trip_ok = st[BIS_TRIP_OK] != 0.0;
    }
    long long F = 1LL << (LU - Lf);  // cell of the stopping level in index units (> 1 only when that level is above 4)
    // ---- the evaluated nodes narrow the bracket: the first failing node inside it closes it, the last passing node before
    // that one opens it (one pass over the sixteen ESS values, no dependent loads)
    int jp = -1, jf = -1, jsame = -1;
    double ev[16];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 16; j++) ev[j] = eff[j];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 15; j >= 0; j--) {
        const long long K = K_first + j * stride;
        const bool aligned = !(K & (F - 1));  // (finer nodes than the stopping level: first round of a wide tolerance)
        const bool inside = K > K_lo && K < K_hi;
        if (aligned && inside && !(ev[j] >= target)) jf = j;
        if (aligned && K == K_lo) jsame = j;
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 16; j++) {
        const long long K = K_first + j * stride;
        const bool aligned = !(K & (F - 1));
        const bool inside = K > K_lo && K < K_hi;
        if (aligned && inside && (jf < 0 || j < jf)) jp = j;  // passes: every failing node inside is at or behind jf
    }
    if (jsame >= 0 && K_lo > 0 && !trip_ok) {  // the lower end re-evaluated for its sums (see the parent test below)
        const int c = bis_col_of_sorted(jsame);
        st[BIS_TRIP_M] = m[jsame], st[BIS_TRIP_S1] = S[2 * c], st[BIS_TRIP_S2] = S[2 * c + 1], trip_ok = true;
        y_lo = y[jsame];
    }
    if (jp >= 0) {
        const int c = bis_col_of_sorted(jp);
        K_lo = K_first + jp * stride, y_lo = y[jp], bmin = beta[jp];
        st[BIS_TRIP_M] = m[jp], st[BIS_TRIP_S1] = S[2 * c], st[BIS_TRIP_S2] = S[2 * c + 1], trip_ok = true;
    }
    if (jf >= 0) K_hi = K_first + jf * stride, y_hi = y[jf], bmax = beta[jf];
    if (window) {
        st[BIS_WINDOWS] += 1.0;
        if (K_hi - K_lo > stride) mode = 1, st[BIS_MISSES] += 1.0;  // the root was not inside: plain steps from here on
    }
    // ---- does the reference's loop stop here?  (its test, on the floats of this bracket and of its parent) ----
    bool done = false;
    for (int guard = 0; guard < 64 && K_hi - K_lo == F; guard++) {
        const double width = bmax - bmin;
        if (width > tol) {  // one level deeper than estimated
            if (LU >= 52) break;  // (node indices are kept in doubles: exact below 2^53; such a tolerance is below the floats' spacing)
            if (F > 1)
                F >>= 1;
            else
                LU++, K_lo <<= 1, K_hi <<= 1, K_first <<= 1, stride <<= 1;
            Lf++;
            break;
        }
        bool parent_stops = false;
        if (Lf >= 1 && !(2.0 * width * (1.0 - 1e-9) > tol)) {  // (a parent is twice as wide up to rounding)
            const long long P_lo = K_lo & ~(2 * F - 1), P_hi = P_lo + 2 * F;
            const double pl = bis_node_beta(P_lo, LU, beta0), ph = bis_node_beta(P_hi, LU, beta0);
            if (!(ph - pl > tol)) {  // the loop never entered this bracket: it stopped on the parent (or above it)
                parent_stops = true;
                Lf--, F <<= 1;
                K_hi = P_hi, bmax = ph;
                if (P_lo != K_lo) K_lo = P_lo, bmin = pl, y_lo = -log(target), trip_ok = false;
            }
        }
        if (!parent_stops) {
            done = trip_ok || K_lo == 0;
            break;
        }
    }
    st[BIS_BMIN] = bmin, st[BIS_BMAX] = bmax;
    st[BIS_LU] = (double)LU, st[BIS_LFIN] = (double)Lf;
    st[BIS_KLO] = (double)K_lo, st[BIS_KHI] = (double)K_hi, st[BIS_YLO] = y_lo, st[BIS_YHI] = y_hi;
    st[BIS_MODE] = (double)mode;
    st[BIS_TRIP_OK] = trip_ok ? 1.0 : 0.0;
    if (done) {
        st[BIS_DONE] = 1.0;
        return;
    }
    // ---- the next sixteen nodes ----
    const long long n = K_hi - K_lo;
    long long S_new = F, Kf_new;
    int window_new = 0;
    if (n == F) {  // lower end without sums (moved up to a parent): its own node first
        Kf_new = K_lo;
    } else {
        while (16 * S_new < n) S_new <<= 1;  // plain step: covers the bracket
        Kf_new = (K_lo / S_new + 1) * S_new;
        double Kr = 0.0, E = 0.0;
        // grid nodes next to the bracket on either side (the bracket's ends are grid nodes or lie outside the grid)
        int jl = -1, jr = -1;
        if (K_lo - stride >= K_first && K_lo - stride <= K_first + 15 * stride && !((K_lo - K_first) % stride))
            jl = (int)((K_lo - K_first) / stride) - 1;
        else if (K_lo > K_first + 15 * stride)
            jl = 15;
        if (K_hi + stride <= K_first + 15 * stride && K_hi + stride >= K_first && !((K_hi - K_first) % stride))
            jr = (int)((K_hi - K_first) / stride) + 1;
        else if (K_hi < K_first)
            jr = 0;
        if (mode == 0 && S_new > F && bis_predict(K_lo, y_lo, K_hi, y_hi, K_first, stride, jl, jr, y, &Kr, &E)) {
            long long S_w = bis_pow2_ceil(4.0 * E / 7.0);
            if (S_w < F) S_w = F;
            if (S_w < S_new) {  // a window pays: seven nodes of margin on either side of the predicted cell
                window_new = 1, S_new = S_w;
                const long long min_first = (K_lo / S_w + 1) * S_w, max_last = ((K_hi - 1) / S_w) * S_w;
                Kf_new = ((long long)floor(Kr / (double)S_w) - 7) * S_w;
                if (Kf_new + 15 * S_w > max_last) Kf_new = max_last - 15 * S_w;
                if (Kf_new < min_first) Kf_new = min_first;
            }
        }
    }
    st[BIS_KFIRST] = (double)Kf_new, st[BIS_STRIDE] = (double)S_new, st[BIS_WINDOW] = (double)window_new;
    const double inv = 1.0 / (1.0 - beta0), b_first = bis_node_beta(Kf_new, LU, beta0);
    st[BIS_C1] = beta0 - b_first;
    st[BIS_C2] = b_first - beta0;
    st[BIS_M1] = m_one * ((b_first - beta0) * inv);
    st[BIS_H] = (1.0 - beta0) * ldexp((double)S_new, -LU);
    st[BIS_DMAX] = m_one * inv;
    st[BIS_NEWPACK] = 1.0;
}
