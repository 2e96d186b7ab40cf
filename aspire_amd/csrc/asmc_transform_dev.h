// asmc_transform_dev.h — device pieces of the preconditioning transforms (asmc_transform.hip) shared with the pCN propose
// kernel that applies the inverse transform to its register-resident proposal (asmc_pcn.hip, PCN_FLOW_PROPOSE_SXT_*).
#pragma once
#include "asmc_common.h"
#include "asmc_pcn_dev.h"  // bm_log_unit

struct TransDev {
    int d;
    const int* kind;      // 0 none, 1 logit, 2 probit
    const int* periodic;  // 1 = wrap into [lower, upper)
    const double* lower;
    const double* upper;
    const double* mean;  // nullptr = no affine stage
    const double* std;
    double eps;
    double unit_logj;    // -sum_{bounded} log(upper - lower)      (forward sign)
    double affine_logj;  // -sum log|std|                          (forward sign)
    int hints;           // ASMC_TR_NO_* bits
};

// numpy's floored modulo for floats (npy_divmod): the result takes the sign of the divisor
__device__ __forceinline__ double floored_mod(double a, double b) {
    double m = fmod(a, b);
    if (b == 0.0) return m;
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

__device__ __forceinline__ double clip(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// One coordinate of the transform chain (shared by both kernels): returns the transformed value, adds the bounded block's
// element term to lj_b and reports whether the coordinate belongs to that block.
struct CoordPar {  // the table row of one coordinate, held in registers by the flat kernel
    int kind, periodic;
    double lo, up, mean, std;
    double inv_w, inv_std;  // 1 / (up - lo), 1 / std (true divisions, once per thread)
};
// a / b from the stored reciprocal y = RN(1 / b): q0 = RN(a y), r = a - b q0 (exact in an FMA), q = RN(q0 + r y) - the
// correctly rounded quotient (Markstein) for three FMAs instead of the ~30-instruction IEEE division sequence, whose
// v_rcp_f64 made the forward transforms ALU-bound
__device__ __forceinline__ double div_by(double a, double b, double y) {
    const double q0 = a * y;
    const double r = fma(-b, q0, a);
    return fma(r, y, q0);
}
// HINTS: ASMC_TR_NO_* bits known at compile time - the branches they rule out are not even compiled (the forward kernel
// with fmod, log, log1p and erfinv all inlined twice needed 256 VGPRs: one wave per SIMD)
template <int DIR, int HINTS>
__device__ __forceinline__ double transform_coord(double v, const CoordPar& c, bool affine, double eps, double& lj_b) {
    const double half_log_2pi = 0.9189385332046727;
    const int kind = c.kind;
    const double lo = c.lo, up = c.up;
    struct {
        double eps;
        bool mean;
    } p = {eps, affine};
    const struct {
        double v;
    } pm = {c.mean}, ps = {c.std};
    if (DIR == 0) {
        if (!(HINTS & ASMC_TR_NO_PERIODIC) && c.periodic) v = lo + floored_mod(v - lo, up - lo);
        if ((HINTS & (ASMC_TR_NO_LOGIT | ASMC_TR_NO_PROBIT)) != (ASMC_TR_NO_LOGIT | ASMC_TR_NO_PROBIT) && kind != 0) {
            double u = div_by(v - lo, up - lo, c.inv_w);
            u = clip(u, p.eps, 1.0 - p.eps);
            if (!(HINTS & ASMC_TR_NO_LOGIT) && ((HINTS & ASMC_TR_NO_PROBIT) || kind == 1)) {
                const double a = log(u), b = log1p(-u);
                v = a - b;
                lj_b += -a - b;
            } else {
                v = erfinv(2.0 * u - 1.0) * 1.4142135623730951;
                lj_b += 0.5 * (2.0 * half_log_2pi + v * v);
            }
        }
        if (p.mean) v = div_by(v - pm.v, ps.v, c.inv_std);
    } else {
        if (p.mean) v = v * ps.v + pm.v;
        if ((HINTS & (ASMC_TR_NO_LOGIT | ASMC_TR_NO_PROBIT)) != (ASMC_TR_NO_LOGIT | ASMC_TR_NO_PROBIT) && kind != 0) {
            double u;
            if (!(HINTS & ASMC_TR_NO_LOGIT) && ((HINTS & ASMC_TR_NO_PROBIT) || kind == 1)) {
                // sigmoid and log u + log(1 - u) from ONE exponential: with e = exp(-|v|), r = 1 / (1 + e):
                // u = r (v >= 0) or e r, and log u + log(1 - u) = -|v| - 2 log(1 + e); on the clamped ends the clamp's own
                // constant.  (The reference's expressions - utils.py:196-245 - cost an exp, a division and two logs; the
                // values agree to a few ulp.)
                const double av = fabs(v);
                const double ex = exp(-av);
                const double x1 = 1.0 + ex;  // (1, 2]
                double r = __builtin_amdgcn_rcp(x1);
                r = fma(fma(-x1, r, 1.0), r, r);
                r = fma(fma(-x1, r, 1.0), r, r);
                u = v >= 0.0 ? r : ex * r;
                const bool clamped = u < p.eps || u > 1.0 - p.eps;
                u = clip(u, p.eps, 1.0 - p.eps);
                lj_b += clamped ? log(p.eps) + log1p(-p.eps) : -av - 2.0 * bm_log_unit(x1);
            } else {
                lj_b += -(0.5 * (2.0 * half_log_2pi + v * v));
                u = 0.5 * (1.0 + erf(v / 1.4142135623730951));
            }
            v = (up - lo) * u + lo;
        }
        if (!(HINTS & ASMC_TR_NO_PERIODIC) && c.periodic) v = lo + floored_mod(v - lo, up - lo);
    }
    return v;
}

