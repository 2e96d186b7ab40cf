// asmc_transform.hip — preconditioning transforms and their log-Jacobians.
//
// Replaces, element-wise on the device, the reference's CompositeTransform (src/aspire/transforms.py:142-316)
// and its parts: PeriodicTransform (:411-436), BoundedTransform / ProbitTransform / LogitTransform (:440-611,
// utils.py:196-245 logit / sigmoid) and AffineTransform (:614-646).  Order as in CompositeTransform:
//   forward  x -> z : periodic wrap, bounded -> unbounded (logit or probit of the unit interval), affine
//   inverse  z -> x : affine, unbounded -> bounded, periodic wrap
// log|det J| is summed per particle in the same grouping as the reference: the bounded block's element terms
// first, then its constant (-/+ sum log(upper - lower)), then the affine constant (-/+ sum log|std|).
// All arithmetic is fp64 whatever the storage type of x (fp64 or fp32).
//
// Layout: rows travel HBM -> padded LDS tile -> one particle per lane and back with coalesced 16-byte
// accesses; the per-dimension parameter table (<= 256 dims x 7 doubles) is read through the scalar cache.
// Traffic: 2 d s + 8 bytes per particle — HBM bound.
#include "asmc_common.h"
#include "asmc_tile.h"
#include "asmc_transform_dev.h"

// DT > 0: the dimension as a compile-time constant (16-byte rows, VEC = 16): the tile copies then divide by constants and
// the per-coordinate loop unrolls; DT = 0: any d at run time
template <typename T, int VEC, int DIR, int DT = 0>
__global__ __launch_bounds__(ASMC_BLOCK) void k_transform(int64_t n, const T* __restrict__ in, T* __restrict__ out,
                                                         double* __restrict__ logj, TransDev p, int waves_per_block) {
    extern __shared__ __align__(16) char smem[];
    const int d = DT > 0 ? DT : p.d;
    const int rowbytes = d * (int)sizeof(T);
    const int ldsrow = lds_row_stride(rowbytes);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tile = smem + (size_t)wave * 64 * ldsrow;
    char* myrow = tile + lane * ldsrow;
    const int64_t n_tiles = (n + 63) / 64;
    const double half_log_2pi = 0.9189385332046727;
    for (int64_t t = (int64_t)blockIdx.x * waves_per_block + wave; t < n_tiles; t += (int64_t)gridDim.x * waves_per_block) {
        const int64_t row0 = t * 64;
        const int64_t valid_bytes = ((n - row0) < 64 ? (n - row0) : 64) * (int64_t)rowbytes;
        tile_load<VEC>(reinterpret_cast<const char*>(in) + row0 * rowbytes, valid_bytes, rowbytes, ldsrow, tile, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row0 + lane < n) {
            double lj_b = 0.0;  // element terms of the bounded block
            bool any_bounded = false;
#pragma unroll 4
            for (int j = 0; j < d; j++) {
                double v = row_get<T>(myrow, j);
                const int kind = p.kind[j];
                const double lo = p.lower[j], up = p.upper[j];
                if (DIR == 0) {  // forward
                    if (p.periodic[j]) v = lo + floored_mod(v - lo, up - lo);
                    if (kind != 0) {
                        any_bounded = true;
                        double u = (v - lo) / (up - lo);
                        u = clip(u, p.eps, 1.0 - p.eps);
                        if (kind == 1) {
                            const double a = log(u), b = log1p(-u);
                            v = a - b;
                            lj_b += -a - b;
                        } else {
                            v = erfinv(2.0 * u - 1.0) * 1.4142135623730951;
                            lj_b += 0.5 * (2.0 * half_log_2pi + v * v);
                        }
                    }
                    if (p.mean) v = (v - p.mean[j]) / p.std[j];
                } else {  // inverse
                    if (p.mean) v = v * p.std[j] + p.mean[j];
                    if (kind != 0) {
                        any_bounded = true;
                        double u;
                        if (kind == 1) {
                            u = 1.0 / (1.0 + exp(-v));
                            u = clip(u, p.eps, 1.0 - p.eps);
                            lj_b += log(u) + log1p(-u);
                        } else {
                            lj_b += -(0.5 * (2.0 * half_log_2pi + v * v));
                            u = 0.5 * (1.0 + erf(v / 1.4142135623730951));
                        }
                        v = (up - lo) * u + lo;
                    }
                    if (p.periodic[j]) v = lo + floored_mod(v - lo, up - lo);
                }
                row_set<T>(myrow, j, v);
            }
            if (logj) {
                double lj = 0.0;
                if (DIR == 0) {
                    if (any_bounded) lj += lj_b + p.unit_logj;
                    if (p.mean) lj += p.affine_logj;
                } else {
                    if (p.mean) lj += -p.affine_logj;
                    if (any_bounded) lj += lj_b + (-p.unit_logj);
                }
                logj[row0 + lane] = lj;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tile_store<VEC>(reinterpret_cast<char*>(out) + row0 * rowbytes, valid_bytes, rowbytes, ldsrow, tile, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Flat form for rows of TPR 16-byte pieces (TPR a power of two <= 64): one piece per thread, fully coalesced 16-byte loads
// and stores with no LDS staging, the per-particle log-Jacobian by a butterfly over the TPR lanes of the row.  The
// element terms are summed in butterfly order instead of coordinate order (differences ~1e-16 relative).
template <typename T, int DIR, int HINTS>
__global__ __launch_bounds__(ASMC_BLOCK) void k_transform_flat(int64_t n, int tpr_log2, const uint4* __restrict__ in,
                                                              uint4* __restrict__ out, double* __restrict__ logj, TransDev p) {
    constexpr int EPT = 16 / (int)sizeof(T);  // elements per thread
    const int tpr = 1 << tpr_log2;
    const int64_t total = n << tpr_log2;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;  // a multiple of tpr: a thread keeps its coordinates
    const int c = (int)(((int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x) & (tpr - 1));
    CoordPar par[EPT];
    bool any_b = false;
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int j = c * EPT + k;
        const double lo = p.lower[j], up = p.upper[j], sd = p.mean ? p.std[j] : 1.0;
        par[k] = CoordPar{p.kind[j], p.periodic[j], lo, up, p.mean ? p.mean[j] : 0.0, sd, 1.0 / (up - lo), 1.0 / sd};
        any_b |= par[k].kind != 0;
    }
    const bool affine = p.mean != nullptr;
    for (int64_t e0 = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; e0 - (threadIdx.x & 63) < total; e0 += stride) {
        const bool valid = e0 < total;  // whole waves stay in the loop for the shuffles
        const int64_t row = e0 >> tpr_log2;
        double lj_b = 0.0;
        if (valid) {
            uint4 q = in[e0];
            T* vals = reinterpret_cast<T*>(&q);
#pragma unroll
            for (int k = 0; k < EPT; k++) vals[k] = (T)transform_coord<DIR, HINTS>((double)vals[k], par[k], affine, p.eps, lj_b);
            out[e0] = q;
        }
        if (logj) {
            int ab = any_b ? 1 : 0;
            for (int o = tpr >> 1; o >= 1; o >>= 1) {
                lj_b += __shfl_xor(lj_b, o, 64);
                ab |= __shfl_xor(ab, o, 64);
            }
            if (valid && c == 0) {
                double lj = 0.0;
                if (DIR == 0) {
                    if (ab) lj += lj_b + p.unit_logj;
                    if (p.mean) lj += p.affine_logj;
                } else {
                    if (p.mean) lj += -p.affine_logj;
                    if (ab) lj += lj_b + (-p.unit_logj);
                }
                logj[row] = lj;
            }
        }
    }
}

template <typename T, int DIR>
static int launch_transform(asmc_ctx* ctx, int64_t n, const T* in, T* out, double* logj, const TransDev& p,
                            hipStream_t st) {
    const int rowbytes = p.d * (int)sizeof(T);
    const size_t per_wave = (size_t)64 * lds_row_stride(rowbytes);
    int wpb = ASMC_BLOCK / 64;
    while (wpb > 1 && per_wave * wpb > 64 * 1024) wpb >>= 1;
    const size_t lds = per_wave * wpb;
    ASMC_REQUIRE(lds <= 64 * 1024, "row too long for one LDS tile");
    const int64_t n_tiles = (n + 63) / 64;
    const int grid = grid_for(n_tiles, wpb, ctx->num_cu * 8);
    const uintptr_t a = (uintptr_t)in | (uintptr_t)out;
    const int vec = (rowbytes % 16 == 0 && a % 16 == 0) ? 16 : (rowbytes % 8 == 0 && a % 8 == 0) ? 8 : 4;
    const char* label = DIR == 0 ? "k_transform_forward" : "k_transform_inverse";
    {
        // rows of a power-of-two number (<= 64) of 16-byte pieces: flat kernel, no LDS
        const int pieces = rowbytes / 16;
        if (rowbytes % 16 == 0 && a % 16 == 0 && pieces >= 1 && pieces <= 64 && (pieces & (pieces - 1)) == 0 &&
            !getenv("ASMC_TRANSFORM_TILED")) {
            int lg = 0;
            while ((1 << lg) < pieces) lg++;
            const int g = grid_for(n * pieces, ASMC_BLOCK, ctx->num_cu * 32);
            switch (p.hints & 7) {
#define FLAT_CASE(HH) \
    case HH: ASMC_LAUNCH(ctx, st, label, (k_transform_flat<T, DIR, HH>), dim3(g), dim3(ASMC_BLOCK), 0, st, n, lg, (const uint4*)in, \
                         (uint4*)out, logj, p); break;
                FLAT_CASE(0)
                FLAT_CASE(1)
                FLAT_CASE(2)
                FLAT_CASE(3)
                FLAT_CASE(4)
                FLAT_CASE(5)
                FLAT_CASE(6)
                FLAT_CASE(7)
#undef FLAT_CASE
            }
            ASMC_LAUNCH_CHECK();
            return ASMC_OK;
        }
    }
    if (vec == 16 && (p.d == 8 || p.d == 16 || p.d == 32 || p.d == 64 || p.d == 128) && !getenv("ASMC_TRANSFORM_GENERIC")) {
        switch (p.d) {
#define TR_CASE(DD) \
    case DD: ASMC_LAUNCH(ctx, st, label, (k_transform<T, 16, DIR, DD>), dim3(grid), dim3(wpb * 64), lds, st, n, in, out, logj, p, wpb); break;
            TR_CASE(8)
            TR_CASE(16)
            TR_CASE(32)
            TR_CASE(64)
            TR_CASE(128)
#undef TR_CASE
        }
    } else if (vec == 16)
        ASMC_LAUNCH(ctx, st, label, (k_transform<T, 16, DIR>), dim3(grid), dim3(wpb * 64), lds, st, n, in, out, logj, p, wpb);
    else if (vec == 8)
        ASMC_LAUNCH(ctx, st, label, (k_transform<T, 8, DIR>), dim3(grid), dim3(wpb * 64), lds, st, n, in, out, logj, p, wpb);
    else
        ASMC_LAUNCH(ctx, st, label, (k_transform<T, 4, DIR>), dim3(grid), dim3(wpb * 64), lds, st, n, in, out, logj, p, wpb);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

static int run_transform(asmc_ctx* ctx, int64_t n, int x_dtype, const void* in, void* out, double* logj,
                         const asmc_transform* t, int dir, asmc_stream stream) {
    ASMC_REQUIRE(ctx && in && out && t, "null pointer");
    ASMC_REQUIRE(n > 0, "n must be positive");
    ASMC_REQUIRE(t->d > 0 && t->d <= ASMC_MAX_DIMS, "bad d");
    ASMC_REQUIRE(t->kind_dev && t->periodic_dev && t->lower_dev && t->upper_dev, "transform tables missing");
    ASMC_REQUIRE((t->mean_dev == nullptr) == (t->std_dev == nullptr), "affine stage needs both mean and std");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    TransDev p;
    p.d = t->d;
    p.kind = t->kind_dev;
    p.periodic = t->periodic_dev;
    p.lower = t->lower_dev;
    p.upper = t->upper_dev;
    p.mean = t->mean_dev;
    p.std = t->std_dev;
    p.eps = t->eps;
    p.unit_logj = t->unit_logj;
    p.affine_logj = t->affine_logj;
    p.hints = t->hints;
    hipStream_t st = as_stream(stream);
    if (x_dtype == ASMC_F64)
        return dir == 0 ? launch_transform<double, 0>(ctx, n, (const double*)in, (double*)out, logj, p, st)
                        : launch_transform<double, 1>(ctx, n, (const double*)in, (double*)out, logj, p, st);
    return dir == 0 ? launch_transform<float, 0>(ctx, n, (const float*)in, (float*)out, logj, p, st)
                    : launch_transform<float, 1>(ctx, n, (const float*)in, (float*)out, logj, p, st);
}

extern "C" {

int asmc_transform_forward(asmc_ctx* ctx, int64_t n, int x_dtype, const void* x_dev, void* z_dev, double* logj_dev,
                           const asmc_transform* t, asmc_stream stream) {
    return run_transform(ctx, n, x_dtype, x_dev, z_dev, logj_dev, t, 0, stream);
}

int asmc_transform_inverse(asmc_ctx* ctx, int64_t n, int x_dtype, const void* z_dev, void* x_dev, double* logj_dev,
                           const asmc_transform* t, asmc_stream stream) {
    return run_transform(ctx, n, x_dtype, z_dev, x_dev, logj_dev, t, 1, stream);
}

}  // extern "C"
