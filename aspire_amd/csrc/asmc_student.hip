// asmc_student.hip — per-particle half of the Student-t reference fit of the tpCN mutation (aspire_amd/student_t.py).
// The EM for a multivariate t runs on a SUBSAMPLE of the particles (m <= 16384 rows gathered on the device); these two
// kernels do everything that touches a particle, the host keeps the d-vector / d x d / scalar algebra (weighted mean,
// Cholesky factor, the root in nu) exactly as it keeps the Cholesky factor of the Gaussian reference.
//   E-step    y = Linv (x - mu), delta = |y|^2, z = (nu + d)/(nu + delta);  sums of z, log z - z and z x
//   scaling   r = sqrt(z) (x - mu')   (its Gram matrix, asmc_centered_gram around 0, is the M-step's scatter matrix)
#include "asmc_common.h"

#define ST_ROWS 64  // rows per block: one wave, one row per lane

// tab = mu[d] | Linv[d*d] (row-major, lower triangle used); partials[block][d + 2] = { sum z, sum (log z - z), sum z x[0..d) }
__global__ __launch_bounds__(ST_ROWS) void k_student_estep(int64_t m, int d, const double* __restrict__ xs,
                                                          const double* __restrict__ tab, double nu,
                                                          double* __restrict__ z_out, double* __restrict__ partials) {
    extern __shared__ double s_dx[];  // [d][64]: centred row of every lane, coordinate-major (conflict-free) | packed lower triangle
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * ST_ROWS + lane;
    const bool valid = i < m;
    const double* mu = tab;
    // The triangle's coefficients are wave-uniform.  Read from global memory they were one scalar load and one wait per
    // FMA of a single dependent chain - 8 256 exposed latencies per row at d = 128, 434 us for 16 384 rows with one wave
    // per CU.  Staged into LDS (66 KB at d = 128, next to the 64 KB of centred rows) the inner loop is two LDS reads and an
    // FMA, and hipcc pipelines the reads across iterations.  Same order of operations: same bits.
    double* s_L = s_dx + (size_t)d * ST_ROWS;
    const int tri = d * (d + 1) / 2;
    const double* Lp = tab + d;  // packed by the host: row j at j (j + 1) / 2
#pragma unroll 8
    for (int e = lane; e < tri; e += ST_ROWS) s_L[e] = Lp[e];
    for (int k = 0; k < d; k++) s_dx[k * ST_ROWS + lane] = valid ? xs[(size_t)i * d + k] - mu[k] : 0.0;
    __syncthreads();
    // four rows at a time: four independent FMA chains share every read of the centred coordinate; each row's own chain
    // still runs k = 0 .. j in order, and delta collects the rows in order
    double delta = 0.0;
    for (int j0 = 0; j0 < d; j0 += 4) {
        const double* r0 = s_L + (size_t)j0 * (j0 + 1) / 2;
        const double* r1 = r0 + j0 + 1;
        const double* r2 = r1 + j0 + 2;
        const double* r3 = r2 + j0 + 3;
        double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
#pragma unroll 4
        for (int k = 0; k <= j0; k++) {
            const double xk = s_dx[k * ST_ROWS + lane];
            y0 = fma(r0[k], xk, y0);
            y1 = fma(r1[k], xk, y1);
            y2 = fma(r2[k], xk, y2);
            y3 = fma(r3[k], xk, y3);
        }
        const int rem = d - j0;  // rows j0 .. j0 + min(4, rem) - 1 exist
        if (rem > 1) {
            const double x1 = s_dx[(j0 + 1) * ST_ROWS + lane];
            y1 = fma(r1[j0 + 1], x1, y1);
            if (rem > 2) {
                y2 = fma(r2[j0 + 1], x1, y2);
                const double x2 = s_dx[(j0 + 2) * ST_ROWS + lane];
                y2 = fma(r2[j0 + 2], x2, y2);
                if (rem > 3) {
                    y3 = fma(r3[j0 + 1], x1, y3);
                    y3 = fma(r3[j0 + 2], x2, y3);
                    y3 = fma(r3[j0 + 3], s_dx[(j0 + 3) * ST_ROWS + lane], y3);
                }
            }
        }
        delta = fma(y0, y0, delta);
        if (rem > 1) delta = fma(y1, y1, delta);
        if (rem > 2) delta = fma(y2, y2, delta);
        if (rem > 3) delta = fma(y3, y3, delta);
    }
    const double z = valid ? (nu + (double)d) / (nu + delta) : 0.0;
    if (valid) z_out[i] = z;
    double* out = partials + (size_t)blockIdx.x * (d + 2);
    const double sz = wave_sum(z), sl = wave_sum(valid ? log(z) - z : 0.0);
    if (lane == 0) out[0] = sz, out[1] = sl;
    for (int k = 0; k < d; k++) {
        const double t = wave_sum(z * (s_dx[k * ST_ROWS + lane] + mu[k]));
        if (lane == 0) out[2 + k] = t;
    }
}

__global__ __launch_bounds__(ASMC_BLOCK) void k_student_scale(int64_t m, int d, const double* __restrict__ xs,
                                                             const double* __restrict__ z, const double* __restrict__ mu,
                                                             double* __restrict__ r) {
    const int64_t total = m * d;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t e = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / d;
        const int k = (int)(e - i * d);
        r[e] = sqrt(z[i]) * (xs[e] - mu[k]);
    }
}

extern "C" {

int asmc_student_estep(asmc_ctx* ctx, int64_t m, int d, const double* xs, const double* mu_host, const double* linv_host,
                       double nu, double* z_dev, double* sums_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && xs && mu_host && linv_host && z_dev && sums_host, "null pointer");
    ASMC_REQUIRE(m > 0 && m <= ASMC_STUDENT_MAX_ROWS, "subsample size out of range");
    ASMC_REQUIRE(d > 0 && d <= ctx->d_max, "bad d");
    ASMC_REQUIRE(nu > 0.0, "nu must be positive");
    hipStream_t st = as_stream(stream);
    const int blocks = (int)((m + ST_ROWS - 1) / ST_ROWS);
    double* d_tab = ctx->d_student;                                   // mu | Linv
    double* d_part = ctx->d_student + (size_t)ctx->d_max * (ctx->d_max + 1);  // [blocks][d + 2]
    double* h = ctx->h_student;
    ASMC_HIP(hipStreamSynchronize(st));  // pinned staging may still be in flight from the previous call
    memcpy(h, mu_host, sizeof(double) * d);
    for (int j = 0; j < d; j++)  // lower triangle, packed: the kernel stages it into LDS with one contiguous copy
        memcpy(h + d + (size_t)j * (j + 1) / 2, linv_host + (size_t)j * d, sizeof(double) * (j + 1));
    ASMC_HIP(hipMemcpyAsync(d_tab, h, sizeof(double) * ((size_t)d * (d + 1) / 2 + d), hipMemcpyHostToDevice, st));
    const size_t dpad = ((size_t)d + 3) & ~(size_t)3;  // the kernel walks four rows at a time (rows >= d are read, never used)
    const size_t lds = sizeof(double) * ((size_t)d * ST_ROWS + dpad * (dpad + 1) / 2);
    static size_t attr_lds = 0;
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_student_estep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    ASMC_LAUNCH(ctx, st, "k_student_estep", k_student_estep, dim3(blocks), dim3(ST_ROWS), lds, st, m, d, xs, (const double*)d_tab, nu,
                z_dev, d_part);
    ASMC_LAUNCH_CHECK();
    ASMC_HIP(hipMemcpyAsync(h, d_part, sizeof(double) * (size_t)blocks * (d + 2), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    for (int c = 0; c < d + 2; c++) {  // block order: deterministic
        double t = 0.0;
        for (int b = 0; b < blocks; b++) t += h[(size_t)b * (d + 2) + c];
        sums_host[c] = t;
    }
    return ASMC_OK;
}

int asmc_student_scale(asmc_ctx* ctx, int64_t m, int d, const double* xs, const double* z_dev, const double* mu_host,
                       double* r_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && xs && z_dev && mu_host && r_dev, "null pointer");
    ASMC_REQUIRE(m > 0 && m <= ASMC_STUDENT_MAX_ROWS && d > 0 && d <= ctx->d_max, "bad sizes");
    hipStream_t st = as_stream(stream);
    double* h = ctx->h_student;
    ASMC_HIP(hipStreamSynchronize(st));
    memcpy(h, mu_host, sizeof(double) * d);
    ASMC_HIP(hipMemcpyAsync(ctx->d_student, h, sizeof(double) * d, hipMemcpyHostToDevice, st));
    const int grid = grid_for(m * d, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS);
    ASMC_LAUNCH(ctx, st, "k_student_scale", k_student_scale, dim3(grid), dim3(ASMC_BLOCK), 0, st, m, d, xs, z_dev,
                (const double*)ctx->d_student, r_dev);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

}  // extern "C"
