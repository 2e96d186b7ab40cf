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
                                                          const double* __restrict__ tab, double nu_val,
                                                          double* __restrict__ z_out, double* __restrict__ partials,
                                                          const double* __restrict__ em, int it) {
    extern __shared__ double s_dx[];
    // em: the state record of the device-side EM (asmc_student_fit): [0] nu, [2] the iteration that converged
    if (em && (double)it > em[2]) return;
    const double nu = em ? em[0] : nu_val;  // [d][64]: centred row of every lane, coordinate-major (conflict-free) | packed lower triangle
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
                                                             double* __restrict__ r, const double* __restrict__ em, int it) {
    if (em && (double)it > em[2]) return;
    const int64_t total = m * d;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t e = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / d;
        const int k = (int)(e - i * d);
        r[e] = sqrt(z[i]) * (xs[e] - mu[k]);
    }
}

// ---- the whole EM on the device (asmc_student_fit) -------------------------------------------------------------------------
// State record em[]: [0] nu  [1] iterations done  [2] the iteration that met the stopping rule (1e9: none yet; kernels of later
// iterations return at once)  [3] -1 when a scale matrix could not be factored  [4] rtol  [5] sum z  [6] sum (log z - z)

// log(x) - digamma(x) for x > 0 without the cancellation of the two (recurrence up to x >= 10, then the asymptotic series)
__device__ __forceinline__ double log_minus_digamma(double x) {
    double acc = 0.0, y = x;
    while (y < 10.0) {
        acc += 1.0 / y;
        y += 1.0;
    }
    const double f = 1.0 / (y * y);
    const double tail = 0.5 / y + f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0)))));
    return acc - log(y / x) + tail;  // (y == x: log 1 = 0)
}

// M-step's vector / scalar half in one block: the block partials of the E-step summed in block order (the host's order), the
// weighted mean into the table the next kernels read, and the root of
//     log(nu/2) - psi(nu/2) + 1 + mean(log z - z) + psi((nu0 + d)/2) - log((nu0 + d)/2) = 0
// (Liu & Rubin; student_t.py) by a 64-way section search on log nu over [1, 1e6]: the left side decreases in nu.
#define NU_MIN_DEV 1.0
#define NU_MAX_DEV 1.0e6
__global__ __launch_bounds__(256) void k_student_mstep(int64_t m, int d, int blocks, const double* __restrict__ partials,
                                                      double* __restrict__ tab, double* __restrict__ em, int it) {
    __shared__ double s_sum[130];
    __shared__ double s_f[64];
    __shared__ double s_lab[2];
    if ((double)it > em[2]) return;
    const int tid = threadIdx.x;
    // every thread takes its copy of the state BEFORE the first barrier: thread 0 rewrites em[0] at the end of the kernel, and a
    // wave that read it late would see the new nu, form another `c` and could take the branch with the barriers alone
    const double nu = em[0], rtol = em[4];
    if (tid < d + 2) {
        double t = 0.0;
        for (int b = 0; b < blocks; b++) t += partials[(size_t)b * (d + 2) + tid];  // block order: the host's sum
        s_sum[tid] = t;
    }
    __syncthreads();
    const double sum_z = s_sum[0], sum_lz = s_sum[1];
    if (tid < d) tab[tid] = s_sum[2 + tid] / sum_z;
    const double c = 1.0 + sum_lz / (double)m - log_minus_digamma(0.5 * (nu + (double)d));
    // f(v) = log(v/2) - psi(v/2) + c, decreasing from +inf to c
    double nu_new;
    const double f_max = log_minus_digamma(0.5 * NU_MAX_DEV) + c, f_min = log_minus_digamma(0.5 * NU_MIN_DEV) + c;
    if (!(c < 0.0) || f_max >= 0.0) {
        nu_new = NU_MAX_DEV;
    } else if (f_min <= 0.0) {
        nu_new = NU_MIN_DEV;
    } else {
        if (tid == 0) s_lab[0] = log(NU_MIN_DEV), s_lab[1] = log(NU_MAX_DEV);
        __syncthreads();
        for (int round = 0; round < 8; round++) {
            const double la = s_lab[0], lb = s_lab[1];
            if (tid < 64) s_f[tid] = log_minus_digamma(0.5 * exp(la + (lb - la) * (double)(tid + 1) / 65.0)) + c;
            __syncthreads();
            if (tid == 0) {
                int k = 0;  // points 1 .. k are still above the root
                while (k < 64 && s_f[k] > 0.0) k++;
                s_lab[0] = la + (lb - la) * (double)k / 65.0;
                s_lab[1] = la + (lb - la) * (double)(k + 1) / 65.0;
            }
            __syncthreads();
        }
        nu_new = exp(0.5 * (s_lab[0] + s_lab[1]));
    }
    if (tid == 0) {
        const bool done = fabs(nu_new - nu) <= rtol * nu;
        em[0] = nu_new;
        em[1] = (double)(it + 1);
        em[5] = sum_z, em[6] = sum_lz;
        if (done) em[2] = (double)it;  // this iteration's scatter matrix is still computed; later iterations are skipped
    }
}

// the scatter matrix of an iteration into the slot the next factorisation reads (unless the EM has stopped before it)
__global__ __launch_bounds__(256) void k_student_keep(int d, const double* __restrict__ gram, double* __restrict__ keep,
                                                     const double* __restrict__ em, int it) {
    if ((double)it > em[2]) return;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < d * d) keep[e] = gram[e];
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
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_student_estep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    ASMC_LAUNCH(ctx, st, "k_student_estep", k_student_estep, dim3(blocks), dim3(ST_ROWS), lds, st, m, d, xs, (const double*)d_tab, nu,
                z_dev, d_part, (const double*)nullptr, 0);
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
                (const double*)ctx->d_student, r_dev, (const double*)nullptr, 0);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// The EM of student_t.fit_student_t_device with every step on the stream: initial moments, then per iteration the factorisation
// of the scale matrix (k_ref_factor), the E-step, the M-step's mean / degrees of freedom (k_student_mstep) and scatter matrix
// (scale + Gram on the matrix cores); iterations behind the one that meets |nu' - nu| <= rtol nu return at once.  One
// synchronisation at the end.  out_dev: (mu | L | Linv) of the final fit in asmc_reference_factor's layout.
// result_host: [0] nu (clamped to [1, 1e6]) [1] iterations [2] factorisation status (0 / -1) [3] converged, then mu[d], Sigma[d*d].
int asmc_student_fit(asmc_ctx* ctx, int64_t m, int d, const double* xs, int max_iter, double rtol, double nu0, double* r_scratch,
                     double* z_scratch, double* out_dev, double* result_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && xs && r_scratch && z_scratch && out_dev && result_host, "null pointer");
    ASMC_REQUIRE(m > 1 && m <= ASMC_STUDENT_MAX_ROWS, "subsample size out of range");
    ASMC_REQUIRE(d > 0 && d <= ctx->d_max && asmc_gram_mm_supported(d, xs) && asmc_gram_mm_supported(d, r_scratch),
                 "shape without the matrix-core Gram kernel (d in {32, 64, 128}, 16-byte aligned rows): the host-driven EM serves it");
    ASMC_REQUIRE(max_iter >= 1 && max_iter <= 256 && nu0 > 0.0 && rtol >= 0.0, "bad EM parameters");
    ASMC_REQUIRE(ctx->gram_pending_d == 0, "an asmc_mean_gram_enqueue is pending: its results would be overwritten");
    hipStream_t st = as_stream(stream);
    const int blocks = (int)((m + ST_ROWS - 1) / ST_ROWS);
    double* d_tab = ctx->d_student;
    double* d_part = ctx->d_student + (size_t)ctx->d_max * (ctx->d_max + 1);
    double* d_em = ctx->d_small + 2320;  // 8 doubles
    double* d_zero = ctx->d_small + 2048;  // the Gram kernel's centre
    double* h = ctx->h_student;
    ASMC_HIP(hipStreamSynchronize(st));  // pinned staging may still be in flight from an earlier call
    h[0] = nu0, h[1] = 0.0, h[2] = 1e9, h[3] = 0.0, h[4] = rtol, h[5] = h[6] = h[7] = 0.0;
    ASMC_HIP(hipMemcpyAsync(d_em, h, sizeof(double) * 8, hipMemcpyHostToDevice, st));
    // initial mean and covariance (ddof = 1): column sums -> centre -> Gram, all on the stream; kept in ctx->d_ref
    int rc = asmc_mean_gram_enqueue(ctx, m, d, ASMC_F64, xs, m, 0, stream);
    if (rc) return rc;
    ctx->gram_pending_d = 0;  // consumed here
    ASMC_HIP(hipMemsetAsync(d_zero, 0, sizeof(double) * d, st));
    const size_t dpad = ((size_t)d + 3) & ~(size_t)3;
    const size_t lds = sizeof(double) * ((size_t)d * ST_ROWS + dpad * (dpad + 1) / 2);
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_student_estep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int sgrid = grid_for(m * d, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS);
    for (int it = 0; it < max_iter; it++) {
        // scale matrix of this iteration -> packed Linv (first iteration: also the mean, from the column sums)
        rc = asmc_ref_factor_launch(ctx, d, it == 0 ? ctx->d_ref : nullptr, ctx->d_ref + 128, (double)m,
                                    it == 0 ? (double)(m - 1) : (double)m, nullptr, nullptr, d_tab, d_em, it, st);
        if (rc) return rc;
        ASMC_LAUNCH(ctx, st, "k_student_estep", k_student_estep, dim3(blocks), dim3(ST_ROWS), lds, st, m, d, xs, (const double*)d_tab, 0.0,
                    z_scratch, d_part, (const double*)d_em, it);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_student_mstep", k_student_mstep, dim3(1), dim3(256), 0, st, m, d, blocks, (const double*)d_part, d_tab, d_em, it);
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_student_scale", k_student_scale, dim3(sgrid), dim3(ASMC_BLOCK), 0, st, m, d, xs, (const double*)z_scratch,
                    (const double*)d_tab, r_scratch, (const double*)d_em, it);
        ASMC_LAUNCH_CHECK();
        int ggrid = 0;
        rc = asmc_gram_mm_launch(ctx, m, d, ASMC_F64, r_scratch, d_zero, &ggrid, st, nullptr, 0.0);
        if (rc) return rc;
        ASMC_LAUNCH(ctx, st, "k_student_keep", k_student_keep, dim3((d * d + 255) / 256), dim3(256), 0, st, d, (const double*)ctx->d_partials,
                    ctx->d_ref + 128, (const double*)d_em, it);
        ASMC_LAUNCH_CHECK();
    }
    // the final fit's factor and inverse for the mutation (mu from the table: sum = mu, n_mean = 1; Sigma = scatter / m)
    double* d_status = ctx->d_small + 2300;
    rc = asmc_ref_factor_launch(ctx, d, d_tab, ctx->d_ref + 128, 1.0, (double)m, out_dev, d_status, nullptr, nullptr, 0, st);
    if (rc) return rc;
    ASMC_HIP(hipMemcpyAsync(h, d_em, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(h + 8, d_status, sizeof(double), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(h + 16, d_tab, sizeof(double) * d, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(h + 16 + 128, ctx->d_ref + 128, sizeof(double) * d * d, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    const double nu = h[0];
    result_host[0] = nu < NU_MIN_DEV ? NU_MIN_DEV : nu > NU_MAX_DEV ? NU_MAX_DEV : nu;
    result_host[1] = h[1];
    result_host[2] = (h[3] < 0.0 || h[8] < 0.0) ? -1.0 : 0.0;
    result_host[3] = h[2] < 1e8 ? 1.0 : 0.0;
    memcpy(result_host + 4, h + 16, sizeof(double) * d);
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++)
            result_host[4 + d + i * d + j] = 0.5 * (h[16 + 128 + i * d + j] / (double)m + h[16 + 128 + j * d + i] / (double)m);
    return ASMC_OK;
}

}  // extern "C"
