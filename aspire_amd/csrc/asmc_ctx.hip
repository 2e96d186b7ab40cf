// asmc_ctx.hip — library context, scratch allocation, error reporting.
#include <stdarg.h>

#include <math.h>

#include "asmc_common.h"

static thread_local char g_err[512] = "";

void asmc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Diagnostic (ASMC_POISON_LDS=1): one block per CU that owns the whole LDS and fills it with 0xFF bytes - NaN as fp64 / fp32 /
// fp16, -1 as an integer.  In front of every launch it turns a read of uninitialised LDS into a repeatable wrong answer (the
// GPU address sanitizer is not available on this pool); the persistent kernels' residency rules are not affected (it ends
// before the next kernel starts).
__global__ __launch_bounds__(256) void k_poison_lds(int words) {
    extern __shared__ unsigned int s_poison[];
    for (int i = threadIdx.x; i < words; i += 256) s_poison[i] = 0xFFFFFFFFu;
    __syncthreads();
    if (s_poison[(threadIdx.x * 977) % words] != 0xFFFFFFFFu) __builtin_trap();  // (keeps the stores alive)
}
void asmc_poison_lds(asmc_ctx* ctx, hipStream_t st) {
    const int bytes = 160 * 1024 - 1024;
    static bool attr_set_dev[ASMC_MAX_DEVICES] = {false}; bool& attr_set = attr_set_dev[asmc_dev_slot(ctx)];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_poison_lds, dim3(4 * ctx->num_cu), dim3(256), bytes, st, bytes / 4);
}

void asmc_prof_begin(asmc_ctx* ctx, const char* label, hipStream_t st) {
    if (!ctx || !ctx->prof_on || ctx->prof_n >= ASMC_PROF_MAX) return;
    ctx->prof_label[ctx->prof_n] = label;
    (void)hipEventRecord(ctx->prof_ev[2 * ctx->prof_n], st);
}

void asmc_prof_end(asmc_ctx* ctx, hipStream_t st) {
    if (!ctx || !ctx->prof_on || ctx->prof_n >= ASMC_PROF_MAX) return;
    (void)hipEventRecord(ctx->prof_ev[2 * ctx->prof_n + 1], st);
    ctx->prof_n++;
}

// The tables of the default noise's Box-Muller transform (asmc_pcn_dev.h bm_pair32), in long double libm:
// [0, 256): (sin, cos)(2 pi (k + 1/2) / 256); [256, 384): (rc_i, 2 ln rc_i), rc_i = fl(1 / c_i), c_i = 1/2 + (i + 1/2) / 256.
void asmc_bm_table_host(double* tab) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < BM_SC_N; k++) {
        const long double ang = two_pi * ((long double)k + 0.5L) / (long double)BM_SC_N;
        tab[2 * k] = (double)sinl(ang);
        tab[2 * k + 1] = (double)cosl(ang);
    }
    for (int i = 0; i < BM_LG_N; i++) {
        const long double c = 0.5L + ((long double)i + 0.5L) / (2.0L * BM_LG_N);
        const double rc = (double)(1.0L / c);
        tab[2 * (BM_SC_N + i)] = rc;
        tab[2 * (BM_SC_N + i) + 1] = (double)(2.0L * logl((long double)rc));
    }
}

extern "C" {

int asmc_profile_enable(asmc_ctx* ctx, int on) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    if (on && !ctx->prof_ev) {
        ctx->prof_ev = new (std::nothrow) hipEvent_t[2 * ASMC_PROF_MAX];
        ctx->prof_label = new (std::nothrow) const char*[ASMC_PROF_MAX];
        if (!ctx->prof_ev || !ctx->prof_label) {
            asmc_set_error("out of host memory");
            return ASMC_ERR_NOMEM;
        }
        for (int i = 0; i < 2 * ASMC_PROF_MAX; i++) ASMC_HIP(hipEventCreate(&ctx->prof_ev[i]));
    }
    ctx->prof_on = on ? 1 : 0;
    if (on) ctx->prof_n = 0;
    return ASMC_OK;
}

int asmc_profile_report(asmc_ctx* ctx, char* buf, int64_t buf_len) {
    ASMC_REQUIRE(ctx && buf && buf_len > 0, "bad arguments");
    buf[0] = 0;
    if (!ctx->prof_ev || ctx->prof_n == 0) return ASMC_OK;
    ASMC_HIP(hipEventSynchronize(ctx->prof_ev[2 * (ctx->prof_n - 1) + 1]));
    // aggregate by label (labels are string literals: pointer or content equality)
    const int n = ctx->prof_n;
    int64_t off = 0;
    char* done = new (std::nothrow) char[n]();
    if (!done) return ASMC_ERR_NOMEM;
    for (int i = 0; i < n; i++) {
        if (done[i]) continue;
        double tot = 0.0;
        int cnt = 0;
        for (int j = i; j < n; j++) {
            if (done[j] || strcmp(ctx->prof_label[j], ctx->prof_label[i]) != 0) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->prof_ev[2 * j], ctx->prof_ev[2 * j + 1]) == hipSuccess) {
                tot += ms;
                cnt++;
            }
            done[j] = 1;
        }
        const int w = snprintf(buf + off, (size_t)(buf_len - off), "%s %d %.6f\n", ctx->prof_label[i], cnt, cnt ? tot / cnt : 0.0);
        if (w < 0 || off + w >= buf_len) break;
        off += w;
    }
    delete[] done;
    ctx->prof_n = 0;
    return ASMC_OK;
}

int asmc_abi_version(void) { return ASMC_ABI_VERSION; }

const char* asmc_last_error(void) { return g_err; }

int asmc_device_count(int* n_out) {
    ASMC_REQUIRE(n_out != nullptr, "null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *n_out = 0;
        asmc_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return ASMC_ERR_HIP;
    }
    *n_out = n;
    return ASMC_OK;
}

int asmc_ctx_create(asmc_ctx** ctx_out, int device, int64_t n_max, int d_max) {
    ASMC_REQUIRE(ctx_out != nullptr, "null ctx_out");
    ASMC_REQUIRE(n_max > 0, "n_max must be positive");
    ASMC_REQUIRE(d_max > 0 && d_max <= ASMC_MAX_DIMS, "d_max out of range");
    *ctx_out = nullptr;
    ASMC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ASMC_HIP(hipGetDeviceProperties(&prop, device));
    asmc_ctx* c = new (std::nothrow) asmc_ctx();
    if (!c) {
        asmc_set_error("out of host memory");
        return ASMC_ERR_NOMEM;
    }
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->n_max = n_max;
    c->d_max = d_max;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->n_tiles_max = (n_max + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE + 1;
    c->gram_blocks = (d_max <= 128 ? 4 : 2) * c->num_cu;  // partial matrices of the moment kernels (d_max^2 doubles each)
    if (c->gram_blocks > ASMC_MAX_BLOCKS) c->gram_blocks = ASMC_MAX_BLOCKS;
    hipError_t e = hipSuccess;
    auto dmalloc = [&](void** p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
    };
    dmalloc((void**)&c->d_partials, sizeof(double) * ASMC_MAX_BLOCKS * ASMC_MAX_BETAS * 2);
    dmalloc((void**)&c->d_small, sizeof(double) * 4096);
    dmalloc((void**)&c->d_keys, sizeof(unsigned long long) * (ASMC_MAX_BETAS + 8));
    dmalloc((void**)&c->d_tiles, sizeof(double) * (size_t)(c->n_tiles_max * 4 + 64));
    dmalloc((void**)&c->d_tiles_i, sizeof(long long) * (size_t)(c->n_tiles_max * 8 + 64));
    c->gram_cap = (size_t)c->gram_blocks * d_max * d_max;
    if (c->gram_cap < (size_t)2048 * 1024) c->gram_cap = (size_t)2048 * 1024;
    dmalloc((void**)&c->d_gram, sizeof(double) * c->gram_cap);
    c->d_mmtab = nullptr;
    c->poison_lds = getenv("ASMC_POISON_LDS") != nullptr;
    c->d_max_pad = d_max <= 4 ? 4 : d_max <= 8 ? 8 : d_max <= 16 ? 16 : d_max <= 32 ? 32 : d_max <= 64 ? 64 : d_max <= 128 ? 128 : 0;
    // operand images of the d = 64 / 128 kernels (147 KB): every context has them - a d <= 32 problem whose proposal flow lives in the
    // 16-particle-group layout (asmc_flow_layout = 1: an autoregressive flow of hidden width 128) runs zero-padded on the D = 64 step
    dmalloc((void**)&c->d_mmtab, sizeof(double) * 2 * 144 * 64);
    dmalloc((void**)&c->d_guide, sizeof(unsigned int) * ((size_t)n_max + 8));
    // the fused flow step's counters sit directly in FRONT of the flag bytes: one memset clears both (asmc_pcn_mutate_flow)
    dmalloc((void**)&c->d_tilectr, ASMC_TILECTR_BYTES + (size_t)n_max + 64);
    c->d_flags = c->d_tilectr ? reinterpret_cast<unsigned char*>(c->d_tilectr) + ASMC_TILECTR_BYTES : nullptr;
    dmalloc((void**)&c->d_gamma, sizeof(double) * ((size_t)ASMC_GAMMA_BATCH * (size_t)n_max + 64 * ASMC_GAMMA_BATCH));
    dmalloc((void**)&c->d_rec, sizeof(double) * 4 * (size_t)n_max);
    const size_t student = (size_t)d_max * (d_max + 1) + (size_t)(ASMC_STUDENT_MAX_ROWS / 64) * (d_max + 2);
    dmalloc((void**)&c->d_student, sizeof(double) * student);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_student, sizeof(double) * student, hipHostMallocDefault);
    dmalloc((void**)&c->d_counts, sizeof(long long) * (size_t)(ASMC_MAX_PCN_STEPS + ASMC_MAX_BLOCKS + n_max / 64 + 8));
    dmalloc((void**)&c->d_rho, sizeof(double) * (ASMC_MAX_PCN_STEPS + 8));
    if (e == hipSuccess) e = hipMemset(c->d_tilectr, 0, ASMC_TILECTR_BYTES);
    dmalloc((void**)&c->d_bar, sizeof(unsigned int) * 1024 * 17);
    if (e == hipSuccess) e = hipMemset(c->d_bar, 0, sizeof(unsigned int) * 1024 * 17);
    dmalloc((void**)&c->d_pcgtab, sizeof(unsigned long long) * (64 * 4 + 8));
    dmalloc((void**)&c->d_select, sizeof(long long) * (2 * (ASMC_SELECT_THREADS / 64) + 8));
    dmalloc((void**)&c->d_ptab, sizeof(double) * (2 * 32 * 32 + 32 + 3 * ASMC_MAX_COMPONENTS * (1 + 2 * 32) + 64));
    dmalloc((void**)&c->d_bmtab, sizeof(double) * 2 * BM_TAB_N);
    dmalloc((void**)&c->d_ref, sizeof(double) * (128 + 128 * 128));
    if (e == hipSuccess) {
        double tab[2 * BM_TAB_N];
        asmc_bm_table_host(tab);
        e = hipMemcpy(c->d_bmtab, tab, sizeof(tab), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_pinned, sizeof(double) * 8192, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_gram, sizeof(double) * (128 + 128 * 128), hipHostMallocDefault);
    if (e != hipSuccess) {
        asmc_set_error("asmc_ctx_create: allocation failed: %s", hipGetErrorString(e));
        asmc_ctx_destroy(c);
        return ASMC_ERR_NOMEM;
    }
    *ctx_out = c;
    return ASMC_OK;
}

int asmc_ctx_destroy(asmc_ctx* c) {
    if (!c) return ASMC_OK;
    (void)hipSetDevice(c->device);
    (void)hipFree(c->d_partials);
    (void)hipFree(c->d_small);
    (void)hipFree(c->d_keys);
    (void)hipFree(c->d_tiles);
    (void)hipFree(c->d_tiles_i);
    (void)hipFree(c->d_gram);
    if (c->d_mmtab) (void)hipFree(c->d_mmtab);
    if (c->d_f16tab) (void)hipFree(c->d_f16tab);
    (void)hipFree(c->d_guide);
    (void)hipFree(c->d_gamma);
    (void)hipFree(c->d_student);
    if (c->h_student) (void)hipHostFree(c->h_student);
    if (c->d_ysoa) (void)hipFree(c->d_ysoa);
    if (c->d_xpad) (void)hipFree(c->d_xpad);
    (void)hipFree(c->d_counts);
    (void)hipFree(c->d_rho);
    (void)hipFree(c->d_tilectr);
    (void)hipFree(c->d_rec);
    (void)hipFree(c->d_bar);
    (void)hipFree(c->d_pcgtab);
    (void)hipFree(c->d_select);
    (void)hipFree(c->d_ptab);
    (void)hipFree(c->d_bmtab);
    (void)hipFree(c->d_ref);
    if (c->prof_ev) {
        for (int i = 0; i < 2 * ASMC_PROF_MAX; i++) (void)hipEventDestroy(c->prof_ev[i]);
        delete[] c->prof_ev;
        delete[] c->prof_label;
    }
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_gram) (void)hipHostFree(c->h_gram);
    if (c->ev_mutate) (void)hipEventDestroy(c->ev_mutate);
    if (c->ev_is) (void)hipEventDestroy(c->ev_is);
    delete c;
    return ASMC_OK;
}

}  // extern "C"
