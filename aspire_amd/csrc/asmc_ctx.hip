// asmc_ctx.hip — library context, scratch allocation, error reporting.
#include <stdarg.h>

#include "asmc_common.h"

static thread_local char g_err[512] = "";

void asmc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

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
    c->gram_blocks = 2 * c->num_cu;
    if (c->gram_blocks > ASMC_MAX_BLOCKS) c->gram_blocks = ASMC_MAX_BLOCKS;
    hipError_t e = hipSuccess;
    auto dmalloc = [&](void** p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
    };
    dmalloc((void**)&c->d_partials, sizeof(double) * ASMC_MAX_BLOCKS * ASMC_MAX_BETAS * 2);
    dmalloc((void**)&c->d_small, sizeof(double) * 4096);
    dmalloc((void**)&c->d_keys, sizeof(unsigned long long) * (ASMC_MAX_BETAS + 8));
    dmalloc((void**)&c->d_tiles, sizeof(double) * (size_t)(c->n_tiles_max * 4 + 64));
    dmalloc((void**)&c->d_tiles_i, sizeof(long long) * (size_t)(c->n_tiles_max * 4 + 64));
    dmalloc((void**)&c->d_gram, sizeof(double) * (size_t)c->gram_blocks * d_max * d_max);
    dmalloc((void**)&c->d_counts, sizeof(long long) * (size_t)(ASMC_MAX_PCN_STEPS + ASMC_MAX_BLOCKS + n_max / 256 + 8));
    dmalloc((void**)&c->d_rho, sizeof(double) * (ASMC_MAX_PCN_STEPS + 8));
    dmalloc((void**)&c->d_pcgtab, sizeof(unsigned long long) * (64 * 4 + 8));
    dmalloc((void**)&c->d_ptab, sizeof(double) * (2 * 32 * 32 + 32 + 3 * ASMC_MAX_COMPONENTS * (1 + 2 * 32) + 64));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_pinned, sizeof(double) * 8192, hipHostMallocDefault);
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
    (void)hipFree(c->d_counts);
    (void)hipFree(c->d_rho);
    (void)hipFree(c->d_pcgtab);
    (void)hipFree(c->d_ptab);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    delete c;
    return ASMC_OK;
}

}  // extern "C"
