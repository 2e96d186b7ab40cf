// asmc_common.h — internals shared by the HIP translation units of libasmc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/asmc.h"

#define ASMC_WAVE 64
#define ASMC_BLOCK 256          // 4 waves: one per SIMD of a CU
#define ASMC_MAX_BLOCKS 2048    // cap for grid-stride reduction kernels (256 CUs x 8)
#define ASMC_SCAN_TILE 2048     // elements per scan tile (256 threads x 8)
#define ASMC_PCN_MAX_GRID (1 << 20)  // blocks per pCN launch (>= 64 particles each): up to 67M particles per rank
#define ASMC_GAMMA_BATCH 8       // Markov steps whose tpCN scale variates one k_gamma_draw launch draws (ctx->d_gamma holds that many)
#define ASMC_MAX_PCN_STEPS 2048 // per asmc_pcn_mutate call (bounded by the pinned staging buffer)
// Box-Muller tables of the default noise (asmc_pcn_dev.h bm_pair32): entries of two doubles
#define BM_SC_N 256                   // [0, 256): (sin, cos)(2 pi (k + 1/2) / 256)
#define BM_LG_N 128                   // [256, 384): (rc_i, 2 ln rc_i)
#define BM_TAB_N (BM_SC_N + BM_LG_N)

void asmc_set_error(const char* fmt, ...);
void asmc_bm_table_host(double* tab);  // asmc_ctx.hip: the 2 * BM_TAB_N doubles (long double libm)
struct asmc_ctx;
int asmc_count_nonfinite_enqueue(asmc_ctx* ctx, int64_t n, const double* v, hipStream_t st);
unsigned long long* asmc_count_slot(asmc_ctx* ctx);  // where the last asmc_count_nonfinite_enqueue leaves {NaN, inf} counts
struct asmc_ctx;
void asmc_prof_begin(asmc_ctx* ctx, const char* label, hipStream_t st);
void asmc_poison_lds(asmc_ctx* ctx, hipStream_t st);  // diagnostic (ASMC_POISON_LDS): every CU's LDS filled with 0xFF bytes
void asmc_prof_end(asmc_ctx* ctx, hipStream_t st);
#define ASMC_PROF_MAX 8192
// every kernel goes through this macro so that bench.py can time individual kernels with HIP events recorded on
// the stream the kernel is launched on
#define ASMC_LAUNCH(ctx, st, label, ...)   \
    do {                                   \
        (ctx)->rec_n = 0; /* any launch may rewrite the arrays ctx->d_rec was packed from */ \
        (ctx)->cs_n = 0;  /* ... or the scratch that holds a gather's column-sum partials */ \
        if ((ctx)->poison_lds) asmc_poison_lds((ctx), (st)); \
        asmc_prof_begin((ctx), (label), (st)); \
        hipLaunchKernelGGL(__VA_ARGS__);   \
        asmc_prof_end((ctx), (st));        \
    } while (0)

#define ASMC_HIP(call)                                                                      \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess) {                                                             \
            asmc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                       \
            return ASMC_ERR_HIP;                                                            \
        }                                                                                   \
    } while (0)

#define ASMC_REQUIRE(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) {                                            \
            asmc_set_error("%s: %s", __func__, msg);              \
            return ASMC_ERR_ARG;                                  \
        }                                                         \
    } while (0)

#define ASMC_LAUNCH_CHECK()                                                                 \
    do {                                                                                    \
        hipError_t _e = hipGetLastError();                                                  \
        if (_e != hipSuccess) {                                                             \
            asmc_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),       \
                           __FILE__, __LINE__);                                             \
            return ASMC_ERR_HIP;                                                            \
        }                                                                                   \
    } while (0)

// fused flow step's counters (2 * ASMC_MAX_PCN_STEPS + 4 words), rounded to 256 bytes: ctx->d_flags starts right behind them
#define ASMC_TILECTR_BYTES ((sizeof(unsigned int) * (2 * ASMC_MAX_PCN_STEPS + 4) + 255) / 256 * 256)
struct asmc_ctx {
    int device;
    int64_t n_max;
    int d_max;
    int num_cu;
    // device scratch (all allocated in asmc_ctx_create)
    double* d_partials;            // [ASMC_MAX_BLOCKS * ASMC_MAX_BETAS * 2] block partial sums
    double* d_small;               // [4096] small result vectors
    unsigned long long* d_keys;    // [ASMC_MAX_BETAS + 8] atomicMax keys + integer counters
    double* d_tiles;               // [n_tiles_max * 4 + 64] scan tile aggregates
    long long* d_tiles_i;          // [n_tiles_max * 8 + 64] integer tile records (exact cdf: info + split; compaction)
    double* d_gram;                // [gram_blocks * d_max * d_max] gram partials
    unsigned int* d_guide;         // [n_max + 8] guide table of the resampling search (importance step: one bucket per cdf entry; asmc_search: one per four)
    unsigned char* d_flags;        // [n_max + 64] accept flags of the split-path pCN step (inside d_tilectr's allocation, behind the counters)
    double* d_gamma;               // [ASMC_GAMMA_BATCH][n_max] tpCN scale variates of the current steps
    // which steps' variates ctx->d_gamma holds (pcn_prepare_gamma)
    int64_t gam_n;
    unsigned long long gam_seed, gam_gid0;
    double gam_shape;
    int gam_noise, gam_count;
    uint32_t gam_step0;
    double* d_rec;                 // [4 * n_max] (ll, lp, lq, 0) records of asmc_gather's source population
    const void* rec_src[3];        // the arrays asmc_importance_step packed into d_rec (k_is_weights writes the records on
    int64_t rec_n;                 //   its way); rec_n != 0: still valid - the next asmc_gather of exactly these skips its packing pass
    unsigned count_gen;            // k_count_nonfinite's count slots are used in turn (asmc_weights.hip)
    uint64_t rec_token;            // generation of d_rec's contents (every pass that writes d_rec bumps it) ...
    const void* rec_hold_src[3];   // ... and the arrays / count / generation of a pack that asmc_normalized_weights_shard did on its
    int64_t rec_hold_n;            //   way: asmc_rec_claim(token) re-validates it for the next asmc_gather when nothing has
    uint64_t rec_hold_token;       //   rewritten d_rec since (the launches in between - the chain's, the search's - do not touch it)
    const void* cs_x;              // the fp64 rows asmc_gather wrote last, whose column-sum partials sit in d_gram [cs_grid][cs_d]
    int64_t cs_n;                  //   (cs_n != 0: still there - an asmc_mean_gram_enqueue of exactly these rows skips its k_colsum pass)
    int cs_d, cs_grid;
    void* d_ysoa;                  // coordinate-major whitened state of a mutation (grown on demand)
    void* d_xpad;                  // zero-padded tables + rows of a mutation whose d has no kernels of its own (grown on demand)
    size_t xpad_bytes;
    int d_max_pad;                 // widest padded problem the scratch of this ctx carries (next supported width >= d_max)
    double* d_student;             // [d_max (d_max + 1) + (ASMC_STUDENT_MAX_ROWS / 64) (d_max + 2)] tpCN fit: tables, partials
    double* h_student;             // pinned staging of the same size
    size_t ysoa_bytes;
    long long* d_counts;           // [ASMC_MAX_PCN_STEPS + max(ASMC_MAX_BLOCKS, n_max/64+1)] accept counts / partials
    double* d_rho;                 // [ASMC_MAX_PCN_STEPS + 8] step-size history on device
    unsigned int* d_tilectr;       // [2 * ASMC_MAX_PCN_STEPS + 2] fused flow-proposal step: tile hand-out counters, blocks-done counters (one each per step), non-finite density count (64-bit)
    unsigned int* d_bar;           // [1024 * 17] arrival counters + the poison cell of the persistent importance-weight kernel's grid barriers (4 KB apart; they only grow)
    unsigned int bar_base[17];     // their values when the next launch starts (top, groups)
    unsigned long long flow_nonfinite;  // non-finite flow densities among the proposals of the last asmc_pcn_mutate_flow
    int poison_lds;                // ASMC_POISON_LDS: a kernel that fills the LDS of every CU with NaN patterns runs in front of EVERY launch
                                   //   (a read of LDS that the kernel itself has not written then shows in the results, run after run)
    int isw_disabled;              // a launch was not fully resident once (barrier time-out): the step-by-step path from now on
    unsigned long long* d_pcgtab;  // [64*4 + 8] PCG64 jump table
    long long* d_select;           // [2 * ASMC_SELECT_THREADS/64 + 8] wave counts, offsets, total of asmc_pcg64_select
    unsigned long long ptab_tag, ysplit_seq;  // who packed d_ptab last (0 = anyone; else the split session's number)
    double* d_ptab;                // [2*32*32 + 32 + 3*8*(1+2*32)] packed pCN parameter block (d <= 32)
    double* d_bmtab;               // [2 * 384] Box-Muller tables of the default noise (asmc_pcn_dev.h bm_pair32)
    double* d_mmtab;               // [2 * 144 * 64] MFMA operand images of L and Linv (d = 64 / 128; NULL when d_max < 64)
    double* d_f16tab;              // resident tables of the flow-proposal step at d = 64 / 128 (asmc_flow16.hip; allocated on first use)
    size_t f16tab_bytes;
    // sharded mutation: accept-count exchange between a step and its adaptation (asmc_pcn_set_count_hook)
    int (*count_hook)(void*, asmc_stream);
    void* count_hook_user;
    long long* count_cell;
    int count_cells;               // cells behind count_cell that one exchange sums (1; a lagged adaptation uses one per step of a block)
    int64_t count_n_global;
    int mutate_defer, mutate_pending_steps;  // asmc_pcn_mutate_flow_enqueue / _result
    hipEvent_t ev_mutate;                    // recorded behind a deferred mutation's read-back
    hipEvent_t ev_is;                        // ... behind asmc_importance_result_enqueue's copies
    int is_result_pending;
    unsigned long long lq_nan;  // NaNs in the carried log q after the last mutation call (asmc_pcn_lq_nan)
    void* rccl_allreduce;  // asmc_pcn_set_count_rccl: the process's ncclAllReduce and a communicator
    void* rccl_comm;
    void* rccl_allgather;  // the process's ncclAllGather (asmc_set_rccl_allgather)
    double* shard_st;              // the search state the one-chain sharded step's passes read (asmc_weights_m2_lse_shard sets it)
    int64_t n_tiles_max;
    int gram_blocks;
    size_t gram_cap;  // doubles in d_gram (>= gram_blocks * d_max^2; the d = 32 matrix-core Gram kernel uses up to 2048 partials)
    unsigned long long pcg_inc[2];  // increment the device jump table was built for
    int pcg_tab_valid;
    // optional per-kernel HIP-event timing (asmc_profile_enable / asmc_profile_report)
    int prof_on;
    int prof_n;
    hipEvent_t* prof_ev;        // [2 * ASMC_PROF_MAX]
    const char** prof_label;    // [ASMC_PROF_MAX]
    // pinned host staging for scalar read-back / small uploads
    double* h_pinned;  // [8192] doubles
    unsigned ref_status_gen;  // asmc_reference_factor: generation of the pinned status cell in use (asmc_pcn.hip)
    double* h_gram;    // [128 + 128 * 128] doubles: asmc_mean_gram's results (sum | Gram) until asmc_mean_gram_fetch
    double* d_ref;     // the same on the device (d_small / d_partials are every other call's scratch): asmc_reference_factor
    int gram_pending_d;  // d of an enqueued, not yet fetched asmc_mean_gram (0: none)
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: the "already raised" guards of the launchers
// are kept per device (a process with contexts on two devices must raise it on both)
#define ASMC_MAX_DEVICES 64
static inline int asmc_dev_slot(const asmc_ctx* ctx) { return ctx->device & (ASMC_MAX_DEVICES - 1); }


// asmc_flow16.hip: flows of more than 32 dimensions (packed layout 1: 16-particle groups, streamed weights)
extern "C" int asmc_flow_layout(int kind, int dims, int hidden);
int64_t asmc_flow16_pack_floats(int kind, int dims, int n_layers, int hidden);
int asmc_flow16_pack(int kind, int dims, int n_layers, int hidden, const float* const* weights_host, const float* const* biases_host,
                     float* packed_host);
int asmc_flow16_logprob(asmc_ctx* ctx, int64_t n, int x_dtype, const void* x, const asmc_coupling* f, double* out, hipStream_t st);
int asmc_flow16_sample(asmc_ctx* ctx, int64_t n, int x_dtype, const asmc_coupling* f, unsigned long long seed, unsigned long long gid0,
                       uint32_t draw_id, void* x_out, double* lq_out, hipStream_t st);
bool asmc_pcn_flow16_ok(const asmc_pcn_params* prm, const asmc_coupling* f);
bool asmc_flow_math_split();

// asmc_weights.hip: the persistent weight kernel of asmc_importance_step (results in ctx->d_small + 2560 .. + 48)
// k_ref_factor on the stream (asmc_pcn.hip; asmc_reference_factor and the Student-t EM of asmc_student.hip)
int asmc_ref_factor_launch(asmc_ctx* ctx, int d, const double* sum, const double* gram, double n_mean, double denom, double* out,
                           double* status, double* tab, double* em, int it, hipStream_t st);
int asmc_gram_mm_launch(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const double* d_center, int* grid_out,
                        hipStream_t st, double* out2, double n_div);
bool asmc_gram_mm_supported(int d, const void* x);
int asmc_is_weights_launch(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq, double beta0,
                           double target_eff, double tol, double* w, double* tiles, double* rec, hipStream_t st);

static inline hipStream_t as_stream(asmc_stream s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// order-preserving map double -> u64 (for atomicMax on doubles; NaNs are filtered by callers)
__device__ __forceinline__ unsigned long long f64_to_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ __forceinline__ double key_to_f64(unsigned long long k) {
    unsigned long long b = (k & 0x8000000000000000ULL) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
    double d;
#if defined(__HIP_DEVICE_COMPILE__)
    d = __longlong_as_double((long long)b);
#else
    memcpy(&d, &b, sizeof(d));
#endif
    return d;
}

static inline int grid_for(int64_t n, int per_block, int cap) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
