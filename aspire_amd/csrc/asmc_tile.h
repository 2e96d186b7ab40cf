// asmc_tile.h — 64-row LDS tiles: coalesced global <-> LDS copies of row-major particle rows and per-lane row access.
// Shared by the pCN / density kernels (asmc_pcn.hip) and the preconditioning transforms (asmc_transform.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__host__ __device__ __forceinline__ int lds_row_stride(int rowbytes) {
    // +16 B breaks the power-of-two stride (bank-conflict free per-lane row reads); rows whose
    // byte length is not a multiple of 16 are padded up to the next multiple of 8 first
    return ((rowbytes + 7) & ~7) + 16;
}

// coalesced copy of a 64-row tile global -> LDS (VEC = bytes per lane per access: 16, 8 or 4)
template <int VEC>
__device__ __forceinline__ void tile_load(const char* __restrict__ g, int64_t valid_bytes, int rowbytes,
                                          int ldsrow, char* lds, int lane) {
    const int tile_bytes = 64 * rowbytes;
    for (int off = lane * VEC; off < tile_bytes; off += 64 * VEC) {
        const int r = off / rowbytes, c = off - r * rowbytes;
        if (VEC == 16) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (off < valid_bytes) v = *reinterpret_cast<const uint4*>(g + off);
            *reinterpret_cast<uint4*>(lds + r * ldsrow + c) = v;
        } else if (VEC == 8) {
            unsigned long long v = 0;
            if (off < valid_bytes) v = *reinterpret_cast<const unsigned long long*>(g + off);
            *reinterpret_cast<unsigned long long*>(lds + r * ldsrow + c) = v;
        } else {
            uint32_t v = 0;
            if (off < valid_bytes) v = *reinterpret_cast<const uint32_t*>(g + off);
            *reinterpret_cast<uint32_t*>(lds + r * ldsrow + c) = v;
        }
    }
}

template <int VEC>
__device__ __forceinline__ void tile_store(char* __restrict__ g, int64_t valid_bytes, int rowbytes,
                                           int ldsrow, const char* lds, int lane) {
    const int tile_bytes = 64 * rowbytes;
    for (int off = lane * VEC; off < tile_bytes; off += 64 * VEC) {
        if (off >= valid_bytes) break;
        const int r = off / rowbytes, c = off - r * rowbytes;
        if (VEC == 16)
            *reinterpret_cast<uint4*>(g + off) = *reinterpret_cast<const uint4*>(lds + r * ldsrow + c);
        else if (VEC == 8)
            *reinterpret_cast<unsigned long long*>(g + off) =
                *reinterpret_cast<const unsigned long long*>(lds + r * ldsrow + c);
        else
            *reinterpret_cast<uint32_t*>(g + off) = *reinterpret_cast<const uint32_t*>(lds + r * ldsrow + c);
    }
}

// coalesced LDS -> global copy of the rows selected by `rowmask` (bit r = row r of the tile): rows whose
// proposal was rejected are simply not written, so a step writes acc_rate * d * s bytes per particle
template <int VEC>
__device__ __forceinline__ void tile_store_rows(char* __restrict__ g, int64_t valid_bytes, int rowbytes, int ldsrow,
                                                const char* lds, int lane, unsigned long long rowmask) {
    const int tile_bytes = 64 * rowbytes;
    for (int off = lane * VEC; off < tile_bytes; off += 64 * VEC) {
        if (off >= valid_bytes) break;
        const int r = off / rowbytes, c = off - r * rowbytes;
        if (!((rowmask >> r) & 1ULL)) continue;
        if (VEC == 16)
            *reinterpret_cast<uint4*>(g + off) = *reinterpret_cast<const uint4*>(lds + r * ldsrow + c);
        else if (VEC == 8)
            *reinterpret_cast<unsigned long long*>(g + off) =
                *reinterpret_cast<const unsigned long long*>(lds + r * ldsrow + c);
        else
            *reinterpret_cast<uint32_t*>(g + off) = *reinterpret_cast<const uint32_t*>(lds + r * ldsrow + c);
    }
}

template <typename T>
__device__ __forceinline__ double row_get(const char* row, int j) {
    return (double)reinterpret_cast<const T*>(row)[j];
}
template <typename T>
__device__ __forceinline__ void row_set(char* row, int j, double v) {
    reinterpret_cast<T*>(row)[j] = (T)v;
}

