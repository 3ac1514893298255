// Shared device helpers for libpcacc_hip.so (gfx950 only: wave64, 256 CUs, 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcacc.h"

#define PCACC_WAVE 64
#define PCACC_CUS 256

#define PCACC_CHECK_LAUNCH()                                   \
    do {                                                       \
        if (hipGetLastError() != hipSuccess) return PCACC_E_LAUNCH; \
    } while (0)

static inline hipStream_t pcacc_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Grid for a grid-stride, HBM-bound kernel: enough workgroups to fill 256 CUs x 8 waves, never more
// than the work (cdna_hip_programming.md guideline 11).
static inline int pcacc_grid(int64_t work_items, int block, int max_blocks = PCACC_CUS * 8)
{
    int64_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

static inline size_t pcacc_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- bf16 <-> f32 (round to nearest even, the conversion torch uses) --------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ---- wave64 / block scans ------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}

// Exclusive prefix sum of `v` over a 256-thread block (4 waves); *total = block sum.
// `lds` must hold 4 ints.  Contains two __syncthreads().
__device__ __forceinline__ int block256_exclusive_scan(int v, int *lds, int *total)
{
    const int incl = wave_inclusive_scan(v);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 63) lds[w] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += (i < w) ? lds[i] : 0;
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + incl - v;
}
