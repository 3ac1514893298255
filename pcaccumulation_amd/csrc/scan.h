// Device-wide chunked scans used by voxelisation, the point->pillar CSR and the per-frame pillar lists.
// Pattern: (1) a counting kernel writes one partial sum per 2048-item chunk, (2) scan_chunk_sums turns
// the partials into exclusive chunk offsets in one 1024-thread workgroup, (3) an assigning kernel
// redoes the chunk-local scan and adds its chunk offset.  Items of a chunk are visited as 8 rows of
// 256 consecutive items, so loads stay coalesced and ranks follow item order.
#pragma once
#include "common.h"

#define PCACC_CHUNK 2048
#define PCACC_CHUNK_ROWS 8

static inline int pcacc_chunks(int64_t n) { return (int)((n + PCACC_CHUNK - 1) / PCACC_CHUNK); }

// In-place exclusive scan of sums[0..n_chunks) by ONE workgroup of 1024 threads; total -> *total_out
// (optionally clamped to `cap` when cap >= 0).
static __global__ __launch_bounds__(1024) void scan_chunk_sums(int *sums, int n_chunks, int *total_out, int cap)
{
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_chunks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < n_chunks) ? sums[i] : 0;
        const int incl = wave_inclusive_scan(v);
        const int w = threadIdx.x >> 6;
        if (lane_id() == 63) wave_tot[w] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            wbase += (k < w) ? wave_tot[k] : 0;
            tot += wave_tot[k];
        }
        const int carry = carry_s;
        if (i < n_chunks) sums[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) {
        int t = carry_s;
        if (cap >= 0 && t > cap) t = cap;
        *total_out = t;
    }
}

// Chunk partial sums of a plain int array.
static __global__ __launch_bounds__(256) void chunk_sums_i32(const int *in, int64_t n, int *sums)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * PCACC_CHUNK;
    int acc = 0;
#pragma unroll
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        acc += (i < n) ? in[i] : 0;
    }
    int tot;
    block256_exclusive_scan(acc, lds, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// out[i] = exclusive prefix sum of in[0..i); out[n] = total when write_total.
// clear_in: the input is zeroed behind the scan (a histogram that becomes the cursor array of the fill pass: saves a memset launch)
static __global__ __launch_bounds__(256) void chunk_scan_i32(const int *in, int64_t n, const int *chunk_offsets,
                                                      int *out, int write_total, int *clear_in = nullptr)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * PCACC_CHUNK;
    int carry = chunk_offsets[blockIdx.x];
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        const int v = (i < n) ? in[i] : 0;
        int tot;
        const int excl = block256_exclusive_scan(v, lds, &tot);
        if (i < n) out[i] = carry + excl;
        if (clear_in && i < n) clear_in[i] = 0;
        if (write_total && i == n - 1) out[n] = carry + excl + v;
        carry += tot;
    }
}
