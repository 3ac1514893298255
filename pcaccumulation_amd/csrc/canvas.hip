// Pillar <-> BEV canvas movement in channels-last layout.
//
// The reference builds the canvas per batch element with `canvas[:, indices] = voxels.t()` on an NCHW
// tensor (models/pillar_encoder.py:144-170): a zero fill of the whole canvas followed by M writes of C
// values each at a stride of nt*ny*nx floats.  Here the canvas is [cell, channel]; a dense inverse table
// cell -> pillar id (built once per forward from the collated coordinates) turns the scatter into ONE
// streaming pass that writes every canvas byte exactly once, in 16-byte pieces, 1 KiB per wave store:
//   HBM traffic = canvas bytes (write) + 4 B/cell (table) + one feature row per occupied cell (read)
// which is the algorithmic minimum (SURVEY.md 8d "pillar scatter").
#include <hip/hip_ext.h>

#include "scan.h"

// ---- coordinates [m,5] = (b,z,y,x,t) -> linear cell index + inverse table ----------------------------
template <typename T>
__global__ __launch_bounds__(256) void cell_index_kernel(const T *__restrict__ coords, int64_t m, int nx, int ny, int nt,
                                                         int64_t n_cells, int32_t *__restrict__ cell, int32_t *cell2pillar)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        const int64_t b = (int64_t)coords[i * 5 + 0];
        const int64_t y = (int64_t)coords[i * 5 + 2];
        const int64_t x = (int64_t)coords[i * 5 + 3];
        const int64_t t = (int64_t)coords[i * 5 + 4];
        // pillar_encoder.py:158: t*nx*ny + y*nx + x inside sample b
        const int64_t c = ((b * nt + t) * ny + y) * nx + x;
        const bool ok = c >= 0 && c < n_cells && x >= 0 && x < nx && y >= 0 && y < ny && t >= 0 && t < nt;
        cell[i] = ok ? (int32_t)c : -1;
        if (ok) atomicMax(&cell2pillar[c], (int32_t)i);
    }
}

extern "C" int pcacc_cell_index(const void *coords, int coords_is_f64, int64_t m, int nx, int ny, int nt, int n_batch,
                                int32_t *cell, int32_t *cell2pillar, void *stream)
{
    if (m < 0 || nx <= 0 || ny <= 0 || nt <= 0 || n_batch <= 0 || !cell2pillar) return PCACC_E_ARG;
    const int64_t n_cells = (int64_t)n_batch * nt * ny * nx;
    if (n_cells >= 0x7fffffffLL || m >= 0x7fffffffLL) return PCACC_E_ARG;
    if (m > 0 && (!coords || !cell)) return PCACC_E_ARG;
    hipStream_t s = pcacc_stream(stream);
    if (hipMemsetAsync(cell2pillar, 0xFF, (size_t)n_cells * 4, s) != hipSuccess) return PCACC_E_LAUNCH;   // -1
    if (m == 0) return PCACC_OK;
    if (coords_is_f64)
        cell_index_kernel<double><<<pcacc_grid(m, 256), 256, 0, s>>>(static_cast<const double *>(coords), m, nx, ny, nt,
                                                                   n_cells, cell, cell2pillar);
    else
        cell_index_kernel<int32_t><<<pcacc_grid(m, 256), 256, 0, s>>>(static_cast<const int32_t *>(coords), m, nx, ny, nt,
                                                                    n_cells, cell, cell2pillar);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- occupied pillars per frame in ascending cell order (stream compaction of cell2pillar) -------------
__global__ __launch_bounds__(256) void fp_count(const int32_t *__restrict__ c2p, int64_t n_cells, int *chunk_sums)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * PCACC_CHUNK;
    int acc = 0;
#pragma unroll
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        acc += __popcll(__ballot(i < n_cells && c2p[i] >= 0));
    }
    if (lane_id() == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(256) void fp_assign(const int32_t *__restrict__ c2p, int64_t n_cells, int64_t cells_per_frame,
                                                 const int *chunk_offsets, int32_t *sorted_pillars, int32_t *frame_offsets)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * PCACC_CHUNK;
    int carry = chunk_offsets[blockIdx.x];
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        const int p = (i < n_cells) ? c2p[i] : -1;
        const int f = p >= 0;
        int tot;
        const int rank = carry + block256_exclusive_scan(f, lds, &tot);
        carry += tot;
        if (f) sorted_pillars[rank] = p;
        if (i < n_cells && (i % cells_per_frame) == 0) frame_offsets[i / cells_per_frame] = rank;
        if (i == n_cells - 1) frame_offsets[n_cells / cells_per_frame] = rank + f;
    }
}

// ---- indices of the non-zero entries of a byte mask (include/pcacc.h: pcacc_compact_mask) ---------------------------------------------------
// 16 mask bytes per lane and load (one uint4): a 2048-entry chunk is half a wave-load; workgroups of 256 threads take 4096 entries = 2 chunks
// of the scan helpers' size -- here a chunk is 4096 entries (own constants: the helpers of scan.h walk 4-byte items).
#define CM_CHUNK 4096
__device__ __forceinline__ uint32_t cm_nonzero_bits(const uint8_t *mask, int64_t i0, int64_t n)      // bit b = mask[i0 + b] != 0, 16 entries
{
    uint32_t bits = 0;
    if (i0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(mask + i0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) bits |= ((w[k] >> (8 * b)) & 0xffu) ? (1u << (4 * k + b)) : 0u;
    } else {
        for (int b = 0; b < 16 && i0 + b < n; ++b) bits |= mask[i0 + b] ? (1u << b) : 0u;
    }
    return bits;
}

__global__ __launch_bounds__(256) void cm_count(const uint8_t *__restrict__ mask, int64_t n, int *chunk_sums)
{
    __shared__ int lds[4];
    const int64_t i0 = (int64_t)blockIdx.x * CM_CHUNK + threadIdx.x * 16;
    int acc = i0 < n ? __popc(cm_nonzero_bits(mask, i0, n)) : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(256) void cm_assign(const uint8_t *__restrict__ mask, int64_t n, const int *__restrict__ chunk_offsets,
                                                 int64_t *__restrict__ indices, int64_t capacity, const int *__restrict__ total)
{
    __shared__ int lds[4];
    // [r6] entries behind the count hold -1, as torch.nonzero_static's tail does: a `capacity` above the true count (a stale host count) then gives a
    // deterministic failure downstream instead of gathers through uninitialised int64 values.  No iteration when capacity == count.
    for (int64_t j = (int64_t)*total + (int64_t)blockIdx.x * 256 + threadIdx.x; j < capacity; j += (int64_t)gridDim.x * 256) indices[j] = -1;
    const int64_t i0 = (int64_t)blockIdx.x * CM_CHUNK + threadIdx.x * 16;
    const uint32_t bits = i0 < n ? cm_nonzero_bits(mask, i0, n) : 0u;
    int tot;
    int64_t rank = chunk_offsets[blockIdx.x] + block256_exclusive_scan(__popc(bits), lds, &tot);
    uint32_t b = bits;
    while (b) {
        const int k = __ffs(b) - 1;
        b &= b - 1;
        if (rank < capacity) indices[rank] = i0 + k;
        ++rank;
    }
}

extern "C" int pcacc_compact_mask_workspace_bytes(int64_t n, size_t *bytes)
{
    if (!bytes || n < 0) return PCACC_E_ARG;
    *bytes = pcacc_align((size_t)((n + CM_CHUNK - 1) / CM_CHUNK + 1) * 4);
    return PCACC_OK;
}

extern "C" int pcacc_compact_mask(const uint8_t *mask, int64_t n, int64_t *indices, int64_t capacity, int32_t *count_out, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    size_t need;
    if (pcacc_compact_mask_workspace_bytes(n, &need) != PCACC_OK || capacity < 0 || n >= 0x7fffffffLL * 16) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (!mask || (capacity > 0 && !indices) || (reinterpret_cast<uintptr_t>(mask) % 16)) return PCACC_E_ARG;
    if (!workspace || workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    int *sums = static_cast<int *>(workspace);
    const int chunks = (int)((n + CM_CHUNK - 1) / CM_CHUNK);
    cm_count<<<chunks, 256, 0, s>>>(mask, n, sums);
    int *total = count_out ? count_out : sums + chunks;         // the workspace holds chunks + 1 words
    scan_chunk_sums<<<1, 1024, 0, s>>>(sums, chunks, total, -1);
    cm_assign<<<chunks, 256, 0, s>>>(mask, n, sums, indices, capacity, total);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_frame_pillars_workspace_bytes(int64_t n_cells, size_t *bytes)
{
    if (!bytes || n_cells < 0) return PCACC_E_ARG;
    *bytes = pcacc_align((size_t)(pcacc_chunks(n_cells) + 1) * 4);
    return PCACC_OK;
}

extern "C" int pcacc_frame_pillars(const int32_t *cell2pillar, int64_t n_cells, int64_t cells_per_frame,
                                   int32_t *sorted_pillars, int32_t *frame_offsets,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    size_t need;
    if (pcacc_frame_pillars_workspace_bytes(n_cells, &need) != PCACC_OK) return PCACC_E_ARG;
    if (n_cells <= 0 || cells_per_frame <= 0 || (n_cells % cells_per_frame) || !cell2pillar || !frame_offsets ||
        !sorted_pillars)
        return PCACC_E_ARG;
    if (!workspace || workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    int *sums = static_cast<int *>(workspace);
    const int chunks = pcacc_chunks(n_cells);
    fp_count<<<chunks, 256, 0, s>>>(cell2pillar, n_cells, sums);
    scan_chunk_sums<<<1, 1024, 0, s>>>(sums, chunks, nullptr, -1);
    fp_assign<<<chunks, 256, 0, s>>>(cell2pillar, n_cells, cells_per_frame, sums, sorted_pillars, frame_offsets);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- A5: the pillar-scatter kernel -----------------------------------------------------------------------
// One 16-byte piece of the canvas per lane per iteration.  For C=32 fp32 a cell is 8 pieces (128 B), so a
// wave covers 8 consecutive cells = 1 KiB contiguous; the 8 lanes of a cell read the same table word
// (one broadcast request) and one contiguous 128-byte feature row.  XCD note: block b runs on XCD b%8 and
// a grid-stride step moves all blocks forward together, so each XCD streams its own 1/8 interleave of the
// canvas with no reuse to lose -- no remap needed for a pure streaming pass.
template <int OUT_BF16>
__global__ __launch_bounds__(256) void pillar_scatter_vec4(const float4 *__restrict__ feats, const int32_t *__restrict__ c2p,
                                                           int64_t n_pieces, int ppc /*pieces per cell, f32 side*/,
                                                           void *__restrict__ canvas)
{
    if (OUT_BF16) {
        // one lane produces 8 bf16 channels = 16 B of output from two float4 of input
        const int opc = ppc / 2;                               // output pieces per cell
        uint4 *out = static_cast<uint4 *>(canvas);
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_pieces; e += (int64_t)gridDim.x * 256) {
            const int64_t cell = e / opc;
            const int sub = (int)(e - cell * opc);
            const int p = c2p[cell];
            uint4 o = make_uint4(0u, 0u, 0u, 0u);
            if (p >= 0) {
                const float4 a = feats[(int64_t)p * ppc + sub * 2];
                const float4 b = feats[(int64_t)p * ppc + sub * 2 + 1];
                o.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
                o.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
                o.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
                o.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
            }
            out[e] = o;
        }
    } else {
        float4 *out = static_cast<float4 *>(canvas);
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_pieces; e += (int64_t)gridDim.x * 256) {
            const int64_t cell = e / ppc;
            const int sub = (int)(e - cell * ppc);
            const int p = c2p[cell];
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p >= 0) o = feats[(int64_t)p * ppc + sub];
            out[e] = o;
        }
    }
}

// bf16 feature rows -> bf16 canvas (the bf16 compute mode: the pillar encoder's last pooling already emits bf16 rows, so the kernel
// moves C*2 B per occupied cell in and C*2 B per cell out -- SURVEY.md 8d's bytes with s = 2 on both sides, no fp32 [M,C] table in
// between).  Two 16-byte pieces per lane and iteration, table words first, then the two row pieces, then the two stores.
__global__ __launch_bounds__(256) void pillar_scatter_rows16(const uint4 *__restrict__ feats, const int32_t *__restrict__ c2p,
                                                             int64_t n_pieces, int ppc /*16-byte pieces per cell*/, uint4 *__restrict__ canvas)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < n_pieces; e0 += 2 * stride) {
        const int64_t e1 = e0 + stride;
        const bool two = e1 < n_pieces;
        const int64_t cell0 = e0 / ppc, cell1 = two ? e1 / ppc : cell0;
        const int p0 = c2p[cell0], p1 = two ? c2p[cell1] : -1;
        uint4 o0 = make_uint4(0u, 0u, 0u, 0u), o1 = o0;
        if (p0 >= 0) o0 = feats[(int64_t)p0 * ppc + (e0 - cell0 * ppc)];
        if (p1 >= 0) o1 = feats[(int64_t)p1 * ppc + (e1 - cell1 * ppc)];
        canvas[e0] = o0;
        if (two) canvas[e1] = o1;
    }
}

// The same fill with K 16-byte pieces per lane and iteration -- K table words, then K row pieces, then K stores in flight per lane -- and, with NT,
// the streaming cache policy on both sides (`nt` loads of the row table, `nt` stores of the canvas: every byte is touched once by this kernel).
typedef uint32_t pcacc_u32x4 __attribute__((ext_vector_type(4)));
template <int K, bool NT, bool NTL = NT>
__global__ __launch_bounds__(256) void pillar_scatter_rows16_k(const pcacc_u32x4 *__restrict__ feats, const int32_t *__restrict__ c2p,
                                                               int64_t n_pieces, int ppc /*16-byte pieces per cell*/, pcacc_u32x4 *__restrict__ canvas)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < n_pieces; e0 += K * stride) {
        int p[K];
        int64_t src[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t e = e0 + k * stride;
            const bool in = e < n_pieces;
            const int64_t cell = in ? e / ppc : 0;
            p[k] = in ? (NTL ? __builtin_nontemporal_load(c2p + cell) : c2p[cell]) : -1;
            src[k] = e - cell * ppc;
        }
        pcacc_u32x4 o[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            o[k] = (pcacc_u32x4){0u, 0u, 0u, 0u};
            if (p[k] >= 0) o[k] = NTL ? __builtin_nontemporal_load(feats + (int64_t)p[k] * ppc + src[k]) : feats[(int64_t)p[k] * ppc + src[k]];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t e = e0 + k * stride;
            if (e < n_pieces) {
                if (NT) __builtin_nontemporal_store(o[k], canvas + e);
                else canvas[e] = o[k];
            }
        }
    }
}

// fp32 rows -> fp32 canvas with the streaming policy on both sides (the fp32 twin canvas of the 'mixed' / fp32x3 modes)
typedef float pcacc_f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pillar_scatter_f32_nt(const pcacc_f32x4v *__restrict__ feats, const int32_t *__restrict__ c2p, int64_t n_pieces,
                                                             int ppc, pcacc_f32x4v *__restrict__ canvas)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_pieces; e += (int64_t)gridDim.x * 256) {
        const int64_t cell = e / ppc;
        const int p = __builtin_nontemporal_load(c2p + cell);
        pcacc_f32x4v o = {0.f, 0.f, 0.f, 0.f};
        if (p >= 0) o = __builtin_nontemporal_load(feats + (int64_t)p * ppc + (e - cell * ppc));
        __builtin_nontemporal_store(o, canvas + e);
    }
}

// narrow canvases (C = 1, 2, 3: occupancy, labels, pillar means) -- one element per lane
template <int OUT_BF16>
__global__ __launch_bounds__(256) void pillar_scatter_scalar(const float *__restrict__ feats, const int32_t *__restrict__ c2p,
                                                             int64_t n_elems, int c, void *__restrict__ canvas)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_elems; e += (int64_t)gridDim.x * 256) {
        const int64_t cell = e / c;
        const int k = (int)(e - cell * c);
        const int p = c2p[cell];
        const float v = (p >= 0) ? feats[(int64_t)p * c + k] : 0.f;
        if (OUT_BF16) static_cast<uint16_t *>(canvas)[e] = f32_to_bf16(v);
        else static_cast<float *>(canvas)[e] = v;
    }
}

// start / stop: optional hipEvents attached to the dispatch itself (hipExtLaunchKernelGGL): their elapsed time is the kernel's own
// begin-to-end interval, the quantity a kernel trace reports -- an event pair recorded around the launch also contains the
// queue's event-processing time (4-5 us on this stack), which is 10 % of this kernel.
static int pillar_scatter_launch(const void *feats, int feats_dtype, const int32_t *cell2pillar, int64_t n_cells, int c, void *canvas,
                                 int dtype, hipEvent_t start, hipEvent_t stop, hipStream_t s)
{
    if (n_cells < 0 || c <= 0 || !cell2pillar || !canvas || (dtype != PCACC_F32 && dtype != PCACC_BF16)) return PCACC_E_ARG;
    if (feats_dtype != PCACC_F32 && !(feats_dtype == PCACC_BF16 && dtype == PCACC_BF16 && c % 8 == 0)) return PCACC_E_ARG;
    if (n_cells == 0) return PCACC_OK;
    int64_t n;
    int width, per_lane = 1;
    const void *fn;
    if (feats_dtype == PCACC_BF16) {
        width = c / 8;
        n = n_cells * width;
        // [r5] default: one piece per lane, streaming (`nt`) loads and stores -- the canvas is written once and read much later, the row table is read
        // once: in the step 35.4 -> 32.7 us (0.66 -> 0.71 of 8 TB/s), behind 1 GiB of dirty lines 51 -> 34 us (0.46 -> 0.69), and the kernels that follow
        // find their operands still cached (step 31.15 -> 30.65 ms, two runs each; profiles/r05_scatter_variants.txt).  '0' = round 4's kernel.
        fn = reinterpret_cast<const void *>(pillar_scatter_rows16_k<1, true>);
        per_lane = 1;
        switch (pcacc_switches().scatter_variant) {               // A/B: PCACC_SCATTER_VARIANT
        case '0': fn = reinterpret_cast<const void *>(pillar_scatter_rows16); per_lane = 2; break;
        case '1': fn = reinterpret_cast<const void *>(pillar_scatter_rows16_k<4, true>); per_lane = 4; break;
        case '2': fn = reinterpret_cast<const void *>(pillar_scatter_rows16_k<2, true>); per_lane = 2; break;
        case '3': fn = reinterpret_cast<const void *>(pillar_scatter_rows16_k<4, false>); per_lane = 4; break;
        case '4': fn = reinterpret_cast<const void *>(pillar_scatter_rows16_k<8, true>); per_lane = 8; break;
        case '5': fn = reinterpret_cast<const void *>(pillar_scatter_rows16_k<1, true>); per_lane = 1; break;
        case '6': fn = reinterpret_cast<const void *>(pillar_scatter_rows16_k<1, true, false>); per_lane = 1; break;     // streaming stores, cached loads
        case '7': fn = reinterpret_cast<const void *>(pillar_scatter_rows16_k<2, true, false>); per_lane = 2; break;
        default: break;
        }
    } else if (dtype == PCACC_F32 ? (c % 4 == 0) : (c % 8 == 0)) {
        width = c / 4;
        n = dtype == PCACC_F32 ? n_cells * width : n_cells * (width / 2);
        fn = dtype == PCACC_F32 ? reinterpret_cast<const void *>(pillar_scatter_vec4<0>) : reinterpret_cast<const void *>(pillar_scatter_vec4<1>);
        // the fp32 canvas is read by the first convolution right behind this fill: cached stores.  The streaming variant (PCACC_SCATTER_VARIANT=f) left the
        // step where it was and slowed the bf16 fill that follows it in the 'mixed' mode from 32.7 to 41 us (profiles/r05_cache_policy_ab.txt)
        if (dtype == PCACC_F32 && pcacc_switches().scatter_variant == 'f')
            fn = reinterpret_cast<const void *>(pillar_scatter_f32_nt);
    } else {
        width = c;
        n = n_cells * c;
        fn = dtype == PCACC_F32 ? reinterpret_cast<const void *>(pillar_scatter_scalar<0>) : reinterpret_cast<const void *>(pillar_scatter_scalar<1>);
    }
    // all kernel families take (features, cell2pillar, n, width, canvas)
    void *args[] = {(void *)&feats, (void *)&cell2pillar, (void *)&n, (void *)&width, (void *)&canvas};
    const int64_t items = (n + per_lane - 1) / per_lane;
    const int blocks_per_cu = pcacc_switches().scatter_blocks > 0 ? pcacc_switches().scatter_blocks : 8;
    if (hipExtLaunchKernel(fn, dim3(pcacc_grid(items, 256, PCACC_CUS * blocks_per_cu)), dim3(256), args, 0, s, start, stop, 0) != hipSuccess) return PCACC_E_LAUNCH;
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_pillar_scatter(const float *feats, const int32_t *cell2pillar, int64_t n_cells, int c,
                                    void *canvas, int dtype, void *stream)
{
    return pillar_scatter_launch(feats, PCACC_F32, cell2pillar, n_cells, c, canvas, dtype, nullptr, nullptr, pcacc_stream(stream));
}

extern "C" int pcacc_pillar_scatter_timed(const float *feats, const int32_t *cell2pillar, int64_t n_cells, int c, void *canvas, int dtype,
                                          void *start_event, void *stop_event, void *stream)
{
    if (!start_event || !stop_event) return PCACC_E_ARG;
    return pillar_scatter_launch(feats, PCACC_F32, cell2pillar, n_cells, c, canvas, dtype, reinterpret_cast<hipEvent_t>(start_event),
                                 reinterpret_cast<hipEvent_t>(stop_event), pcacc_stream(stream));
}

extern "C" int pcacc_pillar_scatter_t(const void *feats, int feats_dtype, const int32_t *cell2pillar, int64_t n_cells, int c, void *canvas,
                                      int dtype, void *start_event, void *stop_event, void *stream)
{
    if ((start_event == nullptr) != (stop_event == nullptr)) return PCACC_E_ARG;
    return pillar_scatter_launch(feats, feats_dtype, cell2pillar, n_cells, c, canvas, dtype, reinterpret_cast<hipEvent_t>(start_event),
                                 reinterpret_cast<hipEvent_t>(stop_event), pcacc_stream(stream));
}

extern "C" int pcacc_timer_create(void **start_event, void **stop_event)
{
    if (!start_event || !stop_event) return PCACC_E_ARG;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return PCACC_E_LAUNCH;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return PCACC_E_LAUNCH; }
    *start_event = a;
    *stop_event = b;
    return PCACC_OK;
}

extern "C" int pcacc_timer_elapsed_us(void *start_event, void *stop_event, float *us)
{
    if (!start_event || !stop_event || !us) return PCACC_E_ARG;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, reinterpret_cast<hipEvent_t>(start_event), reinterpret_cast<hipEvent_t>(stop_event)) != hipSuccess)
        return PCACC_E_LAUNCH;
    *us = ms * 1e3f;
    return PCACC_OK;
}

extern "C" int pcacc_timer_destroy(void *start_event, void *stop_event)
{
    if (start_event) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(start_event));
    if (stop_event) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(stop_event));
    return PCACC_OK;
}

// ---- A6 / backward of A5: row gather -----------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t *__restrict__ src, int words, const int32_t *__restrict__ idx,
                                                          int64_t n_idx, uint32_t *__restrict__ out)
{
    const int64_t total = n_idx * words;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / words;
        const int k = (int)(e - r * words);
        const int64_t j = idx[r];
        out[e] = (j >= 0) ? src[j * words + k] : 0u;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel16(const uint4 *__restrict__ src, int pieces, const int32_t *__restrict__ idx,
                                                            int64_t n_idx, uint4 *__restrict__ out)
{
    const int64_t total = n_idx * pieces;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / pieces;
        const int k = (int)(e - r * pieces);
        const int64_t j = idx[r];
        out[e] = (j >= 0) ? src[j * pieces + k] : make_uint4(0u, 0u, 0u, 0u);
    }
}

extern "C" int pcacc_gather_rows(const void *src, int row_bytes, const int32_t *idx, int64_t n_idx, void *out, void *stream)
{
    if (n_idx < 0 || row_bytes <= 0 || (row_bytes % 4)) return PCACC_E_ARG;
    if (n_idx > 0 && (!src || !idx || !out)) return PCACC_E_ARG;
    if (n_idx == 0) return PCACC_OK;
    hipStream_t s = pcacc_stream(stream);
    if (row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0)) {
        const int pieces = row_bytes / 16;
        gather_rows_kernel16<<<pcacc_grid(n_idx * pieces, 256), 256, 0, s>>>(static_cast<const uint4 *>(src), pieces, idx, n_idx,
                                                                            static_cast<uint4 *>(out));
    } else {
        const int words = row_bytes / 4;
        gather_rows_kernel<<<pcacc_grid(n_idx * words, 256), 256, 0, s>>>(static_cast<const uint32_t *>(src), words, idx, n_idx,
                                                                         static_cast<uint32_t *>(out));
    }
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// Decoder concatenation torch.cat((from_up, from_down), 1) on channels-last rows (models/unet.py:101-113): out[r] = a[r] | b[r], 16 bytes per lane,
// two pieces in flight per lane.  (The library's batched-copy kernel moves the top-level fp32 concatenation of a 4-sequence step, 848 MB, in 170 us.)
__global__ __launch_bounds__(256) void cat2_rows_kernel(const uint4 *__restrict__ a, int pa, const uint4 *__restrict__ b, int pb, int64_t rows,
                                                        uint4 *__restrict__ out)
{
    const int pr = pa + pb;
    const int64_t total = rows * pr, stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += 2 * stride) {
        const int64_t e2 = e + stride;
        const int64_t r0 = e / pr, r1 = e2 / pr;
        const int c0 = (int)(e - r0 * pr), c1 = (int)(e2 - r1 * pr);
        const uint4 v0 = c0 < pa ? a[r0 * pa + c0] : b[r0 * pb + (c0 - pa)];
        uint4 v1 = make_uint4(0, 0, 0, 0);
        if (e2 < total) v1 = c1 < pa ? a[r1 * pa + c1] : b[r1 * pb + (c1 - pa)];
        out[e] = v0;
        if (e2 < total) out[e2] = v1;
    }
}

extern "C" int pcacc_cat2_rows(const void *a, int32_t a_row_bytes, const void *b, int32_t b_row_bytes, int64_t rows, void *out, void *stream)
{
    if (rows < 0 || a_row_bytes <= 0 || b_row_bytes <= 0 || (a_row_bytes % 16) || (b_row_bytes % 16)) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!a || !b || !out || (reinterpret_cast<uintptr_t>(a) % 16) || (reinterpret_cast<uintptr_t>(b) % 16) || (reinterpret_cast<uintptr_t>(out) % 16)) return PCACC_E_ARG;
    const int pa = a_row_bytes / 16, pb = b_row_bytes / 16;
    const int64_t total = rows * (pa + pb);
    cat2_rows_kernel<<<pcacc_grid((total + 1) / 2, 256, PCACC_CUS * 16), 256, 0, pcacc_stream(stream)>>>(static_cast<const uint4 *>(a), pa, static_cast<const uint4 *>(b), pb,
                                                                                                         rows, static_cast<uint4 *>(out));
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" const char *pcacc_target(void) { return "gfx950"; }

static PcaccSwitches g_switches;
static bool g_switches_read = false;

static void switches_read()
{
    const char *e = getenv("PCACC_CONV_RES");
    g_switches.conv_frame_major = getenv("PCACC_CONV_FRAME_MAJOR") != nullptr;
    g_switches.conv_swz_off = getenv("PCACC_CONV_SWZ_OFF") != nullptr;
    g_switches.rows_fm_off = getenv("PCACC_ROWS_FM_OFF") != nullptr;
    g_switches.conv_plan = getenv("PCACC_CONV_PLAN") != nullptr;
    g_switches.conv_res = e ? e[0] : 0;
    const char *xr = getenv("PCACC_XCD_REMAP");
    g_switches.xcd_off = xr && xr[0] == '0';
    const char *v = getenv("PCACC_SCATTER_VARIANT");
    g_switches.scatter_variant = v ? v[0] : 0;
    const char *b = getenv("PCACC_SCATTER_BLOCKS");
    g_switches.scatter_blocks = b ? atoi(b) : 0;
    g_switches_read = true;
}

const PcaccSwitches &pcacc_switches()
{
    if (!g_switches_read) switches_read();        // benign race: every thread writes the same values
    return g_switches;
}

extern "C" int pcacc_reload_switches(void)
{
    switches_read();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Small host arrays (per-sample counts, label offsets, thresholds: a few dozen words) to device memory as KERNEL ARGUMENTS.
// A hipMemcpy from pageable host memory on the compute stream waits for everything queued before it -- each
// `torch.tensor(list, device=...)` in the forward pass was a pipeline drain (20 per step).  A kernel launch carrying the
// words by value is asynchronous like any other launch.
// ---------------------------------------------------------------------------------------------------------------------
struct UploadWords { uint32_t w[240]; };

__global__ void upload_words_kernel(UploadWords words, int n, uint32_t *dst)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = words.w[threadIdx.x];
}

extern "C" int pcacc_upload_words(const uint32_t *host_words, int64_t n, uint32_t *dst, void *stream)
{
    if (n < 0 || (n > 0 && (!host_words || !dst))) return PCACC_E_ARG;
    for (int64_t off = 0; off < n; off += 240) {
        UploadWords words;
        const int m = (int)((n - off) < 240 ? (n - off) : 240);
        for (int i = 0; i < m; ++i) words.w[i] = host_words[off + i];
        upload_words_kernel<<<1, 256, 0, pcacc_stream(stream)>>>(words, m, dst + off);
    }
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// D1. Host data step in front of the path, on the device (libs/dataset.py:147-182, BaseDataset.prep_input before the
// voxeliser): augmentation of the raw points (rigid transform about z, uniform noise, global scale) fused with the crop /
// ground tests into one pass; the caller compacts the kept rows.  float64 like the numpy code it replaces:
//   p = (R p + t);  p += (u - 0.5) * noise;  p *= scale;
//   keep = |x| < crop_xy && |y| < crop_xy && z_min < z < z_max && (!remove_ground || z > ground_z)
// The 3-term products run as an FMA chain (the accumulation order of a K = 3 dgemm), everything else un-fused.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_points_kernel(const double *__restrict__ pts, const double *__restrict__ tsfm,
                                                          const double *__restrict__ noise, double noise_scale, double scale,
                                                          double crop_xy, double z_min, double z_max, int remove_ground, double ground_z,
                                                          int64_t m, double *__restrict__ out, uint8_t *__restrict__ keep)
{
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (tsfm) {
            const double nx = fma(tsfm[2], z, fma(tsfm[1], y, tsfm[0] * x)) + tsfm[9];
            const double ny = fma(tsfm[5], z, fma(tsfm[4], y, tsfm[3] * x)) + tsfm[10];
            const double nz = fma(tsfm[8], z, fma(tsfm[7], y, tsfm[6] * x)) + tsfm[11];
            x = nx; y = ny; z = nz;
        }
        if (noise) {
            x = x + (noise[3 * i] - 0.5) * noise_scale;
            y = y + (noise[3 * i + 1] - 0.5) * noise_scale;
            z = z + (noise[3 * i + 2] - 0.5) * noise_scale;
        }
        if (tsfm || noise) { x = x * scale; y = y * scale; z = z * scale; }
        out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z;
        bool k = fabs(x) < crop_xy && fabs(y) < crop_xy && z < z_max && z > z_min;
        if (remove_ground) k = k && z > ground_z;
        keep[i] = k ? 1 : 0;
    }
}

extern "C" int pcacc_prep_points(const double *points, const double *tsfm12, const double *noise, double noise_scale, double scale,
                                 double crop_xy, double z_min, double z_max, int32_t remove_ground, double ground_z, int64_t m,
                                 double *out_points, uint8_t *keep, void *stream)
{
    if (m < 0) return PCACC_E_ARG;
    if (m == 0) return PCACC_OK;
    if (!points || !out_points || !keep) return PCACC_E_ARG;
    prep_points_kernel<<<pcacc_grid(m, 256), 256, 0, pcacc_stream(stream)>>>(points, tsfm12, noise, noise_scale, scale, crop_xy, z_min,
                                                                            z_max, remove_ground, ground_z, m, out_points, keep);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// A9. Max over the T frames of a [S,T,P] map stack (models/stpn.py:83 `torch.max(x, dim=2)` after the temporal convs), rows of
// P = H*W*C contiguous elements, f32 or bf16.  One streaming pass: 16 bytes per lane and frame, the winning frame (lowest t on
// ties) kept as one byte per element for the backward pass, which writes the whole gradient (zeros included) in one pass.
// The library reduction walks this layout at ~0.2 TB/s.
// ---------------------------------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256) void frames_max_kernel(const void *__restrict__ x, int64_t n_seq, int frames, int64_t plane_vec,
                                                         void *__restrict__ out, uint8_t *__restrict__ arg)
{
    constexpr int V = BF ? 8 : 4;
    const int64_t total = n_seq * plane_vec;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t s = e / plane_vec, p = e - s * plane_vec;
        float best[V];
        uint8_t who[V];
        for (int t = 0; t < frames; ++t) {
            const uint4 raw = reinterpret_cast<const uint4 *>(x)[(s * frames + t) * plane_vec + p];
            float v[V];
            if (BF) {
                v[0] = pcacc_bf16_lo(raw.x), v[1] = pcacc_bf16_hi(raw.x), v[2] = pcacc_bf16_lo(raw.y), v[3] = pcacc_bf16_hi(raw.y);
                v[V - 4] = pcacc_bf16_lo(raw.z), v[V - 3] = pcacc_bf16_hi(raw.z), v[V - 2] = pcacc_bf16_lo(raw.w), v[V - 1] = pcacc_bf16_hi(raw.w);
            } else {
                v[0] = __uint_as_float(raw.x), v[1] = __uint_as_float(raw.y), v[2] = __uint_as_float(raw.z), v[3] = __uint_as_float(raw.w);
            }
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const bool take = t == 0 || v[k] > best[k] || (v[k] != v[k] && best[k] == best[k]);   // NaN propagates like torch.max
                best[k] = take ? v[k] : best[k];
                who[k] = take ? (uint8_t)t : who[k];
            }
        }
        if (BF) {
            reinterpret_cast<uint4 *>(out)[e] = make_uint4(pcacc_pack_bf16x2(best[0], best[1]), pcacc_pack_bf16x2(best[2], best[3]),
                                                           pcacc_pack_bf16x2(best[V - 4], best[V - 3]), pcacc_pack_bf16x2(best[V - 2], best[V - 1]));
            reinterpret_cast<uint2 *>(arg)[e] = make_uint2(who[0] | who[1] << 8 | who[2] << 16 | who[3] << 24,
                                                           who[V - 4] | who[V - 3] << 8 | who[V - 2] << 16 | who[V - 1] << 24);
        } else {
            reinterpret_cast<float4 *>(out)[e] = make_float4(best[0], best[1], best[2], best[3]);
            reinterpret_cast<uint32_t *>(arg)[e] = who[0] | who[1] << 8 | who[2] << 16 | who[3] << 24;
        }
    }
}

template <bool BF>
__global__ __launch_bounds__(256) void frames_max_bwd_kernel(const void *__restrict__ grad_out, const uint8_t *__restrict__ arg, int64_t n_seq,
                                                             int frames, int64_t plane_vec, void *__restrict__ grad_x)
{
    const int64_t total = n_seq * plane_vec;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t s = e / plane_vec, p = e - s * plane_vec;
        const uint4 g = reinterpret_cast<const uint4 *>(grad_out)[e];
        uint32_t w0, w1 = 0;
        if (BF) {
            const uint2 a = reinterpret_cast<const uint2 *>(arg)[e];
            w0 = a.x, w1 = a.y;
        } else {
            w0 = reinterpret_cast<const uint32_t *>(arg)[e];
        }
        for (int t = 0; t < frames; ++t) {
            uint4 o;
            if (BF) {
                const uint32_t m0 = ((w0 & 0xff) == (uint32_t)t ? 0xffffu : 0u) | (((w0 >> 8) & 0xff) == (uint32_t)t ? 0xffff0000u : 0u);
                const uint32_t m1 = (((w0 >> 16) & 0xff) == (uint32_t)t ? 0xffffu : 0u) | ((w0 >> 24) == (uint32_t)t ? 0xffff0000u : 0u);
                const uint32_t m2 = ((w1 & 0xff) == (uint32_t)t ? 0xffffu : 0u) | (((w1 >> 8) & 0xff) == (uint32_t)t ? 0xffff0000u : 0u);
                const uint32_t m3 = (((w1 >> 16) & 0xff) == (uint32_t)t ? 0xffffu : 0u) | ((w1 >> 24) == (uint32_t)t ? 0xffff0000u : 0u);
                o = make_uint4(g.x & m0, g.y & m1, g.z & m2, g.w & m3);
            } else {
                o = make_uint4((w0 & 0xff) == (uint32_t)t ? g.x : 0u, ((w0 >> 8) & 0xff) == (uint32_t)t ? g.y : 0u,
                               ((w0 >> 16) & 0xff) == (uint32_t)t ? g.z : 0u, (w0 >> 24) == (uint32_t)t ? g.w : 0u);
            }
            reinterpret_cast<uint4 *>(grad_x)[(s * frames + t) * plane_vec + p] = o;
        }
    }
}

static int frames_max_args(int dtype, int64_t n_seq, int frames, int64_t plane, int64_t *plane_vec)
{
    if (dtype != PCACC_F32 && dtype != PCACC_BF16) return PCACC_E_ARG;
    const int v = dtype == PCACC_BF16 ? 8 : 4;
    if (n_seq < 0 || frames <= 0 || frames > 255 || plane <= 0 || plane % v) return PCACC_E_ARG;
    *plane_vec = plane / v;
    return PCACC_OK;
}

extern "C" int pcacc_frames_max(const void *x, int dtype, int64_t n_seq, int32_t frames, int64_t plane, void *out, uint8_t *arg, void *stream)
{
    int64_t pv;
    if (frames_max_args(dtype, n_seq, frames, plane, &pv) != PCACC_OK) return PCACC_E_ARG;
    if (n_seq == 0) return PCACC_OK;
    if (!x || !out || !arg) return PCACC_E_ARG;
    const int grid = pcacc_grid(n_seq * pv, 256, PCACC_CUS * 16);
    if (dtype == PCACC_BF16) frames_max_kernel<true><<<grid, 256, 0, pcacc_stream(stream)>>>(x, n_seq, frames, pv, out, arg);
    else frames_max_kernel<false><<<grid, 256, 0, pcacc_stream(stream)>>>(x, n_seq, frames, pv, out, arg);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_frames_max_backward(const void *grad_out, const uint8_t *arg, int dtype, int64_t n_seq, int32_t frames, int64_t plane,
                                         void *grad_x, void *stream)
{
    int64_t pv;
    if (frames_max_args(dtype, n_seq, frames, plane, &pv) != PCACC_OK) return PCACC_E_ARG;
    if (n_seq == 0) return PCACC_OK;
    if (!grad_out || !arg || !grad_x) return PCACC_E_ARG;
    const int grid = pcacc_grid(n_seq * pv, 256, PCACC_CUS * 16);
    if (dtype == PCACC_BF16) frames_max_bwd_kernel<true><<<grid, 256, 0, pcacc_stream(stream)>>>(grad_out, arg, n_seq, frames, pv, grad_x);
    else frames_max_bwd_kernel<false><<<grid, 256, 0, pcacc_stream(stream)>>>(grad_out, arg, n_seq, frames, pv, grad_x);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
