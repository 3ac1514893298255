// Bilinear sampling kernels on channels-last feature maps.
//
// With [map, y, x, channel] storage each of the 4 taps of a sample is ONE contiguous run of C elements,
// so C/4 adjacent lanes fetch it with a single coalesced request.  The reference instead packs the points
// into fake H x W sampling grids and replicates the whole feature map once per grid
// (models/pillar_encoder.py:252-264); none of that traffic exists here.
//
// grid_sample semantics reproduced (ATen grid_sampler_2d, bilinear, align_corners=False):
//   pix = ((g + 1) * size - 1) / 2;  'border': clamp pix to [0, size-1];  'zeros': drop outside corners
//   weights: nw=(xe-x)(ye-y) ne=(x-xw)(ye-y) sw=(xe-x)(y-yn) se=(x-xw)(y-yn), xw=floor(x), xe=xw+1 ...
#include "common.h"

struct Taps {
    int x0, y0;          // north-west corner
    float w00, w01, w10, w11;   // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    bool vx0, vx1, vy0, vy1;    // corner inside the map
};

template <bool BORDER>
__device__ __forceinline__ Taps make_taps(float gx, float gy, int w, int h)
{
#pragma clang fp contract(off)
    float x = ((gx + 1.0f) * (float)w - 1.0f) / 2.0f;
    float y = ((gy + 1.0f) * (float)h - 1.0f) / 2.0f;
    if (BORDER) {
        x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
        y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
    }
    const float xw = floorf(x), yn = floorf(y);
    const float xe = xw + 1.0f, ys = yn + 1.0f;
    Taps t;
    t.w00 = (xe - x) * (ys - y);
    t.w01 = (x - xw) * (ys - y);
    t.w10 = (xe - x) * (y - yn);
    t.w11 = (x - xw) * (y - yn);
    // NaN / huge coordinates: every validity test fails, the sample is zero (never an OOB access)
    t.vx0 = xw >= 0.0f && xw <= (float)(w - 1);
    t.vx1 = xe >= 0.0f && xe <= (float)(w - 1);
    t.vy0 = yn >= 0.0f && yn <= (float)(h - 1);
    t.vy1 = ys >= 0.0f && ys <= (float)(h - 1);
    t.x0 = t.vx0 ? (int)xw : (t.vx1 ? (int)xe - 1 : 0);
    t.y0 = t.vy0 ? (int)yn : (t.vy1 ? (int)ys - 1 : 0);
    return t;
}

template <int BF16>
__device__ __forceinline__ float4 load4(const void *base, int64_t elem_off)
{
    if (BF16) {
        const uint2 r = *reinterpret_cast<const uint2 *>(static_cast<const uint16_t *>(base) + elem_off);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                           __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4 *>(static_cast<const float *>(base) + elem_off);
}

template <int BF16>
__device__ __forceinline__ void store4(void *base, int64_t elem_off, float4 v)
{
    if (BF16) {
        uint2 r;
        r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
        r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
        *reinterpret_cast<uint2 *>(static_cast<uint16_t *>(base) + elem_off) = r;
    } else {
        *reinterpret_cast<float4 *>(static_cast<float *>(base) + elem_off) = v;
    }
}

__device__ __forceinline__ void axpy4(float4 &acc, float a, const float4 &v)
{
    acc.x += a * v.x; acc.y += a * v.y; acc.z += a * v.z; acc.w += a * v.w;
}

template <int BF16>
__device__ __forceinline__ float4 sample4(const void *map, int w, int c, const Taps &t, int ch)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t r0 = ((int64_t)t.y0 * w + t.x0) * c + ch;
    const int64_t r1 = r0 + (int64_t)w * c;
    if (t.vy0 && t.vx0) axpy4(acc, t.w00, load4<BF16>(map, r0));
    if (t.vy0 && t.vx1) axpy4(acc, t.w01, load4<BF16>(map, r0 + c));
    if (t.vy1 && t.vx0) axpy4(acc, t.w10, load4<BF16>(map, r1));
    if (t.vy1 && t.vx1) axpy4(acc, t.w11, load4<BF16>(map, r1 + c));
    return acc;
}

// ---- A11 forward ---------------------------------------------------------------------------------------
template <int BF16>
__global__ __launch_bounds__(256) void bilinear_gather_kernel(const void *__restrict__ fmap, int n_maps, int h, int w, int c,
                                                              const float *__restrict__ pts, const int32_t *__restrict__ map_idx,
                                                              int64_t k, float xs, float ys, float *__restrict__ out)
{
    const int lpp = c / 4;
    const int64_t total = k * lpp;
    const int esz = BF16 ? 2 : 4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / lpp;
        const int ch = (int)(e - i * lpp) * 4;
        const int mi = map_idx[i];
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mi >= 0 && mi < n_maps) {
            const float gx = __fdiv_rn(pts[i * 3 + 0], xs);       // pillar_encoder.py:248-249
            const float gy = __fdiv_rn(pts[i * 3 + 1], ys);
            const Taps t = make_taps<true>(gx, gy, w, h);
            r = sample4<BF16>(static_cast<const char *>(fmap) + (int64_t)mi * h * w * c * esz, w, c, t, ch);
        }
        *reinterpret_cast<float4 *>(out + i * c + ch) = r;
    }
}

extern "C" int pcacc_bilinear_gather(const void *fmap, int dtype, int n_maps, int h, int w, int c,
                                     const float *points, const int32_t *map_idx, int64_t k,
                                     float x_scale, float y_scale, float *out, void *stream)
{
    if (k < 0 || n_maps <= 0 || h <= 0 || w <= 0 || c <= 0 || (c % 4) || (dtype != PCACC_F32 && dtype != PCACC_BF16)) return PCACC_E_ARG;
    if (k > 0 && (!fmap || !points || !map_idx || !out)) return PCACC_E_ARG;
    if (k == 0) return PCACC_OK;
    hipStream_t s = pcacc_stream(stream);
    const int grid = pcacc_grid(k * (c / 4), 256);
    if (dtype == PCACC_BF16)
        bilinear_gather_kernel<1><<<grid, 256, 0, s>>>(fmap, n_maps, h, w, c, points, map_idx, k, x_scale, y_scale, out);
    else
        bilinear_gather_kernel<0><<<grid, 256, 0, s>>>(fmap, n_maps, h, w, c, points, map_idx, k, x_scale, y_scale, out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- A11 backward (w.r.t. the feature map): 4 weighted scatter-adds of the C-vector, fp32 HW atomics -----
__device__ __forceinline__ void atomic_axpy4(float *dst, float a, const float4 &g)
{
    atomicAdd(dst + 0, a * g.x); atomicAdd(dst + 1, a * g.y); atomicAdd(dst + 2, a * g.z); atomicAdd(dst + 3, a * g.w);
}

__global__ __launch_bounds__(256) void bilinear_gather_bwd_kernel(const float *__restrict__ grad_out, int n_maps, int h, int w, int c,
                                                                  const float *__restrict__ pts, const int32_t *__restrict__ map_idx,
                                                                  int64_t k, float xs, float ys, float *grad_fmap)
{
    const int lpp = c / 4;
    const int64_t total = k * lpp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / lpp;
        const int ch = (int)(e - i * lpp) * 4;
        const int mi = map_idx[i];
        if (mi < 0 || mi >= n_maps) continue;
        const float gx = __fdiv_rn(pts[i * 3 + 0], xs);
        const float gy = __fdiv_rn(pts[i * 3 + 1], ys);
        const Taps t = make_taps<true>(gx, gy, w, h);
        const float4 g = *reinterpret_cast<const float4 *>(grad_out + i * c + ch);
        float *base = grad_fmap + (int64_t)mi * h * w * c;
        const int64_t r0 = ((int64_t)t.y0 * w + t.x0) * c + ch;
        const int64_t r1 = r0 + (int64_t)w * c;
        if (t.vy0 && t.vx0) atomic_axpy4(base + r0, t.w00, g);
        if (t.vy0 && t.vx1) atomic_axpy4(base + r0 + c, t.w01, g);
        if (t.vy1 && t.vx0) atomic_axpy4(base + r1, t.w10, g);
        if (t.vy1 && t.vx1) atomic_axpy4(base + r1 + c, t.w11, g);
    }
}

extern "C" int pcacc_bilinear_gather_backward(const float *grad_out, int n_maps, int h, int w, int c,
                                              const float *points, const int32_t *map_idx, int64_t k,
                                              float x_scale, float y_scale, float *grad_fmap, void *stream)
{
    if (k < 0 || n_maps <= 0 || h <= 0 || w <= 0 || c <= 0 || (c % 4)) return PCACC_E_ARG;
    if (k > 0 && (!grad_out || !points || !map_idx || !grad_fmap)) return PCACC_E_ARG;
    if (k == 0) return PCACC_OK;
    bilinear_gather_bwd_kernel<<<pcacc_grid(k * (c / 4), 256), 256, 0, pcacc_stream(stream)>>>(
        grad_out, n_maps, h, w, c, points, map_idx, k, x_scale, y_scale, grad_fmap);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- A11 backward without atomics: points sorted by the cell of their upper-left tap (CSR from pcacc_csr_build), then one
// lane group per OUTPUT cell sums the contributions of the four neighbouring base cells in index order.  The atomic
// version issues 16 fp32 atomics per (point, 4 channels): 106 M atomics for 416 k points x 64 channels = 1.5 ms at the
// 72 G atomics/s the L2 sustains on scattered addresses; this one reads every gradient row four times (coalesced) and
// writes every map cell once; sums run in index order (bit-reproducible for cells with <= 64 points, the CSR's sorted case).
__global__ __launch_bounds__(256) void bilinear_base_cell_kernel(const float *__restrict__ pts, const int32_t *__restrict__ map_idx,
                                                                 int64_t k, int n_maps, int h, int w, float xs, float ys,
                                                                 int32_t *__restrict__ key)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < k; i += (int64_t)gridDim.x * 256) {
        const int mi = map_idx[i];
        int32_t cell = n_maps * h * w;                                           // points of no map go to a spare segment
        if (mi >= 0 && mi < n_maps) {
            const Taps t = make_taps<true>(__fdiv_rn(pts[i * 3 + 0], xs), __fdiv_rn(pts[i * 3 + 1], ys), w, h);
            cell = (mi * h + t.y0) * w + t.x0;
        }
        key[i] = cell;
    }
}

// Step 1 of the sorted backward: per point in CSR order, its four tap weights (0 where the tap is outside the map) and its
// gradient row -- so that the per-cell loops below read consecutive rows with no dependent index loads and no divisions.
template <int G_BF16>
__global__ __launch_bounds__(256) void bilinear_sorted_prep_kernel(const void *__restrict__ grad_out, int c, int h, int w,
                                                                   const float *__restrict__ pts, const int32_t *__restrict__ order,
                                                                   int64_t k, float xs, float ys, float4 *__restrict__ wts,
                                                                   void *__restrict__ g_sorted, int *__restrict__ crowded)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) crowded[0] = 0;                     // the work list of the per-cell pass that follows starts empty
    const int lpp = c / 4;
    const int64_t total = k * lpp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t q = e / lpp;
        const int ch = (int)(e - q * lpp) * 4;
        const int64_t i = order[q];
        store4<G_BF16>(g_sorted, q * c + ch, load4<G_BF16>(grad_out, i * c + ch));
        if (ch == 0) {
            const Taps t = make_taps<true>(__fdiv_rn(pts[i * 3 + 0], xs), __fdiv_rn(pts[i * 3 + 1], ys), w, h);
            wts[q] = make_float4((t.vy0 && t.vx0) ? t.w00 : 0.f, (t.vy0 && t.vx1) ? t.w01 : 0.f, (t.vy1 && t.vx0) ? t.w10 : 0.f,
                                 (t.vy1 && t.vx1) ? t.w11 : 0.f);
        }
    }
}

// [r5] cells are walked in 4 x 4 tiles (one tile = the 16 cells x c / 4 lanes a 64-channel workgroup pass covers): a point's gradient row is read by the
// four cells around it, and with row-major cells the two in the next map row were 288 cells = another workgroup, usually another XCD, away -- every row
// came out of HBM / the Infinity Cache twice (PMC round 4: 1.83 x the algorithmic bytes); inside a tile the second reader finds it in the CU's cache.
// The six segment offsets a cell needs are loaded up front (independent loads) instead of two behind each tap's bounds test.  Same taps in the same
// order per cell: results bit-identical to the row-major walk.
#define BG_CROWD 8                    // a cell with a longer segment is summed by the crowded-cell kernel
#define BG_U 16                       // its batch: points whose weights and rows are in flight together
// segment bounds of the four base cells (x - tx, y - ty) of a cell; a base outside the map gets the empty range
__device__ __forceinline__ void bg_bounds(const int32_t *__restrict__ seg_offsets, int64_t cell, int x, int y, int w, int (&b)[4], int (&en)[4])
{
    const int o_c = seg_offsets[cell], o_c1 = seg_offsets[cell + 1];
    const int o_l = x > 0 ? seg_offsets[cell - 1] : o_c;
    b[0] = o_c; en[0] = o_c1;
    b[1] = o_l; en[1] = o_c;
    if (y > 0) {
        const int o_u = seg_offsets[cell - w], o_u1 = seg_offsets[cell - w + 1];
        const int o_ul = x > 0 ? seg_offsets[cell - w - 1] : o_u;
        b[2] = o_u; en[2] = o_u1;
        b[3] = o_ul; en[3] = o_u;
    } else {
        b[2] = en[2] = b[3] = en[3] = 0;
    }
}

template <int G_BF16, int OUT_BF16>
__global__ __launch_bounds__(256) void bilinear_gather_bwd_sorted_kernel(const void *__restrict__ g_sorted, const float4 *__restrict__ wts,
                                                                         int n_maps, int h, int w, int c,
                                                                         const int32_t *__restrict__ seg_offsets,
                                                                         void *__restrict__ grad_fmap, int *__restrict__ crowded)
{
    const int lpp = c / 4;
    const int tw = (w + 3) >> 2, th = (h + 3) >> 2;
    const int64_t total = (int64_t)n_maps * th * tw * 16 * lpp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / lpp;                                               // tile-major cell number
        const int ch = (int)(e - k * lpp) * 4;
        const int64_t tile = k >> 4;
        const int ci = (int)(k & 15);
        const int tx0 = (int)(tile % tw), ty0 = (int)((tile / tw) % th), mi = (int)(tile / ((int64_t)tw * th));
        const int x = tx0 * 4 + (ci & 3), y = ty0 * 4 + (ci >> 2);
        if (x >= w || y >= h) continue;
        const int64_t cell = ((int64_t)mi * h + y) * w + x;
        int b[4], en[4];
        bg_bounds(seg_offsets, cell, x, y, w, b, en);
        // [r5] crowded cells -- a segment of more than BG_CROWD points: the step's points sit in a few boxes, tens to hundreds per cell -- go to a work list
        // and are summed by bilinear_gather_bwd_crowded_kernel (same order, loads in batches); here they would hold a lane group for hundreds of dependent
        // round trips, and giving THIS kernel the registers for batches cost the sparse cells 20 % (occupancy 8 -> 5)
        if (max(max(en[0] - b[0], en[1] - b[1]), max(en[2] - b[2], en[3] - b[3])) > BG_CROWD) {
            if (ch == 0) crowded[1 + atomicAdd(crowded, 1)] = (int)cell;
            continue;
        }
        // [r5] the FIRST point of each of the four segments is fetched up front -- tap weights and gradient row, eight independent loads from clamped
        // positions -- before anything is added: with ~1 point per cell that is the whole cell in two memory round trips (offsets, then rows) instead of a
        // weights -> row chain per tap, one after the other (in the step, on cold data, the kernel ran at 0.55 TB/s: 235 us against 79 us on a warm
        // benchmark loop).  Sums run in the same order as before -- tap by tap, points in segment order, zero weights skipped: bit-identical.
        float4 w0[4], g0[4];
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            const int q = b[tap] < en[tap] ? b[tap] : 0;                         // row 0 exists (the launcher handles k = 0)
            w0[tap] = wts[q];
            g0[tap] = load4<G_BF16>(g_sorted, (int64_t)q * c + ch);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {                                      // this cell as tap (ty, tx) of base cell (y-ty, x-tx)
            if (b[tap] >= en[tap]) continue;
            {
                const float wt = tap == 0 ? w0[tap].x : tap == 1 ? w0[tap].y : tap == 2 ? w0[tap].z : w0[tap].w;
                if (wt != 0.f) {                                                 // 0: outside the map (or an exact zero weight)
                    const float4 g = g0[tap];
                    acc.x += wt * g.x; acc.y += wt * g.y; acc.z += wt * g.z; acc.w += wt * g.w;
                }
            }
            for (int q = b[tap] + 1; q < en[tap]; ++q) {
                const float4 w4 = wts[q];
                const float wt = tap == 0 ? w4.x : tap == 1 ? w4.y : tap == 2 ? w4.z : w4.w;
                if (wt == 0.f) continue;
                const float4 g = load4<G_BF16>(g_sorted, (int64_t)q * c + ch);
                acc.x += wt * g.x; acc.y += wt * g.y; acc.z += wt * g.z; acc.w += wt * g.w;
            }
        }
        store4<OUT_BF16>(grad_fmap, cell * c + ch, acc);
    }
}

// The crowded cells of the work list (crowded[0] = their number): c / 4 lanes per cell as above, the same sums in the same order -- tap by tap, points in
// segment order, zero weights skipped -- with the loads of BG_U points in flight at a time.  Cells are independent, so the order of the list does not matter.
template <int G_BF16, int OUT_BF16>
__global__ __launch_bounds__(256) void bilinear_gather_bwd_crowded_kernel(const void *__restrict__ g_sorted, const float4 *__restrict__ wts, int h, int w, int c,
                                                                          const int32_t *__restrict__ seg_offsets, void *__restrict__ grad_fmap,
                                                                          const int *__restrict__ crowded)
{
    const int lpp = c / 4;
    const int64_t total = (int64_t)crowded[0] * lpp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / lpp;
        const int ch = (int)(e - k * lpp) * 4;
        const int64_t cell = crowded[1 + k];
        const int x = (int)(cell % w), y = (int)((cell / w) % h);
        int b[4], en[4];
        bg_bounds(seg_offsets, cell, x, y, w, b, en);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            for (int q = b[tap]; q < en[tap]; q += BG_U) {                       // the last batch loads its tail from the segment's last row and skips it
                float4 wq[BG_U], gq[BG_U];
#pragma unroll
                for (int u = 0; u < BG_U; ++u) {
                    const int qq = min(q + u, en[tap] - 1);
                    wq[u] = wts[qq];
                    gq[u] = load4<G_BF16>(g_sorted, (int64_t)qq * c + ch);
                }
#pragma unroll
                for (int u = 0; u < BG_U; ++u) {
                    const float wt = tap == 0 ? wq[u].x : tap == 1 ? wq[u].y : tap == 2 ? wq[u].z : wq[u].w;
                    if (q + u >= en[tap] || wt == 0.f) continue;
                    acc.x += wt * gq[u].x; acc.y += wt * gq[u].y; acc.z += wt * gq[u].z; acc.w += wt * gq[u].w;
                }
            }
        }
        store4<OUT_BF16>(grad_fmap, cell * c + ch, acc);
    }
}

extern "C" int pcacc_bilinear_base_cells(const float *points, const int32_t *map_idx, int64_t k, int n_maps, int h, int w,
                                         float x_scale, float y_scale, int32_t *cell, void *stream)
{
    if (k < 0 || n_maps <= 0 || h <= 0 || w <= 0 || (int64_t)n_maps * h * w >= 0x7fffffff) return PCACC_E_ARG;
    if (k == 0) return PCACC_OK;
    if (!points || !map_idx || !cell) return PCACC_E_ARG;
    bilinear_base_cell_kernel<<<pcacc_grid(k, 256), 256, 0, pcacc_stream(stream)>>>(points, map_idx, k, n_maps, h, w, x_scale, y_scale, cell);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_bilinear_sorted_workspace_bytes(int64_t k, int c, int grad_dtype, size_t *bytes)
{
    if (!bytes || k < 0 || c <= 0 || (c % 4) || (grad_dtype != PCACC_F32 && grad_dtype != PCACC_BF16)) return PCACC_E_ARG;
    // tap weights, gradient rows in CSR order, and the work list of crowded cells: its count + at most 4 k / (BG_CROWD + 1) cells (a crowded cell reads a segment
    // of more than BG_CROWD points, and a segment is read by four cells)
    *bytes = pcacc_align((size_t)k * 16) + pcacc_align((size_t)k * c * (grad_dtype == PCACC_BF16 ? 2 : 4)) + pcacc_align(((size_t)k / 2 + 8) * 4);
    return PCACC_OK;
}

extern "C" int pcacc_bilinear_gather_backward_sorted(const void *grad_out, int grad_dtype, int n_maps, int h, int w, int c,
                                                     const float *points, const int32_t *seg_offsets, const int32_t *order, int64_t k,
                                                     float x_scale, float y_scale, void *grad_fmap, int out_dtype, void *workspace,
                                                     size_t workspace_bytes, void *stream)
{
    if (n_maps <= 0 || h <= 0 || w <= 0 || c <= 0 || (c % 4) || k < 0) return PCACC_E_ARG;
    if ((grad_dtype != PCACC_F32 && grad_dtype != PCACC_BF16) || (out_dtype != PCACC_F32 && out_dtype != PCACC_BF16)) return PCACC_E_ARG;
    if (!grad_out || !points || !seg_offsets || !order || !grad_fmap || !workspace) return PCACC_E_ARG;
    size_t need = 0;
    pcacc_bilinear_sorted_workspace_bytes(k, c, grad_dtype, &need);
    if (workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    float4 *wts = reinterpret_cast<float4 *>(workspace);
    void *g_sorted = static_cast<char *>(workspace) + pcacc_align((size_t)k * 16);
    int *crowded = reinterpret_cast<int *>(static_cast<char *>(g_sorted) + pcacc_align((size_t)k * c * (grad_dtype == PCACC_BF16 ? 2 : 4)));
    if (k > 0) {
        const int pgrid = pcacc_grid(k * (c / 4), 256);
        if (grad_dtype == PCACC_BF16) bilinear_sorted_prep_kernel<1><<<pgrid, 256, 0, s>>>(grad_out, c, h, w, points, order, k, x_scale, y_scale, wts, g_sorted, crowded);
        else bilinear_sorted_prep_kernel<0><<<pgrid, 256, 0, s>>>(grad_out, c, h, w, points, order, k, x_scale, y_scale, wts, g_sorted, crowded);
    }
    if (k == 0) {                                                               // no point: every cell's sum is empty (the kernel reads row 0 unconditionally)
        const size_t esz = out_dtype == PCACC_BF16 ? 2 : 4;
        if (hipMemsetAsync(grad_fmap, 0, (size_t)n_maps * h * w * c * esz, s) != hipSuccess) return PCACC_E_LAUNCH;
        return PCACC_OK;
    }
    // many short workgroups: the per-cell loops are as long as the cell is crowded, a fine grid evens that out
    const int grid = pcacc_grid((int64_t)n_maps * ((h + 3) / 4) * ((w + 3) / 4) * 16 * (c / 4), 256, PCACC_CUS * 64);
    const int cgrid = pcacc_grid(((int64_t)k / 2 + 8) * (c / 4), 256, PCACC_CUS * 8);
#define BGS(GB, OB) do { bilinear_gather_bwd_sorted_kernel<GB, OB><<<grid, 256, 0, s>>>(g_sorted, wts, n_maps, h, w, c, seg_offsets, grad_fmap, crowded); \
                         bilinear_gather_bwd_crowded_kernel<GB, OB><<<cgrid, 256, 0, s>>>(g_sorted, wts, h, w, c, seg_offsets, grad_fmap, crowded); } while (0)
    if (grad_dtype == PCACC_BF16) { if (out_dtype == PCACC_BF16) BGS(1, 1); else BGS(1, 0); }
    else { if (out_dtype == PCACC_BF16) BGS(0, 1); else BGS(0, 0); }
#undef BGS
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- A9: ego-motion BEV warp -------------------------------------------------------------------------------
template <int BF16>
__global__ __launch_bounds__(256) void bev_warp_kernel(const void *__restrict__ bev, int n_batch, int nt, int h, int w, int c,
                                                       const float *__restrict__ inv_pose, float x_reso, float y_reso,
                                                       float x_min, float y_min, void *__restrict__ out, uint16_t *__restrict__ out16 = nullptr)
{
    const int lpp = c / 4;
    const int64_t total = (int64_t)n_batch * nt * h * w * lpp;
    const int esz = BF16 ? 2 : 4;
    const int64_t frame_elems = (int64_t)h * w * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int ch = (int)(e % lpp) * 4;
        int64_t q = e / lpp;
        const int x = (int)(q % w); q /= w;
        const int y = (int)(q % h); q /= h;
        const int t = (int)(q % nt);
        const int b = (int)(q / nt);
        const int64_t dst = (((int64_t)(b * nt + t) * h + y) * w + x) * c + ch;
        float4 r;
        if (t == 0) {
            // motionnet.py:111: bev_feats[batch_idx, frame_idx:frame_idx+1] with frame_idx leaked = nt-1
            r = load4<BF16>(bev, (((int64_t)(b * nt + nt - 1) * h + y) * w + x) * c + ch);
        } else {
#pragma clang fp contract(off)
            const float *p = inv_pose + (int64_t)(b * nt + t) * 16;
            const float mx = ((float)x + 0.5f) * x_reso + x_min;          // motionnet.py:60-68
            const float my = ((float)y + 0.5f) * y_reso + y_min;
            const float tx = (p[0] * mx + p[1] * my) + p[3];              // pose[:2,:2] @ grid + pose[:2,3:4]
            const float ty = (p[4] * mx + p[5] * my) + p[7];
            const Taps tp = make_taps<false>(tx / fabsf(x_min), ty / fabsf(y_min), w, h);
            r = sample4<BF16>(static_cast<const char *>(bev) + (int64_t)(b * nt + t) * frame_elems * esz, w, c, tp, ch);
        }
        store4<BF16>(out, dst, r);
        // [r6] 'mixed' mode: the bf16 shadow of the warped map (what the bf16 backward of the first temporal convolution reads) from the same registers --
        // it used to be a conversion pass of its own over the 212 MB map (ops._EnterMixed)
        if (!BF16 && out16) *reinterpret_cast<uint2 *>(out16 + dst) = make_uint2(pcacc_pack_bf16x2(r.x, r.y), pcacc_pack_bf16x2(r.z, r.w));
    }
}

extern "C" int pcacc_bev_warp_dual(const float *bev, int n_batch, int nt, int h, int w, int c, const float *inv_pose, float x_reso, float y_reso, float x_min,
                                   float y_min, float *out, uint16_t *out16, void *stream)
{
    if (n_batch <= 0 || nt <= 0 || h <= 0 || w <= 0 || c <= 0 || (c % 4) || !bev || !inv_pose || !out || !out16) return PCACC_E_ARG;
    const int64_t total = (int64_t)n_batch * nt * h * w * (c / 4);
    bev_warp_kernel<0><<<pcacc_grid(total, 256), 256, 0, pcacc_stream(stream)>>>(bev, n_batch, nt, h, w, c, inv_pose, x_reso, y_reso, x_min, y_min, out, out16);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_bev_warp(const void *bev, int dtype, int n_batch, int nt, int h, int w, int c,
                              const float *inv_pose, float x_reso, float y_reso, float x_min, float y_min,
                              void *out, void *stream)
{
    if (n_batch <= 0 || nt <= 0 || h <= 0 || w <= 0 || c <= 0 || (c % 4) || (dtype != PCACC_F32 && dtype != PCACC_BF16)) return PCACC_E_ARG;
    if (!bev || !inv_pose || !out) return PCACC_E_ARG;
    hipStream_t s = pcacc_stream(stream);
    const int64_t total = (int64_t)n_batch * nt * h * w * (c / 4);
    if (dtype == PCACC_BF16)
        bev_warp_kernel<1><<<pcacc_grid(total, 256), 256, 0, s>>>(bev, n_batch, nt, h, w, c, inv_pose, x_reso, y_reso, x_min, y_min, out);
    else
        bev_warp_kernel<0><<<pcacc_grid(total, 256), 256, 0, s>>>(bev, n_batch, nt, h, w, c, inv_pose, x_reso, y_reso, x_min, y_min, out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- A10: per-point rigid transform ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rigid_transform_kernel(const float *__restrict__ pts, const int32_t *__restrict__ frame_idx,
                                                              const float *__restrict__ tsfm, int64_t n, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float *m = tsfm + (int64_t)frame_idx[i] * 16;
        const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        out[i * 3 + 0] = m[0] * x + m[1] * y + m[2] * z + m[3];
        out[i * 3 + 1] = m[4] * x + m[5] * y + m[6] * z + m[7];
        out[i * 3 + 2] = m[8] * x + m[9] * y + m[10] * z + m[11];
    }
}

extern "C" int pcacc_rigid_transform(const float *points, const int32_t *frame_idx, const float *tsfm, int64_t n,
                                     float *out, void *stream)
{
    if (n < 0 || (n > 0 && (!points || !frame_idx || !tsfm || !out))) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    rigid_transform_kernel<<<pcacc_grid(n, 256), 256, 0, pcacc_stream(stream)>>>(points, frame_idx, tsfm, n, out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
