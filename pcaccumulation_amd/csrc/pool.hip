// Tail of a U-Net encoder stage on channels-last bf16 rows: models/unet.py:60-71 (DownConv: conv, ReLU, conv, ReLU, 2x2 max-pool,
// returning the pooled map AND the map before the pool as the decoder's skip connection).
//
// Forward: the 2x2 / stride-2 max-pool (floor mode, as nn.MaxPool2d(2, 2)); no index tensor is kept.
// Backward: the stage's second convolution wants d(loss)/d(its ReLU output) = [un-pooled gradient of the pooled map] + [gradient of
// the skip connection], zeroed where that output is not positive.  Autograd with library ops spends three full-resolution
// element-wise passes on it (max_pool2d backward through an int64 index map, the add of the two branches, threshold_backward:
// 8 + reads and writes of the map); here it is one pass: per 2x2 window the four outputs y, the four skip gradients and the one
// pooled gradient are read, the window's first maximum (scan order, strict >, what the library's forward picks) gets the pooled
// gradient, and the masked sums are written.
#include "common.h"

__device__ __forceinline__ float pool_lo(uint32_t v) { return pcacc_bf16_lo(v); }
__device__ __forceinline__ float pool_hi(uint32_t v) { return pcacc_bf16_hi(v); }

// max of two packed bf16 pairs, element-wise (NaN: a NaN in `b` wins, as `val > max || isnan(val)` does)
__device__ __forceinline__ uint32_t pool_max2(uint32_t a, uint32_t b)
{
    const float al = pool_lo(a), bl = pool_lo(b), ah = pool_hi(a), bh = pool_hi(b);
    const uint32_t lo = (bl > al || bl != bl) ? (b & 0xffffu) : (a & 0xffffu);
    const uint32_t hi = (bh > ah || bh != bh) ? (b & 0xffff0000u) : (a & 0xffff0000u);
    return lo | hi;
}

__global__ __launch_bounds__(256) void maxpool2x2_kernel(const uint4 *__restrict__ x, int64_t n_img, int h, int w, int c8,
                                                         uint4 *__restrict__ out)
{
    const int h2 = h / 2, w2 = w / 2;
    const int64_t total = n_img * h2 * w2 * c8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c8);
        int64_t q = e / c8;
        const int xo = (int)(q % w2); q /= w2;
        const int yo = (int)(q % h2);
        const int64_t img = q / h2;
        const int64_t base = ((img * h + 2 * yo) * w + 2 * xo) * c8 + c;
        const uint4 a = x[base], b = x[base + c8], d = x[base + (int64_t)w * c8], f = x[base + (int64_t)w * c8 + c8];
        uint4 m;
        m.x = pool_max2(pool_max2(pool_max2(a.x, b.x), d.x), f.x);
        m.y = pool_max2(pool_max2(pool_max2(a.y, b.y), d.y), f.y);
        m.z = pool_max2(pool_max2(pool_max2(a.z, b.z), d.z), f.z);
        m.w = pool_max2(pool_max2(pool_max2(a.w, b.w), d.w), f.w);
        out[e] = m;
    }
}

// one packed pair of the four window positions: which position holds the first maximum of each half, then
// out_i = y_i > 0 ? skip_i + (i is that position ? pooled : 0) : 0, rounded to bf16 once
__device__ __forceinline__ void pool_bwd2(const uint32_t (&y)[4], const uint32_t (&gs)[4], uint32_t gp, uint32_t (&o)[4])
{
    float best_l = pool_lo(y[0]), best_h = pool_hi(y[0]);
    int arg_l = 0, arg_h = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const float vl = pool_lo(y[i]), vh = pool_hi(y[i]);
        if (vl > best_l || vl != vl) { best_l = vl; arg_l = i; }
        if (vh > best_h || vh != vh) { best_h = vh; arg_h = i; }
    }
    const float pl = pool_lo(gp), ph = pool_hi(gp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float l = pool_lo(gs[i]) + (arg_l == i ? pl : 0.f), hv = pool_hi(gs[i]) + (arg_h == i ? ph : 0.f);
        if (!(pool_lo(y[i]) > 0.f)) l = 0.f;
        if (!(pool_hi(y[i]) > 0.f)) hv = 0.f;
        o[i] = pcacc_pack_bf16x2(l, hv);
    }
}

// one lane per (2x2 window, 8 channels); pixels of the odd last row / column (outside every window) take the skip gradient only
template <bool HAS_POOL, bool HAS_SKIP>
__global__ __launch_bounds__(256) void pool_skip_relu_bwd_kernel(const uint4 *__restrict__ y, const uint4 *__restrict__ g_pool,
                                                                 const uint4 *__restrict__ g_skip, int64_t n_img, int h, int w, int c8,
                                                                 uint4 *__restrict__ out, int gs_pitch)
{
    // gs_pitch: 16-byte units between consecutive pixels of g_skip (c8 for a dense map; 2 c8 for a channel slice of the decoder's
    // concatenation gradient, read in place instead of through a contiguous copy)
    const int h2 = (h + 1) / 2, w2 = (w + 1) / 2, hp = h / 2, wp = w / 2;
    const int64_t total = n_img * h2 * w2 * c8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c8);
        int64_t q = e / c8;
        const int xo = (int)(q % w2); q /= w2;
        const int yo = (int)(q % h2);
        const int64_t img = q / h2;
        const bool in_x = 2 * xo + 1 < w, in_y = 2 * yo + 1 < h;
        const int64_t base = ((img * h + 2 * yo) * w + 2 * xo) * c8 + c;
        const int64_t off[4] = {0, c8, (int64_t)w * c8, (int64_t)w * c8 + c8};
        const int64_t gbase = ((img * h + 2 * yo) * w + 2 * xo) * gs_pitch + c;
        const int64_t goff[4] = {0, gs_pitch, (int64_t)w * gs_pitch, (int64_t)w * gs_pitch + gs_pitch};
        const bool ok[4] = {true, in_x, in_y, in_x && in_y};
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        uint4 yv[4], gs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            yv[i] = ok[i] ? y[base + off[i]] : zero;
            gs[i] = (HAS_SKIP && ok[i]) ? g_skip[gbase + goff[i]] : zero;
        }
        uint4 gp = zero;
        if (HAS_POOL && in_x && in_y && yo < hp && xo < wp) gp = g_pool[((img * hp + yo) * wp + xo) * c8 + c];
        uint4 o[4];
        {
            uint32_t a[4], b[4], r[4];
#define POOL_LANE(F)                                                                                                   \
            a[0] = yv[0].F; a[1] = yv[1].F; a[2] = yv[2].F; a[3] = yv[3].F;                                            \
            b[0] = gs[0].F; b[1] = gs[1].F; b[2] = gs[2].F; b[3] = gs[3].F;                                            \
            pool_bwd2(a, b, gp.F, r);                                                                                  \
            o[0].F = r[0]; o[1].F = r[1]; o[2].F = r[2]; o[3].F = r[3];
            POOL_LANE(x) POOL_LANE(y) POOL_LANE(z) POOL_LANE(w)
#undef POOL_LANE
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) out[base + off[i]] = o[i];
    }
}

extern "C" int pcacc_maxpool2x2_bf16(const uint16_t *x, int64_t n_img, int32_t h, int32_t w, int32_t c, uint16_t *out, void *stream)
{
    if (n_img < 0 || h < 2 || w < 2 || c <= 0 || (c % 8)) return PCACC_E_ARG;
    if (n_img == 0) return PCACC_OK;
    if (!x || !out) return PCACC_E_ARG;
    const int64_t total = n_img * (h / 2) * (w / 2) * (c / 8);
    maxpool2x2_kernel<<<pcacc_grid(total, 256, PCACC_CUS * 16), 256, 0, pcacc_stream(stream)>>>(
        reinterpret_cast<const uint4 *>(x), n_img, h, w, c / 8, reinterpret_cast<uint4 *>(out));
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

static int pool_skip_bwd_bf16(const uint16_t *y, const uint16_t *grad_pooled, const uint16_t *grad_skip, int64_t skip_pitch, int64_t n_img,
                              int32_t h, int32_t w, int32_t c, uint16_t *grad_y, void *stream)
{
    if (n_img < 0 || h < 2 || w < 2 || c <= 0 || (c % 8) || skip_pitch < c || (skip_pitch % 8) || skip_pitch / 8 > 0x7fffffff) return PCACC_E_ARG;
    if (grad_skip && (reinterpret_cast<uintptr_t>(grad_skip) & 15)) return PCACC_E_ARG;
    const int gs_pitch = (int)(skip_pitch / 8);
    if (n_img == 0) return PCACC_OK;
    if (!y || !grad_y) return PCACC_E_ARG;
    const int64_t total = n_img * ((h + 1) / 2) * ((w + 1) / 2) * (c / 8);
    const int grid = pcacc_grid(total, 256, PCACC_CUS * 16);
    hipStream_t s = pcacc_stream(stream);
    const uint4 *yy = reinterpret_cast<const uint4 *>(y), *gp = reinterpret_cast<const uint4 *>(grad_pooled),
                *gs = reinterpret_cast<const uint4 *>(grad_skip);
    uint4 *o = reinterpret_cast<uint4 *>(grad_y);
    if (gp && gs) pool_skip_relu_bwd_kernel<true, true><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 8, o, gs_pitch);
    else if (gp) pool_skip_relu_bwd_kernel<true, false><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 8, o, gs_pitch);
    else if (gs) pool_skip_relu_bwd_kernel<false, true><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 8, o, gs_pitch);
    else pool_skip_relu_bwd_kernel<false, false><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 8, o, gs_pitch);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_pool_skip_relu_backward_bf16(const uint16_t *y, const uint16_t *grad_pooled, const uint16_t *grad_skip, int64_t n_img,
                                                  int32_t h, int32_t w, int32_t c, uint16_t *grad_y, void *stream)
{
    return pool_skip_bwd_bf16(y, grad_pooled, grad_skip, c, n_img, h, w, c, grad_y, stream);
}

// grad_skip read in place from a wider map: skip_pitch = elements between consecutive pixels (>= c, a multiple of 8 / 4 for bf16 / f32; the
// pointer 16-byte aligned) -- the skip half of the decoder's concatenation gradient (models/unet.py:101-113) without a contiguous copy
extern "C" int pcacc_pool_skip_relu_backward_strided_bf16(const uint16_t *y, const uint16_t *grad_pooled, const uint16_t *grad_skip,
                                                          int64_t skip_pitch, int64_t n_img, int32_t h, int32_t w, int32_t c, uint16_t *grad_y,
                                                          void *stream)
{
    return pool_skip_bwd_bf16(y, grad_pooled, grad_skip, skip_pitch, n_img, h, w, c, grad_y, stream);
}

// ---- the same two passes on fp32 rows (compute modes fp32x3 / fp32): 4 channels per lane -------------------------------------------------
__device__ __forceinline__ float pool_fmax(float a, float b) { return (b > a || b != b) ? b : a; }

__global__ __launch_bounds__(256) void maxpool2x2_f32_kernel(const float4 *__restrict__ x, int64_t n_img, int h, int w, int c4, float4 *__restrict__ out)
{
    const int h2 = h / 2, w2 = w / 2;
    const int64_t total = n_img * h2 * w2 * c4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c4);
        int64_t q = e / c4;
        const int xo = (int)(q % w2); q /= w2;
        const int yo = (int)(q % h2);
        const int64_t img = q / h2;
        const int64_t base = ((img * h + 2 * yo) * w + 2 * xo) * c4 + c;
        const float4 a = x[base], b = x[base + c4], d = x[base + (int64_t)w * c4], f = x[base + (int64_t)w * c4 + c4];
        out[e] = make_float4(pool_fmax(pool_fmax(pool_fmax(a.x, b.x), d.x), f.x), pool_fmax(pool_fmax(pool_fmax(a.y, b.y), d.y), f.y),
                             pool_fmax(pool_fmax(pool_fmax(a.z, b.z), d.z), f.z), pool_fmax(pool_fmax(pool_fmax(a.w, b.w), d.w), f.w));
    }
}

__device__ __forceinline__ void pool_bwd1(const float (&y)[4], const float (&gs)[4], float gp, float (&o)[4])
{
    float best = y[0];
    int arg = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (y[i] > best || y[i] != y[i]) { best = y[i]; arg = i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (y[i] > 0.f) ? gs[i] + (arg == i ? gp : 0.f) : 0.f;
}

// out_amax (256 zeroed slots, pcacc_absmax256 layout, or NULL): the largest magnitude written -- the scale of the split kernels that read grad_y
template <bool HAS_POOL, bool HAS_SKIP>
__global__ __launch_bounds__(256) void pool_skip_relu_bwd_f32_kernel(const float4 *__restrict__ y, const float4 *__restrict__ g_pool,
                                                                     const float4 *__restrict__ g_skip, int64_t n_img, int h, int w, int c4,
                                                                     float4 *__restrict__ out, float *__restrict__ out_amax, int gs_pitch)
{
    const int h2 = (h + 1) / 2, w2 = (w + 1) / 2, hp = h / 2, wp = w / 2;
    const int64_t total = n_img * h2 * w2 * c4;
    float mx = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c4);
        int64_t q = e / c4;
        const int xo = (int)(q % w2); q /= w2;
        const int yo = (int)(q % h2);
        const int64_t img = q / h2;
        const bool in_x = 2 * xo + 1 < w, in_y = 2 * yo + 1 < h;
        const int64_t base = ((img * h + 2 * yo) * w + 2 * xo) * c4 + c;
        const int64_t off[4] = {0, c4, (int64_t)w * c4, (int64_t)w * c4 + c4};
        const int64_t gbase = ((img * h + 2 * yo) * w + 2 * xo) * gs_pitch + c;
        const int64_t goff[4] = {0, gs_pitch, (int64_t)w * gs_pitch, (int64_t)w * gs_pitch + gs_pitch};
        const bool ok[4] = {true, in_x, in_y, in_x && in_y};
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 yv[4], gs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            yv[i] = ok[i] ? y[base + off[i]] : zero;
            gs[i] = (HAS_SKIP && ok[i]) ? g_skip[gbase + goff[i]] : zero;
        }
        float4 gp = zero;
        if (HAS_POOL && in_x && in_y && yo < hp && xo < wp) gp = g_pool[((img * hp + yo) * wp + xo) * c4 + c];
        float4 o[4];
        {
            float a[4], b[4], r[4];
#define POOL_LANE(F)                                                                                                   \
            a[0] = yv[0].F; a[1] = yv[1].F; a[2] = yv[2].F; a[3] = yv[3].F;                                            \
            b[0] = gs[0].F; b[1] = gs[1].F; b[2] = gs[2].F; b[3] = gs[3].F;                                            \
            pool_bwd1(a, b, gp.F, r);                                                                                  \
            o[0].F = r[0]; o[1].F = r[1]; o[2].F = r[2]; o[3].F = r[3];
            POOL_LANE(x) POOL_LANE(y) POOL_LANE(z) POOL_LANE(w)
#undef POOL_LANE
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) {
                out[base + off[i]] = o[i];
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[i].x), fabsf(o[i].y))), fmaxf(fabsf(o[i].z), fabsf(o[i].w)));
                if (!(o[i].x == o[i].x && o[i].y == o[i].y && o[i].z == o[i].z && o[i].w == o[i].w)) mx = __builtin_inff();
            }
    }
    if (out_amax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(out_amax) + (blockIdx.x & 255), __float_as_uint(mx));
    }
}

extern "C" int pcacc_maxpool2x2_f32(const float *x, int64_t n_img, int32_t h, int32_t w, int32_t c, float *out, void *stream)
{
    if (n_img < 0 || h < 2 || w < 2 || c <= 0 || (c % 4)) return PCACC_E_ARG;
    if (n_img == 0) return PCACC_OK;
    if (!x || !out) return PCACC_E_ARG;
    const int64_t total = n_img * (h / 2) * (w / 2) * (c / 4);
    maxpool2x2_f32_kernel<<<pcacc_grid(total, 256, PCACC_CUS * 16), 256, 0, pcacc_stream(stream)>>>(
        reinterpret_cast<const float4 *>(x), n_img, h, w, c / 4, reinterpret_cast<float4 *>(out));
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

static int pool_skip_bwd_f32(const float *y, const float *grad_pooled, const float *grad_skip, int64_t skip_pitch, int64_t n_img, int32_t h,
                             int32_t w, int32_t c, float *grad_y, float *out_amax, void *stream)
{
    if (n_img < 0 || h < 2 || w < 2 || c <= 0 || (c % 4) || skip_pitch < c || (skip_pitch % 4) || skip_pitch / 4 > 0x7fffffff) return PCACC_E_ARG;
    if (grad_skip && (reinterpret_cast<uintptr_t>(grad_skip) & 15)) return PCACC_E_ARG;
    const int gs_pitch = (int)(skip_pitch / 4);
    if (n_img == 0) return PCACC_OK;
    if (!y || !grad_y) return PCACC_E_ARG;
    const int64_t total = n_img * ((h + 1) / 2) * ((w + 1) / 2) * (c / 4);
    const int grid = pcacc_grid(total, 256, PCACC_CUS * 16);
    hipStream_t s = pcacc_stream(stream);
    const float4 *yy = reinterpret_cast<const float4 *>(y), *gp = reinterpret_cast<const float4 *>(grad_pooled),
                 *gs = reinterpret_cast<const float4 *>(grad_skip);
    float4 *o = reinterpret_cast<float4 *>(grad_y);
    if (gp && gs) pool_skip_relu_bwd_f32_kernel<true, true><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 4, o, out_amax, gs_pitch);
    else if (gp) pool_skip_relu_bwd_f32_kernel<true, false><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 4, o, out_amax, gs_pitch);
    else if (gs) pool_skip_relu_bwd_f32_kernel<false, true><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 4, o, out_amax, gs_pitch);
    else pool_skip_relu_bwd_f32_kernel<false, false><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 4, o, out_amax, gs_pitch);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_pool_skip_relu_backward_f32(const float *y, const float *grad_pooled, const float *grad_skip, int64_t n_img, int32_t h,
                                                 int32_t w, int32_t c, float *grad_y, float *out_amax, void *stream)
{
    return pool_skip_bwd_f32(y, grad_pooled, grad_skip, c, n_img, h, w, c, grad_y, out_amax, stream);
}

extern "C" int pcacc_pool_skip_relu_backward_strided_f32(const float *y, const float *grad_pooled, const float *grad_skip, int64_t skip_pitch,
                                                         int64_t n_img, int32_t h, int32_t w, int32_t c, float *grad_y, float *out_amax,
                                                         void *stream)
{
    return pool_skip_bwd_f32(y, grad_pooled, grad_skip, skip_pitch, n_img, h, w, c, grad_y, out_amax, stream);
}


// ---- 'mixed' compute mode: y f32 (the forward's own values), gradients bf16 -----------------------------------------------------------------
// The window's winner must be the one the fp32 forward picked.  Recomputed from a bf16 copy of y, two window values that differ by less than
// 2^-8 relative tie and the gradient goes to the first of them: about 1 % of the windows route their whole gradient element to a neighbouring
// pixel -- an error of 100 % on 1 % of the elements is a noise of sqrt(0.01) = 10 % of the gradient's norm per pooling stage (measured: +3.3 % on
// the pillar encoder's gradient norms behind four stages, i.e. ~25 % noise).  One lane per (2x2 window, 8 channels): two float4 of y, one uint4
// of each gradient per position.
template <bool HAS_POOL, bool HAS_SKIP>
__global__ __launch_bounds__(256) void pool_skip_relu_bwd_y32_kernel(const float4 *__restrict__ y, const uint4 *__restrict__ g_pool,
                                                                     const uint4 *__restrict__ g_skip, int64_t n_img, int h, int w, int c8,
                                                                     uint4 *__restrict__ out, int gs_pitch)
{
    const int h2 = (h + 1) / 2, w2 = (w + 1) / 2, hp = h / 2, wp = w / 2;
    const int64_t total = n_img * h2 * w2 * c8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c8);
        int64_t q = e / c8;
        const int xo = (int)(q % w2); q /= w2;
        const int yo = (int)(q % h2);
        const int64_t img = q / h2;
        const bool in_x = 2 * xo + 1 < w, in_y = 2 * yo + 1 < h;
        const int64_t base = ((img * h + 2 * yo) * w + 2 * xo) * c8 + c;
        const int64_t off[4] = {0, c8, (int64_t)w * c8, (int64_t)w * c8 + c8};
        const int64_t gbase = ((img * h + 2 * yo) * w + 2 * xo) * gs_pitch + c;
        const int64_t goff[4] = {0, gs_pitch, (int64_t)w * gs_pitch, (int64_t)w * gs_pitch + gs_pitch};
        const bool ok[4] = {true, in_x, in_y, in_x && in_y};
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        const float4 fzero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 ya[4], yb[4];
        uint4 gs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ya[i] = ok[i] ? y[2 * (base + off[i])] : fzero;
            yb[i] = ok[i] ? y[2 * (base + off[i]) + 1] : fzero;
            gs[i] = (HAS_SKIP && ok[i]) ? g_skip[gbase + goff[i]] : zero;
        }
        uint4 gp = zero;
        if (HAS_POOL && in_x && in_y && yo < hp && xo < wp) gp = g_pool[((img * hp + yo) * wp + xo) * c8 + c];
        uint4 o[4];
        {
            float a[4], b[4], r0[4], r1[4];
#define POOL_PAIR(F, YA, YB)                                                                                           \
            a[0] = YA(0); a[1] = YA(1); a[2] = YA(2); a[3] = YA(3);                                                    \
            b[0] = pool_lo(gs[0].F); b[1] = pool_lo(gs[1].F); b[2] = pool_lo(gs[2].F); b[3] = pool_lo(gs[3].F);        \
            pool_bwd1(a, b, pool_lo(gp.F), r0);                                                                        \
            a[0] = YB(0); a[1] = YB(1); a[2] = YB(2); a[3] = YB(3);                                                    \
            b[0] = pool_hi(gs[0].F); b[1] = pool_hi(gs[1].F); b[2] = pool_hi(gs[2].F); b[3] = pool_hi(gs[3].F);        \
            pool_bwd1(a, b, pool_hi(gp.F), r1);                                                                        \
            o[0].F = pcacc_pack_bf16x2(r0[0], r1[0]); o[1].F = pcacc_pack_bf16x2(r0[1], r1[1]);                        \
            o[2].F = pcacc_pack_bf16x2(r0[2], r1[2]); o[3].F = pcacc_pack_bf16x2(r0[3], r1[3]);
#define YAX(i) ya[i].x
#define YAY(i) ya[i].y
#define YAZ(i) ya[i].z
#define YAW(i) ya[i].w
#define YBX(i) yb[i].x
#define YBY(i) yb[i].y
#define YBZ(i) yb[i].z
#define YBW(i) yb[i].w
            POOL_PAIR(x, YAX, YAY) POOL_PAIR(y, YAZ, YAW) POOL_PAIR(z, YBX, YBY) POOL_PAIR(w, YBZ, YBW)
#undef POOL_PAIR
#undef YAX
#undef YAY
#undef YAZ
#undef YAW
#undef YBX
#undef YBY
#undef YBZ
#undef YBW
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) out[base + off[i]] = o[i];
    }
}

// y [n,h,w,c] f32; grad_pooled [n,h/2,w/2,c] / grad_skip (pixel pitch skip_pitch elements) / grad_y bf16; c % 8 == 0
extern "C" int pcacc_pool_skip_relu_backward_strided_y32(const float *y, const uint16_t *grad_pooled, const uint16_t *grad_skip, int64_t skip_pitch,
                                                         int64_t n_img, int32_t h, int32_t w, int32_t c, uint16_t *grad_y, void *stream)
{
    if (n_img < 0 || h < 2 || w < 2 || c <= 0 || (c % 8) || skip_pitch < c || (skip_pitch % 8) || skip_pitch / 8 > 0x7fffffff) return PCACC_E_ARG;
    if (grad_skip && (reinterpret_cast<uintptr_t>(grad_skip) & 15)) return PCACC_E_ARG;
    const int gs_pitch = (int)(skip_pitch / 8);
    if (n_img == 0) return PCACC_OK;
    if (!y || !grad_y) return PCACC_E_ARG;
    const int64_t total = n_img * ((h + 1) / 2) * ((w + 1) / 2) * (c / 8);
    const int grid = pcacc_grid(total, 256, PCACC_CUS * 16);
    hipStream_t s = pcacc_stream(stream);
    const float4 *yy = reinterpret_cast<const float4 *>(y);
    const uint4 *gp = reinterpret_cast<const uint4 *>(grad_pooled), *gs = reinterpret_cast<const uint4 *>(grad_skip);
    uint4 *o = reinterpret_cast<uint4 *>(grad_y);
    if (gp && gs) pool_skip_relu_bwd_y32_kernel<true, true><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 8, o, gs_pitch);
    else if (gp) pool_skip_relu_bwd_y32_kernel<true, false><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 8, o, gs_pitch);
    else if (gs) pool_skip_relu_bwd_y32_kernel<false, true><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 8, o, gs_pitch);
    else pool_skip_relu_bwd_y32_kernel<false, false><<<grid, 256, 0, s>>>(yy, gp, gs, n_img, h, w, c / 8, o, gs_pitch);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
