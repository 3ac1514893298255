// The last convolution of a SegHead2D (models/unet.py:259-277: Conv2d(mid, out_channel, 3, padding 1) with out_channel = 2 for the
// fg / bg head): 3x3, stride 1, zero padding 1, c_in = 32 or 64 channels in, 1..4 channels out, channels-last rows.  576 multiply-adds
// per pixel and 8 bytes out against 128 bytes in: an HBM / L1 stream, not a matrix-core problem -- MFMA tiles would be 94 % padding
// (the library ran it as three implicit-GEMM launches plus layout transposes).  Exact fp32 arithmetic on f32 or bf16 inputs, so the
// same three kernels serve the fp32, fp32x3 and bf16 compute modes.
//   forward : y[px][co]      = b[co] + sum_{tap, ci} x[px + tap][ci] w[co][ci][tap]
//   dgrad   : dx[px][ci]     = sum_{tap, co} dy[px - tap][co] w[co][ci][tap]
//   wgrad   : dw[co][ci][tap] = sum_px dy[px][co] x[px + tap][ci],  db[co] = sum_px dy[px][co]
// Lane layout: c_in / 4 lanes per pixel, each owning 4 consecutive input channels (one 16-byte / 8-byte load per tap: the lanes of a
// pixel read one contiguous row); a 256-thread workgroup covers 256 / (c_in / 4) consecutive pixels.
#include "common.h"

#define HC_THREADS 256
#define HC_MAXCO 4

__device__ __forceinline__ float4 hc_ld4(const void *p, int bf16, int64_t i4) { return pcacc_ld4(p, bf16 != 0, i4); }

template <int CI, int COM>
__global__ __launch_bounds__(HC_THREADS) void head_conv_fwd_kernel(const void *__restrict__ x, int x_bf16, const float *__restrict__ w,
                                                                   const float *__restrict__ bias, float *__restrict__ y, int n_img, int h,
                                                                   int wd, int co, int64_t ws_o, int64_t ws_i, int64_t ws_y, int64_t ws_x)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP;        // lanes per pixel, pixels per workgroup
    __shared__ float wl[9][CI][COM];                      // [tap][ci][co]
    for (int e = threadIdx.x; e < 9 * CI * COM; e += HC_THREADS) {
        const int c = e % COM, ci = (e / COM) % CI, tap = e / (COM * CI);
        wl[tap][ci][c] = c < co ? w[c * ws_o + ci * ws_i + (tap / 3) * ws_y + (tap % 3) * ws_x] : 0.f;
    }
    __syncthreads();
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int64_t n_px = (int64_t)n_img * h * wd;
    for (int64_t px = (int64_t)blockIdx.x * PPB + slot; px < n_px; px += (int64_t)gridDim.x * PPB) {
        const unsigned pq = (unsigned)px / (unsigned)wd;                 // 32-bit: n_img * h * w < 2^31 (checked by the host)
        const int xx = (int)((unsigned)px - pq * (unsigned)wd), yy = (int)(pq % (unsigned)h);
        float acc[COM];
#pragma unroll
        for (int c = 0; c < COM; ++c) acc[c] = 0.f;
        float4 v[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // always load (from the pixel itself when the neighbour is outside), then select: loads under a branch are issued and
            // waited for one at a time (9 serial memory latencies per pixel: 185 us for this launch, 743 us for the weight gradient)
            const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
            const bool ok = (unsigned)y2 < (unsigned)h && (unsigned)x2 < (unsigned)wd;
            v[tap] = hc_ld4(x, x_bf16, (ok ? px + (int64_t)(tap / 3 - 1) * wd + (tap % 3 - 1) : px) * LPP + l);
            if (!ok) v[tap] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float *wt = &wl[tap][4 * l][0];
#pragma unroll
            for (int c = 0; c < COM; ++c)
                acc[c] += v[tap].x * wt[c] + v[tap].y * wt[COM + c] + v[tap].z * wt[2 * COM + c] + v[tap].w * wt[3 * COM + c];
        }
#pragma unroll
        for (int d = 1; d < LPP; d <<= 1)
#pragma unroll
            for (int c = 0; c < COM; ++c) acc[c] += __shfl_xor(acc[c], d, 64);
        if (l < co) {
            float r = acc[0];
#pragma unroll
            for (int c = 1; c < COM; ++c) r = l == c ? acc[c] : r;
            y[px * co + l] = r + (bias ? bias[l] : 0.f);
        }
    }
}

template <int CI, int COM>
__global__ __launch_bounds__(HC_THREADS) void head_conv_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w, void *__restrict__ dx,
                                                                     int dx_bf16, int n_img, int h, int wd, int co, int64_t ws_o, int64_t ws_i,
                                                                     int64_t ws_y, int64_t ws_x)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP;
    __shared__ float wl[9][COM][CI];                      // [tap][co][ci]
    for (int e = threadIdx.x; e < 9 * COM * CI; e += HC_THREADS) {
        const int ci = e % CI, c = (e / CI) % COM, tap = e / (CI * COM);
        wl[tap][c][ci] = c < co ? w[c * ws_o + ci * ws_i + (tap / 3) * ws_y + (tap % 3) * ws_x] : 0.f;
    }
    __syncthreads();
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int64_t n_px = (int64_t)n_img * h * wd;
    for (int64_t px = (int64_t)blockIdx.x * PPB + slot; px < n_px; px += (int64_t)gridDim.x * PPB) {
        const unsigned pq = (unsigned)px / (unsigned)wd;                 // 32-bit: n_img * h * w < 2^31 (checked by the host)
        const int xx = (int)((unsigned)px - pq * (unsigned)wd), yy = (int)(pq % (unsigned)h);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float gv[9][COM];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {                    // output pixel px - tap offset saw this pixel through tap `tap`
            const int y2 = yy - (tap / 3 - 1), x2 = xx - (tap % 3 - 1);
            const bool ok = (unsigned)y2 < (unsigned)h && (unsigned)x2 < (unsigned)wd;
            const float *g = dy + (ok ? px - (int64_t)(tap / 3 - 1) * wd - (tap % 3 - 1) : px) * co;      // unconditional loads, see the forward kernel
#pragma unroll
            for (int c = 0; c < COM; ++c) gv[tap][c] = (c < co && ok) ? g[c < co ? c : 0] : 0.f;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < COM; ++c) {
                const float4 wv = *reinterpret_cast<const float4 *>(&wl[tap][c][4 * l]);
                acc.x += gv[tap][c] * wv.x; acc.y += gv[tap][c] * wv.y; acc.z += gv[tap][c] * wv.z; acc.w += gv[tap][c] * wv.w;
            }
        pcacc_st4(dx, dx_bf16 != 0, px * LPP + l, acc);
    }
}

// per-workgroup partial sums of dw (+ db) go to a workspace slot; a second launch sums the slots in a fixed order (one atomic per element
// and workgroup from ~10^3 workgroups onto 578 words serialised in L2: 707 us for this launch)
template <int CI, int COM>
__global__ __launch_bounds__(HC_THREADS) void head_conv_wgrad_kernel(const float *__restrict__ dy, const void *__restrict__ x, int x_bf16,
                                                                     float *__restrict__ partial, int n_img, int h, int wd, int co)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP, SPW = 64 / LPP;      // pixel slots per wave
    __shared__ float red[HC_THREADS / 64][LPP][9 * COM * 4 + COM];
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP, wave = threadIdx.x >> 6;
    const int64_t n_px = (int64_t)n_img * h * wd;
    float acc[9][COM][4], bs[COM];
#pragma unroll
    for (int c = 0; c < COM; ++c) bs[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COM; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][c][q] = 0.f;
    for (int64_t px = (int64_t)blockIdx.x * PPB + slot; px < n_px; px += (int64_t)gridDim.x * PPB) {
        const unsigned pq = (unsigned)px / (unsigned)wd;                 // 32-bit: n_img * h * w < 2^31 (checked by the host)
        const int xx = (int)((unsigned)px - pq * (unsigned)wd), yy = (int)(pq % (unsigned)h);
        float g[COM];
        float4 v[9];
#pragma unroll
        for (int c = 0; c < COM; ++c) g[c] = c < co ? dy[px * co + c] : 0.f;
#pragma unroll
        for (int c = 0; c < COM; ++c) bs[c] += g[c];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
            const bool ok = (unsigned)y2 < (unsigned)h && (unsigned)x2 < (unsigned)wd;
            v[tap] = hc_ld4(x, x_bf16, (ok ? px + (int64_t)(tap / 3 - 1) * wd + (tap % 3 - 1) : px) * LPP + l);      // unconditional, see the forward kernel
            if (!ok) v[tap] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < COM; ++c) {
                acc[tap][c][0] += g[c] * v[tap].x; acc[tap][c][1] += g[c] * v[tap].y; acc[tap][c][2] += g[c] * v[tap].z; acc[tap][c][3] += g[c] * v[tap].w;
            }
    }
    // the pixel slots of a wave (lanes l, l + LPP, ...) hold partial sums of the same outputs
#pragma unroll
    for (int d = LPP; d < 64; d <<= 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < COM; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[t][c][q] += __shfl_xor(acc[t][c][q], d, 64);
#pragma unroll
        for (int c = 0; c < COM; ++c) bs[c] += __shfl_xor(bs[c], d, 64);
    }
    (void)SPW;
    if ((threadIdx.x & 63) < LPP) {
        float *r = red[wave][l];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < COM; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) r[(t * COM + c) * 4 + q] = acc[t][c][q];
#pragma unroll
        for (int c = 0; c < COM; ++c) r[9 * COM * 4 + c] = bs[c];
    }
    __syncthreads();
    // slot of this workgroup: dw [co][ci][3][3] (element (c, ci = 4 l + q, tap)) followed by db [co]
    float *mine = partial + (int64_t)blockIdx.x * (co * CI * 9 + co);
    for (int e = threadIdx.x; e < co * CI * 9; e += HC_THREADS) {
        const int tap = e % 9, ci = (e / 9) % CI, c = e / (9 * CI);
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < HC_THREADS / 64; ++wv) s += red[wv][ci / 4][(tap * COM + c) * 4 + (ci & 3)];
        mine[e] = s;
    }
    if (threadIdx.x < co) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < HC_THREADS / 64; ++wv) s += red[wv][0][9 * COM * 4 + threadIdx.x];      // every lane group saw every pixel's dy: take group 0
        mine[co * CI * 9 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void head_conv_wgrad_reduce_kernel(const float *__restrict__ partial, int n_parts, int n_w, int co,
                                                                     float *__restrict__ dw, float *__restrict__ db)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6, elems = n_w + co;
    __shared__ float sm[4][64];
    float s0 = 0.f, s1 = 0.f;
    if (e < elems) {
        int p = grp;
        for (; p + 4 < n_parts; p += 8) {
            s0 += partial[(int64_t)p * elems + e];
            s1 += partial[(int64_t)(p + 4) * elems + e];
        }
        if (p < n_parts) s0 += partial[(int64_t)p * elems + e];
    }
    sm[grp][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (grp == 0 && e < elems) {
        const float t = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
        if (e < n_w) dw[e] = t;
        else if (db) db[e - n_w] = t;
    }
}

static bool hc_ok(int n_img, int h, int w, int c_in, int c_out)
{
    return n_img >= 1 && h >= 1 && w >= 1 && (int64_t)n_img * h * w < 0x7fffffffLL && (c_in == 32 || c_in == 64) && c_out >= 1 && c_out <= HC_MAXCO;
}

extern "C" int pcacc_head_conv3x3_supported(int32_t c_in, int32_t c_out) { return hc_ok(1, 1, 1, c_in, c_out) ? 1 : 0; }

// x [n,h,w,c_in] f32 (x_dtype 0) or bf16 (1); w f32 [c_out][c_in][3][3] through strides (host, elements: o, i, y, x); y [n,h,w,c_out] f32
extern "C" int pcacc_head_conv3x3_forward(const void *x, int32_t x_dtype, const float *w, const int64_t *w_strides, const float *bias, float *y,
                                          int32_t n_img, int32_t h, int32_t wd, int32_t c_in, int32_t c_out, void *stream)
{
    if (!x || !w || !w_strides || !y || !hc_ok(n_img, h, wd, c_in, c_out)) return PCACC_E_ARG;
    const int64_t n_px = (int64_t)n_img * h * wd;
    const int ppb = HC_THREADS / (c_in / 4);
    const int grid = pcacc_grid(n_px, ppb, PCACC_CUS * 8);
#define HC_FWD(CIV, COV) hipLaunchKernelGGL((head_conv_fwd_kernel<CIV, COV>), dim3(grid), dim3(HC_THREADS), 0, pcacc_stream(stream), x, x_dtype, w, bias, y, n_img, \
                                            h, wd, c_out, w_strides[0], w_strides[1], w_strides[2], w_strides[3])
    if (c_in == 32 && c_out <= 2) HC_FWD(32, 2); else if (c_in == 32) HC_FWD(32, 4); else if (c_out <= 2) HC_FWD(64, 2); else HC_FWD(64, 4);
#undef HC_FWD
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// dy [n,h,w,c_out] f32 -> dx [n,h,w,c_in] f32 / bf16
extern "C" int pcacc_head_conv3x3_dgrad(const float *dy, const float *w, const int64_t *w_strides, void *dx, int32_t dx_dtype, int32_t n_img, int32_t h,
                                        int32_t wd, int32_t c_in, int32_t c_out, void *stream)
{
    if (!dy || !w || !w_strides || !dx || !hc_ok(n_img, h, wd, c_in, c_out)) return PCACC_E_ARG;
    const int64_t n_px = (int64_t)n_img * h * wd;
    const int ppb = HC_THREADS / (c_in / 4);
    const int grid = pcacc_grid(n_px, ppb, PCACC_CUS * 8);
#define HC_DG(CIV, COV) hipLaunchKernelGGL((head_conv_dgrad_kernel<CIV, COV>), dim3(grid), dim3(HC_THREADS), 0, pcacc_stream(stream), dy, w, dx, dx_dtype, n_img, \
                                           h, wd, c_out, w_strides[0], w_strides[1], w_strides[2], w_strides[3])
    if (c_in == 32 && c_out <= 2) HC_DG(32, 2); else if (c_in == 32) HC_DG(32, 4); else if (c_out <= 2) HC_DG(64, 2); else HC_DG(64, 4);
#undef HC_DG
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

static int hc_wgrad_grid(int64_t n_px, int c_in) { return pcacc_grid(n_px, HC_THREADS / (c_in / 4), PCACC_CUS * 3); }

extern "C" int pcacc_head_conv3x3_wgrad_workspace_bytes(int32_t n_img, int32_t h, int32_t wd, int32_t c_in, int32_t c_out, size_t *bytes)
{
    if (!bytes || !hc_ok(n_img, h, wd, c_in, c_out)) return PCACC_E_ARG;
    *bytes = (size_t)hc_wgrad_grid((int64_t)n_img * h * wd, c_in) * (c_out * c_in * 9 + c_out) * sizeof(float);
    return PCACC_OK;
}

// dw [c_out][c_in][3][3] f32 contiguous and db [c_out] f32 (NULL = not wanted)
extern "C" int pcacc_head_conv3x3_wgrad(const float *dy, const void *x, int32_t x_dtype, float *dw, float *db, int32_t n_img, int32_t h, int32_t wd,
                                        int32_t c_in, int32_t c_out, void *workspace, size_t workspace_bytes, void *stream)
{
    size_t need;
    if (!dy || !x || !dw || !workspace || pcacc_head_conv3x3_wgrad_workspace_bytes(n_img, h, wd, c_in, c_out, &need) != PCACC_OK) return PCACC_E_ARG;
    if (workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t st = pcacc_stream(stream);
    const int grid = hc_wgrad_grid((int64_t)n_img * h * wd, c_in);
    float *partial = static_cast<float *>(workspace);
#define HC_WG(CIV, COV) hipLaunchKernelGGL((head_conv_wgrad_kernel<CIV, COV>), dim3(grid), dim3(HC_THREADS), 0, st, dy, x, x_dtype, partial, n_img, h, wd, c_out)
    if (c_in == 32 && c_out <= 2) HC_WG(32, 2); else if (c_in == 32) HC_WG(32, 4); else if (c_out <= 2) HC_WG(64, 2); else HC_WG(64, 4);
#undef HC_WG
    const int n_w = c_out * c_in * 9;
    hipLaunchKernelGGL(head_conv_wgrad_reduce_kernel, dim3((n_w + c_out + 63) / 64), dim3(256), 0, st, partial, grid, n_w, c_out, dw, db);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
