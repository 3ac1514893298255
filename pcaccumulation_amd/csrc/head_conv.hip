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

template <int CI>
__global__ __launch_bounds__(HC_THREADS) void head_conv_fwd_kernel(const void *__restrict__ x, int x_bf16, const float *__restrict__ w,
                                                                   const float *__restrict__ bias, float *__restrict__ y, int n_img, int h,
                                                                   int wd, int co, int64_t ws_o, int64_t ws_i, int64_t ws_y, int64_t ws_x)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP;        // lanes per pixel, pixels per workgroup
    __shared__ float wl[9][CI][HC_MAXCO];                      // [tap][ci][co]
    for (int e = threadIdx.x; e < 9 * CI * HC_MAXCO; e += HC_THREADS) {
        const int c = e % HC_MAXCO, ci = (e / HC_MAXCO) % CI, tap = e / (HC_MAXCO * CI);
        wl[tap][ci][c] = c < co ? w[c * ws_o + ci * ws_i + (tap / 3) * ws_y + (tap % 3) * ws_x] : 0.f;
    }
    __syncthreads();
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int64_t n_px = (int64_t)n_img * h * wd;
    for (int64_t px = (int64_t)blockIdx.x * PPB + slot; px < n_px; px += (int64_t)gridDim.x * PPB) {
        const int xx = (int)(px % wd), yy = (int)((px / wd) % h);
        float acc[HC_MAXCO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
            if ((unsigned)y2 < (unsigned)h && (unsigned)x2 < (unsigned)wd) {
                const float4 v = hc_ld4(x, x_bf16, (px + (int64_t)(tap / 3 - 1) * wd + (tap % 3 - 1)) * LPP + l);
                const float *wt = &wl[tap][4 * l][0];
#pragma unroll
                for (int c = 0; c < HC_MAXCO; ++c) acc[c] += v.x * wt[c] + v.y * wt[HC_MAXCO + c] + v.z * wt[2 * HC_MAXCO + c] + v.w * wt[3 * HC_MAXCO + c];
            }
        }
#pragma unroll
        for (int d = 1; d < LPP; d <<= 1)
#pragma unroll
            for (int c = 0; c < HC_MAXCO; ++c) acc[c] += __shfl_xor(acc[c], d, 64);
        if (l < co) y[px * co + l] = (l == 0 ? acc[0] : l == 1 ? acc[1] : l == 2 ? acc[2] : acc[3]) + (bias ? bias[l] : 0.f);
    }
}

template <int CI>
__global__ __launch_bounds__(HC_THREADS) void head_conv_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w, void *__restrict__ dx,
                                                                     int dx_bf16, int n_img, int h, int wd, int co, int64_t ws_o, int64_t ws_i,
                                                                     int64_t ws_y, int64_t ws_x)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP;
    __shared__ float wl[9][HC_MAXCO][CI];                      // [tap][co][ci]
    for (int e = threadIdx.x; e < 9 * HC_MAXCO * CI; e += HC_THREADS) {
        const int ci = e % CI, c = (e / CI) % HC_MAXCO, tap = e / (CI * HC_MAXCO);
        wl[tap][c][ci] = c < co ? w[c * ws_o + ci * ws_i + (tap / 3) * ws_y + (tap % 3) * ws_x] : 0.f;
    }
    __syncthreads();
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int64_t n_px = (int64_t)n_img * h * wd;
    for (int64_t px = (int64_t)blockIdx.x * PPB + slot; px < n_px; px += (int64_t)gridDim.x * PPB) {
        const int xx = (int)(px % wd), yy = (int)((px / wd) % h);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {                    // output pixel px - tap offset saw this pixel through tap `tap`
            const int y2 = yy - (tap / 3 - 1), x2 = xx - (tap % 3 - 1);
            if ((unsigned)y2 < (unsigned)h && (unsigned)x2 < (unsigned)wd) {
                const float *g = dy + (px - (int64_t)(tap / 3 - 1) * wd - (tap % 3 - 1)) * co;
                for (int c = 0; c < co; ++c) {
                    const float gv = g[c];
                    const float4 wv = *reinterpret_cast<const float4 *>(&wl[tap][c][4 * l]);
                    acc.x += gv * wv.x; acc.y += gv * wv.y; acc.z += gv * wv.z; acc.w += gv * wv.w;
                }
            }
        }
        pcacc_st4(dx, dx_bf16 != 0, px * LPP + l, acc);
    }
}

// per-workgroup partial sums of dw (+ db), added to the zero-filled result with one atomic per element and workgroup
template <int CI>
__global__ __launch_bounds__(HC_THREADS) void head_conv_wgrad_kernel(const float *__restrict__ dy, const void *__restrict__ x, int x_bf16,
                                                                     float *__restrict__ dw, float *__restrict__ db, int n_img, int h, int wd,
                                                                     int co)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP, SPW = 64 / LPP;      // pixel slots per wave
    __shared__ float red[HC_THREADS / 64][LPP][9 * HC_MAXCO * 4 + HC_MAXCO];
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP, wave = threadIdx.x >> 6;
    const int64_t n_px = (int64_t)n_img * h * wd;
    float acc[9][HC_MAXCO][4], bs[HC_MAXCO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < HC_MAXCO; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][c][q] = 0.f;
    for (int64_t px = (int64_t)blockIdx.x * PPB + slot; px < n_px; px += (int64_t)gridDim.x * PPB) {
        const int xx = (int)(px % wd), yy = (int)((px / wd) % h);
        float g[HC_MAXCO];
#pragma unroll
        for (int c = 0; c < HC_MAXCO; ++c) g[c] = c < co ? dy[px * co + c] : 0.f;
#pragma unroll
        for (int c = 0; c < HC_MAXCO; ++c) bs[c] += g[c];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
            if ((unsigned)y2 < (unsigned)h && (unsigned)x2 < (unsigned)wd) {
                const float4 v = hc_ld4(x, x_bf16, (px + (int64_t)(tap / 3 - 1) * wd + (tap % 3 - 1)) * LPP + l);
#pragma unroll
                for (int c = 0; c < HC_MAXCO; ++c) {
                    acc[tap][c][0] += g[c] * v.x; acc[tap][c][1] += g[c] * v.y; acc[tap][c][2] += g[c] * v.z; acc[tap][c][3] += g[c] * v.w;
                }
            }
        }
    }
    // the pixel slots of a wave (lanes l, l + LPP, ...) hold partial sums of the same outputs
#pragma unroll
    for (int d = LPP; d < 64; d <<= 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < HC_MAXCO; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[t][c][q] += __shfl_xor(acc[t][c][q], d, 64);
#pragma unroll
        for (int c = 0; c < HC_MAXCO; ++c) bs[c] += __shfl_xor(bs[c], d, 64);
    }
    (void)SPW;
    if ((threadIdx.x & 63) < LPP) {
        float *r = red[wave][l];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < HC_MAXCO; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) r[(t * HC_MAXCO + c) * 4 + q] = acc[t][c][q];
#pragma unroll
        for (int c = 0; c < HC_MAXCO; ++c) r[9 * HC_MAXCO * 4 + c] = bs[c];
    }
    __syncthreads();
    // dw [co][ci][3][3] (contiguous): element (c, ci = 4 l + q, tap)
    for (int e = threadIdx.x; e < co * CI * 9; e += HC_THREADS) {
        const int tap = e % 9, ci = (e / 9) % CI, c = e / (9 * CI);
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < HC_THREADS / 64; ++wv) s += red[wv][ci / 4][(tap * HC_MAXCO + c) * 4 + (ci & 3)];
        atomicAdd(&dw[e], s);
    }
    if (threadIdx.x < co && db) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < HC_THREADS / 64; ++wv) s += red[wv][0][9 * HC_MAXCO * 4 + threadIdx.x];      // every lane group saw every pixel's dy: take group 0
        atomicAdd(&db[threadIdx.x], s);
    }
}

static bool hc_ok(int n_img, int h, int w, int c_in, int c_out) { return n_img >= 1 && h >= 1 && w >= 1 && (c_in == 32 || c_in == 64) && c_out >= 1 && c_out <= HC_MAXCO; }

extern "C" int pcacc_head_conv3x3_supported(int32_t c_in, int32_t c_out) { return hc_ok(1, 1, 1, c_in, c_out) ? 1 : 0; }

// x [n,h,w,c_in] f32 (x_dtype 0) or bf16 (1); w f32 [c_out][c_in][3][3] through strides (host, elements: o, i, y, x); y [n,h,w,c_out] f32
extern "C" int pcacc_head_conv3x3_forward(const void *x, int32_t x_dtype, const float *w, const int64_t *w_strides, const float *bias, float *y,
                                          int32_t n_img, int32_t h, int32_t wd, int32_t c_in, int32_t c_out, void *stream)
{
    if (!x || !w || !w_strides || !y || !hc_ok(n_img, h, wd, c_in, c_out)) return PCACC_E_ARG;
    const int64_t n_px = (int64_t)n_img * h * wd;
    const int ppb = HC_THREADS / (c_in / 4);
    const int grid = pcacc_grid(n_px, ppb, PCACC_CUS * 8);
    if (c_in == 32)
        hipLaunchKernelGGL(head_conv_fwd_kernel<32>, dim3(grid), dim3(HC_THREADS), 0, pcacc_stream(stream), x, x_dtype, w, bias, y, n_img, h, wd, c_out,
                           w_strides[0], w_strides[1], w_strides[2], w_strides[3]);
    else
        hipLaunchKernelGGL(head_conv_fwd_kernel<64>, dim3(grid), dim3(HC_THREADS), 0, pcacc_stream(stream), x, x_dtype, w, bias, y, n_img, h, wd, c_out,
                           w_strides[0], w_strides[1], w_strides[2], w_strides[3]);
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
    if (c_in == 32)
        hipLaunchKernelGGL(head_conv_dgrad_kernel<32>, dim3(grid), dim3(HC_THREADS), 0, pcacc_stream(stream), dy, w, dx, dx_dtype, n_img, h, wd, c_out,
                           w_strides[0], w_strides[1], w_strides[2], w_strides[3]);
    else
        hipLaunchKernelGGL(head_conv_dgrad_kernel<64>, dim3(grid), dim3(HC_THREADS), 0, pcacc_stream(stream), dy, w, dx, dx_dtype, n_img, h, wd, c_out,
                           w_strides[0], w_strides[1], w_strides[2], w_strides[3]);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// dw [c_out][c_in][3][3] f32 contiguous and db [c_out] f32 (NULL = not wanted): cleared here, then accumulated
extern "C" int pcacc_head_conv3x3_wgrad(const float *dy, const void *x, int32_t x_dtype, float *dw, float *db, int32_t n_img, int32_t h, int32_t wd,
                                        int32_t c_in, int32_t c_out, void *stream)
{
    if (!dy || !x || !dw || !hc_ok(n_img, h, wd, c_in, c_out)) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    if (hipMemsetAsync(dw, 0, (size_t)c_out * c_in * 9 * sizeof(float), st) != hipSuccess) return PCACC_E_LAUNCH;
    if (db && hipMemsetAsync(db, 0, (size_t)c_out * sizeof(float), st) != hipSuccess) return PCACC_E_LAUNCH;
    const int64_t n_px = (int64_t)n_img * h * wd;
    const int ppb = HC_THREADS / (c_in / 4);
    const int grid = pcacc_grid(n_px, ppb, PCACC_CUS * 4);
    if (c_in == 32)
        hipLaunchKernelGGL(head_conv_wgrad_kernel<32>, dim3(grid), dim3(HC_THREADS), 0, st, dy, x, x_dtype, dw, db, n_img, h, wd, c_out);
    else
        hipLaunchKernelGGL(head_conv_wgrad_kernel<64>, dim3(grid), dim3(HC_THREADS), 0, st, dy, x, x_dtype, dw, db, n_img, h, wd, c_out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
