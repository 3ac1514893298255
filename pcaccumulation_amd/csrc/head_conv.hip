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
#include <type_traits>
#include "common.h"

#define HC_THREADS 256
#define HC_MAXCO 4

__device__ __forceinline__ float4 hc_ld4(const void *p, int bf16, int64_t i4) { return pcacc_ld4(p, bf16 != 0, i4); }

// forward / weight gradient: a workgroup walks 8 x 32-pixel tiles; a tile's input patch (10 x 34 pixels, converted to f32) is staged in LDS
// once -- read straight from global memory, every input row went through L1 three times and through L2 about three times (the rows above
// and below belong to other workgroups): 252 us forward / 240 us weight gradient for a 212 MB map.
#define HC_TR 8
#define HC_TC 32
#define HC_PW (HC_TC + 2)
#define HC_PH (HC_TR + 2)

struct HcTile { int img, y0, x0; };
__device__ __forceinline__ HcTile hc_tile(int t, int tiles_y, int tiles_x)
{
    const int per_img = tiles_y * tiles_x, img = t / per_img, r = t - img * per_img;
    return HcTile{img, (r / tiles_x) * HC_TR, (r % tiles_x) * HC_TC};
}

template <int CI>
__device__ __forceinline__ void hc_stage(const void *__restrict__ x, int x_bf16, const HcTile &t, int h, int wd, float *patch)
{
    constexpr int LPP = CI / 4;
    for (int e = threadIdx.x; e < HC_PH * HC_PW * LPP; e += HC_THREADS) {
        const int pp = e / LPP, l4 = e - pp * LPP;
        const int py = pp / HC_PW, pxx = pp - py * HC_PW;
        const int y = t.y0 - 1 + py, xx = t.x0 - 1 + pxx;
        const bool ok = (unsigned)y < (unsigned)h && (unsigned)xx < (unsigned)wd;
        const int yc = min(max(y, 0), h - 1), xc = min(max(xx, 0), wd - 1);                     // always load, then select (see the data gradient)
        float4 v = hc_ld4(x, x_bf16, (((int64_t)t.img * h + yc) * wd + xc) * LPP + l4);
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(patch + pp * CI + l4 * 4) = v;
    }
}

// [r5] the same staging in two halves, so that the NEXT tile's rows travel in registers while the current tile is computed (forward and weight gradient
// were load -> barrier -> compute -> barrier per tile, the load latency covered only by the two other workgroups of the CU): hc_fetch issues a thread's
// loads back to back from clamped positions (raw element type: 8 bytes per piece for bf16 rows), hc_put converts, zeroes what lies outside and stores.
template <int CI, bool BF> struct HcRaw { typedef typename std::conditional<BF, uint2, float4>::type T; };
template <int CI> struct HcPieces { static constexpr int N = (HC_PH * HC_PW * (CI / 4) + HC_THREADS - 1) / HC_THREADS; };

template <int CI, bool BF>
__device__ __forceinline__ void hc_fetch(const void *__restrict__ x, const HcTile &t, int h, int wd, typename HcRaw<CI, BF>::T (&r)[HcPieces<CI>::N])
{
    constexpr int LPP = CI / 4;
    const typename HcRaw<CI, BF>::T *src = static_cast<const typename HcRaw<CI, BF>::T *>(x);
#pragma unroll
    for (int q = 0; q < HcPieces<CI>::N; ++q) {
        const int e = min((int)threadIdx.x + q * HC_THREADS, HC_PH * HC_PW * LPP - 1);
        const int pp = e / LPP, l4 = e - pp * LPP;
        const int py = pp / HC_PW, pxx = pp - py * HC_PW;
        const int yc = min(max(t.y0 - 1 + py, 0), h - 1), xc = min(max(t.x0 - 1 + pxx, 0), wd - 1);
        r[q] = src[(((int64_t)t.img * h + yc) * wd + xc) * LPP + l4];
    }
}

template <int CI, bool BF>
__device__ __forceinline__ void hc_put(const typename HcRaw<CI, BF>::T (&r)[HcPieces<CI>::N], const HcTile &t, int h, int wd, float *patch)
{
    constexpr int LPP = CI / 4;
#pragma unroll
    for (int q = 0; q < HcPieces<CI>::N; ++q) {
        const int e = threadIdx.x + q * HC_THREADS;
        if (e >= HC_PH * HC_PW * LPP) break;
        const int pp = e / LPP, l4 = e - pp * LPP;
        const int py = pp / HC_PW, pxx = pp - py * HC_PW;
        const int y = t.y0 - 1 + py, xx = t.x0 - 1 + pxx;
        const bool ok = (unsigned)y < (unsigned)h && (unsigned)xx < (unsigned)wd;
        float4 v;
        if constexpr (BF) v = make_float4(pcacc_bf16_lo(r[q].x), pcacc_bf16_hi(r[q].x), pcacc_bf16_lo(r[q].y), pcacc_bf16_hi(r[q].y));
        else v = r[q];
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(patch + pp * CI + l4 * 4) = v;
    }
}

template <int CI, int COM, bool BF>
__global__ __launch_bounds__(HC_THREADS) void head_conv_fwd_kernel(const void *__restrict__ x, int x_bf16, const float *__restrict__ w,
                                                                   const float *__restrict__ bias, float *__restrict__ y, int n_img, int h,
                                                                   int wd, int co, int64_t ws_o, int64_t ws_i, int64_t ws_y, int64_t ws_x,
                                                                   int tiles_y, int tiles_x)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP;        // lanes per pixel, pixels per pass
    extern __shared__ __attribute__((aligned(16))) float hc_lds[];
    float *patch = hc_lds;                                     // [HC_PH][HC_PW][CI]
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    float4 wr[9][COM];                                         // w[c][4 l .. 4 l + 3][tap]: the lane's weights stay in registers
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < COM; ++c) {
            const float *wp = w + c * ws_o + (4 * l) * ws_i + (tap / 3) * ws_y + (tap % 3) * ws_x;
            wr[tap][c] = c < co ? make_float4(wp[0], wp[ws_i], wp[2 * ws_i], wp[3 * ws_i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    const int n_tiles = n_img * tiles_y * tiles_x;
    constexpr bool PIPE = CI == 32;                            // 64 channels: 22 pieces per thread in flight would cost the kernel its occupancy -- staged in place
    typename HcRaw<CI, BF>::T raw[HcPieces<CI>::N];
    if (PIPE && (int)blockIdx.x < n_tiles) hc_fetch<CI, BF>(x, hc_tile(blockIdx.x, tiles_y, tiles_x), h, wd, raw);
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const HcTile tl = hc_tile(t, tiles_y, tiles_x);
        __syncthreads();                                       // the previous tile's readers are done
        if constexpr (PIPE) hc_put<CI, BF>(raw, tl, h, wd, patch); else hc_stage<CI>(x, x_bf16, tl, h, wd, patch);
        __syncthreads();
        if (PIPE && t + (int)gridDim.x < n_tiles) hc_fetch<CI, BF>(x, hc_tile(t + gridDim.x, tiles_y, tiles_x), h, wd, raw);
        for (int q = slot; q < HC_TR * HC_TC; q += PPB) {
            const int row = q / HC_TC, col = q % HC_TC;
            float acc[COM];
#pragma unroll
            for (int c = 0; c < COM; ++c) acc[c] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float4 v = *reinterpret_cast<const float4 *>(patch + ((row + tap / 3) * HC_PW + col + tap % 3) * CI + 4 * l);
#pragma unroll
                for (int c = 0; c < COM; ++c) acc[c] += v.x * wr[tap][c].x + v.y * wr[tap][c].y + v.z * wr[tap][c].z + v.w * wr[tap][c].w;
            }
#pragma unroll
            for (int d = 1; d < LPP; d <<= 1)
#pragma unroll
                for (int c = 0; c < COM; ++c) acc[c] += __shfl_xor(acc[c], d, 64);
            const int yy = tl.y0 + row, xx = tl.x0 + col;
            if (l < co && yy < h && xx < wd) {
                float r = acc[0];
#pragma unroll
                for (int c = 1; c < COM; ++c) r = l == c ? acc[c] : r;
                y[(((int64_t)tl.img * h + yy) * wd + xx) * co + l] = r + (bias ? bias[l] : 0.f);
            }
        }
    }
}

// data gradient: 8 bytes in, 128 bytes out per pixel.  The lane's 9 x c_out x 4 weights live in registers; a tile's dy patch (10 x 34 pixels x c_out)
// is staged in LDS, so the nine reads per pixel are LDS broadcasts instead of dependent global loads
template <int CI, int COM>
__global__ __launch_bounds__(HC_THREADS) void head_conv_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w, void *__restrict__ dx,
                                                                     int dx_bf16, int n_img, int h, int wd, int co, int64_t ws_o, int64_t ws_i,
                                                                     int64_t ws_y, int64_t ws_x, int tiles_y, int tiles_x)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP;
    __shared__ float gp[HC_PH * HC_PW * COM];
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    float4 wr[9][COM];                                         // w[c][4 l .. 4 l + 3][tap]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < COM; ++c) {
            const float *wp = w + c * ws_o + (4 * l) * ws_i + (tap / 3) * ws_y + (tap % 3) * ws_x;
            wr[tap][c] = c < co ? make_float4(wp[0], wp[ws_i], wp[2 * ws_i], wp[3 * ws_i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    const int n_tiles = n_img * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const HcTile tl = hc_tile(t, tiles_y, tiles_x);
        __syncthreads();
        for (int e = threadIdx.x; e < HC_PH * HC_PW * COM; e += HC_THREADS) {
            const int pp = e / COM, c = e - pp * COM;
            const int py = pp / HC_PW, pxx = pp - py * HC_PW;
            const int y = tl.y0 - 1 + py, xx = tl.x0 - 1 + pxx;
            const bool ok = (unsigned)y < (unsigned)h && (unsigned)xx < (unsigned)wd && c < co;
            const int yc = min(max(y, 0), h - 1), xc = min(max(xx, 0), wd - 1);
            const float v = dy[(((int64_t)tl.img * h + yc) * wd + xc) * co + (c < co ? c : 0)];       // always load, then select
            gp[e] = ok ? v : 0.f;
        }
        __syncthreads();
        for (int q = slot; q < HC_TR * HC_TC; q += PPB) {
            const int row = q / HC_TC, col = q % HC_TC;
            const int yy = tl.y0 + row, xx = tl.x0 + col;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {                // output pixel (this - tap offset) saw this pixel through tap `tap`
                const float *g = gp + ((row + 2 - tap / 3) * HC_PW + col + 2 - tap % 3) * COM;
#pragma unroll
                for (int c = 0; c < COM; ++c) {
                    const float gv = g[c];
                    acc.x += gv * wr[tap][c].x; acc.y += gv * wr[tap][c].y; acc.z += gv * wr[tap][c].z; acc.w += gv * wr[tap][c].w;
                }
            }
            if (yy < h && xx < wd) pcacc_st4(dx, dx_bf16 != 0, (((int64_t)tl.img * h + yy) * wd + xx) * LPP + l, acc);
        }
    }
}

// per-workgroup partial sums of dw (+ db) go to a workspace slot; a second launch sums the slots in a fixed order (one atomic per element
// and workgroup from ~10^3 workgroups onto 578 words serialised in L2: 707 us for this launch)
template <int CI, int COM, bool BF>
__global__ __launch_bounds__(HC_THREADS) void head_conv_wgrad_kernel(const float *__restrict__ dy, const void *__restrict__ x, int x_bf16,
                                                                     float *__restrict__ partial, int n_img, int h, int wd, int co, int tiles_y,
                                                                     int tiles_x)
{
    constexpr int LPP = CI / 4, PPB = HC_THREADS / LPP;
    extern __shared__ __attribute__((aligned(16))) float hc_lds[];
    float *patch = hc_lds;                                     // [HC_PH][HC_PW][CI]; afterwards the cross-wave reduction [4][LPP][9 COM 4 + COM]
    constexpr int RED = 9 * COM * 4 + COM;
    static_assert(4 * LPP * RED <= HC_PH * HC_PW * CI, "reduction scratch fits the patch");
    const int l = threadIdx.x % LPP, slot = threadIdx.x / LPP, wave = threadIdx.x >> 6;
    float acc[9][COM][4], bs[COM];
#pragma unroll
    for (int c = 0; c < COM; ++c) bs[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COM; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][c][q] = 0.f;
    const int n_tiles = n_img * tiles_y * tiles_x;
    constexpr bool PIPE = CI == 32 && COM == 4;                // measured: 32 -> 4 272 -> 196 us; 32 -> 2 loses a workgroup per CU to the extra registers (121 -> 135 us): staged in place
    typename HcRaw<CI, BF>::T raw[HcPieces<CI>::N];
    if (PIPE && (int)blockIdx.x < n_tiles) hc_fetch<CI, BF>(x, hc_tile(blockIdx.x, tiles_y, tiles_x), h, wd, raw);
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const HcTile tl = hc_tile(t, tiles_y, tiles_x);
        __syncthreads();
        if constexpr (PIPE) hc_put<CI, BF>(raw, tl, h, wd, patch); else hc_stage<CI>(x, x_bf16, tl, h, wd, patch);
        __syncthreads();
        if (PIPE && t + (int)gridDim.x < n_tiles) hc_fetch<CI, BF>(x, hc_tile(t + gridDim.x, tiles_y, tiles_x), h, wd, raw);
        // [r5] two-output head: the pixel's gradient pair for the NEXT pass is loaded while this pass multiplies (it was a global load at the head of every
        // pass, used at once)
        auto dy_at = [&](int q) {
            const int yy = tl.y0 + q / HC_TC, xx = tl.x0 + q % HC_TC;
            return dy + (((int64_t)tl.img * h + min(yy, h - 1)) * wd + min(xx, wd - 1)) * co;
        };
        // [r6] HC_DY_AHEAD (default OFF).  Round 5 loaded the two-output head's gradient pair one pixel row ahead (`g_next`, one 8-byte load kept in a register pair
        // across the loop's back edge).  tests/test_determinism.py found the ONE gradient of the model that still differed between two staged steps -- this
        // kernel's dW, and only its second output channel (the odd register of the pair): with the pair carried across the back edge the kernel's result, on
        // the same operands (also on private copies of them), differed from its result on a quiet device in 8 of 256 calls made while the other stream kept
        // the device busy, and in 0 of 256 with plain per-pass loads (tools/r06_diag_headconv3.py, profiles/r06_headconv_race.txt).  The compiled loop waits
        // for the early load right after issuing it (s_waitcnt vmcnt(0) in front of the first use), so the early load never overlapped anything: nothing is
        // lost with it.  The mechanism below the ISA is not known; the construct is gone.
#ifndef HC_DY_AHEAD
#define HC_DY_AHEAD 0
#endif
        constexpr bool AHEAD = COM == 2 && HC_DY_AHEAD;
        float2 g_next = make_float2(0.f, 0.f);
        if (AHEAD && co == 2) g_next = *reinterpret_cast<const float2 *>(dy_at(slot));
        for (int q = slot; q < HC_TR * HC_TC; q += PPB) {
            const int row = q / HC_TC, col = q % HC_TC;
            const int yy = tl.y0 + row, xx = tl.x0 + col;
            const bool ok = yy < h && xx < wd;
            const float *gp = dy_at(q);
            float g[COM];
            if (AHEAD && co == 2) {
                const float2 g2 = g_next;
                if (q + PPB < HC_TR * HC_TC) g_next = *reinterpret_cast<const float2 *>(dy_at(q + PPB));
                g[0] = ok ? g2.x : 0.f;
                g[1] = ok ? g2.y : 0.f;
            } else {
#pragma unroll
                for (int c = 0; c < COM; ++c) g[c] = (c < co && ok) ? gp[c < co ? c : 0] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < COM; ++c) bs[c] += g[c];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float4 v = *reinterpret_cast<const float4 *>(patch + ((row + tap / 3) * HC_PW + col + tap % 3) * CI + 4 * l);
#pragma unroll
                for (int c = 0; c < COM; ++c) {
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
            for (int c = 0; c < COM; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[t][c][q] += __shfl_xor(acc[t][c][q], d, 64);
#pragma unroll
        for (int c = 0; c < COM; ++c) bs[c] += __shfl_xor(bs[c], d, 64);
    }
    __syncthreads();                                           // the last tile's readers are done with the patch
    float *red = patch;
    if ((threadIdx.x & 63) < LPP) {
        float *r = red + (wave * LPP + l) * RED;
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
        for (int wv = 0; wv < HC_THREADS / 64; ++wv) s += red[(wv * LPP + ci / 4) * RED + (tap * COM + c) * 4 + (ci & 3)];
        mine[e] = s;
    }
    if (threadIdx.x < co) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < HC_THREADS / 64; ++wv) s += red[(wv * LPP) * RED + 9 * COM * 4 + threadIdx.x];      // every lane group saw every pixel's dy: take group 0
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
    const int tiles_y = (h + HC_TR - 1) / HC_TR, tiles_x = (wd + HC_TC - 1) / HC_TC;
    const int64_t n_tiles = (int64_t)n_img * tiles_y * tiles_x;
    const int grid = (int)(n_tiles < PCACC_CUS * 6 ? n_tiles : PCACC_CUS * 6);
#define HC_FWD(CIV, COV)                                                                                                                        \
    do {                                                                                                                                        \
        const size_t lds = (size_t)(HC_PH * HC_PW * CIV) * sizeof(float);                                                       \
        auto kern = x_dtype ? head_conv_fwd_kernel<CIV, COV, true> : head_conv_fwd_kernel<CIV, COV, false>;                                     \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)     \
            return PCACC_E_LAUNCH;                                                                                                              \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(HC_THREADS), lds, pcacc_stream(stream), x, x_dtype, w, bias, y, n_img, h, wd, c_out, w_strides[0], \
                           w_strides[1], w_strides[2], w_strides[3], tiles_y, tiles_x);                                                         \
    } while (0)
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
    const int tiles_y = (h + HC_TR - 1) / HC_TR, tiles_x = (wd + HC_TC - 1) / HC_TC;
    const int64_t n_tiles = (int64_t)n_img * tiles_y * tiles_x;
    const int grid = (int)(n_tiles < PCACC_CUS * 8 ? n_tiles : PCACC_CUS * 8);
#define HC_DG(CIV, COV) hipLaunchKernelGGL((head_conv_dgrad_kernel<CIV, COV>), dim3(grid), dim3(HC_THREADS), 0, pcacc_stream(stream), dy, w, dx, dx_dtype, n_img, \
                                           h, wd, c_out, w_strides[0], w_strides[1], w_strides[2], w_strides[3], tiles_y, tiles_x)
    if (c_in == 32 && c_out <= 2) HC_DG(32, 2); else if (c_in == 32) HC_DG(32, 4); else if (c_out <= 2) HC_DG(64, 2); else HC_DG(64, 4);
#undef HC_DG
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

static int hc_wgrad_grid(int n_img, int h, int wd)
{
    const int64_t n_tiles = (int64_t)n_img * ((h + HC_TR - 1) / HC_TR) * ((wd + HC_TC - 1) / HC_TC);
    return (int)(n_tiles < PCACC_CUS * 3 ? n_tiles : PCACC_CUS * 3);
}

extern "C" int pcacc_head_conv3x3_wgrad_workspace_bytes(int32_t n_img, int32_t h, int32_t wd, int32_t c_in, int32_t c_out, size_t *bytes)
{
    if (!bytes || !hc_ok(n_img, h, wd, c_in, c_out)) return PCACC_E_ARG;
    *bytes = (size_t)hc_wgrad_grid(n_img, h, wd) * (c_out * c_in * 9 + c_out) * sizeof(float);
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
    const int grid = hc_wgrad_grid(n_img, h, wd);
    const int tiles_y = (h + HC_TR - 1) / HC_TR, tiles_x = (wd + HC_TC - 1) / HC_TC;
    float *partial = static_cast<float *>(workspace);
#define HC_WG(CIV, COV)                                                                                                                         \
    do {                                                                                                                                        \
        const size_t lds = (size_t)(HC_PH * HC_PW * CIV) * sizeof(float);                                                                       \
        auto kern = x_dtype ? head_conv_wgrad_kernel<CIV, COV, true> : head_conv_wgrad_kernel<CIV, COV, false>;                                 \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)     \
            return PCACC_E_LAUNCH;                                                                                                              \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(HC_THREADS), lds, st, dy, x, x_dtype, partial, n_img, h, wd, c_out, tiles_y, tiles_x);        \
    } while (0)
    if (c_in == 32 && c_out <= 2) HC_WG(32, 2); else if (c_in == 32) HC_WG(32, 4); else if (c_out <= 2) HC_WG(64, 2); else HC_WG(64, 4);
#undef HC_WG
    const int n_w = c_out * c_in * 9;
    hipLaunchKernelGGL(head_conv_wgrad_reduce_kernel, dim3((n_w + c_out + 63) / 64), dim3(256), 0, st, partial, grid, n_w, c_out, dw, db);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
