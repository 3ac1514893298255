// A ResnetBlockFC of the pillar encoder (models/pillar_encoder.py:13-55, sizes 64 -> 32 -> 32 with a linear shortcut) as ONE kernel
// forward and ONE backward over bf16 point rows:
//     h   = relu(x) W0^T + b0            [rows, 32]
//     out = relu(h) W1^T + b1 + x Ws^T   [rows, 32]
// x is [rows, 64], either contiguous or -- blocks 1.. of the encoder, models/pillar_encoder.py:116-118 -- the virtual concatenation
// cat(xa[row], pooled[p2v[row]]) of two 32-wide pieces.  As three row-linear launches the block moves 1.85 GB per 3.2 M rows (fc_0
// and the shortcut both read x, fc_1 reads both results); here x is read once and `out` written once (0.6 GB), plus relu(h) kept for
// the backward (0.2 GB).  The backward reads x, relu(h) and d(out) once and produces d(x) -- already split into its two pieces -- and
// all five parameter gradients in the same pass: three more data-gradient launches, three weight-gradient launches and the add of the
// two d(x) contributions disappear.
//
// Matrix-core layout as in mlp_mfma.hip: v_mfma_f32_32x32x16_bf16 with A = 32 output features x 16 k (weights, bf16 in LDS) and
// B = 16 k x 32 rows (a wave's quarter of the staged 128-row tile), so D has lane = row and register quads = 4 consecutive features;
// results pass through LDS and leave in 16-byte coalesced stores.  Weight gradients reduce over rows: both operands come through the
// hardware transpose read (ds_read_b64_tr_b16) of the row-major tiles.  Every wave works on its own 32 rows of the tile in every
// step, so the only workgroup barriers are around tile staging and the output tile.
#include "common.h"

typedef __bf16 pb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pb_f32x16 __attribute__((ext_vector_type(16)));
typedef short pb_s16x4 __attribute__((ext_vector_type(4)));
union pb_frag { pb_bf16x8 v; pb_s16x4 h[2]; uint32_t u[4]; uint4 q; };

#define PB_TILE 128
#define PB_THREADS 256
#define PB_PARTIAL 5184          // floats per partial slot: dW1 [32][32], db1 [32], dWs [32][64], dW0 [32][64], db0 [32]
#define PB_OFF_W1 0
#define PB_OFF_B1 1024
#define PB_OFF_WS 1056
#define PB_OFF_W0 3104
#define PB_OFF_B0 5152

struct PbPieces {                 // second half of x: b[idx[row]] (32 columns); b == NULL: x is the contiguous [rows, 64] array `xa`
    const uint16_t *b;
    const int32_t *idx;
};

__device__ __forceinline__ uint32_t pb_relu2(uint32_t v)
{
    const uint32_t neg = (v >> 15) & 0x00010001u;
    return v & ~(neg * 0xffffu);
}
__device__ __forceinline__ pb_frag pb_relu(pb_frag f)
{
    f.u[0] = pb_relu2(f.u[0]); f.u[1] = pb_relu2(f.u[1]); f.u[2] = pb_relu2(f.u[2]); f.u[3] = pb_relu2(f.u[3]);
    return f;
}
__device__ __forceinline__ bool pb_pos(uint32_t half) { return half != 0 && half <= 0x7f80u; }       // bf16 > 0 (not NaN)

// 16-byte piece `c` (0..1023) of the x tile: row c / 8 of the tile, columns 8 (c % 8) ..
// `prow` = xp.idx[row], fetched one tile earlier (the index load and the row load it feeds are a tile apart)
template <bool GATHER>
__device__ __forceinline__ uint4 pb_load_x(const uint16_t *__restrict__ xa, const PbPieces &xp, int64_t row, int col, int prow)
{
    if (!GATHER) return *reinterpret_cast<const uint4 *>(xa + row * 64 + col);
    if (col < 32) return *reinterpret_cast<const uint4 *>(xa + row * 32 + col);
    return *reinterpret_cast<const uint4 *>(xp.b + (int64_t)prow * 32 + (col - 32));
}

// fp32 [n][k] weights -> bf16 LDS rows of stride `ld`; transposed: dst[k][n] = src[n][k]
__device__ __forceinline__ void pb_stage_weights(const float *__restrict__ src, int n, int k, uint16_t *dst, int ld, bool transposed)
{
    for (int e = threadIdx.x; e < n * k; e += PB_THREADS) {
        const int r = e / k, c = e % k;
        const uint16_t v = f32_to_bf16(src[e]);
        if (transposed) dst[c * ld + r] = v;
        else dst[r * ld + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
template <bool GATHER>
__global__ __launch_bounds__(PB_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void pfn_block_fwd_kernel(const uint16_t *__restrict__ xa, PbPieces xp, const float *__restrict__ W0,
                                                                   const float *__restrict__ b0, const float *__restrict__ Ws,
                                                                   const float *__restrict__ W1, const float *__restrict__ b1,
                                                                   uint16_t *__restrict__ out, uint16_t *__restrict__ hr, int64_t rows)
{
    constexpr int XS = 72, HS = 40;
    __shared__ __attribute__((aligned(16))) uint16_t xs[PB_TILE * XS];          // the x tile; later the output tile [128][HS]
    __shared__ __attribute__((aligned(16))) uint16_t hs[PB_TILE * HS];          // relu(h)
    __shared__ __attribute__((aligned(16))) uint16_t w0s[32 * XS], wss[32 * XS], w1s[32 * HS];
    __shared__ float b0s[32], b1s[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    pb_stage_weights(W0, 32, 64, w0s, XS, false);
    pb_stage_weights(Ws, 32, 64, wss, XS, false);
    pb_stage_weights(W1, 32, 32, w1s, HS, false);
    if (threadIdx.x < 32) { b0s[threadIdx.x] = b0 ? b0[threadIdx.x] : 0.f; b1s[threadIdx.x] = b1 ? b1[threadIdx.x] : 0.f; }

    const int64_t n_tiles = (rows + PB_TILE - 1) / PB_TILE;
    uint4 xreg[4];
    int prow[4] = {0, 0, 0, 0};
    auto fetch_rows = [&](int64_t tile) {                                     // pillar rows of the gathered half, one tile ahead of their use
        if (!GATHER) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            const int64_t row = tile * PB_TILE + (c >> 3);
            prow[q] = ((c & 7) >= 4 && row < rows) ? xp.idx[row] : 0;
        }
    };
    auto fetch = [&](int64_t tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            const int64_t row = tile * PB_TILE + (c >> 3);
            xreg[q] = row < rows ? pb_load_x<GATHER>(xa, xp, row, (c & 7) * 8, prow[q]) : make_uint4(0, 0, 0, 0);
        }
    };
    int64_t tile = blockIdx.x;
    fetch_rows(tile);
    if (tile < n_tiles) fetch(tile);
    fetch_rows(tile + gridDim.x);
    const int myrow = wave * 32 + lp;
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                                      // the previous output tile has left xs
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            *reinterpret_cast<uint4 *>(xs + (c >> 3) * XS + (c & 7) * 8) = xreg[q];
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);              // in flight during the rest of this tile
        fetch_rows(tile + 2 * (int64_t)gridDim.x);

        pb_f32x16 acc_h, acc_o;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_h[r] = 0.f; acc_o[r] = 0.f; }
        const uint16_t *xrow = xs + myrow * XS + lh * 8;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            pb_frag bx;
            bx.q = *reinterpret_cast<const uint4 *>(xrow + kc * 16);
            const pb_frag br = pb_relu(bx);
            const pb_bf16x8 a0 = *reinterpret_cast<const pb_bf16x8 *>(w0s + lp * XS + lh * 8 + kc * 16);
            const pb_bf16x8 as = *reinterpret_cast<const pb_bf16x8 *>(wss + lp * XS + lh * 8 + kc * 16);
            acc_h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, br.v, acc_h, 0, 0, 0);
            acc_o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as, bx.v, acc_o, 0, 0, 0);
        }
        uint16_t *hrow = hs + myrow * HS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * g + 4 * lh;
            uint2 pk;
            pk.x = pcacc_pack_bf16x2(fmaxf(acc_h[4 * g] + b0s[c], 0.f), fmaxf(acc_h[4 * g + 1] + b0s[c + 1], 0.f));
            pk.y = pcacc_pack_bf16x2(fmaxf(acc_h[4 * g + 2] + b0s[c + 2], 0.f), fmaxf(acc_h[4 * g + 3] + b0s[c + 3], 0.f));
            *reinterpret_cast<uint2 *>(hrow + c) = pk;
        }
        __syncthreads();                                                      // relu(h) complete; every wave is done reading xs
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const pb_bf16x8 bh = *reinterpret_cast<const pb_bf16x8 *>(hrow + lh * 8 + kc * 16);
            const pb_bf16x8 a1 = *reinterpret_cast<const pb_bf16x8 *>(w1s + lp * HS + lh * 8 + kc * 16);
            acc_o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc_o, 0, 0, 0);
        }
        uint16_t *orow = xs + myrow * HS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * g + 4 * lh;
            uint2 pk;
            pk.x = pcacc_pack_bf16x2(acc_o[4 * g] + b1s[c], acc_o[4 * g + 1] + b1s[c + 1]);
            pk.y = pcacc_pack_bf16x2(acc_o[4 * g + 2] + b1s[c + 2], acc_o[4 * g + 3] + b1s[c + 3]);
            *reinterpret_cast<uint2 *>(orow + c) = pk;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {                                         // relu(h) for the backward: 512 sixteen-byte pieces
            const int c = threadIdx.x + q * PB_THREADS;
            const int64_t row = tile * PB_TILE + (c >> 2);
            if (hr && row < rows) *reinterpret_cast<uint4 *>(hr + row * 32 + (c & 3) * 8) = *reinterpret_cast<const uint4 *>(hs + (c >> 2) * HS + (c & 3) * 8);
        }
        __syncthreads();                                                      // the output tile is staged
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            const int64_t row = tile * PB_TILE + (c >> 2);
            if (row < rows) *reinterpret_cast<uint4 *>(out + row * 32 + (c & 3) * 8) = *reinterpret_cast<const uint4 *>(xs + (c >> 2) * HS + (c & 3) * 8);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
#define PB_TR(p) __builtin_amdgcn_ds_read_tr16_b64_v4i16((pb_s16x4 __attribute__((address_space(3))) *)(p))

template <bool GATHER>
__global__ __launch_bounds__(PB_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void pfn_block_bwd_kernel(const uint16_t *__restrict__ xa, PbPieces xp, const uint16_t *__restrict__ hr,
                                                                   const uint16_t *__restrict__ gout, const float *__restrict__ W0,
                                                                   const float *__restrict__ Ws, const float *__restrict__ W1,
                                                                   uint16_t *__restrict__ gxa, uint16_t *__restrict__ gxb,
                                                                   float *__restrict__ partial, int64_t rows, float *__restrict__ gp_zero)
{
    if (blockIdx.x == 0)                                                      // the reduce launch adds into the parameter gradients
        for (int e = threadIdx.x; e < PB_PARTIAL; e += PB_THREADS) gp_zero[e] = 0.f;
    // row strides (elements): 96 / 32 keep the transpose reads conflict-free (pcacc_tr_stride); OS = the d(x) tile
    constexpr int XS = 96, GS = 32, WS = 40, OS = 72;
    __shared__ __attribute__((aligned(16))) uint16_t xs[PB_TILE * XS];                    // x
    __shared__ __attribute__((aligned(16))) uint16_t ghg[3 * PB_TILE * GS];               // d(out) | relu(h) | d(h); later the d(x) tile
    __shared__ __attribute__((aligned(16))) uint16_t w1t[32 * WS], w0t[64 * WS], wst[64 * WS];   // W1^T [h][o], W0^T [i][h], Ws^T [i][o]
    uint16_t *gs = ghg, *hrs = ghg + PB_TILE * GS, *ghs = ghg + 2 * PB_TILE * GS, *os = ghg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    pb_stage_weights(W1, 32, 32, w1t, WS, true);
    pb_stage_weights(W0, 32, 64, w0t, WS, true);
    pb_stage_weights(Ws, 32, 64, wst, WS, true);

    pb_f32x16 acc_w1, acc_ws0, acc_ws1, acc_w00, acc_w01;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_w1[r] = 0.f; acc_ws0[r] = 0.f; acc_ws1[r] = 0.f; acc_w00[r] = 0.f; acc_w01[r] = 0.f; }
    float bias_sum = 0.f;                  // lane l < 32: column l of d(out), l >= 32: column l - 32 of d(h), over this wave's rows

    const int64_t n_tiles = (rows + PB_TILE - 1) / PB_TILE;
    uint4 xreg[4], greg[2], hreg[2];
    int prow[4] = {0, 0, 0, 0};
    auto fetch_rows = [&](int64_t tile) {
        if (!GATHER) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            const int64_t row = tile * PB_TILE + (c >> 3);
            prow[q] = ((c & 7) >= 4 && row < rows) ? xp.idx[row] : 0;
        }
    };
    auto fetch = [&](int64_t tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            const int64_t row = tile * PB_TILE + (c >> 3);
            xreg[q] = row < rows ? pb_load_x<GATHER>(xa, xp, row, (c & 7) * 8, prow[q]) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            const int64_t row = tile * PB_TILE + (c >> 2);
            const bool ok = row < rows;
            greg[q] = ok ? *reinterpret_cast<const uint4 *>(gout + row * 32 + (c & 3) * 8) : make_uint4(0, 0, 0, 0);
            hreg[q] = ok ? *reinterpret_cast<const uint4 *>(hr + row * 32 + (c & 3) * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    int64_t tile = blockIdx.x;
    fetch_rows(tile);
    if (tile < n_tiles) fetch(tile);
    fetch_rows(tile + gridDim.x);
    const int myrow = wave * 32 + lp;
    const int g4 = lane >> 4, li = lane & 15;
    const int tr_row = (g4 >> 1) * 8 + (li >> 2), tr_col = (g4 & 1) * 16 + (li & 3) * 4;
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                                      // the previous d(x) tile has left the LDS
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            *reinterpret_cast<uint4 *>(xs + (c >> 3) * XS + (c & 7) * 8) = xreg[q];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            *reinterpret_cast<uint4 *>(gs + (c >> 2) * GS + (c & 3) * 8) = greg[q];
            *reinterpret_cast<uint4 *>(hrs + (c >> 2) * GS + (c & 3) * 8) = hreg[q];
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
        fetch_rows(tile + 2 * (int64_t)gridDim.x);

        // A. d(h) = (d(out) W1) where h > 0
        pb_bf16x8 bg[2];
        pb_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bg[kc] = *reinterpret_cast<const pb_bf16x8 *>(gs + myrow * GS + lh * 8 + kc * 16);
            const pb_bf16x8 a = *reinterpret_cast<const pb_bf16x8 *>(w1t + lp * WS + lh * 8 + kc * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bg[kc], acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * g + 4 * lh;
            const uint2 hq = *reinterpret_cast<const uint2 *>(hrs + myrow * GS + c);
            uint2 pk;
            pk.x = pcacc_pack_bf16x2(pb_pos(hq.x & 0xffffu) ? acc[4 * g] : 0.f, pb_pos(hq.x >> 16) ? acc[4 * g + 1] : 0.f);
            pk.y = pcacc_pack_bf16x2(pb_pos(hq.y & 0xffffu) ? acc[4 * g + 2] : 0.f, pb_pos(hq.y >> 16) ? acc[4 * g + 3] : 0.f);
            *reinterpret_cast<uint2 *>(ghs + myrow * GS + c) = pk;
        }
        __syncthreads();                                                      // d(h) rows visible (same wave reads them back, other lanes)

        // C. parameter gradients over this wave's 32 rows: A = 32 features x 16 rows of d(out) / d(h), B = 16 rows x 32 features
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            const int r0 = wave * 32 + s * 16 + tr_row;
            pb_frag a_g, a_gh, b_hr, b_x0, b_x1;
            a_g.h[0] = PB_TR(gs + r0 * GS + tr_col);       a_g.h[1] = PB_TR(gs + (r0 + 4) * GS + tr_col);
            a_gh.h[0] = PB_TR(ghs + r0 * GS + tr_col);     a_gh.h[1] = PB_TR(ghs + (r0 + 4) * GS + tr_col);
            b_hr.h[0] = PB_TR(hrs + r0 * GS + tr_col);     b_hr.h[1] = PB_TR(hrs + (r0 + 4) * GS + tr_col);
            b_x0.h[0] = PB_TR(xs + r0 * XS + tr_col);      b_x0.h[1] = PB_TR(xs + (r0 + 4) * XS + tr_col);
            b_x1.h[0] = PB_TR(xs + r0 * XS + 32 + tr_col); b_x1.h[1] = PB_TR(xs + (r0 + 4) * XS + 32 + tr_col);
            const pb_frag b_r0 = pb_relu(b_x0), b_r1 = pb_relu(b_x1);
            acc_w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_g.v, b_hr.v, acc_w1, 0, 0, 0);
            acc_ws0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_g.v, b_x0.v, acc_ws0, 0, 0, 0);
            acc_ws1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_g.v, b_x1.v, acc_ws1, 0, 0, 0);
            acc_w00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_gh.v, b_r0.v, acc_w00, 0, 0, 0);
            acc_w01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_gh.v, b_r1.v, acc_w01, 0, 0, 0);
        }
        {                                                                     // bias gradients: column sums of d(out) and d(h)
            const uint16_t *col = (lh ? ghs : gs) + wave * 32 * GS + lp;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 32; r += 2) { s0 += bf16_to_f32(col[r * GS]); s1 += bf16_to_f32(col[(r + 1) * GS]); }
            bias_sum += s0 + s1;
        }

        // B. d(x) = (d(h) W0) where x > 0, + d(out) Ws
        uint2 pk[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            pb_f32x16 a1, a2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a1[r] = 0.f; a2[r] = 0.f; }
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const pb_bf16x8 bgh = *reinterpret_cast<const pb_bf16x8 *>(ghs + myrow * GS + lh * 8 + kc * 16);
                const pb_bf16x8 a0 = *reinterpret_cast<const pb_bf16x8 *>(w0t + (nt * 32 + lp) * WS + lh * 8 + kc * 16);
                const pb_bf16x8 as = *reinterpret_cast<const pb_bf16x8 *>(wst + (nt * 32 + lp) * WS + lh * 8 + kc * 16);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bgh, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as, bg[kc], a2, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 8 * g + 4 * lh;
                const uint2 xq = *reinterpret_cast<const uint2 *>(xs + myrow * XS + nt * 32 + c);
                pk[nt][g].x = pcacc_pack_bf16x2((pb_pos(xq.x & 0xffffu) ? a1[4 * g] : 0.f) + a2[4 * g],
                                                (pb_pos(xq.x >> 16) ? a1[4 * g + 1] : 0.f) + a2[4 * g + 1]);
                pk[nt][g].y = pcacc_pack_bf16x2((pb_pos(xq.y & 0xffffu) ? a1[4 * g + 2] : 0.f) + a2[4 * g + 2],
                                                (pb_pos(xq.y >> 16) ? a1[4 * g + 3] : 0.f) + a2[4 * g + 3]);
            }
        }
        __syncthreads();                                                      // every wave is done with d(out) / relu(h) / d(h)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2 *>(os + myrow * OS + nt * 32 + 8 * g + 4 * lh) = pk[nt][g];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PB_THREADS;
            const int64_t row = tile * PB_TILE + (c >> 3);
            const int col = (c & 7) * 8;
            if (row >= rows) continue;
            const uint4 v = *reinterpret_cast<const uint4 *>(os + (c >> 3) * OS + col);
            if (!gxb) *reinterpret_cast<uint4 *>(gxa + row * 64 + col) = v;
            else if (col < 32) *reinterpret_cast<uint4 *>(gxa + row * 32 + col) = v;
            else *reinterpret_cast<uint4 *>(gxb + row * 32 + (col - 32)) = v;
        }
    }
    // partial slot of this wave: D has lane = column (B feature), register r = A feature (r & 3) + 8 (r >> 2) + 4 lh
    float *mine = partial + ((int64_t)blockIdx.x * 4 + wave) * PB_PARTIAL;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = (r & 3) + 8 * (r >> 2) + 4 * lh;
        mine[PB_OFF_W1 + f * 32 + lp] = acc_w1[r];
        mine[PB_OFF_WS + f * 64 + lp] = acc_ws0[r];
        mine[PB_OFF_WS + f * 64 + 32 + lp] = acc_ws1[r];
        mine[PB_OFF_W0 + f * 64 + lp] = acc_w00[r];
        mine[PB_OFF_W0 + f * 64 + 32 + lp] = acc_w01[r];
    }
    mine[(lh ? PB_OFF_B0 : PB_OFF_B1) + lp] = bias_sum;
}

// out[e] = sum over the partial slots in a fixed order (common.h: pcacc_reduce_partials) -- run-to-run identical
__global__ __launch_bounds__(1024) void pfn_block_reduce_kernel(const float *__restrict__ partial, int n_parts, float *__restrict__ out)
{
    pcacc_reduce_partials<16>(partial, n_parts, PB_PARTIAL, [&](int e, float v) { out[e] = v; });
}

static int pb_grid(int64_t rows, int per_cu)
{
    const int64_t n_tiles = (rows + PB_TILE - 1) / PB_TILE;
    const int64_t grid = (int64_t)PCACC_CUS * per_cu;
    return (int)(grid > n_tiles ? n_tiles : grid);
}

extern "C" int pcacc_pfn_block_forward(const uint16_t *xa, const uint16_t *pooled, const int32_t *p2v, const float *w0, const float *b0,
                                       const float *ws, const float *w1, const float *b1, uint16_t *out, uint16_t *relu_h, int64_t rows,
                                       void *stream)
{
    if (rows < 0 || (pooled && !p2v)) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!xa || !w0 || !ws || !w1 || !out) return PCACC_E_ARG;
    if (pooled) pfn_block_fwd_kernel<true><<<pb_grid(rows, 3), PB_THREADS, 0, pcacc_stream(stream)>>>(xa, PbPieces{pooled, p2v}, w0, b0, ws, w1, b1, out, relu_h, rows);
    else pfn_block_fwd_kernel<false><<<pb_grid(rows, 3), PB_THREADS, 0, pcacc_stream(stream)>>>(xa, PbPieces{pooled, p2v}, w0, b0, ws, w1, b1, out, relu_h, rows);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_pfn_block_backward_workspace_bytes(int64_t rows, size_t *bytes)
{
    if (!bytes || rows < 0) return PCACC_E_ARG;
    *bytes = (size_t)pb_grid(rows, 2) * 4 * PB_PARTIAL * sizeof(float);
    return PCACC_OK;
}

extern "C" int pcacc_pfn_block_backward(const uint16_t *xa, const uint16_t *pooled, const int32_t *p2v, const uint16_t *relu_h,
                                        const uint16_t *grad_out, const float *w0, const float *ws, const float *w1, uint16_t *grad_xa,
                                        uint16_t *grad_xb, float *grad_params, int64_t rows, void *workspace, size_t workspace_bytes,
                                        void *stream)
{
    if (rows < 0 || !grad_params || (pooled && (!p2v || !grad_xb)) || (!pooled && grad_xb)) return PCACC_E_ARG;
    hipStream_t s = pcacc_stream(stream);
    if (rows == 0) {
        if (hipMemsetAsync(grad_params, 0, PB_PARTIAL * sizeof(float), s) != hipSuccess) return PCACC_E_LAUNCH;
        return PCACC_OK;
    }
    if (!xa || !relu_h || !grad_out || !w0 || !ws || !w1 || !grad_xa || !workspace) return PCACC_E_ARG;
    const int grid = pb_grid(rows, 2);
    if (workspace_bytes < (size_t)grid * 4 * PB_PARTIAL * sizeof(float)) return PCACC_E_WORKSPACE;
    float *partial = reinterpret_cast<float *>(workspace);
    if (pooled) pfn_block_bwd_kernel<true><<<grid, PB_THREADS, 0, s>>>(xa, PbPieces{pooled, p2v}, relu_h, grad_out, w0, ws, w1, grad_xa, grad_xb, partial, rows, grad_params);
    else pfn_block_bwd_kernel<false><<<grid, PB_THREADS, 0, s>>>(xa, PbPieces{pooled, p2v}, relu_h, grad_out, w0, ws, w1, grad_xa, grad_xb, partial, rows, grad_params);
    pfn_block_reduce_kernel<<<(PB_PARTIAL + 15) / 16, 1024, 0, s>>>(partial, grid * 4, grad_params);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
