// Per-point linear layers ("rows x skinny weight") at HBM speed  (SURVEY.md 8a row A4 and the per-point heads of A10).
//
// The pillar encoder, the STPN point heads and the TubeNet embeddings are nn.Linear layers applied to 10^5..10^6 rows with
// 3..128 input and 2..128 output features (models/pillar_encoder.py:112-121, models/stpn.py:94-102).  As library GEMMs these
// skinny shapes ran at 0.5 TB/s (hipBLASLt, 0.6-1.1 ms per call, 39 ms per training step, profiles/r01_bench_v2).  They are
// HBM-bound streams: read a row, multiply by a weight matrix that fits in the scalar cache, write a row.
//
//   rows_linear  : Y = [relu]( [relu|mask](X) @ W^T + b [+ residual] ) [masked]      -- forward AND backward-data (W^T passed in)
//   rows_wgrad   : dW_aug[N, K+1] += dYeff^T @ [Xeff | 1]                            -- weight + bias gradient, fp32 MFMA
//
// rows_linear: one row per lane.  The 128-row tile is staged through LDS so that global loads/stores are fully coalesced
// (row stride +1 float: conflict-free for both the row-major fill and the per-lane row read); the weight index is
// wave-uniform, so weights come through the scalar cache as SGPR operands of v_fma (no LDS traffic for them).
// rows_wgrad: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain): A = dY^T (n x 2 rows), B = X (2 rows x k), each half-wave loads
// one contiguous 128-byte row piece; per-wave 32x32 tiles are reduced with fp32 atomics into the zero-filled output.
#include <type_traits>
#include "common.h"

// rows per tile = threads per workgroup: 128 when max(K,N) <= 64 (33 KB of LDS), 64 otherwise (33 KB at 128 features)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// flags
#define MLP_PRE_RELU 1        // X := max(X, 0) on load
#define MLP_POST_RELU 2       // Y := max(Y, 0) before the store
// element types (bit set = bf16, clear = f32) of the five row tensors: the arithmetic is fp32 either way
#define MLP_X_BF16 1
#define MLP_INMASK_BF16 2
#define MLP_RES_BF16 4
#define MLP_OUTMASK_BF16 8
#define MLP_Y_BF16 16

__device__ __forceinline__ float mlp_ld1(const void *p, bool bf, int64_t i)
{
    return bf ? bf16_to_f32(reinterpret_cast<const uint16_t *>(p)[i]) : reinterpret_cast<const float *>(p)[i];
}
__device__ __forceinline__ void mlp_st1(void *p, bool bf, int64_t i, float v)
{
    if (!bf) reinterpret_cast<float *>(p)[i] = v;
    else reinterpret_cast<uint16_t *>(p)[i] = f32_to_bf16(v);
}

template <int K, int MLP_ROWS>
__global__ __launch_bounds__(MLP_ROWS) void rows_linear_kernel(const void *__restrict__ X, const void *__restrict__ in_mask,
                                                               const float *__restrict__ W, const float *__restrict__ bias,
                                                               const void *__restrict__ residual, const void *__restrict__ out_mask,
                                                               void *__restrict__ Y, int64_t rows, int N, int flags, int dt)
{
    const bool x_bf = dt & MLP_X_BF16, im_bf = dt & MLP_INMASK_BF16, r_bf = dt & MLP_RES_BF16, om_bf = dt & MLP_OUTMASK_BF16,
               y_bf = dt & MLP_Y_BF16;
    extern __shared__ __attribute__((aligned(16))) float tile[];       // MLP_ROWS x (max(K,N)+1)
    const int tid = threadIdx.x;
    const int n_tiles = (int)((rows + MLP_ROWS - 1) / MLP_ROWS);
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t row0 = (int64_t)t * MLP_ROWS;
        const int64_t n_in = min((int64_t)MLP_ROWS, rows - row0) * K;
        const int NS = (N % 4 == 0) ? N + 4 : N + 1;
        // 1. coalesced fill of the input tile (relu / mask applied here).  KS / NS = row strides in LDS: +4 floats when the
        //    rows are moved as float4 (keeps 16-byte alignment, conflict-free for ds_read_b128), +1 float otherwise.
        constexpr int KS = (K % 4 == 0) ? K + 4 : K + 1;
        if (K % 4 == 0) {
            const int64_t g4 = row0 * K / 4;
            for (int i4 = tid; i4 < MLP_ROWS * K / 4; i4 += MLP_ROWS) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((int64_t)i4 * 4 < n_in) {
                    v = pcacc_ld4(X, x_bf, g4 + i4);
                    if (flags & MLP_PRE_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (in_mask) {
                        const float4 mk = pcacc_ld4(in_mask, im_bf, g4 + i4);
                        if (!(mk.x > 0.f)) v.x = 0.f;
                        if (!(mk.y > 0.f)) v.y = 0.f;
                        if (!(mk.z > 0.f)) v.z = 0.f;
                        if (!(mk.w > 0.f)) v.w = 0.f;
                    }
                }
                const int e = i4 * 4;
                *reinterpret_cast<float4 *>(&tile[(e / K) * KS + (e % K)]) = v;
            }
        } else {
            for (int idx = tid; idx < MLP_ROWS * K; idx += MLP_ROWS) {
                float v = 0.f;
                if (idx < n_in) {
                    v = mlp_ld1(X, x_bf, row0 * K + idx);
                    if (flags & MLP_PRE_RELU) v = fmaxf(v, 0.f);
                    if (in_mask && !(mlp_ld1(in_mask, im_bf, row0 * K + idx) > 0.f)) v = 0.f;
                }
                tile[(idx / K) * KS + (idx % K)] = v;
            }
        }
        __syncthreads();
        float x[K];
#pragma unroll
        for (int k = 0; k < K; ++k) x[k] = tile[tid * KS + k];
        __syncthreads();
        // 2. N outputs per row, 8 at a time; weights are wave-uniform -> scalar loads.  For even K the dot product is
        //    accumulated as two interleaved partial sums (even / odd k) in one v_pk_fma_f32 per pair: the fp32 VALU peak
        //    of gfx950 (157 TF) is only reachable with packed FMA, and the 128x128 layers are VALU-bound, not HBM-bound.
        for (int n0 = 0; n0 < N; n0 += 8) {
            float acc[8];
            if (K % 2 == 0 && K >= 32) {
                f32x2 acc2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc2[j] = (f32x2){0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (n0 + j < N) {
                        const f32x2 *w2 = reinterpret_cast<const f32x2 *>(W + (int64_t)(n0 + j) * K);
#pragma unroll
                        for (int k = 0; k < K / 2; ++k)
                            acc2[j] = __builtin_elementwise_fma((f32x2){x[2 * k], x[2 * k + 1]}, w2[k], acc2[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = (acc2[j].x + acc2[j].y) + ((bias && n0 + j < N) ? bias[n0 + j] : 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = (bias && n0 + j < N) ? bias[n0 + j] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (n0 + j < N) {
                        const float *w = W + (int64_t)(n0 + j) * K;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[j] = fmaf(x[k], w[k], acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (n0 + j < N) tile[tid * NS + n0 + j] = acc[j];
        }
        __syncthreads();
        // 3. coalesced store of the output tile (residual / relu / mask applied here)
        const int64_t n_out = min((int64_t)MLP_ROWS, rows - row0) * N;
        if (N % 4 == 0) {
            const int64_t g4 = row0 * N / 4;
            for (int i4 = tid; (int64_t)i4 * 4 < n_out; i4 += MLP_ROWS) {
                const int e = i4 * 4;
                float4 v = *reinterpret_cast<const float4 *>(&tile[(e / N) * NS + (e % N)]);
                if (residual) { const float4 r = pcacc_ld4(residual, r_bf, g4 + i4); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                if (flags & MLP_POST_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (out_mask) {
                    const float4 mk = pcacc_ld4(out_mask, om_bf, g4 + i4);
                    if (!(mk.x > 0.f)) v.x = 0.f;
                    if (!(mk.y > 0.f)) v.y = 0.f;
                    if (!(mk.z > 0.f)) v.z = 0.f;
                    if (!(mk.w > 0.f)) v.w = 0.f;
                }
                pcacc_st4(Y, y_bf, g4 + i4, v);
            }
        } else {
            for (int idx = tid; idx < n_out; idx += MLP_ROWS) {
                float v = tile[(idx / N) * NS + (idx % N)];
                if (residual) v += mlp_ld1(residual, r_bf, row0 * N + idx);
                if (flags & MLP_POST_RELU) v = fmaxf(v, 0.f);
                if (out_mask && !(mlp_ld1(out_mask, om_bf, row0 * N + idx) > 0.f)) v = 0.f;
                mlp_st1(Y, y_bf, row0 * N + idx, v);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The edges of the point chains: few inputs (the 9 pillar-encoder features, 2/3-d positional encodings, and the backward-data
// pass of the 2-output heads: k <= 9) or few outputs (the 2-output heads: n <= 4).  One row per lane wastes the machine there
// (a 64-lane workgroup per 33 KB of LDS: 0.5-1.3 TB/s); these are plain streaming kernels instead.
//   few inputs : one lane per (row, 8 consecutive outputs); the row's k inputs come through L1 (the n/8 lanes of a row are
//                adjacent), the weights from LDS; 16- or 32-byte stores.  Same fp32 FMA order over k as rows_linear_kernel.
//   few outputs: k/8 lanes per row, each 8 consecutive inputs (16- or 32-byte loads), partial dot products folded with
//                xor-shuffles inside the lane group.
// ---------------------------------------------------------------------------------------------------------------------
// Few inputs, G = N / 8 lanes per row (a power of two <= 16): a lane owns 8 output columns for the whole launch -- their 8 x K weights
// and biases live in registers -- and walks the rows.  The K inputs of a row are loaded once by the row's lanes (lane g takes elements
// g, g + G, ...: one coalesced load instruction per wave covers 64 / G rows) and handed round with wave shuffles: no LDS staging, no
// barrier, two rows in flight per lane.  (r02: the former version -- K global loads of 4 bytes per lane and 8 x K weight reads from LDS --
// ran the 3.2 M x 9 -> 64 layer of the pillar encoder in 297 us, address-unit- and LDS-bound; the HBM floor is 65 us.)
#define FEW_U 4                     // rows in flight per lane in the few-feature streaming kernels
template <int K, int G, bool HALVES>
__global__ __launch_bounds__(256) void rows_linear_fewk_kernel(const void *__restrict__ X, const void *__restrict__ in_mask,
                                                               const float *__restrict__ W, const float *__restrict__ bias,
                                                               const void *__restrict__ residual, const void *__restrict__ out_mask,
                                                               void *__restrict__ Y, int64_t rows, int flags, int dt,
                                                               uint16_t *__restrict__ Y16 = nullptr, float *__restrict__ out_amax = nullptr)
{
    const bool x_bf = dt & MLP_X_BF16, im_bf = dt & MLP_INMASK_BF16, r_bf = dt & MLP_RES_BF16, om_bf = dt & MLP_OUTMASK_BF16,
               y_bf = dt & MLP_Y_BF16;
    constexpr int N = 8 * G, RB = 256 / G, NL = (K + G - 1) / G;      // rows per workgroup pass, loads per lane and row
    float omax = 0.f;                                                 // Y16 / out_amax ('mixed' mode, fp32 Y): bf16 copy of Y and its 256 partial maxima
    const int g = threadIdx.x % G, rsub = threadIdx.x / G;
    const int lane = threadIdx.x & 63, rowbase = lane - g;
    // [r5] fp32 rows: the lane's eight outputs are channels 4g .. 4g+3 of each HALF of the row (was 8g .. 8g+7): one store instruction of the row's G lanes
    // then covers a contiguous half row (128 B at N = 64: whole cache lines) instead of every other 16 bytes of the whole row -- 398 -> 357 us for the
    // 3.2 M x 9 -> 64 layer with its bf16 copy, 284 -> 234 us without (profiles/r05_fewk_stores_ab.txt).  bf16-only rows keep 8g .. 8g+7: one 16-byte
    // store per lane beats two 8-byte ones there (160 against 171 us).
    constexpr bool halves = HALVES;                                  // the launcher passes HALVES = fp32 rows
    float w[8][K], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = halves ? (j >> 2) * (4 * G) + 4 * g + (j & 3) : 8 * g + j;
        b[j] = bias ? bias[ch] : 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) w[j][k] = W[ch * K + k];
    }
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t r0 = (int64_t)blockIdx.x * RB + rsub; r0 < rows; r0 += FEW_U * stride) {      // the lanes of a row share r0: the shuffles below stay inside the row
        float part[FEW_U][NL];
        if (!x_bf && !in_mask) {
            // [r5] the common case (fp32 rows, no mask) as straight-line code: FEW_U x NL unconditional loads from clamped indices go out back to back, the
            // range test is a select behind them.  The general loop below guards each load with a branch (and picks the element type at run time): the ISA was
            // `global_load, s_waitcnt vmcnt(0)` once per element, eight round trips in series per pass of a lane.
            const float *x32 = static_cast<const float *>(X);
#pragma unroll
            for (int u = 0; u < FEW_U; ++u) {
                const int64_t row = r0 + u * stride;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int k = g + l * G;
                    const bool ok = row < rows && k < K;
                    float v = x32[ok ? row * K + k : 0];
                    if (flags & MLP_PRE_RELU) v = fmaxf(v, 0.f);
                    part[u][l] = ok ? v : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < FEW_U; ++u) {
                const int64_t row = r0 + u * stride;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int k = g + l * G;
                    float v = 0.f;
                    if (row < rows && k < K) {
                        v = mlp_ld1(X, x_bf, row * K + k);
                        if (flags & MLP_PRE_RELU) v = fmaxf(v, 0.f);
                        if (in_mask && !(mlp_ld1(in_mask, im_bf, row * K + k) > 0.f)) v = 0.f;
                    }
                    part[u][l] = v;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FEW_U; ++u) {
            const int64_t row = r0 + u * stride;
            float x[K];
#pragma unroll
            for (int k = 0; k < K; ++k) x[k] = __shfl(part[u][k / G], rowbase + (k % G), 64);
            if (row >= rows) continue;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = b[j];
#pragma unroll
                for (int k = 0; k < K; ++k) acc[j] = fmaf(x[k], w[j][k], acc[j]);
            }
            // groups of four channels: fp32 rows q4 (first half of the row) and q4 + G (second half); bf16 rows the lane's two adjacent groups
            const int64_t q4 = row * (N / 4) + (halves ? g : 2 * g);
            constexpr int hstep = halves ? G : 1;
            if (y_bf && !residual && !out_mask) {                     // the common case: one 16-byte store of 8 bf16
                if (flags & MLP_POST_RELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
                }
                reinterpret_cast<uint4 *>(Y)[q4 / 2] = make_uint4(pcacc_pack_bf16x2(acc[0], acc[1]), pcacc_pack_bf16x2(acc[2], acc[3]),
                                                                 pcacc_pack_bf16x2(acc[4], acc[5]), pcacc_pack_bf16x2(acc[6], acc[7]));
                continue;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t i4 = q4 + h * hstep;
                float4 v = make_float4(acc[4 * h], acc[4 * h + 1], acc[4 * h + 2], acc[4 * h + 3]);
                if (residual) { const float4 r = pcacc_ld4(residual, r_bf, i4); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                if (flags & MLP_POST_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (out_mask) {
                    const float4 mk = pcacc_ld4(out_mask, om_bf, i4);
                    if (!(mk.x > 0.f)) v.x = 0.f;
                    if (!(mk.y > 0.f)) v.y = 0.f;
                    if (!(mk.z > 0.f)) v.z = 0.f;
                    if (!(mk.w > 0.f)) v.w = 0.f;
                }
                pcacc_st4(Y, y_bf, i4, v);
                if (Y16) reinterpret_cast<uint2 *>(Y16)[i4] = make_uint2(pcacc_pack_bf16x2(v.x, v.y), pcacc_pack_bf16x2(v.z, v.w));
                if (out_amax) {
                    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) omax = __builtin_inff();
                }
            }
        }
    }
    if (out_amax) {                                                   // uniform: one atomic per wave into one of the 256 slots (zeroed by the caller)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) omax = fmaxf(omax, __shfl_xor(omax, d, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(out_amax) + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255), __float_as_uint(omax));
    }
}

template <int K, int N>
__global__ __launch_bounds__(256) void rows_linear_fewn_kernel(const void *__restrict__ X, const void *__restrict__ in_mask,
                                                               const float *__restrict__ W, const float *__restrict__ bias,
                                                               const void *__restrict__ residual, const void *__restrict__ out_mask,
                                                               void *__restrict__ Y, int64_t rows, int flags, int dt)
{
    const bool x_bf = dt & MLP_X_BF16, im_bf = dt & MLP_INMASK_BF16, r_bf = dt & MLP_RES_BF16, om_bf = dt & MLP_OUTMASK_BF16,
               y_bf = dt & MLP_Y_BF16;
    constexpr int LANES = K / 8;                                   // lanes per row (4, 8 or 16)
    const int sub = threadIdx.x % LANES;
    float w[N][8];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) w[j][c] = W[j * K + sub * 8 + c];
    const int64_t total = rows * LANES;
    const int64_t padded = (total + 255) / 256 * 256;              // whole waves stay in the loop: the shuffles need all lanes
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < padded; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / LANES;
        const bool live = row < rows;
        float x[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                const int64_t g4 = (row * K + sub * 8) / 4 + h;
                v = pcacc_ld4(X, x_bf, g4);
                if (flags & MLP_PRE_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (in_mask) {
                    const float4 mk = pcacc_ld4(in_mask, im_bf, g4);
                    if (!(mk.x > 0.f)) v.x = 0.f;
                    if (!(mk.y > 0.f)) v.y = 0.f;
                    if (!(mk.z > 0.f)) v.z = 0.f;
                    if (!(mk.w > 0.f)) v.w = 0.f;
                }
            }
            x[4 * h] = v.x, x[4 * h + 1] = v.y, x[4 * h + 2] = v.z, x[4 * h + 3] = v.w;
        }
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) a = fmaf(x[c], w[j][c], a);
#pragma unroll
            for (int d = LANES / 2; d; d >>= 1) a += __shfl_xor(a, d, 64);
            acc[j] = a;
        }
        if (live && sub == 0) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float v = acc[j] + (bias ? bias[j] : 0.f);
                if (residual) v += mlp_ld1(residual, r_bf, row * N + j);
                if (flags & MLP_POST_RELU) v = fmaxf(v, 0.f);
                if (out_mask && !(mlp_ld1(out_mask, om_bf, row * N + j) > 0.f)) v = 0.f;
                mlp_st1(Y, y_bf, row * N + j, v);
            }
        }
    }
}

static bool mlp_k_supported(int k) { return k == 2 || k == 3 || k == 4 || k == 9 || k == 32 || k == 64 || k == 128; }

static int rows_linear_any(const void *x, const void *in_mask, const float *w, const float *bias, const void *residual,
                           const void *out_mask, void *y, int64_t rows, int k, int n, int flags, int dt, void *stream, uint16_t *y16 = nullptr,
                           float *y_amax = nullptr)
{
    if (rows < 0 || n <= 0 || n > 128 || !mlp_k_supported(k)) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!x || !w || !y) return PCACC_E_ARG;
    hipStream_t s = pcacc_stream(stream);
    if (k <= 9 && (n == 8 || n == 16 || n == 32 || n == 64 || n == 128)) {   // few inputs: lane per (row, 8 outputs)
        const int grid = pcacc_grid(rows * (n / 8), 256 * FEW_U, PCACC_CUS * 8);
#define FEWK(KK, GG) do { if (dt & MLP_Y_BF16) rows_linear_fewk_kernel<KK, GG, false><<<grid, 256, 0, s>>>(x, in_mask, w, bias, residual, out_mask, y, rows, flags, dt, y16, y_amax); \
                          else rows_linear_fewk_kernel<KK, GG, true><<<grid, 256, 0, s>>>(x, in_mask, w, bias, residual, out_mask, y, rows, flags, dt, y16, y_amax); } while (0)
#define FEWK_G(KK) do { switch (n) { case 8: FEWK(KK, 1); break; case 16: FEWK(KK, 2); break; case 32: FEWK(KK, 4); break; \
                                     case 64: FEWK(KK, 8); break; default: FEWK(KK, 16); break; } } while (0)
        if (k == 2) FEWK_G(2);
        else if (k == 3) FEWK_G(3);
        else if (k == 4) FEWK_G(4);
        else FEWK_G(9);
#undef FEWK_G
#undef FEWK
        PCACC_CHECK_LAUNCH();
        return PCACC_OK;
    }
    if (y16 || y_amax) return PCACC_E_ARG;                         // the second output exists in the few-input kernel only
    if (n <= 2 && k >= 32) {                                       // few outputs: k/8 lanes per row
        const int grid = pcacc_grid(rows * (k / 8), 256, PCACC_CUS * 16);
#define FEWN(KK, NN) rows_linear_fewn_kernel<KK, NN><<<grid, 256, 0, s>>>(x, in_mask, w, bias, residual, out_mask, y, rows, flags, dt)
        if (n == 1) { if (k == 32) FEWN(32, 1); else if (k == 64) FEWN(64, 1); else FEWN(128, 1); }
        else { if (k == 32) FEWN(32, 2); else if (k == 64) FEWN(64, 2); else FEWN(128, 2); }
#undef FEWN
        PCACC_CHECK_LAUNCH();
        return PCACC_OK;
    }
    const int feat = k > n ? k : n;
    const int tile_rows = feat <= 64 ? 128 : 64;
    const int n_tiles = (int)((rows + tile_rows - 1) / tile_rows);
    const int grid = n_tiles < PCACC_CUS * 16 ? n_tiles : PCACC_CUS * 16;
    const size_t lds = (size_t)tile_rows * (feat + 4) * sizeof(float);
#define LAUNCH(KK)                                                                                                        \
    do {                                                                                                                  \
        if (tile_rows == 128)                                                                                             \
            rows_linear_kernel<KK, 128><<<grid, 128, lds, s>>>(x, in_mask, w, bias, residual, out_mask, y, rows, n, flags, dt); \
        else                                                                                                              \
            rows_linear_kernel<KK, 64><<<grid, 64, lds, s>>>(x, in_mask, w, bias, residual, out_mask, y, rows, n, flags, dt);  \
    } while (0)
    switch (k) {
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        case 4: LAUNCH(4); break;
        case 9: LAUNCH(9); break;
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        case 128: LAUNCH(128); break;
        default: return PCACC_E_ARG;
    }
#undef LAUNCH
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_rows_linear(const float *x, const float *in_mask, const float *w, const float *bias, const float *residual,
                                 const float *out_mask, float *y, int64_t rows, int k, int n, int flags, void *stream)
{
    return rows_linear_any(x, in_mask, w, bias, residual, out_mask, y, rows, k, n, flags, 0, stream);
}

// 'mixed' mode, few inputs (k <= 9: the 9 -> 64 position layer of the pillar encoder, models/pillar_encoder.py:100-108): fp32 rows in exact fp32
// arithmetic (no matrix cores: 9 FMAs per output), the bf16 shadow of the result and its 256 partial maxima from the same store phase
// (y_amax zero-filled by the caller) -- a copy pass and a maximum pass over 3.2 M x 64 values less.
extern "C" int pcacc_rows_linear_few_dual(const float *x, const float *w, const float *bias, const float *residual, float *y, uint16_t *y16,
                                          float *y_amax, int64_t rows, int k, int n, int flags, void *stream)
{
    if (k > 9 || !(n == 8 || n == 16 || n == 32 || n == 64 || n == 128) || !y16 || !y_amax) return PCACC_E_ARG;
    return rows_linear_any(x, nullptr, w, bias, residual, nullptr, y, rows, k, n, flags, 0, stream, y16, y_amax);
}

extern "C" int pcacc_rows_linear_mixed(const void *x, const void *in_mask, const float *w, const float *bias, const void *residual,
                                       const void *out_mask, void *y, int64_t rows, int k, int n, int flags, int dtypes, void *stream)
{
    if (dtypes & ~31) return PCACC_E_ARG;
    return rows_linear_any(x, in_mask, w, bias, residual, out_mask, y, rows, k, n, flags, dtypes, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight / bias gradient.  dW_aug[n][k] (k in [0,K]; column K is the bias gradient) = sum_r dY[r][n] * Xaug[r][k],
// with Xaug[r][K] = 1.  dy_mask: dY is zeroed where dy_mask <= 0 (post-ReLU layers); x_relu: X := max(X,0) (pre-ReLU layers).
//
// A workgroup (4 waves) walks chunks of WG_ROWS rows: the chunk's dY and X rows are contiguous in HBM and are staged
// into LDS with 16-byte coalesced loads (masks / ReLU applied on the way), so every row is read from HBM exactly once
// however many 32x32 output tiles there are.  Tiles are dealt round-robin to the waves; each tile is a chain of
// v_mfma_f32_32x32x2_f32 (A = dY^T: n x 2 rows, B = X: 2 rows x k) whose operands are two ds_read_b32 per MFMA.
// Workgroup partials go to the zero-filled output with one fp32 atomic per element per workgroup.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
// tiles per wave (template parameter WG_MAX_TILES): 1 when there are <= 4 output tiles (keeps the register footprint of the
// common 32/64-wide layers small: 4 workgroups per CU overlap staging and MFMA), 2 for <= 8, 6 for the 128 x 129 case.

template <int WG_ROWS, int WG_MAX_TILES>
__global__ __launch_bounds__(256) void rows_wgrad_kernel(const void *__restrict__ dY, const void *__restrict__ dy_mask,
                                                         const void *__restrict__ X, int x_relu, int64_t rows, int K, int N,
                                                         int k_tiles, int n_tile_total, float *dW, int dt)
{
    const bool dy_bf = dt & 1, m_bf = dt & 2, x_bf = dt & 4;                 // bf16 flags of dY, dy_mask, X
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [WG_ROWS][NS] dY, then [WG_ROWS][KS] X
    const int NS = N + 4, KS = K + 4;                                // +4: rows stay 16-byte aligned, bank-spread
    float *sdy = lds, *sx = lds + WG_ROWS * NS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, li = lane & 31;
    f32x16 acc[WG_MAX_TILES];
#pragma unroll
    for (int t = 0; t < WG_MAX_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int64_t n_chunks = (rows + WG_ROWS - 1) / WG_ROWS;
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int64_t row0 = ch * WG_ROWS;
        const int nrow = (int)min((int64_t)WG_ROWS, rows - row0);
        // stage dY (N floats per row) and X (K floats per row); scalar path when the width is not a multiple of 4
        if ((N & 3) == 0) {
            const int64_t g4 = row0 * N / 4;
            for (int i = threadIdx.x; i < WG_ROWS * N / 4; i += 256) {
                const int e = i * 4, r = e / N, c = e % N;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < nrow) {
                    v = pcacc_ld4(dY, dy_bf, g4 + i);
                    if (dy_mask) { const float4 mk = pcacc_ld4(dy_mask, m_bf, g4 + i); if (!(mk.x > 0.f)) v.x = 0.f; if (!(mk.y > 0.f)) v.y = 0.f; if (!(mk.z > 0.f)) v.z = 0.f; if (!(mk.w > 0.f)) v.w = 0.f; }
                }
                *reinterpret_cast<float4 *>(&sdy[r * NS + c]) = v;
            }
        } else {
            for (int i = threadIdx.x; i < WG_ROWS * N; i += 256) {
                const int r = i / N, c = i % N;
                float v = 0.f;
                if (r < nrow) { v = mlp_ld1(dY, dy_bf, row0 * N + i); if (dy_mask && !(mlp_ld1(dy_mask, m_bf, row0 * N + i) > 0.f)) v = 0.f; }
                sdy[r * NS + c] = v;
            }
        }
        if ((K & 3) == 0) {
            const int64_t g4 = row0 * K / 4;
            for (int i = threadIdx.x; i < WG_ROWS * K / 4; i += 256) {
                const int e = i * 4, r = e / K, c = e % K;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < nrow) {
                    v = pcacc_ld4(X, x_bf, g4 + i);
                    if (x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                *reinterpret_cast<float4 *>(&sx[r * KS + c]) = v;
            }
        } else {
            for (int i = threadIdx.x; i < WG_ROWS * K; i += 256) {
                const int r = i / K, c = i % K;
                float v = 0.f;
                if (r < nrow) { v = mlp_ld1(X, x_bf, row0 * K + i); if (x_relu) v = fmaxf(v, 0.f); }
                sx[r * KS + c] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < WG_MAX_TILES; ++t) {
            const int tile = wave + 4 * t;                           // uniform per wave
            if (tile < n_tile_total) {
                const int n = (tile / k_tiles) * 32 + li;
                const int k = (tile % k_tiles) * 32 + li;
                const bool nv = n < N;
                const int kind = k < K ? 0 : (k == K ? 1 : 2);       // data column / ones column (bias) / padding
#pragma unroll 4
                for (int r = 0; r < WG_ROWS; r += 2) {
                    const int rr = r + half;
                    const float a = nv ? sdy[rr * NS + n] : 0.f;
                    const float b = kind == 0 ? sx[rr * KS + k] : ((kind == 1 && rr < nrow) ? 1.0f : 0.f);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    const int KA = K + 1;
#pragma unroll
    for (int t = 0; t < WG_MAX_TILES; ++t) {
        const int tile = wave + 4 * t;
        if (tile < n_tile_total) {
            const int nb = (tile / k_tiles) * 32, k = (tile % k_tiles) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < N && k < KA) atomicAdd(&dW[(int64_t)n * KA + k], acc[t][r]);
            }
        }
    }
}

// One of the two operands has only a few features (the 9 / 4 / 3 / 2 inputs of a first layer, or the 1 / 2 outputs of a head):
// the 32 x 32 MFMA tiles above are mostly padding there and the kernel is LDS-bound at 0.3-1.1 TB/s.  Streamed instead: the WIDE
// operand (C = 32 / 64 / 128 features, rows of C4N = C / 4 lanes, 8- or 16-byte loads) against the NARROW one (F <= 9 features: the
// lanes of a row load one element each and hand them round with wave shuffles); each lane keeps its 4 x F partial products, the
// column sums of its 4 wide features and the sums of the narrow features.  Folded over the lanes of a wave that own the same
// columns (xor-shuffles) and over the 4 waves (LDS); workgroup partials go to the workspace and a second kernel adds them up in
// a fixed order (fp32 atomics on the ~10^3 output words from ~10^3 workgroups serialise in L2: 1.3 ms measured).
//   wide = dY, narrow = X (few inputs) : dW[n][k] = P[n][k], bias column = wide sums
//   wide = X, narrow = dY (few outputs): dW[n][k] = P[k][n], bias column = narrow sums
#define FEW_WU 2                    // (4 rows in flight cost the weight-gradient kernel half its occupancy: 500 us instead of 308)
template <int F, int C4N, int FAST = 0>                                  // FAST: 0 the general loop, 1 / 2 the straight-line bf16 form without / with a mask
__global__ __launch_bounds__(256) void rows_wgrad_few_kernel(const void *__restrict__ wide, const void *__restrict__ wide_mask, int wide_relu,
                                                             const void *__restrict__ narrow, const void *__restrict__ narrow_mask,
                                                             int narrow_relu, int64_t rows, int wide_is_dy, int flags_bf,
                                                             float *__restrict__ partial)
{
    const bool w_bf = flags_bf & 1, wm_bf = flags_bf & 2, n_bf = flags_bf & 4, nm_bf = flags_bf & 8;
    constexpr int V = 4 * F + 4 + F;                                   // values a lane accumulates
    constexpr int C = 4 * C4N, RB = 256 / C4N, NL = (F + C4N - 1) / C4N;
    __shared__ float red[4 * C4N * V];
    const int c4 = threadIdx.x % C4N, rsub = threadIdx.x / C4N;
    const int lane = threadIdx.x & 63, rowbase = lane - c4;
    float acc[F][4], wsum[4] = {0.f, 0.f, 0.f, 0.f}, ssum[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { ssum[f] = 0.f; acc[f][0] = acc[f][1] = acc[f][2] = acc[f][3] = 0.f; }
    const int64_t stride = (int64_t)gridDim.x * RB;
    auto consume = [&](const float4 &wv, const float (&pt)[NL]) {
        wsum[0] += wv.x; wsum[1] += wv.y; wsum[2] += wv.z; wsum[3] += wv.w;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float v = __shfl(pt[f / C4N], rowbase + (f % C4N), 64);
            ssum[f] += v;
            acc[f][0] = fmaf(v, wv.x, acc[f][0]);
            acc[f][1] = fmaf(v, wv.y, acc[f][1]);
            acc[f][2] = fmaf(v, wv.z, acc[f][2]);
            acc[f][3] = fmaf(v, wv.w, acc[f][3]);
        }
    };
    // [r5] the common calls (narrow operand fp32 and unmasked; wide operand and its mask of one element type) as straight-line code: the loads of
    // FEW_WF rows -- wide row piece, its mask piece, the narrow element -- go out back to back from clamped row numbers, conversion / ReLU / mask / range
    // test follow as selects.  The general loop below guards every load with a branch and picks element types at run time: `global_load, s_waitcnt
    // vmcnt(0)` per element and two rows in flight (the 3.2 M x 64 x 9 weight gradient of the pillar encoder's position layer ran at 2.1 TB/s).
    constexpr int FEW_WF = 4;
    // (fp32 wide rows already carry 16 bytes per lane and load: there the four-row form costs occupancy and measured 7 - 20 % slower -- they keep the loop below)
    // Own instantiations (FAST, chosen by the launcher): inside one kernel the two forms shared a register budget and the general loop lost a third of its speed.
    auto fast_pass = [&](auto bf_tag, auto mask_tag) {
        constexpr bool BF = decltype(bf_tag)::value, MASKED = decltype(mask_tag)::value;
        using R = typename std::conditional<BF, uint2, float4>::type;
        auto to_f4 = [](const R &r) {
            if constexpr (BF) return make_float4(pcacc_bf16_lo(r.x), pcacc_bf16_hi(r.x), pcacc_bf16_lo(r.y), pcacc_bf16_hi(r.y));
            else return r;
        };
        const R *wp = static_cast<const R *>(wide), *mp = static_cast<const R *>(wide_mask);
        const float *np = static_cast<const float *>(narrow);
        for (int64_t r0 = (int64_t)blockIdx.x * RB + rsub; r0 < rows; r0 += FEW_WF * stride) {
            R wr[FEW_WF], mr[FEW_WF];
            float nr[FEW_WF][NL];
#pragma unroll
            for (int u = 0; u < FEW_WF; ++u) {
                const int64_t row = r0 + u * stride, rc = row < rows ? row : rows - 1;
                wr[u] = wp[rc * C4N + c4];
                if constexpr (MASKED) mr[u] = mp[rc * C4N + c4];
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int f = c4 + l * C4N;
                    nr[u][l] = np[f < F ? rc * F + f : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < FEW_WF; ++u) {
                const bool in = r0 + u * stride < rows;
                float4 wv = to_f4(wr[u]);
                if (wide_relu) { wv.x = fmaxf(wv.x, 0.f); wv.y = fmaxf(wv.y, 0.f); wv.z = fmaxf(wv.z, 0.f); wv.w = fmaxf(wv.w, 0.f); }
                if constexpr (MASKED) {
                    const float4 mk = to_f4(mr[u]);
                    if (!(mk.x > 0.f)) wv.x = 0.f;
                    if (!(mk.y > 0.f)) wv.y = 0.f;
                    if (!(mk.z > 0.f)) wv.z = 0.f;
                    if (!(mk.w > 0.f)) wv.w = 0.f;
                }
                if (!in) wv = make_float4(0.f, 0.f, 0.f, 0.f);
                float pt[NL];
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    float v = nr[u][l];
                    if (narrow_relu) v = fmaxf(v, 0.f);
                    pt[l] = (in && c4 + l * C4N < F) ? v : 0.f;
                }
                consume(wv, pt);
            }
        }
    };
    if constexpr (FAST == 2) fast_pass(std::true_type(), std::true_type());
    else if constexpr (FAST == 1) fast_pass(std::true_type(), std::false_type());
    else
    for (int64_t r0 = (int64_t)blockIdx.x * RB + rsub; r0 < rows; r0 += FEW_WU * stride) {       // the lanes of a row share r0
        float4 w[FEW_WU];
        float part[FEW_WU][NL];
#pragma unroll
        for (int u = 0; u < FEW_WU; ++u) {
            const int64_t row = r0 + u * stride;
            w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows) {
                w[u] = pcacc_ld4(wide, w_bf, row * C4N + c4);
                if (wide_relu) { w[u].x = fmaxf(w[u].x, 0.f); w[u].y = fmaxf(w[u].y, 0.f); w[u].z = fmaxf(w[u].z, 0.f); w[u].w = fmaxf(w[u].w, 0.f); }
                if (wide_mask) {
                    const float4 mk = pcacc_ld4(wide_mask, wm_bf, row * C4N + c4);
                    if (!(mk.x > 0.f)) w[u].x = 0.f;
                    if (!(mk.y > 0.f)) w[u].y = 0.f;
                    if (!(mk.z > 0.f)) w[u].z = 0.f;
                    if (!(mk.w > 0.f)) w[u].w = 0.f;
                }
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int f = c4 + l * C4N;
                float v = 0.f;
                if (row < rows && f < F) {
                    v = mlp_ld1(narrow, n_bf, row * F + f);
                    if (narrow_relu) v = fmaxf(v, 0.f);
                    if (narrow_mask && !(mlp_ld1(narrow_mask, nm_bf, row * F + f) > 0.f)) v = 0.f;
                }
                part[u][l] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < FEW_WU; ++u) consume(w[u], part[u]);
    }
    // lanes c4, c4 + C4N, ... of a wave own the same columns
    float vals[V];
#pragma unroll
    for (int f = 0; f < F; ++f) { vals[4 * f] = acc[f][0]; vals[4 * f + 1] = acc[f][1]; vals[4 * f + 2] = acc[f][2]; vals[4 * f + 3] = acc[f][3]; vals[4 * F + 4 + f] = ssum[f]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) vals[4 * F + j] = wsum[j];
#pragma unroll
    for (int d = C4N; d < 64; d <<= 1)
#pragma unroll
        for (int i = 0; i < V; ++i) vals[i] += __shfl_xor(vals[i], d, 64);
    const int wave = threadIdx.x >> 6;
    if (lane < C4N)
#pragma unroll
        for (int i = 0; i < V; ++i) red[(wave * C4N + lane) * V + i] = vals[i];
    __syncthreads();
    if (threadIdx.x < C4N) {
#pragma unroll
        for (int i = 0; i < V; ++i)
            vals[i] = (red[threadIdx.x * V + i] + red[(C4N + threadIdx.x) * V + i]) + (red[(2 * C4N + threadIdx.x) * V + i] + red[(3 * C4N + threadIdx.x) * V + i]);
        if (wide_is_dy) {                                              // dW [C][F + 1]
            float *out = partial + (int64_t)blockIdx.x * C * (F + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float *o = out + (c4 * 4 + j) * (F + 1);
#pragma unroll
                for (int f = 0; f < F; ++f) o[f] = vals[4 * f + j];
                o[F] = vals[4 * F + j];
            }
        } else {                                                       // dW [F][C + 1]
            float *out = partial + (int64_t)blockIdx.x * F * (C + 1);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                float *o = out + f * (C + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[c4 * 4 + j] = vals[4 * f + j];
                if (c4 == 0) o[C] = vals[4 * F + 4 + f];
            }
        }
    }
}

// out[e] = sum over the workgroup partials: 16 elements x 16 slices of the partials per workgroup (a fixed order: run-to-run identical).  With 64
// elements x 4 slices a thread walked up to 512 partial slots one dependent load after the other: 37 us for a 64 x 10 gradient.
__global__ __launch_bounds__(256) void rows_wgrad_few_reduce_kernel(const float *__restrict__ partial, int n_parts, int elems, float *__restrict__ out,
                                                                    int split_k)
{
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < elems) {
        int p = slice;
        for (; p + 16 < n_parts; p += 32) {
            s0 += partial[(int64_t)p * elems + e];
            s1 += partial[(int64_t)(p + 16) * elems + e];
        }
        if (p < n_parts) s0 += partial[(int64_t)p * elems + e];
    }
    red[slice][el] = s0 + s1;
    __syncthreads();
    if (slice == 0 && e < elems) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][el];
        int o = e;
        if (split_k > 0) {                                                    // dW [n][k] then the bias gradients [n], both contiguous
            const int row = e / (split_k + 1), col = e % (split_k + 1);
            o = col < split_k ? row * split_k + col : (elems / (split_k + 1)) * split_k + row;
        }
        out[o] = v;
    }
}

static bool wgrad_few_width(int c) { return c == 32 || c == 64 || c == 128; }
static bool wgrad_few_feats(int f) { return f == 1 || f == 2 || f == 3 || f == 4 || f == 9; }
static int wgrad_few_grid(int64_t rows, int c)
{
    const int rb = 256 / (c / 4);
    const int64_t tiles = (rows + FEW_WU * rb - 1) / (FEW_WU * rb);
    return (int)(tiles < PCACC_CUS * 4 ? (tiles < 1 ? 1 : tiles) : PCACC_CUS * 4);
}

extern "C" int pcacc_rows_wgrad_few_supported(int32_t k, int32_t n)
{
    return ((wgrad_few_feats(k) && wgrad_few_width(n)) || (wgrad_few_feats(n) && wgrad_few_width(k))) ? 1 : 0;
}

extern "C" int pcacc_rows_wgrad_few_workspace_bytes(int64_t rows, int32_t k, int32_t n, size_t *bytes)
{
    if (!bytes || rows < 0 || !pcacc_rows_wgrad_few_supported(k, n)) return PCACC_E_ARG;
    const bool few_in = wgrad_few_feats(k) && wgrad_few_width(n);
    *bytes = (size_t)wgrad_few_grid(rows, few_in ? n : k) * n * (k + 1) * sizeof(float);
    return PCACC_OK;
}

extern "C" int pcacc_rows_wgrad_few(const void *dy, const void *dy_mask, const void *x, int32_t x_relu, int64_t rows, int32_t k, int32_t n,
                                    float *dw_aug, int32_t dt, void *workspace, size_t workspace_bytes, void *stream)
{
    if (rows < 0 || !dw_aug || (dt & ~7) || !pcacc_rows_wgrad_few_supported(k, n)) return PCACC_E_ARG;
    const int split_k = (x_relu & 2) ? k : 0;                                 // flags: bit 0 = ReLU on X, bit 1 = split result layout
    x_relu &= 1;
    hipStream_t s = pcacc_stream(stream);
    const int elems = n * (k + 1);
    if (rows == 0) {
        if (hipMemsetAsync(dw_aug, 0, (size_t)elems * sizeof(float), s) != hipSuccess) return PCACC_E_LAUNCH;
        return PCACC_OK;
    }
    if (!dy || !x || !workspace) return PCACC_E_ARG;
    const bool few_in = wgrad_few_feats(k) && wgrad_few_width(n);
    const int c = few_in ? n : k, f = few_in ? k : n;
    const int grid = wgrad_few_grid(rows, c);
    if (workspace_bytes < (size_t)grid * elems * sizeof(float)) return PCACC_E_WORKSPACE;
    float *partial = reinterpret_cast<float *>(workspace);
    // bf16 flags: wide, wide mask, narrow, narrow mask (dt: bit 0 dY, bit 1 dy_mask, bit 2 X)
    const int fb = few_in ? (dt & 7) : (((dt & 4) ? 1 : 0) | ((dt & 1) ? 4 : 0) | ((dt & 2) ? 8 : 0));
    // few inputs, bf16 dY (and mask), fp32 X: the straight-line form (rows_wgrad_few_kernel, FAST)
    const bool fast = few_in && (fb & 1) && !(fb & 4) && (!dy_mask || (fb & 2));
#define WG_FEW(FF, CC)                                                                                                                     \
    do {                                                                                                                                   \
        if (few_in && fast && dy_mask) rows_wgrad_few_kernel<FF, CC, 2><<<grid, 256, 0, s>>>(dy, dy_mask, 0, x, nullptr, x_relu, rows, 1, fb, partial); \
        else if (few_in && fast) rows_wgrad_few_kernel<FF, CC, 1><<<grid, 256, 0, s>>>(dy, dy_mask, 0, x, nullptr, x_relu, rows, 1, fb, partial); \
        else if (few_in) rows_wgrad_few_kernel<FF, CC><<<grid, 256, 0, s>>>(dy, dy_mask, 0, x, nullptr, x_relu, rows, 1, fb, partial);     \
        else rows_wgrad_few_kernel<FF, CC><<<grid, 256, 0, s>>>(x, nullptr, x_relu, dy, dy_mask, 0, rows, 0, fb, partial);                 \
    } while (0)
#define WG_FEW_C(FF) do { if (c == 32) WG_FEW(FF, 8); else if (c == 64) WG_FEW(FF, 16); else WG_FEW(FF, 32); } while (0)
    switch (f) {
        case 1: WG_FEW_C(1); break;
        case 2: WG_FEW_C(2); break;
        case 3: WG_FEW_C(3); break;
        case 4: WG_FEW_C(4); break;
        default: WG_FEW_C(9); break;
    }
#undef WG_FEW_C
#undef WG_FEW
    rows_wgrad_few_reduce_kernel<<<(elems + 15) / 16, 256, 0, s>>>(partial, grid, elems, dw_aug, split_k);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

static int rows_wgrad_any(const void *dy, const void *dy_mask, const void *x, int x_relu, int64_t rows, int k, int n, float *dw_aug,
                          int dt, void *stream)
{
    if (rows < 0 || k <= 0 || n <= 0 || k > 128 || n > 128 || !dw_aug) return PCACC_E_ARG;
    hipStream_t s = pcacc_stream(stream);
    if (hipMemsetAsync(dw_aug, 0, (size_t)n * (k + 1) * sizeof(float), s) != hipSuccess) return PCACC_E_LAUNCH;
    if (rows == 0) return PCACC_OK;
    if (!dy || !x) return PCACC_E_ARG;
    const int k_tiles = (k + 1 + 31) / 32, n_tiles = (n + 31) / 32;
    const int total = k_tiles * n_tiles;
    if (total > 24) return PCACC_E_ARG;
    // rows per staged chunk: as many as ~48 KB of LDS hold (more MFMA work per barrier pair), 32 for the 128-wide layers
    const int wg_rows = (n + k + 8) * 4 * 96 <= 49152 ? 96 : 32;
    const int64_t n_chunks = (rows + wg_rows - 1) / wg_rows;
    int grid = PCACC_CUS * (total <= 4 ? 4 : 2);
    if (grid > n_chunks) grid = (int)n_chunks;
    const size_t lds = (size_t)wg_rows * (n + 4 + k + 4) * sizeof(float);
#define WG_LAUNCH(R, T) rows_wgrad_kernel<R, T><<<grid, 256, lds, s>>>(dy, dy_mask, x, x_relu, rows, k, n, k_tiles, total, dw_aug, dt)
    if (wg_rows == 96) {
        if (total <= 4) WG_LAUNCH(96, 1);
        else if (total <= 8) WG_LAUNCH(96, 2);
        else WG_LAUNCH(96, 6);
    } else {
        if (total <= 4) WG_LAUNCH(32, 1);
        else if (total <= 8) WG_LAUNCH(32, 2);
        else WG_LAUNCH(32, 6);
    }
#undef WG_LAUNCH
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_rows_wgrad(const float *dy, const float *dy_mask, const float *x, int x_relu, int64_t rows, int k, int n,
                                float *dw_aug, void *stream)
{
    return rows_wgrad_any(dy, dy_mask, x, x_relu, rows, k, n, dw_aug, 0, stream);
}

extern "C" int pcacc_rows_wgrad_mixed(const void *dy, const void *dy_mask, const void *x, int x_relu, int64_t rows, int k, int n,
                                      float *dw_aug, int dtypes, void *stream)
{
    if (dtypes & ~7) return PCACC_E_ARG;
    return rows_wgrad_any(dy, dy_mask, x, x_relu, rows, k, n, dw_aug, dtypes, stream);
}
