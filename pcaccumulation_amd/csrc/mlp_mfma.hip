// Per-point linear layers on the bf16 matrix cores, bf16 rows in and out (the bf16 compute mode of the point MLPs:
// STPN point heads models/stpn.py:94-102, TubeNet embeddings models/tpointnet.py:176-196, pillar encoder blocks
// models/pillar_encoder.py:13-55).  Same contract as rows_linear in mlp.hip,
//     Y = [relu]( [relu|mask](X) @ W^T + b [+ residual] ) [masked],
// with X, Y, residual and the two masks stored as bf16, W and b as fp32 (rounded to bf16 once per launch), fp32
// accumulation.  At 128 features the fp32 version is VALU-bound (0.57 ms per 3 M rows); here the multiply is ~5 % of the
// matrix-core peak and the kernel streams rows at HBM speed with half the bytes.
//
// Workgroup = 4 waves, persistent over 128-row tiles.  v_mfma_f32_32x32x16_bf16 with A = 32 output features x 16 k (weights,
// resident in LDS), B = 16 k x 32 rows (the staged tile): D comes out with lane = row, so the result is written to LDS as
// 8-byte channel quads and leaves through fully coalesced 16-byte stores, where residual / ReLU / output mask are applied.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define MM_TILE 128
#define MM_THREADS 256
#define MM_PRE_RELU 1
#define MM_POST_RELU 2

// max(x, 0) on two packed bf16: clear the halves whose sign bit is set
__device__ __forceinline__ uint32_t mm_relu2(uint32_t v)
{
    const uint32_t neg = (v >> 15) & 0x00010001u;
    return v & ~(neg * 0xffffu);
}

// keep the halves of v whose mask half is > 0
__device__ __forceinline__ uint32_t mm_mask2(uint32_t v, uint32_t m)
{
    const uint32_t lo = m & 0xffffu, hi = m >> 16;                        // > 0: sign clear, not zero, not NaN
    const uint32_t lo_ok = (lo != 0 && lo <= 0x7f80u) ? 0x0000ffffu : 0u;
    const uint32_t hi_ok = (hi != 0 && hi <= 0x7f80u) ? 0xffff0000u : 0u;
    return v & (lo_ok | hi_ok);
}

__device__ __forceinline__ uint4 mm_relu8(uint4 v) { return make_uint4(mm_relu2(v.x), mm_relu2(v.y), mm_relu2(v.z), mm_relu2(v.w)); }
__device__ __forceinline__ uint4 mm_mask8(uint4 v, uint4 m)
{
    return make_uint4(mm_mask2(v.x, m.x), mm_mask2(v.y, m.y), mm_mask2(v.z, m.z), mm_mask2(v.w, m.w));
}


// A row made of two pieces ("virtual concatenation"): columns [0,ka) from a[row], columns [ka,K) from b[idx[row]] (idx == NULL:
// b[row]).  b == NULL: the plain contiguous [rows,K] layout of `a`.  Lets the PFN blocks consume cat(point row, pooled
// pillar row) (models/pillar_encoder.py:116-118) without materialising the gather or the concatenation.
struct RowPieces {
    const uint16_t *b;
    const int32_t *idx;
    int ka;
};
__device__ __forceinline__ const uint16_t *row_piece(const uint16_t *a, const RowPieces &s, int K, int64_t row, int col)
{
    if (!s.b) return a + row * K + col;
    if (col < s.ka) return a + row * s.ka + col;
    return s.b + (s.idx ? (int64_t)s.idx[row] : row) * (K - s.ka) + (col - s.ka);
}

template <int K, int CT>
__global__ __launch_bounds__(MM_THREADS) void rows_linear_bf16_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ in_mask,
                                                                      const float *__restrict__ W, const float *__restrict__ bias,
                                                                      const uint16_t *__restrict__ residual,
                                                                      const uint16_t *__restrict__ out_mask, uint16_t *__restrict__ Y,
                                                                      int64_t rows, int flags, RowPieces xs2, RowPieces ms2,
                                                                      uint16_t *__restrict__ Y2, int na)
{
    constexpr int N = CT * 32;
    constexpr int XS = K + 8, YS = N + 8;                      // padded LDS row lengths (elements)
    constexpr int REGION = MM_TILE * (XS > YS ? XS : YS);      // input tile, later the output tile
    constexpr int X_CHUNKS = MM_TILE * K / 8;                  // 16-byte pieces of an input tile
    constexpr int X_PER_THREAD = X_CHUNKS / MM_THREADS;
    constexpr int Y_CHUNKS = MM_TILE * N / 8;
    constexpr int Y_PER_THREAD = Y_CHUNKS / MM_THREADS;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *xs = lds;
    uint16_t *ws = lds + REGION;                               // [N][XS]
    float *bias_l = reinterpret_cast<float *>(ws + N * XS);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;

    for (int e = threadIdx.x; e < N * K / 2; e += MM_THREADS) {               // weights: fp32 pairs -> packed bf16
        const int n = (2 * e) / K, k = (2 * e) % K;
        const float2 w2 = *reinterpret_cast<const float2 *>(W + (int64_t)n * K + k);
        *reinterpret_cast<uint32_t *>(ws + n * XS + k) = pcacc_pack_bf16x2(w2.x, w2.y);
    }
    if (threadIdx.x < N) bias_l[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;

    const int64_t n_tiles = (rows + MM_TILE - 1) / MM_TILE;
    uint4 xreg[X_PER_THREAD], mreg[X_PER_THREAD];
    auto fetch = [&](int64_t tile) {
        const int64_t base = tile * MM_TILE * K;                              // element offset of the tile
        const int64_t limit = rows * K;
#pragma unroll
        for (int q = 0; q < X_PER_THREAD; ++q) {
            const int64_t e = base + (int64_t)(threadIdx.x + q * MM_THREADS) * 8;
            uint4 v = make_uint4(0, 0, 0, 0), m = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
            if (e < limit) {
                v = *reinterpret_cast<const uint4 *>(row_piece(X, xs2, K, e / K, (int)(e % K)));
                if (in_mask) m = *reinterpret_cast<const uint4 *>(in_mask + e);
            }
            xreg[q] = v;
            mreg[q] = m;
        }
    };

    int64_t tile = blockIdx.x;
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                                      // the previous tile's output left the region
#pragma unroll
        for (int q = 0; q < X_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * MM_THREADS;
            uint4 v = xreg[q];
            if (flags & MM_PRE_RELU) v = mm_relu8(v);
            if (in_mask) v = mm_mask8(v, mreg[q]);
            *reinterpret_cast<uint4 *>(xs + (c / (K / 8)) * XS + (c % (K / 8)) * 8) = v;
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);              // in flight during the MFMAs and the store phase

        f32x16_t acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const uint16_t *xrow = xs + (wave * 32 + lp) * XS + lh * 8;
        const uint16_t *wrow = ws + lp * XS + lh * 8;
#pragma unroll
        for (int kc = 0; kc < K / 16; ++kc) {
            const bf16x8_t b = *reinterpret_cast<const bf16x8_t *>(xrow + kc * 16);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const bf16x8_t a = *reinterpret_cast<const bf16x8_t *>(wrow + ct * 32 * XS + kc * 16);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ct], 0, 0, 0);
            }
        }
        __syncthreads();                                                      // every wave is done reading the input tile
        uint16_t *yrow = xs + (wave * 32 + lp) * YS;                          // lane = row; quads of 4 consecutive features
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = ct * 32 + 8 * g + 4 * lh;
                const float4 bv = *reinterpret_cast<const float4 *>(bias_l + c);
                uint2 pk;
                pk.x = pcacc_pack_bf16x2(acc[ct][4 * g] + bv.x, acc[ct][4 * g + 1] + bv.y);
                pk.y = pcacc_pack_bf16x2(acc[ct][4 * g + 2] + bv.z, acc[ct][4 * g + 3] + bv.w);
                *reinterpret_cast<uint2 *>(yrow + c) = pk;
            }
        __syncthreads();
        const int64_t ybase = tile * MM_TILE * N, ylimit = rows * N;
#pragma unroll
        for (int q = 0; q < Y_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * MM_THREADS;
            const int64_t e = ybase + (int64_t)c * 8;
            if (e >= ylimit) continue;
            uint4 v = *reinterpret_cast<const uint4 *>(xs + (c / (N / 8)) * YS + (c % (N / 8)) * 8);
            if (residual) {
                const uint4 r = *reinterpret_cast<const uint4 *>(residual + e);
                v.x = pcacc_pack_bf16x2(pcacc_bf16_lo(v.x) + pcacc_bf16_lo(r.x), pcacc_bf16_hi(v.x) + pcacc_bf16_hi(r.x));
                v.y = pcacc_pack_bf16x2(pcacc_bf16_lo(v.y) + pcacc_bf16_lo(r.y), pcacc_bf16_hi(v.y) + pcacc_bf16_hi(r.y));
                v.z = pcacc_pack_bf16x2(pcacc_bf16_lo(v.z) + pcacc_bf16_lo(r.z), pcacc_bf16_hi(v.z) + pcacc_bf16_hi(r.z));
                v.w = pcacc_pack_bf16x2(pcacc_bf16_lo(v.w) + pcacc_bf16_lo(r.w), pcacc_bf16_hi(v.w) + pcacc_bf16_hi(r.w));
            }
            if (flags & MM_POST_RELU) v = mm_relu8(v);
            const int64_t row = e / N;
            const int col = (int)(e % N);
            if (out_mask) v = mm_mask8(v, *reinterpret_cast<const uint4 *>(row_piece(out_mask, ms2, N, row, col)));
            if (!Y2) *reinterpret_cast<uint4 *>(Y + e) = v;
            else if (col < na) *reinterpret_cast<uint4 *>(Y + row * na + col) = v;
            else *reinterpret_cast<uint4 *>(Y2 + row * (N - na) + (col - na)) = v;
        }
    }
}

template <int K, int CT>
static int mm_launch(const uint16_t *x, const uint16_t *in_mask, const float *w, const float *bias, const uint16_t *residual,
                     const uint16_t *out_mask, uint16_t *y, int64_t rows, int flags, hipStream_t st, RowPieces xs2 = RowPieces{nullptr, nullptr, 0},
                     RowPieces ms2 = RowPieces{nullptr, nullptr, 0}, uint16_t *y2 = nullptr, int na = 0)
{
    constexpr int N = CT * 32;
    constexpr int XS = K + 8, YS = N + 8;
    const size_t lds = (size_t)(MM_TILE * (XS > YS ? XS : YS) + N * XS) * sizeof(uint16_t) + N * sizeof(float);
    auto kern = rows_linear_bf16_kernel<K, CT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    const int64_t n_tiles = (rows + MM_TILE - 1) / MM_TILE;
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    int64_t grid = (int64_t)PCACC_CUS * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(MM_THREADS), lds, st, x, in_mask, w, bias, residual, out_mask, y, rows, flags, xs2, ms2, y2, na);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_rows_linear_bf16(const uint16_t *x, const uint16_t *in_mask, const float *w, const float *bias,
                                      const uint16_t *residual, const uint16_t *out_mask, uint16_t *y, int64_t rows, int32_t k,
                                      int32_t n, int32_t flags, void *stream)
{
    if (rows < 0 || (k != 32 && k != 64 && k != 128) || (n != 32 && n != 64 && n != 128)) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!x || !w || !y) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
#define MM_CASE(KK, CTV) \
    if (k == KK && n == CTV * 32) return mm_launch<KK, CTV>(x, in_mask, w, bias, residual, out_mask, y, rows, flags, st)
    MM_CASE(32, 1); MM_CASE(32, 2); MM_CASE(32, 4);
    MM_CASE(64, 1); MM_CASE(64, 2); MM_CASE(64, 4);
    MM_CASE(128, 1); MM_CASE(128, 2); MM_CASE(128, 4);
#undef MM_CASE
    return PCACC_E_ARG;
}

// The same layer on rows made of two pieces (see RowPieces).  Forward: x = cat(xa [rows,ka], xb[b_index] [.,k-ka]).  Backward-data
// (w = W^T, x = the output gradient, xb = NULL): the [rows,n] result leaves as y [rows,na] and y2 [rows,n-na], masked where the
// forward input cat(out_mask_a, out_mask_b[b_index]) was <= 0.
extern "C" int pcacc_rows_linear_cat_bf16(const uint16_t *xa, const uint16_t *xb, const int32_t *b_index, int32_t ka, const uint16_t *in_mask,
                                          const float *w, const float *bias, const uint16_t *residual, const uint16_t *out_mask_a,
                                          const uint16_t *out_mask_b, uint16_t *y, uint16_t *y2, int32_t na, int64_t rows, int32_t k, int32_t n,
                                          int32_t flags, void *stream)
{
    if (rows < 0 || (k != 32 && k != 64 && k != 128) || (n != 32 && n != 64 && n != 128)) return PCACC_E_ARG;
    if (xb && (ka <= 0 || ka >= k || ka % 8)) return PCACC_E_ARG;
    if (y2 && (na <= 0 || na >= n || na % 8)) return PCACC_E_ARG;
    if (out_mask_b && (!out_mask_a || !y2)) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!xa || !w || !y) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    const RowPieces xs2{xb, b_index, ka}, ms2{out_mask_b, b_index, na};
#define MM_CASE(KK, CTV) \
    if (k == KK && n == CTV * 32) return mm_launch<KK, CTV>(xa, in_mask, w, bias, residual, out_mask_a, y, rows, flags, st, xs2, ms2, y2, na)
    MM_CASE(32, 1); MM_CASE(32, 2); MM_CASE(32, 4);
    MM_CASE(64, 1); MM_CASE(64, 2); MM_CASE(64, 4);
    MM_CASE(128, 1); MM_CASE(128, 2); MM_CASE(128, 4);
#undef MM_CASE
    return PCACC_E_ARG;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight / bias gradient from bf16 rows on the bf16 matrix cores:  dW_aug[n][k] (k in [0,K]; column K = bias gradient)
//   = sum_r dYeff[r][n] * Xaug[r][k],  Xaug[r][K] = 1,  dYeff = dY masked where dy_mask <= 0,  X optionally ReLU'd.
// The reduction runs over rows, so both MFMA operands need 8 consecutive ROWS of one column per lane: the tiles are staged
// row-major (coalesced, masks applied on the way) and the fragments come through the hardware transpose read
// (ds_read_b64_tr_b16, two per fragment) instead of eight 16-bit gathers.  k and n are multiples of 32.
// Workgroup partials are written to a workspace with plain stores and summed by a second launch: with ~1000 workgroups
// an atomic per element per workgroup is 17 M same-address atomics for a 128 x 129 gradient, longer than the products.
// ---------------------------------------------------------------------------------------------------------------------
// Rows per staged tile: 64 for the widest layers (128 features each side: 4 sixteen-byte pieces of dY and of X per thread), more for the
// narrow ones so that every thread still has 4 + 4 pieces in flight and a barrier pair is paid per 128 / 256 rows instead of per 64
// (the 32 -> 32 layers of the pillar encoder ran 49 tile iterations of 8 MFMAs each per workgroup: barrier-bound at 3.6 TB/s).
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
union wg_frag { bf16x8_t v; wg_s16x4 h[2]; uint16_t e[8]; };

template <int MAX_TILES, int WG_R, int NW = 4>
__global__ __launch_bounds__(NW * 64) void rows_wgrad_bf16_kernel(const uint16_t *__restrict__ dY, const uint16_t *__restrict__ dy_mask,
                                                              const uint16_t *__restrict__ X, int x_relu, int64_t rows, int K, int N,
                                                              int k_tiles, int n_tile_total, int tiles_par, float *partial, RowPieces xs2,
                                                              float *__restrict__ dw_zero)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t wlds[];
    if (blockIdx.x == 0)                                                       // the reduce launch adds into dW: cleared here, not by a memset
        for (int e = threadIdx.x; e < N * (K + 1); e += NW * 64) dw_zero[e] = 0.f;
    const int NS = pcacc_tr_stride(N), KS = pcacc_tr_stride(K);
    uint16_t *sdy = wlds, *sx = wlds + WG_R * NS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    f32x16_t acc[MAX_TILES];
#pragma unroll
    for (int t = 0; t < MAX_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Staging: up to 4 sixteen-byte pieces of dY and of X per thread (64 rows x 128 features), fetched one tile ahead.
    const int64_t n_chunks = (rows + WG_R - 1) / WG_R;
    const int ny = WG_R * N / 8, nx = WG_R * K / 8;                              // pieces per tile
    constexpr int PIECES = 16 / NW;                                            // 16-byte pieces of dY and of X per thread and tile
    uint4 yreg[PIECES], mreg[PIECES], xreg[PIECES];
    const int kshift = __ffs(K) - 1;                                           // two-piece rows: K is a power of two
    int prow[PIECES] = {};                                                // rows of the second piece for the NEXT fetch: the index
    auto fetch_rows = [&](int64_t ch) {                                        // load and the row load it feeds are a tile apart
        if (!xs2.b) return;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int i = threadIdx.x + q * NW * 64;
            const int64_t row = ch * WG_R + ((i * 8) >> kshift);
            prow[q] = (i < nx && row < rows && ch < n_chunks) ? (xs2.idx ? xs2.idx[row] : (int)row) : 0;
        }
    };
    auto fetch = [&](int64_t ch) {
        const int64_t row0 = ch * WG_R;
        const int64_t lim_n = (rows - row0) * N, lim_k = (rows - row0) * K;      // elements of this tile that exist
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int i = threadIdx.x + q * NW * 64;
            uint4 v = make_uint4(0, 0, 0, 0), m = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
            if (i < ny && (int64_t)i * 8 < lim_n) {
                v = *reinterpret_cast<const uint4 *>(dY + row0 * N + (int64_t)i * 8);
                if (dy_mask) m = *reinterpret_cast<const uint4 *>(dy_mask + row0 * N + (int64_t)i * 8);
            }
            yreg[q] = v;
            mreg[q] = m;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (i < nx && (int64_t)i * 8 < lim_k) {
                const int col = (i * 8) & (K - 1);
                const uint16_t *src = X + row0 * K + (int64_t)i * 8;
                if (xs2.b)                                                       // two-piece rows: the gathered row's index was
                    src = col < xs2.ka ? X + (row0 + ((i * 8) >> kshift)) * xs2.ka + col   // fetched one tile earlier (prow)
                                       : xs2.b + (int64_t)prow[q] * (K - xs2.ka) + (col - xs2.ka);
                u = *reinterpret_cast<const uint4 *>(src);
            }
            xreg[q] = u;
        }
    };
    int64_t ch = blockIdx.x;
    fetch_rows(ch);
    if (ch < n_chunks) fetch(ch);
    fetch_rows(ch + gridDim.x);
    for (; ch < n_chunks; ch += gridDim.x) {
        const int64_t row0 = ch * WG_R;
        __syncthreads();                                                         // the previous tile's gathers are done
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int i = threadIdx.x + q * NW * 64;
            if (i < ny) {
                const int e = i * 8;
                uint4 v = yreg[q];
                if (dy_mask) v = mm_mask8(v, mreg[q]);
                uint2 *dst = reinterpret_cast<uint2 *>(sdy + (e / N) * NS + e % N);
                dst[0] = make_uint2(v.x, v.y);
                dst[1] = make_uint2(v.z, v.w);
            }
            if (i < nx) {
                const int e = i * 8;
                uint4 v = xreg[q];
                if (x_relu) v = mm_relu8(v);
                uint2 *dst = reinterpret_cast<uint2 *>(sx + (e / K) * KS + e % K);
                dst[0] = make_uint2(v.x, v.y);
                dst[1] = make_uint2(v.z, v.w);
            }
        }
        __syncthreads();
        if (ch + gridDim.x < n_chunks) fetch(ch + gridDim.x);
        fetch_rows(ch + 2 * (int64_t)gridDim.x);
        const int nrow = (int)min((int64_t)WG_R, rows - row0);
        // Fragments through the LDS transpose read: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x
        // 16-column block whose 16 four-element pieces the lanes address (piece i = row i>>2, columns 4*(i&3)..+3).  Groups
        // 0/1 cover columns 0-15 / 16-31 of rows 0-7, groups 2/3 the same columns of rows 8-15: two reads give the lane its
        // 8 consecutive rows of one column, which is the MFMA operand (k = rows).
        const int g = lane >> 4, li = lane & 15;
        const int tr_row = (g >> 1) * 8 + (li >> 2), tr_col = (g & 1) * 16 + (li & 3) * 4;
        // few output tiles (tiles_par = 1 or 2 of them in parallel): the waves split the rows of the staged tile instead of idling
        // (row group = wave / tiles_par; every group keeps its own partial slot)
        const int rgroups = NW / tiles_par, rgrp = wave / tiles_par;
        const int r_lo = rgrp * (WG_R / rgroups), r_hi = r_lo + WG_R / rgroups;
#pragma unroll
        for (int t = 0; t < MAX_TILES; ++t) {
            const int tile = wave % tiles_par + tiles_par * t;                   // uniform per wave
            if (tile < n_tile_total) {
                const int nt = tile / k_tiles, kt = tile % k_tiles;
                const bool ones = kt * 32 >= K;                                  // the tile that holds the bias column (k == K)
                const uint16_t *pa = sdy + tr_row * NS + nt * 32 + tr_col;
                const uint16_t *pb = sx + tr_row * KS + (ones ? 0 : kt * 32) + tr_col;
                for (int r0 = r_lo; r0 < r_hi; r0 += 16) {
                    wg_frag a, b;
                    a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3))) *)(pa + r0 * NS));
                    a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3))) *)(pa + (r0 + 4) * NS));
                    if (!ones) {
                        b.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3))) *)(pb + r0 * KS));
                        b.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3))) *)(pb + (r0 + 4) * KS));
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            b.e[j] = (kt * 32 + lp == K && r0 + 8 * lh + j < nrow) ? (uint16_t)0x3f80 : (uint16_t)0;
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[t], 0, 0, 0);
                }
            }
        }
    }
    const int KA = K + 1;
    float *mine = partial + ((int64_t)blockIdx.x * (NW / tiles_par) + wave / tiles_par) * N * KA;
#pragma unroll
    for (int t = 0; t < MAX_TILES; ++t) {
        const int tile = wave % tiles_par + tiles_par * t;
        if (tile < n_tile_total) {
            const int nb = (tile / k_tiles) * 32, k = (tile % k_tiles) * 32 + lp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < N && k < KA) mine[(int64_t)n * KA + k] = acc[t][r];
            }
        }
    }
}

// out[e] = sum over the workgroup partials in a fixed order (common.h: pcacc_reduce_partials) -- run-to-run identical
// split_k > 0: the [n][split_k + 1] result is written as dW [n][split_k] followed by the bias gradients [n], both contiguous
template <int EL>
__global__ __launch_bounds__(1024) void rows_wgrad_reduce_kernel(const float *__restrict__ partial, int n_parts, int elems, float *__restrict__ out, int split_k)
{
    pcacc_reduce_partials<EL>(partial, n_parts, elems, [&](int e, float v) {
        int o = e;
        if (split_k > 0) {
            const int row = e / (split_k + 1), col = e % (split_k + 1);
            o = col < split_k ? row * split_k + col : (elems / (split_k + 1)) * split_k + row;
        }
        out[o] = v;
    });
}

static int wgrad_bf16_tile_rows(int k, int n) { return k + n <= 64 ? 256 : (k + n <= 128 ? 128 : 64); }
static int wgrad_bf16_waves(int total) { return total > 12 ? 8 : 4; }       // 20 tiles (128 x 129): 8 waves x 3 accumulators, 2 workgroups per CU
static int wgrad_bf16_tiles_par(int total) { return total <= 1 ? 1 : (total <= 2 ? 2 : (total > 12 ? 8 : 4)); }

static int wgrad_bf16_grid(int64_t rows, int tile_rows)
{
    const int64_t n_chunks = (rows + tile_rows - 1) / tile_rows;
    int64_t grid = PCACC_CUS * 4;
    return (int)(grid > n_chunks ? n_chunks : grid);
}

extern "C" int pcacc_rows_wgrad_bf16_workspace_bytes(int64_t rows, int32_t k, int32_t n, size_t *bytes)
{
    if (!bytes || rows < 0 || k <= 0 || n <= 0) return PCACC_E_ARG;
    const int total = ((k + 1 + 31) / 32) * ((n + 31) / 32);
    *bytes = (size_t)(rows > 0 ? wgrad_bf16_grid(rows, wgrad_bf16_tile_rows(k, n)) : 0) * (wgrad_bf16_waves(total) / wgrad_bf16_tiles_par(total)) * n * (k + 1) * sizeof(float);
    return PCACC_OK;
}

static int rows_wgrad_bf16_any(const uint16_t *dy, const uint16_t *dy_mask, const uint16_t *x, RowPieces xs2, int32_t x_relu, int64_t rows,
                               int32_t k, int32_t n, float *dw_aug, void *workspace, size_t workspace_bytes, void *stream)
{
    if (rows < 0 || k <= 0 || n <= 0 || k > 128 || n > 128 || (k % 32) || (n % 32) || !dw_aug) return PCACC_E_ARG;
    const int split_k = (x_relu & 2) ? k : 0;                                 // flags: bit 0 = ReLU on X, bit 1 = split result layout
    x_relu &= 1;
    hipStream_t st = pcacc_stream(stream);
    if (rows == 0) {
        if (hipMemsetAsync(dw_aug, 0, (size_t)n * (k + 1) * sizeof(float), st) != hipSuccess) return PCACC_E_LAUNCH;
        return PCACC_OK;
    }
    if (!dy || !x || !workspace) return PCACC_E_ARG;
    const int k_tiles = (k + 1 + 31) / 32, n_tiles = (n + 31) / 32;
    const int total = k_tiles * n_tiles;
    if (total > 24) return PCACC_E_ARG;
    const int tile_rows = wgrad_bf16_tile_rows(k, n), tiles_par = wgrad_bf16_tiles_par(total), parts_per_wg = wgrad_bf16_waves(total) / tiles_par;
    const int grid = wgrad_bf16_grid(rows, tile_rows);
    const int elems = n * (k + 1);
    if (workspace_bytes < (size_t)grid * parts_per_wg * elems * sizeof(float)) return PCACC_E_WORKSPACE;
    float *partial = reinterpret_cast<float *>(workspace);
    const size_t lds = (size_t)tile_rows * (pcacc_tr_stride(n) + pcacc_tr_stride(k)) * sizeof(uint16_t);
#define WGB(T, R) rows_wgrad_bf16_kernel<T, R><<<grid, 256, lds, st>>>(dy, dy_mask, x, x_relu, rows, k, n, k_tiles, total, tiles_par, partial, xs2, dw_aug)
    if (total <= 4) { if (tile_rows == 256) WGB(1, 256); else if (tile_rows == 128) WGB(1, 128); else WGB(1, 64); }
    else if (total <= 8) { if (tile_rows == 128) WGB(2, 128); else WGB(2, 64); }
    else if (total <= 12) WGB(3, 64);
    else rows_wgrad_bf16_kernel<3, 64, 8><<<grid, 512, lds, st>>>(dy, dy_mask, x, x_relu, rows, k, n, k_tiles, total, tiles_par, partial, xs2, dw_aug);
#undef WGB
    if (pcacc_reduce_el(elems) == 64) rows_wgrad_reduce_kernel<64><<<(elems + 63) / 64, 1024, 0, st>>>(partial, grid * parts_per_wg, elems, dw_aug, split_k);
    else rows_wgrad_reduce_kernel<16><<<(elems + 15) / 16, 1024, 0, st>>>(partial, grid * parts_per_wg, elems, dw_aug, split_k);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_rows_wgrad_bf16(const uint16_t *dy, const uint16_t *dy_mask, const uint16_t *x, int32_t x_relu, int64_t rows,
                                     int32_t k, int32_t n, float *dw_aug, void *workspace, size_t workspace_bytes, void *stream)
{
    return rows_wgrad_bf16_any(dy, dy_mask, x, RowPieces{nullptr, nullptr, 0}, x_relu, rows, k, n, dw_aug, workspace, workspace_bytes, stream);
}

// x = cat(xa [rows,ka], xb[b_index] [.,k-ka]) (see RowPieces); workspace as pcacc_rows_wgrad_bf16_workspace_bytes(rows, k, n)
extern "C" int pcacc_rows_wgrad_cat_bf16(const uint16_t *dy, const uint16_t *dy_mask, const uint16_t *xa, const uint16_t *xb,
                                         const int32_t *b_index, int32_t ka, int32_t x_relu, int64_t rows, int32_t k, int32_t n,
                                         float *dw_aug, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!xb || ka <= 0 || ka >= k || ka % 8 || (k & (k - 1))) return PCACC_E_ARG;
    return rows_wgrad_bf16_any(dy, dy_mask, xa, RowPieces{xb, b_index, ka}, x_relu, rows, k, n, dw_aug, workspace, workspace_bytes, stream);
}
