// Per-point linear layers at fp32 accuracy on the 16-bit matrix cores ("fp32x3" compute mode): fp32 rows in and out, every product
// formed from fp16 hi / lo halves of power-of-two scaled operands (three v_mfma_f32_32x32x16_f16 per fragment pair, fp32 accumulation;
// see conv_split.hip for the arithmetic and its error bound).  Same contracts as the bf16 kernels of mlp_mfma.hip,
//     Y = [relu]( [relu|mask](X) @ W^T + b [+ residual] ) [masked],          dW_aug = dYeff^T @ [Xeff | 1],
// on the pillar encoder (models/pillar_encoder.py:13-55,113-122), the STPN point heads (models/stpn.py:94-102) and the TubeNet
// embeddings (models/tpointnet.py:176-196) -- nn.Linear in fp32 in the reference.  The fp32 mode ran these layers on the fp32 vector
// units / the fp32 MFMA (rows_linear_kernel, rows_wgrad_kernel in mlp.hip): 28 ms of a 67 ms step at 0.5 - 2.7 TB/s; at a third of the
// fp16 matrix rate they are plain HBM streams.
//
// Scales: the row tensors come with their absolute maxima (pcacc_absmax256, one scale per tensor; two-piece rows take the larger of the
// two pieces' maxima); the weight matrix is scaled per output row inside the kernel (a workgroup stages the whole [N][K] matrix anyway).
#include "common.h"

typedef _Float16 ms_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ms_f16x2 __attribute__((ext_vector_type(2)));
typedef float ms_f32x16 __attribute__((ext_vector_type(16)));
typedef short ms_s16x4 __attribute__((ext_vector_type(4)));
union ms_frag { ms_f16x8 v; ms_s16x4 h[2]; uint16_t e[8]; };

#define MS_TILE 128
#define MS_THREADS 256
#define MS_PRE_RELU 1
#define MS_POST_RELU 2

__device__ __forceinline__ float ms_scale_of(float amax)
{
    if (!(amax > 0.f) || !(amax < __builtin_inff())) return 1.f;
    int k;
    frexpf(amax, &k);
    return ldexpf(1.f, 14 - k);
}
// largest of the 256 partial maxima of one tensor (and of a second one when given): wave-uniform, no LDS
__device__ __forceinline__ float ms_amax(const float *__restrict__ parts, const float *__restrict__ parts2)
{
    const int lane = threadIdx.x & 63;
    float m = fmaxf(fmaxf(parts[lane], parts[lane + 64]), fmaxf(parts[lane + 128], parts[lane + 192]));
    if (parts2) m = fmaxf(m, fmaxf(fmaxf(parts2[lane], parts2[lane + 64]), fmaxf(parts2[lane + 128], parts2[lane + 192])));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    return m;
}
__device__ __forceinline__ uint32_t ms_pack(float a, float b)
{
    const pcacc_f32x2 f = {a, b};
    const ms_f16x2 r = __builtin_convertvector(f, ms_f16x2);
    return *reinterpret_cast<const uint32_t *>(&r);
}
#ifdef PCACC_X3_EXPERIMENT
__device__ int ms_xword;                                     // common.h: precision-map experiment build
extern "C" int pcacc_x3_experiment_rows(int word, void *stream)
{
    if (hipStreamSynchronize(pcacc_stream(stream)) != hipSuccess) return PCACC_E_LAUNCH;     // kernels already queued keep the word they were launched under
    return hipMemcpyToSymbol(HIP_SYMBOL(ms_xword), &word, sizeof(int)) == hipSuccess ? PCACC_OK : PCACC_E_LAUNCH;
}
#endif
// WEIGHT: the operand is a weight (the experiment build treats activations and weights separately; no difference in the shipped library)
template <bool WEIGHT = false>
__device__ __forceinline__ void ms_split2(float a, float b, uint32_t &hi, uint32_t &lo)
{
#ifdef PCACC_X3_EXPERIMENT
    const bool drop = pcacc_x_apply(WEIGHT ? PCACC_X_W(ms_xword) : PCACC_X_ACT(ms_xword), a, b);
#endif
    hi = ms_pack(a, b);
    const pcacc_f32x2 back = __builtin_convertvector(*reinterpret_cast<const ms_f16x2 *>(&hi), pcacc_f32x2);
    lo = ms_pack(a - back[0], b - back[1]);
#ifdef PCACC_X3_EXPERIMENT
    if (drop) lo = 0u;
#endif
}
template <bool WEIGHT = false>
__device__ __forceinline__ void ms_split8(const float4 &a, const float4 &b, float s, uint4 &hi, uint4 &lo)
{
    ms_split2<WEIGHT>(a.x * s, a.y * s, hi.x, lo.x);
    ms_split2<WEIGHT>(a.z * s, a.w * s, hi.y, lo.y);
    ms_split2<WEIGHT>(b.x * s, b.y * s, hi.z, lo.z);
    ms_split2<WEIGHT>(b.z * s, b.w * s, hi.w, lo.w);
}
__device__ __forceinline__ float4 ms_relu4(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }
__device__ __forceinline__ float4 ms_mask4(float4 v, float4 m)
{
    return make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
}

// A row made of two pieces ("virtual concatenation"): columns [0,ka) from a[row], columns [ka,K) from b[idx[row]] (idx == NULL: b[row]);
// b == NULL: the plain contiguous [rows,K] layout of `a` (RowPieces of mlp_mfma.hip on fp32 rows).
struct MsPieces {
    const float *b;
    const int32_t *idx;
    int ka;
};
__device__ __forceinline__ const float *ms_piece(const float *a, const MsPieces &s, int K, int64_t row, int col)
{
    if (!s.b) return a + row * K + col;
    if (col < s.ka) return a + row * s.ka + col;
    return s.b + (s.idx ? (int64_t)s.idx[row] : row) * (K - s.ka) + (col - s.ka);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Workgroup = 4 waves, persistent over 128-row tiles; A = 32 output features x 16 k (weights, hi / lo planes resident in LDS), B = 16 k x
// 32 rows (the staged tile, hi / lo planes): D comes out with lane = row; the result passes through LDS as fp32 and leaves in fully
// coalesced 16-byte stores, where residual / ReLU / output mask are applied.
template <int K, int CT>
__global__ __launch_bounds__(MS_THREADS) void rows_linear_split_kernel(const float *__restrict__ X, const float *__restrict__ x_amax,
                                                                       const float *__restrict__ x_amax2, const float *__restrict__ in_mask,
                                                                       const float *__restrict__ W, const float *__restrict__ bias,
                                                                       const float *__restrict__ residual, const float *__restrict__ out_mask,
                                                                       float *__restrict__ Y, int64_t rows, int flags, MsPieces xs2, MsPieces ms2,
                                                                       float *__restrict__ Y2, int na, float *__restrict__ y_amax)
{
    constexpr int N = CT * 32;
    constexpr int XS = K + 8;                                  // padded LDS row (elements) of the 16-bit planes
    constexpr int YS = N + 4;                                  // padded LDS row (floats) of the output tile
    constexpr int XPLANE = MS_TILE * XS, WPLANE = N * XS;
    constexpr int REGION_B = (2 * XPLANE * 2 > MS_TILE * YS * 4) ? 2 * XPLANE * 2 : MS_TILE * YS * 4;   // input planes, later the output tile
    constexpr int X_CHUNKS = MS_TILE * K / 8;                  // 8-element pieces of an input tile
    constexpr int X_PER_THREAD = X_CHUNKS / MS_THREADS;
    constexpr int Y_CHUNKS = MS_TILE * N / 4;                  // 4-element (16-byte) pieces of an output tile
    constexpr int Y_PER_THREAD = Y_CHUNKS / MS_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t *xs = reinterpret_cast<uint16_t *>(lds_raw);                    // [2][MS_TILE][XS]
    float *ys = reinterpret_cast<float *>(lds_raw);                          // [MS_TILE][YS]  (after the MFMAs)
    uint16_t *ws = reinterpret_cast<uint16_t *>(lds_raw + REGION_B);         // [2][N][XS]
    float *bias_l = reinterpret_cast<float *>(ws + 2 * WPLANE);              // [N]
    float *invt = bias_l + N;                                                // [N]  1 / (row scale of W)
    unsigned *wmax = reinterpret_cast<unsigned *>(invt + N);                 // [N]  row maxima of |W| (float bits)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;

    // weights: per-row maximum -> power-of-two row scale -> hi / lo planes
    if (threadIdx.x < N) wmax[threadIdx.x] = 0u;
    __syncthreads();
    for (int e = threadIdx.x; e < N * K; e += MS_THREADS) {
        const float v = W[e];
        atomicMax(&wmax[e / K], (v != v) ? 0x7f800000u : __float_as_uint(fabsf(v)));
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * K / 2; e += MS_THREADS) {
        const int n = (2 * e) / K, k = (2 * e) % K;
        const float t = ms_scale_of(__uint_as_float(wmax[n]));
        const float2 w2 = *reinterpret_cast<const float2 *>(W + (int64_t)n * K + k);
        uint32_t hi, lo;
        ms_split2<true>(w2.x * t, w2.y * t, hi, lo);
        *reinterpret_cast<uint32_t *>(ws + n * XS + k) = hi;
        *reinterpret_cast<uint32_t *>(ws + WPLANE + n * XS + k) = lo;
    }
    if (threadIdx.x < N) {
        bias_l[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
        invt[threadIdx.x] = 1.f / ms_scale_of(__uint_as_float(wmax[threadIdx.x]));
    }
    const float sx = ms_scale_of(ms_amax(x_amax, x_amax2));
    const float inv_sx = 1.f / sx;

    const int64_t n_tiles = (rows + MS_TILE - 1) / MS_TILE;
    float4 xreg[X_PER_THREAD][2];
    int xok = 0;                                                              // bit q: piece q of xreg is a real row of X
    int64_t xbase = 0;                                                        // element offset of the fetched tile
    auto fetch = [&](int64_t tile) {
        const int64_t base = tile * MS_TILE * K;                              // element offset of the tile
        const int64_t limit = rows * K;
        // [r5] every load of the tile is issued before anything reads a loaded value: rows past the end are read from row 0 and zeroed when the tile is staged
        // (xok carries the flags), the optional ReLU / mask are applied there too -- a per-piece `if (e < limit) { load; relu; mask }` had made the fetch a chain
        // of `global_load, s_waitcnt vmcnt(0)` (see conv3x3_split_res_kernel)
        int okb = 0;
        if (!xs2.b) {                                                         // one-piece rows (uniform): straight-line code, the loads go out back to back
#pragma unroll
            for (int q = 0; q < X_PER_THREAD; ++q) {
                const int64_t e = base + (int64_t)(threadIdx.x + q * MS_THREADS) * 8;
                const bool ok = e < limit;
                const float *src = X + (ok ? e : 0);
                xreg[q][0] = *reinterpret_cast<const float4 *>(src);
                xreg[q][1] = *reinterpret_cast<const float4 *>(src + 4);
                okb |= ok ? (1 << q) : 0;
            }
        } else {                                                              // two-piece rows (a row index may sit between the pieces: a dependent load anyway)
#pragma unroll
            for (int q = 0; q < X_PER_THREAD; ++q) {
                const int64_t e = base + (int64_t)(threadIdx.x + q * MS_THREADS) * 8;
                const bool ok = e < limit;
                const int64_t ec = ok ? e : 0;
                const float *src = ms_piece(X, xs2, K, ec / K, (int)(ec % K));
                xreg[q][0] = *reinterpret_cast<const float4 *>(src);
                xreg[q][1] = *reinterpret_cast<const float4 *>(src + 4);
                okb |= ok ? (1 << q) : 0;
            }
        }
        xok = okb;
        xbase = base;
    };
    // the staged form of piece q of the fetched tile: out-of-range rows zero, optional ReLU, optional mask (the order the fetch applied them in)
    auto staged = [&](int q, float4 &a, float4 &b) __attribute__((always_inline)) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool ok = (xok >> q) & 1;
        a = ok ? xreg[q][0] : z;
        b = ok ? xreg[q][1] : z;
        if (flags & MS_PRE_RELU) { a = ms_relu4(a); b = ms_relu4(b); }
        if (in_mask) {                                                        // uniform; the masked layers of the fp32x3 backward
            const int64_t e = xbase + (int64_t)(threadIdx.x + q * MS_THREADS) * 8;
            if (ok) {
                a = ms_mask4(a, *reinterpret_cast<const float4 *>(in_mask + e));
                b = ms_mask4(b, *reinterpret_cast<const float4 *>(in_mask + e + 4));
            }
        }
    };

    float omax = 0.f;                                                         // |output| maximum of this thread (NaN -> inf)
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                                      // the previous tile's output left the region
#pragma unroll
        for (int q = 0; q < X_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * MS_THREADS;
            uint4 hi, lo;
            float4 xa, xb;
            staged(q, xa, xb);
            ms_split8(xa, xb, sx, hi, lo);
            uint16_t *dst = xs + (c / (K / 8)) * XS + (c % (K / 8)) * 8;
            *reinterpret_cast<uint4 *>(dst) = hi;
            *reinterpret_cast<uint4 *>(dst + XPLANE) = lo;
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);              // in flight during the MFMAs and the store phase

        ms_f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const uint16_t *xrow = xs + (wave * 32 + lp) * XS + lh * 8;
        const uint16_t *wrow = ws + lp * XS + lh * 8;
#pragma unroll
        for (int kc = 0; kc < K / 16; ++kc) {
            const ms_f16x8 bh = *reinterpret_cast<const ms_f16x8 *>(xrow + kc * 16);
            const ms_f16x8 bl = *reinterpret_cast<const ms_f16x8 *>(xrow + XPLANE + kc * 16);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const ms_f16x8 ah = *reinterpret_cast<const ms_f16x8 *>(wrow + ct * 32 * XS + kc * 16);
                const ms_f16x8 al = *reinterpret_cast<const ms_f16x8 *>(wrow + WPLANE + ct * 32 * XS + kc * 16);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[ct], 0, 0, 0);
            }
        }
        __syncthreads();                                                      // every wave is done reading the input planes
        float *yrow = ys + (wave * 32 + lp) * YS;                             // lane = row; quads of 4 consecutive features
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = ct * 32 + 8 * g + 4 * lh;
                const float4 bv = *reinterpret_cast<const float4 *>(bias_l + c);
                const float4 sc = *reinterpret_cast<const float4 *>(invt + c);
                *reinterpret_cast<float4 *>(yrow + c) = make_float4(acc[ct][4 * g] * (sc.x * inv_sx) + bv.x, acc[ct][4 * g + 1] * (sc.y * inv_sx) + bv.y,
                                                                    acc[ct][4 * g + 2] * (sc.z * inv_sx) + bv.z, acc[ct][4 * g + 3] * (sc.w * inv_sx) + bv.w);
            }
        __syncthreads();
        const int64_t ybase = tile * MS_TILE * N, ylimit = rows * N;
#pragma unroll
        for (int q = 0; q < Y_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * MS_THREADS;
            const int64_t e = ybase + (int64_t)c * 4;
            if (e >= ylimit) continue;
            float4 v = *reinterpret_cast<const float4 *>(ys + (c / (N / 4)) * YS + (c % (N / 4)) * 4);
            if (residual) {
                const float4 r = *reinterpret_cast<const float4 *>(residual + e);
                v = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
            }
            if (flags & MS_POST_RELU) v = ms_relu4(v);
            const int64_t row = e / N;
            const int col = (int)(e % N);
            if (out_mask) v = ms_mask4(v, *reinterpret_cast<const float4 *>(ms_piece(out_mask, ms2, N, row, col)));
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) omax = __builtin_inff();
            if (!Y2 || na < 0) {                                              // na < 0: Y2 is the bf16 SHADOW of Y ('mixed' mode), same offsets
                *reinterpret_cast<float4 *>(Y + e) = v;
                if (Y2) *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(Y2) + e) = make_uint2(pcacc_pack_bf16x2(v.x, v.y), pcacc_pack_bf16x2(v.z, v.w));
            } else if (col < na) *reinterpret_cast<float4 *>(Y + row * na + col) = v;
            else *reinterpret_cast<float4 *>(Y2 + row * (N - na) + (col - na)) = v;
        }
    }
    if (y_amax) {                                                             // uniform: the result's maximum for its consumer's scale --
#pragma unroll                                                                // one atomic per wave into one of 256 slots (zeroed by the caller)
        for (int d = 32; d >= 1; d >>= 1) omax = fmaxf(omax, __shfl_xor(omax, d, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(y_amax) + (blockIdx.x & 255), __float_as_uint(omax));
    }
}

// The wide layers (K x N = 128 x 128, 128 x 64, 64 x 128): in the kernel above their weight planes alone take 35 - 70 KB of LDS next to the
// 70 KB of staged rows -- one workgroup per CU, whose load / split / MFMA / store phases then run in series (2.0 TB/s on the 128 x 128
// layers of the STPN heads).  Here a WAVE owns a 32-feature tile (and, for N = 64, half of the rows): its weight fragments -- K/16 x (hi, lo)
// -- are split once from global memory into registers and stay there, the LDS holds only the rows, and two workgroups share a CU out of phase.
template <int K, int CT>
__global__ __launch_bounds__(MS_THREADS, 2) void rows_linear_split_fm_kernel(const float *__restrict__ X, const float *__restrict__ x_amax,
                                                                             const float *__restrict__ x_amax2, const float *__restrict__ in_mask,
                                                                             const float *__restrict__ W, const float *__restrict__ bias,
                                                                             const float *__restrict__ residual, const float *__restrict__ out_mask,
                                                                             float *__restrict__ Y, int64_t rows, int flags, MsPieces xs2, MsPieces ms2,
                                                                             float *__restrict__ Y2, int na, float *__restrict__ y_amax)
{
    constexpr int N = CT * 32, KC = K / 16;
    constexpr int RG = 4 / CT, RT = 4 / RG;                    // row groups among the 4 waves, 32-row tiles per wave
    constexpr int XS = K + 8, YS = N + 4;
    constexpr int XPLANE = MS_TILE * XS;
    constexpr int REGION_B = (2 * XPLANE * 2 > MS_TILE * YS * 4) ? 2 * XPLANE * 2 : MS_TILE * YS * 4;
    constexpr int X_PER_THREAD = MS_TILE * K / 8 / MS_THREADS, Y_PER_THREAD = MS_TILE * N / 4 / MS_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t *xs = reinterpret_cast<uint16_t *>(lds_raw);                    // [2][MS_TILE][XS]
    float *ys = reinterpret_cast<float *>(lds_raw);                          // [MS_TILE][YS]  (after the MFMAs)
    float *bias_l = reinterpret_cast<float *>(lds_raw + REGION_B);           // [N]
    float *invt = bias_l + N;                                                // [N]  1 / (row scale of W)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int ct = wave % CT, rg = wave / CT;

    // this wave's weight rows n = ct * 32 + lp: row maximum -> power-of-two scale -> hi / lo fragments in registers
    ms_f16x8 wh[KC], wl[KC];
    {
        const float *wrow = W + (int64_t)(ct * 32 + lp) * K + lh * 8;
        float m = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {                      // two passes over the (cached) row instead of 2 KC live float4s
            const float4 a = *reinterpret_cast<const float4 *>(wrow + kc * 16), b = *reinterpret_cast<const float4 *>(wrow + kc * 16 + 4);
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                               fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            if (!(a.x == a.x && a.y == a.y && a.z == a.z && a.w == a.w && b.x == b.x && b.y == b.y && b.z == b.z && b.w == b.w)) m = __builtin_inff();
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));                   // the row's other k-half
        const float t = ms_scale_of(m);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            uint4 hi, lo;
            ms_split8<true>(*reinterpret_cast<const float4 *>(wrow + kc * 16), *reinterpret_cast<const float4 *>(wrow + kc * 16 + 4), t, hi, lo);
            wh[kc] = *reinterpret_cast<const ms_f16x8 *>(&hi);
            wl[kc] = *reinterpret_cast<const ms_f16x8 *>(&lo);
        }
        if (rg == 0 && lh == 0) {
            invt[ct * 32 + lp] = 1.f / t;
            bias_l[ct * 32 + lp] = bias ? bias[ct * 32 + lp] : 0.f;
        }
    }
    const float sx = ms_scale_of(ms_amax(x_amax, x_amax2));
    const float inv_sx = 1.f / sx;

    const int64_t n_tiles = (rows + MS_TILE - 1) / MS_TILE;
    float4 xreg[X_PER_THREAD][2];
    int xok = 0;                                                              // bit q: piece q of xreg is a real row of X
    int64_t xbase = 0;                                                        // element offset of the fetched tile
    auto fetch = [&](int64_t tile) {
        const int64_t base = tile * MS_TILE * K;
        const int64_t limit = rows * K;
        // [r5] every load of the tile is issued before anything reads a loaded value: rows past the end are read from row 0 and zeroed when the tile is staged
        // (xok carries the flags), the optional ReLU / mask are applied there too -- a per-piece `if (e < limit) { load; relu; mask }` had made the fetch a chain
        // of `global_load, s_waitcnt vmcnt(0)` (see conv3x3_split_res_kernel)
        int okb = 0;
        if (!xs2.b) {                                                         // one-piece rows (uniform): straight-line code, the loads go out back to back
#pragma unroll
            for (int q = 0; q < X_PER_THREAD; ++q) {
                const int64_t e = base + (int64_t)(threadIdx.x + q * MS_THREADS) * 8;
                const bool ok = e < limit;
                const float *src = X + (ok ? e : 0);
                xreg[q][0] = *reinterpret_cast<const float4 *>(src);
                xreg[q][1] = *reinterpret_cast<const float4 *>(src + 4);
                okb |= ok ? (1 << q) : 0;
            }
        } else {                                                              // two-piece rows (a row index may sit between the pieces: a dependent load anyway)
#pragma unroll
            for (int q = 0; q < X_PER_THREAD; ++q) {
                const int64_t e = base + (int64_t)(threadIdx.x + q * MS_THREADS) * 8;
                const bool ok = e < limit;
                const int64_t ec = ok ? e : 0;
                const float *src = ms_piece(X, xs2, K, ec / K, (int)(ec % K));
                xreg[q][0] = *reinterpret_cast<const float4 *>(src);
                xreg[q][1] = *reinterpret_cast<const float4 *>(src + 4);
                okb |= ok ? (1 << q) : 0;
            }
        }
        xok = okb;
        xbase = base;
    };
    // the staged form of piece q of the fetched tile: out-of-range rows zero, optional ReLU, optional mask (the order the fetch applied them in)
    auto staged = [&](int q, float4 &a, float4 &b) __attribute__((always_inline)) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool ok = (xok >> q) & 1;
        a = ok ? xreg[q][0] : z;
        b = ok ? xreg[q][1] : z;
        if (flags & MS_PRE_RELU) { a = ms_relu4(a); b = ms_relu4(b); }
        if (in_mask) {                                                        // uniform; the masked layers of the fp32x3 backward
            const int64_t e = xbase + (int64_t)(threadIdx.x + q * MS_THREADS) * 8;
            if (ok) {
                a = ms_mask4(a, *reinterpret_cast<const float4 *>(in_mask + e));
                b = ms_mask4(b, *reinterpret_cast<const float4 *>(in_mask + e + 4));
            }
        }
    };

    float omax = 0.f;
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                                      // the previous tile's output left the region (first pass: invt / bias written)
#pragma unroll
        for (int q = 0; q < X_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * MS_THREADS;
            uint4 hi, lo;
            float4 xa, xb;
            staged(q, xa, xb);
            ms_split8(xa, xb, sx, hi, lo);
            uint16_t *dst = xs + (c / (K / 8)) * XS + (c % (K / 8)) * 8;
            *reinterpret_cast<uint4 *>(dst) = hi;
            *reinterpret_cast<uint4 *>(dst + XPLANE) = lo;
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);              // in flight during the MFMAs and the store phase

        ms_f32x16 acc[RT];
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const uint16_t *xrow = xs + (rg * RT * 32 + lp) * XS + lh * 8;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                const ms_f16x8 bh = *reinterpret_cast<const ms_f16x8 *>(xrow + j * 32 * XS + kc * 16);
                const ms_f16x8 bl = *reinterpret_cast<const ms_f16x8 *>(xrow + XPLANE + j * 32 * XS + kc * 16);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kc], bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[kc], bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kc], bh, acc[j], 0, 0, 0);
            }
        __syncthreads();                                                      // every wave is done reading the input planes
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            float *yrow = ys + ((rg * RT + j) * 32 + lp) * YS + ct * 32;      // lane = row; quads of 4 consecutive features
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 8 * g + 4 * lh;
                const float4 bv = *reinterpret_cast<const float4 *>(bias_l + ct * 32 + c);
                const float4 sc = *reinterpret_cast<const float4 *>(invt + ct * 32 + c);
                *reinterpret_cast<float4 *>(yrow + c) = make_float4(acc[j][4 * g] * (sc.x * inv_sx) + bv.x, acc[j][4 * g + 1] * (sc.y * inv_sx) + bv.y,
                                                                    acc[j][4 * g + 2] * (sc.z * inv_sx) + bv.z, acc[j][4 * g + 3] * (sc.w * inv_sx) + bv.w);
            }
        }
        __syncthreads();
        const int64_t ybase = tile * MS_TILE * N, ylimit = rows * N;
#pragma unroll
        for (int q = 0; q < Y_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * MS_THREADS;
            const int64_t e = ybase + (int64_t)c * 4;
            if (e >= ylimit) continue;
            float4 v = *reinterpret_cast<const float4 *>(ys + (c / (N / 4)) * YS + (c % (N / 4)) * 4);
            if (residual) {
                const float4 r = *reinterpret_cast<const float4 *>(residual + e);
                v = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
            }
            if (flags & MS_POST_RELU) v = ms_relu4(v);
            const int64_t row = e / N;
            const int col = (int)(e % N);
            if (out_mask) v = ms_mask4(v, *reinterpret_cast<const float4 *>(ms_piece(out_mask, ms2, N, row, col)));
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) omax = __builtin_inff();
            if (!Y2 || na < 0) {                                              // na < 0: Y2 is the bf16 SHADOW of Y ('mixed' mode), same offsets
                *reinterpret_cast<float4 *>(Y + e) = v;
                if (Y2) *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(Y2) + e) = make_uint2(pcacc_pack_bf16x2(v.x, v.y), pcacc_pack_bf16x2(v.z, v.w));
            } else if (col < na) *reinterpret_cast<float4 *>(Y + row * na + col) = v;
            else *reinterpret_cast<float4 *>(Y2 + row * (N - na) + (col - na)) = v;
        }
    }
    if (y_amax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) omax = fmaxf(omax, __shfl_xor(omax, d, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(y_amax) + (blockIdx.x & 255), __float_as_uint(omax));
    }
}

template <int K, int CT>
static int ms_launch_fm(const float *x, const float *x_amax, const float *x_amax2, const float *in_mask, const float *w, const float *bias,
                        const float *residual, const float *out_mask, float *y, int64_t rows, int flags, hipStream_t st, MsPieces xs2, MsPieces ms2,
                        float *y2, int na, float *y_amax)
{
    constexpr int N = CT * 32, XS = K + 8, YS = N + 4;
    constexpr size_t region = (size_t)(2 * MS_TILE * XS * 2 > MS_TILE * YS * 4 ? 2 * MS_TILE * XS * 2 : MS_TILE * YS * 4);
    const size_t lds = region + (size_t)2 * N * 4;
    auto kern = rows_linear_split_fm_kernel<K, CT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    const int64_t n_tiles = (rows + MS_TILE - 1) / MS_TILE;
    int64_t grid = (int64_t)PCACC_CUS * 2;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(MS_THREADS), lds, st, x, x_amax, x_amax2, in_mask, w, bias, residual, out_mask, y, rows, flags,
                       xs2, ms2, y2, na, y_amax);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

template <int K, int CT>
static size_t ms_linear_lds()
{
    constexpr int N = CT * 32, XS = K + 8, YS = N + 4;
    constexpr size_t region = (size_t)(2 * MS_TILE * XS * 2 > MS_TILE * YS * 4 ? 2 * MS_TILE * XS * 2 : MS_TILE * YS * 4);
    return region + (size_t)2 * N * XS * 2 + 3 * N * 4;
}

template <int K, int CT>
static int ms_launch(const float *x, const float *x_amax, const float *x_amax2, const float *in_mask, const float *w, const float *bias,
                     const float *residual, const float *out_mask, float *y, int64_t rows, int flags, hipStream_t st, MsPieces xs2, MsPieces ms2,
                     float *y2, int na, float *y_amax)
{
    const size_t lds = ms_linear_lds<K, CT>();
    auto kern = rows_linear_split_kernel<K, CT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    const int64_t n_tiles = (rows + MS_TILE - 1) / MS_TILE;
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 3 ? 3 : (per_cu < 1 ? 1 : per_cu);
    int64_t grid = (int64_t)PCACC_CUS * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(MS_THREADS), lds, st, x, x_amax, x_amax2, in_mask, w, bias, residual, out_mask, y, rows, flags,
                       xs2, ms2, y2, na, y_amax);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

static int ms_dispatch(const float *x, const float *x_amax, const float *x_amax2, const float *in_mask, const float *w, const float *bias,
                       const float *residual, const float *out_mask, float *y, int64_t rows, int k, int n, int flags, hipStream_t st,
                       MsPieces xs2, MsPieces ms2, float *y2, int na, float *y_amax)
{
#define MS_CASE(KK, CTV) \
    if (k == KK && n == CTV * 32) return ms_launch<KK, CTV>(x, x_amax, x_amax2, in_mask, w, bias, residual, out_mask, y, rows, flags, st, xs2, ms2, y2, na, y_amax)
#define MS_CASE_FM(KK, CTV) \
    if (k == KK && n == CTV * 32 && !pcacc_switches().rows_fm_off) return ms_launch_fm<KK, CTV>(x, x_amax, x_amax2, in_mask, w, bias, residual, out_mask, y, rows, flags, st, xs2, ms2, y2, na, y_amax)
    MS_CASE_FM(128, 4); MS_CASE_FM(128, 2); MS_CASE_FM(64, 4);              // weights in registers, two workgroups per CU
#undef MS_CASE_FM
    MS_CASE(32, 1); MS_CASE(32, 2); MS_CASE(32, 4);
    MS_CASE(64, 1); MS_CASE(64, 2); MS_CASE(64, 4);
    MS_CASE(128, 1); MS_CASE(128, 2); MS_CASE(128, 4);
#undef MS_CASE
    return PCACC_E_ARG;
}

extern "C" int pcacc_rows_linear_split(const float *x, const float *x_amax, const float *in_mask, const float *w, const float *bias,
                                       const float *residual, const float *out_mask, float *y, float *y_amax, int64_t rows, int32_t k, int32_t n,
                                       int32_t flags, void *stream)
{
    if (rows < 0 || (k != 32 && k != 64 && k != 128) || (n != 32 && n != 64 && n != 128)) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!x || !x_amax || !w || !y) return PCACC_E_ARG;
    const MsPieces none{nullptr, nullptr, 0};
    return ms_dispatch(x, x_amax, nullptr, in_mask, w, bias, residual, out_mask, y, rows, k, n, flags, pcacc_stream(stream), none, none, nullptr, 0, y_amax);
}

// 'mixed' compute mode: the same layer with y16 = bf16(y) [rows,n] as a second output of the same epilogue (the shadow the bf16 backward reads)
extern "C" int pcacc_rows_linear_split_dual(const float *x, const float *x_amax, const float *in_mask, const float *w, const float *bias,
                                            const float *residual, const float *out_mask, float *y, uint16_t *y16, float *y_amax, int64_t rows,
                                            int32_t k, int32_t n, int32_t flags, void *stream)
{
    if (rows < 0 || (k != 32 && k != 64 && k != 128) || (n != 32 && n != 64 && n != 128)) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!x || !x_amax || !w || !y || !y16) return PCACC_E_ARG;
    const MsPieces none{nullptr, nullptr, 0};
    return ms_dispatch(x, x_amax, nullptr, in_mask, w, bias, residual, out_mask, y, rows, k, n, flags, pcacc_stream(stream), none, none,
                       reinterpret_cast<float *>(y16), -1, y_amax);
}

// The same layer on rows made of two pieces (see MsPieces; the fp32 twin of pcacc_rows_linear_cat_bf16).  Forward: x = cat(xa [rows,ka],
// xb[b_index] [.,k-ka]).  Backward-data (w = W^T, x = the output gradient, xb = NULL): the [rows,n] result leaves as y [rows,na] and
// y2 [rows,n-na], masked where the forward input cat(out_mask_a, out_mask_b[b_index]) was <= 0.
extern "C" int pcacc_rows_linear_cat_split(const float *xa, const float *xa_amax, const float *xb, const float *xb_amax, const int32_t *b_index,
                                           int32_t ka, const float *in_mask, const float *w, const float *bias, const float *residual,
                                           const float *out_mask_a, const float *out_mask_b, float *y, float *y2, int32_t na, float *y_amax,
                                           int64_t rows, int32_t k, int32_t n, int32_t flags, void *stream)
{
    if (rows < 0 || (k != 32 && k != 64 && k != 128) || (n != 32 && n != 64 && n != 128)) return PCACC_E_ARG;
    if (xb && (ka <= 0 || ka >= k || ka % 8 || !xb_amax)) return PCACC_E_ARG;
    if (y2 && (na <= 0 || na >= n || na % 8)) return PCACC_E_ARG;
    if (out_mask_b && (!out_mask_a || !y2)) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!xa || !xa_amax || !w || !y) return PCACC_E_ARG;
    const MsPieces xs2{xb, b_index, ka}, ms2{out_mask_b, b_index, na};
    return ms_dispatch(xa, xa_amax, xb ? xb_amax : nullptr, in_mask, w, bias, residual, out_mask_a, y, rows, k, n, flags, pcacc_stream(stream), xs2,
                       ms2, y2, na, y_amax);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Weight / bias gradient from fp32 rows:  dW_aug[n][k] (k in [0,K]; column K = bias gradient) = sum_r dYeff[r][n] * Xaug[r][k], Xaug[r][K] = 1.
// The scheme of rows_wgrad_bf16_kernel (mlp_mfma.hip): the reduction runs over rows, tiles are staged row-major (masks, ReLU, scale and
// the hi / lo split applied on the way) and the fragments -- 8 consecutive ROWS of one column per lane -- come through the hardware
// transpose read (ds_read_b64_tr_b16).  Partials per workgroup go to a workspace; the reduce launch sums them and removes the scales.
template <int MAX_TILES, int WG_R, int NW = 4>
__global__ __launch_bounds__(NW * 64) void rows_wgrad_split_kernel(const float *__restrict__ dY, const float *__restrict__ dy_amax,
                                                                   const float *__restrict__ dy_mask, const float *__restrict__ X,
                                                                   const float *__restrict__ x_amax, const float *__restrict__ x_amax2, int x_relu,
                                                                   int64_t rows, int K, int N, int k_tiles, int n_tile_total, int tiles_par,
                                                                   float *partial, MsPieces xs2, float *__restrict__ dw_zero)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t wlds[];
    if (blockIdx.x == 0)                                                       // the reduce launch adds into dW: cleared here, not by a memset
        for (int e = threadIdx.x; e < N * (K + 1); e += NW * 64) dw_zero[e] = 0.f;
    const int NS = pcacc_tr_stride(N), KS = pcacc_tr_stride(K);
    const int yplane = WG_R * NS, xplane = WG_R * KS;
    uint16_t *sdy = wlds, *sx = wlds + 2 * yplane;             // [2][WG_R][NS], [2][WG_R][KS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const float sy = ms_scale_of(ms_amax(dy_amax, nullptr)), sxs = ms_scale_of(ms_amax(x_amax, x_amax2));
    ms_f32x16 acc[MAX_TILES];
#pragma unroll
    for (int t = 0; t < MAX_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int64_t n_chunks = (rows + WG_R - 1) / WG_R;
    const int ny = WG_R * N / 8, nx = WG_R * K / 8;                              // 8-element pieces per tile
    constexpr int PIECES = 16 / NW;
    float4 yreg[PIECES][2], xreg[PIECES][2];
    const int kshift = __ffs(K) - 1;                                           // K is a power of two
    int prow[PIECES] = {};
    auto fetch_rows = [&](int64_t ch) {                                        // pillar rows of the gathered half, one tile ahead of their use
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
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int i = threadIdx.x + q * NW * 64;
            float4 a = z, b = z;
            if (i < ny && (int64_t)i * 8 < lim_n) {
                const float *src = dY + row0 * N + (int64_t)i * 8;
                a = *reinterpret_cast<const float4 *>(src);
                b = *reinterpret_cast<const float4 *>(src + 4);
                if (dy_mask) {
                    const float *m = dy_mask + row0 * N + (int64_t)i * 8;
                    a = ms_mask4(a, *reinterpret_cast<const float4 *>(m));
                    b = ms_mask4(b, *reinterpret_cast<const float4 *>(m + 4));
                }
            }
            yreg[q][0] = a;
            yreg[q][1] = b;
            float4 c = z, d = z;
            if (i < nx && (int64_t)i * 8 < lim_k) {
                const int col = (i * 8) & (K - 1);
                const float *src = X + row0 * K + (int64_t)i * 8;
                if (xs2.b)
                    src = col < xs2.ka ? X + (row0 + ((i * 8) >> kshift)) * xs2.ka + col
                                       : xs2.b + (int64_t)prow[q] * (K - xs2.ka) + (col - xs2.ka);
                c = *reinterpret_cast<const float4 *>(src);
                d = *reinterpret_cast<const float4 *>(src + 4);
                if (x_relu) { c = ms_relu4(c); d = ms_relu4(d); }
            }
            xreg[q][0] = c;
            xreg[q][1] = d;
        }
    };
    int64_t ch = blockIdx.x;
    fetch_rows(ch);
    if (ch < n_chunks) fetch(ch);
    fetch_rows(ch + gridDim.x);
    for (; ch < n_chunks; ch += gridDim.x) {
        const int64_t row0 = ch * WG_R;
        __syncthreads();                                                         // the previous tile's fragment reads are done
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int i = threadIdx.x + q * NW * 64;
            if (i < ny) {
                const int e = i * 8;
                uint4 hi, lo;
                ms_split8(yreg[q][0], yreg[q][1], sy, hi, lo);
                uint2 *dst = reinterpret_cast<uint2 *>(sdy + (e / N) * NS + e % N);
                dst[0] = make_uint2(hi.x, hi.y);
                dst[1] = make_uint2(hi.z, hi.w);
                uint2 *dl = reinterpret_cast<uint2 *>(sdy + yplane + (e / N) * NS + e % N);
                dl[0] = make_uint2(lo.x, lo.y);
                dl[1] = make_uint2(lo.z, lo.w);
            }
            if (i < nx) {
                const int e = i * 8;
                uint4 hi, lo;
                ms_split8(xreg[q][0], xreg[q][1], sxs, hi, lo);
                uint2 *dst = reinterpret_cast<uint2 *>(sx + (e / K) * KS + e % K);
                dst[0] = make_uint2(hi.x, hi.y);
                dst[1] = make_uint2(hi.z, hi.w);
                uint2 *dl = reinterpret_cast<uint2 *>(sx + xplane + (e / K) * KS + e % K);
                dl[0] = make_uint2(lo.x, lo.y);
                dl[1] = make_uint2(lo.z, lo.w);
            }
        }
        __syncthreads();
        if (ch + gridDim.x < n_chunks) fetch(ch + gridDim.x);
        fetch_rows(ch + 2 * (int64_t)gridDim.x);
        const int nrow = (int)min((int64_t)WG_R, rows - row0);
        const int g = lane >> 4, li = lane & 15;
        const int tr_row = (g >> 1) * 8 + (li >> 2), tr_col = (g & 1) * 16 + (li & 3) * 4;
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
                    ms_frag ah, al, bh, bl;
                    ah.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms_s16x4 __attribute__((address_space(3))) *)(pa + r0 * NS));
                    ah.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms_s16x4 __attribute__((address_space(3))) *)(pa + (r0 + 4) * NS));
                    al.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms_s16x4 __attribute__((address_space(3))) *)(pa + yplane + r0 * NS));
                    al.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms_s16x4 __attribute__((address_space(3))) *)(pa + yplane + (r0 + 4) * NS));
                    if (!ones) {
                        bh.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms_s16x4 __attribute__((address_space(3))) *)(pb + r0 * KS));
                        bh.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms_s16x4 __attribute__((address_space(3))) *)(pb + (r0 + 4) * KS));
                        bl.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms_s16x4 __attribute__((address_space(3))) *)(pb + xplane + r0 * KS));
                        bl.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms_s16x4 __attribute__((address_space(3))) *)(pb + xplane + (r0 + 4) * KS));
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bl.v, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh.v, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh.v, acc[t], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            bh.e[j] = (kt * 32 + lp == K && r0 + 8 * lh + j < nrow) ? (uint16_t)0x3c00 : (uint16_t)0;     // fp16 1.0
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh.v, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh.v, acc[t], 0, 0, 0);
                    }
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

// out[e] = (sum over the workgroup partials in a fixed order: common.h pcacc_reduce_partials -- run-to-run identical) / the operand scales.
// split_k > 0: the [n][split_k + 1] result is written as dW [n][split_k] followed by the bias gradients [n], both contiguous.
template <int EL>
__global__ __launch_bounds__(1024) void rows_wgrad_split_reduce_kernel(const float *__restrict__ partial, int n_parts, int elems, int ka,
                                                                       const float *__restrict__ dy_amax, const float *__restrict__ x_amax,
                                                                       const float *__restrict__ x_amax2, float *__restrict__ out, int split_k)
{
    const float inv_y = 1.f / ms_scale_of(ms_amax(dy_amax, nullptr)), inv_yx = inv_y / ms_scale_of(ms_amax(x_amax, x_amax2));
    pcacc_reduce_partials<EL>(partial, n_parts, elems, [&](int e, float v) {
        const int row = e / ka, col = e % ka;
        int o = e;
        if (split_k > 0) o = col < split_k ? row * split_k + col : (elems / ka) * split_k + row;
        out[o] = v * (col == ka - 1 ? inv_y : inv_yx);
    });
}

static int wsplit_tile_rows(int k, int n) { return (k <= 64 && n <= 64) ? 128 : 64; }
static int wsplit_waves(int total) { return total > 12 ? 8 : 4; }
static int wsplit_tiles_par(int total) { return total <= 1 ? 1 : (total <= 2 ? 2 : (total > 12 ? 8 : 4)); }
static int wsplit_grid(int64_t rows, int tile_rows, int k, int n)
{
    const int64_t n_chunks = (rows + tile_rows - 1) / tile_rows;
    const size_t lds = (size_t)2 * tile_rows * (pcacc_tr_stride(n) + pcacc_tr_stride(k)) * sizeof(uint16_t);
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 3 ? 3 : (per_cu < 1 ? 1 : per_cu);
    int64_t grid = (int64_t)PCACC_CUS * per_cu;
    return (int)(grid > n_chunks ? n_chunks : grid);
}

extern "C" int pcacc_rows_wgrad_split_workspace_bytes(int64_t rows, int32_t k, int32_t n, size_t *bytes)
{
    if (!bytes || rows < 0 || k <= 0 || n <= 0 || k > 128 || n > 128) return PCACC_E_ARG;
    const int total = ((k + 1 + 31) / 32) * ((n + 31) / 32);
    *bytes = (size_t)(rows > 0 ? wsplit_grid(rows, wsplit_tile_rows(k, n), k, n) : 0) * (wsplit_waves(total) / wsplit_tiles_par(total)) * n * (k + 1) *
             sizeof(float);
    return PCACC_OK;
}

static int rows_wgrad_split_any(const float *dy, const float *dy_amax, const float *dy_mask, const float *x, const float *x_amax,
                                const float *x_amax2, MsPieces xs2, int32_t x_relu, int64_t rows, int32_t k, int32_t n, float *dw_aug,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    if (rows < 0 || k <= 0 || n <= 0 || k > 128 || n > 128 || (k % 32) || (n % 32) || (k & (k - 1)) || !dw_aug) return PCACC_E_ARG;
    const int split_k = (x_relu & 2) ? k : 0;                                 // flags: bit 0 = ReLU on X, bit 1 = split result layout
    x_relu &= 1;
    hipStream_t st = pcacc_stream(stream);
    if (rows == 0) {
        if (hipMemsetAsync(dw_aug, 0, (size_t)n * (k + 1) * sizeof(float), st) != hipSuccess) return PCACC_E_LAUNCH;
        return PCACC_OK;
    }
    if (!dy || !dy_amax || !x || !x_amax || !workspace) return PCACC_E_ARG;
    const int k_tiles = (k + 1 + 31) / 32, n_tiles = (n + 31) / 32;
    const int total = k_tiles * n_tiles;
    if (total > 24) return PCACC_E_ARG;
    const int tile_rows = wsplit_tile_rows(k, n), tiles_par = wsplit_tiles_par(total), parts_per_wg = wsplit_waves(total) / tiles_par;
    const int grid = wsplit_grid(rows, tile_rows, k, n);
    const int elems = n * (k + 1);
    if (workspace_bytes < (size_t)grid * parts_per_wg * elems * sizeof(float)) return PCACC_E_WORKSPACE;
    float *partial = reinterpret_cast<float *>(workspace);
    const size_t lds = (size_t)2 * tile_rows * (pcacc_tr_stride(n) + pcacc_tr_stride(k)) * sizeof(uint16_t);
#define WSP(T, R, NWV)                                                                                                                  \
    do {                                                                                                                                \
        auto kern = rows_wgrad_split_kernel<T, R, NWV>;                                                                                 \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                                   (int)lds) != hipSuccess)                                                             \
            return PCACC_E_LAUNCH;                                                                                                      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, st, dy, dy_amax, dy_mask, x, x_amax, x_amax2, x_relu, rows, k, n, k_tiles, total, \
                           tiles_par, partial, xs2, dw_aug);                                                                            \
    } while (0)
    if (total <= 4) { if (tile_rows == 128) WSP(1, 128, 4); else WSP(1, 64, 4); }
    else if (total <= 8) { if (tile_rows == 128) WSP(2, 128, 4); else WSP(2, 64, 4); }
    else if (total <= 12) WSP(3, 64, 4);
    else WSP(3, 64, 8);
#undef WSP
    if (pcacc_reduce_el(elems) == 64)
        hipLaunchKernelGGL(rows_wgrad_split_reduce_kernel<64>, dim3((elems + 63) / 64), dim3(1024), 0, st, partial, grid * parts_per_wg, elems, k + 1,
                           dy_amax, x_amax, x_amax2, dw_aug, split_k);
    else
        hipLaunchKernelGGL(rows_wgrad_split_reduce_kernel<16>, dim3((elems + 15) / 16), dim3(1024), 0, st, partial, grid * parts_per_wg, elems, k + 1,
                           dy_amax, x_amax, x_amax2, dw_aug, split_k);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_rows_wgrad_split(const float *dy, const float *dy_amax, const float *dy_mask, const float *x, const float *x_amax,
                                      int32_t x_relu, int64_t rows, int32_t k, int32_t n, float *dw_aug, void *workspace, size_t workspace_bytes,
                                      void *stream)
{
    return rows_wgrad_split_any(dy, dy_amax, dy_mask, x, x_amax, nullptr, MsPieces{nullptr, nullptr, 0}, x_relu, rows, k, n, dw_aug, workspace,
                                workspace_bytes, stream);
}

// x = cat(xa [rows,ka], xb[b_index] [.,k-ka]) (see MsPieces); workspace as pcacc_rows_wgrad_split_workspace_bytes(rows, k, n)
extern "C" int pcacc_rows_wgrad_cat_split(const float *dy, const float *dy_amax, const float *dy_mask, const float *xa, const float *xa_amax,
                                          const float *xb, const float *xb_amax, const int32_t *b_index, int32_t ka, int32_t x_relu, int64_t rows,
                                          int32_t k, int32_t n, float *dw_aug, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!xb || !xb_amax || ka <= 0 || ka >= k || ka % 8 || (k & (k - 1))) return PCACC_E_ARG;
    return rows_wgrad_split_any(dy, dy_amax, dy_mask, xa, xa_amax, xb_amax, MsPieces{xb, b_index, ka}, x_relu, rows, k, n, dw_aug, workspace,
                                workspace_bytes, stream);
}
