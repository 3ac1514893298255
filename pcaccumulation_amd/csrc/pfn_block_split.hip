// A ResnetBlockFC of the pillar encoder (models/pillar_encoder.py:13-55, sizes 64 -> 32 -> 32 with a linear shortcut) in the fp32x3
// compute mode: fp32 point rows, every product on the 16-bit matrix cores from scaled fp16 hi / lo halves (conv_split.hip has the
// arithmetic).  The fp32 twin of pfn_block.hip's forward, and of the data half of its backward:
//     h   = relu(x) W0^T + b0            [rows, 32]
//     out = relu(h) W1^T + b1 + x Ws^T   [rows, 32]
// x = [rows, 64] contiguous or the virtual concatenation cat(xa[row], pooled[p2v[row]]) (models/pillar_encoder.py:116-118).
// As separate row-linear launches the block moved 3.3 GB forward per 3.2 M rows (fc_0 and the shortcut both read x, fc_1 reads both
// results) and more backward (three data gradients, the add of the two d(x) contributions); here:
//   forward  reads x once (256 B / row), writes out and relu(h) (128 B each, relu(h) is the weight gradient's operand) and two sign
//            masks (x > 0: 64 bits, h > 0: 32 bits per row) -- all the data-gradient kernel needs of x and h;
//   dgrad    reads d(out) (128 B) and the masks (12 B), writes d(x) -- already split into its two pieces -- and d(h) (the operand of fc_0's
//            weight gradient): 524 B / row.  The three weight gradients run on the row weight-gradient kernels of mlp_split.hip.
// Scales: x and d(out) come with their absolute maxima (one scale per tensor); the in-kernel intermediates relu(h) / d(h) are scaled per
// wave (a wave's 32 rows are the only consumers of its intermediate: the exact maximum is at hand in the accumulators); each weight
// matrix gets one scale when a workgroup stages it.
#include "common.h"

typedef _Float16 pbs_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pbs_f16x2 __attribute__((ext_vector_type(2)));
typedef float pbs_f32x16 __attribute__((ext_vector_type(16)));
union pbs_frag { pbs_f16x8 v; uint4 q; uint32_t u[4]; };

#define PBS_TILE 128
#define PBS_THREADS 256

struct PbsPieces {                // second half of x: b[idx[row]] (32 columns); b == NULL: x is the contiguous [rows, 64] array `xa`
    const float *b;
    const int32_t *idx;
};

__device__ __forceinline__ float pbs_scale_of(float amax)
{
    if (!(amax > 0.f) || !(amax < __builtin_inff())) return 1.f;
    int k;
    frexpf(amax, &k);
    return ldexpf(1.f, 14 - k);
}
__device__ __forceinline__ float pbs_amax(const float *__restrict__ parts, const float *__restrict__ parts2)
{
    const int lane = threadIdx.x & 63;
    float m = fmaxf(fmaxf(parts[lane], parts[lane + 64]), fmaxf(parts[lane + 128], parts[lane + 192]));
    if (parts2) m = fmaxf(m, fmaxf(fmaxf(parts2[lane], parts2[lane + 64]), fmaxf(parts2[lane + 128], parts2[lane + 192])));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    return m;
}
__device__ __forceinline__ float pbs_wave_max(float m)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    return m;
}
__device__ __forceinline__ uint32_t pbs_pack(float a, float b)
{
    const pcacc_f32x2 f = {a, b};
    const pbs_f16x2 r = __builtin_convertvector(f, pbs_f16x2);
    return *reinterpret_cast<const uint32_t *>(&r);
}
#ifdef PCACC_X3_EXPERIMENT
__device__ int pbs_xword;                                    // common.h: precision-map experiment build
extern "C" int pcacc_x3_experiment_pfn(int word, void *stream)
{
    if (hipStreamSynchronize(pcacc_stream(stream)) != hipSuccess) return PCACC_E_LAUNCH;     // kernels already queued keep the word they were launched under
    return hipMemcpyToSymbol(HIP_SYMBOL(pbs_xword), &word, sizeof(int)) == hipSuccess ? PCACC_OK : PCACC_E_LAUNCH;
}
#endif
__device__ __forceinline__ void pbs_split2(float a, float b, uint32_t &hi, uint32_t &lo)
{
#ifdef PCACC_X3_EXPERIMENT
    const bool drop = pcacc_x_apply(PCACC_X_ACT(pbs_xword), a, b);
#endif
    hi = pbs_pack(a, b);
    const pcacc_f32x2 back = __builtin_convertvector(*reinterpret_cast<const pbs_f16x2 *>(&hi), pcacc_f32x2);
    lo = pbs_pack(a - back[0], b - back[1]);
#ifdef PCACC_X3_EXPERIMENT
    if (drop) lo = 0u;
#endif
}
__device__ __forceinline__ void pbs_split8(const float4 &a, const float4 &b, float s, uint4 &hi, uint4 &lo)
{
    pbs_split2(a.x * s, a.y * s, hi.x, lo.x);
    pbs_split2(a.z * s, a.w * s, hi.y, lo.y);
    pbs_split2(b.x * s, b.y * s, hi.z, lo.z);
    pbs_split2(b.z * s, b.w * s, hi.w, lo.w);
}
// max(x, 0) on a split fragment: the sign of x is the sign of its hi half (rounding keeps the sign, a lo half never outweighs its hi)
__device__ __forceinline__ void pbs_relu(pbs_frag &hi, pbs_frag &lo)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t keep = ~(((hi.u[i] >> 15) & 0x00010001u) * 0xffffu);
        hi.u[i] &= keep;
        lo.u[i] &= keep;
    }
}

// fp32 [n][k] weights -> one power-of-two scale for the matrix (hi + lo halves carry 22 bits down to 2^-17 of the largest entry and an
// absolute 2^-38 of it below: a per-row scale would buy nothing), hi / lo planes with row stride `ld` (plane stride `plane`).
// transposed: the staged matrix is src^T.  Returns 1 / scale.  `wmax`: one word of LDS scratch.
__device__ __forceinline__ float pbs_stage_weights(const float *__restrict__ src, int n, int k, bool transposed, uint16_t *dst, int ld, int plane,
                                                   unsigned *wmax)
{
    if (threadIdx.x == 0) *wmax = 0u;
    __syncthreads();
    unsigned m = 0u;
    for (int e = threadIdx.x; e < n * k; e += PBS_THREADS) {
        const float v = src[e];
        const unsigned b = (v != v) ? 0x7f800000u : __float_as_uint(fabsf(v));
        m = b > m ? b : m;
    }
    atomicMax(wmax, m);
    __syncthreads();
    const float t = pbs_scale_of(__uint_as_float(*wmax));
    for (int e = threadIdx.x; e < n * k; e += PBS_THREADS) {
        const int sr = e / k, sc = e % k;
        const int r = transposed ? sc : sr, c = transposed ? sr : sc;
        float v = src[e] * t;
#ifdef PCACC_X3_EXPERIMENT
        float v2 = v;
        const bool xdrop = pcacc_x_apply(PCACC_X_W(pbs_xword), v, v2);
#endif
        const _Float16 hi = (_Float16)v;
        _Float16 lo = (_Float16)(v - (float)hi);
#ifdef PCACC_X3_EXPERIMENT
        if (xdrop) lo = (_Float16)0.f;
#endif
        dst[r * ld + c] = *reinterpret_cast<const uint16_t *>(&hi);
        dst[plane + r * ld + c] = *reinterpret_cast<const uint16_t *>(&lo);
    }
    __syncthreads();
    return 1.f / t;
}

template <bool GATHER>
__device__ __forceinline__ const float *pbs_x_ptr(const float *__restrict__ xa, const PbsPieces &xp, int64_t row, int col, int prow)
{
    if (!GATHER) return xa + row * 64 + col;
    if (col < 32) return xa + row * 32 + col;
    return xp.b + (int64_t)prow * 32 + (col - 32);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
template <bool GATHER>
__global__ __launch_bounds__(PBS_THREADS, 2) void pfn_block_split_fwd_kernel(const float *__restrict__ xa, PbsPieces xp, const float *__restrict__ xa_amax,
                                                                          const float *__restrict__ xb_amax, const float *__restrict__ W0,
                                                                          const float *__restrict__ b0, const float *__restrict__ Ws,
                                                                          const float *__restrict__ W1, const float *__restrict__ b1,
                                                                          float *__restrict__ out, float *__restrict__ hr, uint64_t *__restrict__ xmask,
                                                                          uint32_t *__restrict__ hmask, float *__restrict__ out_amax,
                                                                          float *__restrict__ hr_amax, int64_t rows, uint16_t *__restrict__ out16,
                                                                          uint16_t *__restrict__ hr16)
{
    constexpr int XS = 72, HS = 40, OS = 36;
    constexpr int XPL = PBS_TILE * XS, HPL = PBS_TILE * HS, W64 = 32 * XS, W32 = 32 * HS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);                       // [2][128][XS]; later the out tile [128][OS] f32
    uint16_t *hs = xs + 2 * XPL;                                             // [2][128][HS]; later the relu(h) tile [128][OS] f32
    uint16_t *w0s = hs + 2 * HPL, *wss = w0s + 2 * W64, *w1s = wss + 2 * W64;  // [2][32][XS], [2][32][XS], [2][32][HS]
    float *fl = reinterpret_cast<float *>(w1s + 2 * W32);
    float *b0s = fl, *b1s = fl + 32;
    unsigned *wmax = reinterpret_cast<unsigned *>(fl + 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const float it0 = pbs_stage_weights(W0, 32, 64, false, w0s, XS, W64, wmax);
    const float its = pbs_stage_weights(Ws, 32, 64, false, wss, XS, W64, wmax);
    const float it1 = pbs_stage_weights(W1, 32, 32, false, w1s, HS, W32, wmax);
    if (threadIdx.x < 32) { b0s[threadIdx.x] = b0 ? b0[threadIdx.x] : 0.f; b1s[threadIdx.x] = b1 ? b1[threadIdx.x] : 0.f; }
    const float sx = pbs_scale_of(pbs_amax(xa_amax, GATHER ? xb_amax : nullptr));
    const float k0 = it0 / sx, ks = its / sx;
    __syncthreads();

    const int64_t n_tiles = (rows + PBS_TILE - 1) / PBS_TILE;
    float4 xreg[4][2];
    int prow[4] = {0, 0, 0, 0};
    auto fetch_rows = [&](int64_t tile) {                                     // pillar rows of the gathered half, one tile ahead of their use
        if (!GATHER) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PBS_THREADS;
            const int64_t row = tile * PBS_TILE + (c >> 3);
            prow[q] = ((c & 7) >= 4 && row < rows) ? xp.idx[row] : 0;
        }
    };
    auto fetch = [&](int64_t tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PBS_THREADS;                      // 8-channel chunk: row c >> 3, columns 8 (c & 7) ..
            const int64_t row = tile * PBS_TILE + (c >> 3);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (row < rows) {
                const float *src = pbs_x_ptr<GATHER>(xa, xp, row, (c & 7) * 8, prow[q]);
                a = *reinterpret_cast<const float4 *>(src);
                b = *reinterpret_cast<const float4 *>(src + 4);
            }
            xreg[q][0] = a;
            xreg[q][1] = b;
        }
    };
    float omax = 0.f, hmax_all = 0.f;
    int64_t tile = blockIdx.x;
    fetch_rows(tile);
    if (tile < n_tiles) fetch(tile);
    fetch_rows(tile + gridDim.x);
    const int myrow = wave * 32 + lp;
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                                      // the previous tile's output tiles have left the LDS
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PBS_THREADS;
            uint4 hi, lo;
            pbs_split8(xreg[q][0], xreg[q][1], sx, hi, lo);
            uint16_t *dst = xs + (c >> 3) * XS + (c & 7) * 8;
            *reinterpret_cast<uint4 *>(dst) = hi;
            *reinterpret_cast<uint4 *>(dst + XPL) = lo;
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);              // in flight during the rest of this tile
        fetch_rows(tile + 2 * (int64_t)gridDim.x);

        pbs_f32x16 acc_h, acc_s, acc_1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_h[r] = 0.f; acc_s[r] = 0.f; acc_1[r] = 0.f; }
        const uint16_t *xrow = xs + myrow * XS + lh * 8;
        uint32_t xlo = 0, xhi = 0;                                            // x > 0, this lane's 32 of the row's 64 channels (before the lane's shift)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            pbs_frag bh, bl, rh, rl;
            bh.q = *reinterpret_cast<const uint4 *>(xrow + kc * 16);
            bl.q = *reinterpret_cast<const uint4 *>(xrow + XPL + kc * 16);
            rh = bh; rl = bl;
            pbs_relu(rh, rl);
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                     // channel kc*16 + lh*8 + 2i (+1): positive iff the relu kept it and it is not zero
                const uint32_t h2 = rh.u[i] | rl.u[i];
                const uint32_t two = ((h2 & 0x00007fffu) ? 1u : 0u) | ((h2 & 0x7fff0000u) ? 2u : 0u);
                if (kc < 2) xlo |= two << (kc * 16 + 2 * i);
                else xhi |= two << ((kc - 2) * 16 + 2 * i);
            }
            const pbs_f16x8 a0h = *reinterpret_cast<const pbs_f16x8 *>(w0s + lp * XS + lh * 8 + kc * 16);
            const pbs_f16x8 a0l = *reinterpret_cast<const pbs_f16x8 *>(w0s + W64 + lp * XS + lh * 8 + kc * 16);
            const pbs_f16x8 ash = *reinterpret_cast<const pbs_f16x8 *>(wss + lp * XS + lh * 8 + kc * 16);
            const pbs_f16x8 asl = *reinterpret_cast<const pbs_f16x8 *>(wss + W64 + lp * XS + lh * 8 + kc * 16);
            acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, rl.v, acc_h, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ash, bl.v, acc_s, 0, 0, 0);
            acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, rh.v, acc_h, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_f16(asl, bh.v, acc_s, 0, 0, 0);
            acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, rh.v, acc_h, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ash, bh.v, acc_s, 0, 0, 0);
        }
        xlo <<= 8 * lh;
        xhi <<= 8 * lh;
        xlo |= __shfl_xor(xlo, 32, 64);                                       // the other half-wave holds the row's other 32 channels
        xhi |= __shfl_xor(xhi, 32, 64);
        const uint64_t xbits = ((uint64_t)xhi << 32) | xlo;
        // relu(h): value, per-wave maximum -> this wave's scale, split into the wave's rows of the hs planes
        float hv[16], hm = 0.f;
        uint32_t hbits = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 8 * g + 4 * lh + q;
                const float v = fmaxf(acc_h[4 * g + q] * k0 + b0s[c], 0.f);
                hv[4 * g + q] = v;
                hm = fmaxf(hm, v);
                if (v != v) hm = __builtin_inff();
                if (v > 0.f) hbits |= 1u << (8 * g + q);
            }
        hbits <<= 4 * lh;
        hbits |= __shfl_xor(hbits, 32, 64);
        hm = pbs_wave_max(hm);
        hmax_all = fmaxf(hmax_all, hm);
        const float sh = pbs_scale_of(hm), k1 = it1 / sh;
        uint16_t *hrow = hs + myrow * HS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 ph, pl;
            pbs_split2(hv[4 * g] * sh, hv[4 * g + 1] * sh, ph.x, pl.x);
            pbs_split2(hv[4 * g + 2] * sh, hv[4 * g + 3] * sh, ph.y, pl.y);
            *reinterpret_cast<uint2 *>(hrow + 8 * g + 4 * lh) = ph;
            *reinterpret_cast<uint2 *>(hrow + HPL + 8 * g + 4 * lh) = pl;
        }
        __syncthreads();                                                      // relu(h) planes complete; every wave is done reading the x planes
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const pbs_f16x8 bh = *reinterpret_cast<const pbs_f16x8 *>(hrow + lh * 8 + kc * 16);
            const pbs_f16x8 bl = *reinterpret_cast<const pbs_f16x8 *>(hrow + HPL + lh * 8 + kc * 16);
            const pbs_f16x8 a1h = *reinterpret_cast<const pbs_f16x8 *>(w1s + lp * HS + lh * 8 + kc * 16);
            const pbs_f16x8 a1l = *reinterpret_cast<const pbs_f16x8 *>(w1s + W32 + lp * HS + lh * 8 + kc * 16);
            acc_1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, acc_1, 0, 0, 0);
            acc_1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, acc_1, 0, 0, 0);
            acc_1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, acc_1, 0, 0, 0);
        }
        __syncthreads();                                                      // every wave is done with the relu(h) planes
        float *orow = reinterpret_cast<float *>(xs) + myrow * OS, *hrow32 = reinterpret_cast<float *>(hs) + myrow * OS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * g + 4 * lh;
            float4 o, hq;
            o.x = acc_s[4 * g] * ks + acc_1[4 * g] * k1 + b1s[c];
            o.y = acc_s[4 * g + 1] * ks + acc_1[4 * g + 1] * k1 + b1s[c + 1];
            o.z = acc_s[4 * g + 2] * ks + acc_1[4 * g + 2] * k1 + b1s[c + 2];
            o.w = acc_s[4 * g + 3] * ks + acc_1[4 * g + 3] * k1 + b1s[c + 3];
            hq = make_float4(hv[4 * g], hv[4 * g + 1], hv[4 * g + 2], hv[4 * g + 3]);
            *reinterpret_cast<float4 *>(orow + c) = o;
            *reinterpret_cast<float4 *>(hrow32 + c) = hq;
        }
        {
            const int64_t row = tile * PBS_TILE + myrow;
            if (lh == 0 && row < rows && xmask) { xmask[row] = xbits; hmask[row] = hbits; }
        }
        __syncthreads();                                                      // the two output tiles are staged
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PBS_THREADS;                      // 16-byte piece: row c >> 3, columns 4 (c & 7) ..
            const int64_t row = tile * PBS_TILE + (c >> 3);
            if (row >= rows) continue;
            const float4 o = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(xs) + (c >> 3) * OS + (c & 7) * 4);
            *reinterpret_cast<float4 *>(out + row * 32 + (c & 7) * 4) = o;
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            if (!(o.x == o.x && o.y == o.y && o.z == o.z && o.w == o.w)) omax = __builtin_inff();
            if (out16)                                                        // 'mixed' mode: the bf16 shadows the bf16 backward kernel reads
                *reinterpret_cast<uint2 *>(out16 + row * 32 + (c & 7) * 4) = make_uint2(pcacc_pack_bf16x2(o.x, o.y), pcacc_pack_bf16x2(o.z, o.w));
            if (hr || hr16) {
                const float4 hq = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(hs) + (c >> 3) * OS + (c & 7) * 4);
                if (hr) *reinterpret_cast<float4 *>(hr + row * 32 + (c & 7) * 4) = hq;
                if (hr16) *reinterpret_cast<uint2 *>(hr16 + row * 32 + (c & 7) * 4) = make_uint2(pcacc_pack_bf16x2(hq.x, hq.y), pcacc_pack_bf16x2(hq.z, hq.w));
            }
        }
    }
    omax = pbs_wave_max(omax);
    if (lane == 0) {
        if (out_amax) atomicMax(reinterpret_cast<unsigned *>(out_amax) + (blockIdx.x & 255), __float_as_uint(omax));
        if (hr_amax) atomicMax(reinterpret_cast<unsigned *>(hr_amax) + (blockIdx.x & 255), __float_as_uint(hmax_all));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// data half of the backward:  d(h) = (d(out) W1) where h > 0;   d(x) = (d(h) W0) where x > 0, + d(out) Ws
template <bool SPLIT_OUT>
__global__ __launch_bounds__(PBS_THREADS, 2) void pfn_block_split_dgrad_kernel(const float *__restrict__ gout, const float *__restrict__ g_amax,
                                                                            const uint64_t *__restrict__ xmask, const uint32_t *__restrict__ hmask,
                                                                            const float *__restrict__ W0, const float *__restrict__ Ws,
                                                                            const float *__restrict__ W1, float *__restrict__ gxa,
                                                                            float *__restrict__ gxb, float *__restrict__ dh, float *__restrict__ gx_amax,
                                                                            float *__restrict__ dh_amax, int64_t rows)
{
    constexpr int GS = 40, XO = 68, HO = 36;
    constexpr int GPL = PBS_TILE * GS, W32 = 32 * GS, W64 = 64 * GS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *gs = reinterpret_cast<uint16_t *>(smem);                       // [2][128][GS] d(out);   region A = gs + ds planes, later the d(x) / d(h) tiles
    uint16_t *ds = gs + 2 * GPL;                                             // [2][128][GS] d(h)
    uint16_t *w1t = ds + 2 * GPL, *w0t = w1t + 2 * W32, *wst = w0t + 2 * W64;  // W1^T [32 h][32 o], W0^T [64 i][32 h], Ws^T [64 i][32 o]
    float *fl = reinterpret_cast<float *>(wst + 2 * W64);
    unsigned *wmax = reinterpret_cast<unsigned *>(fl);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const float it1 = pbs_stage_weights(W1, 32, 32, true, w1t, GS, W32, wmax);
    const float it0 = pbs_stage_weights(W0, 32, 64, true, w0t, GS, W64, wmax);
    const float its = pbs_stage_weights(Ws, 32, 64, true, wst, GS, W64, wmax);
    const float sg = pbs_scale_of(pbs_amax(g_amax, nullptr));
    const float k1 = it1 / sg, ks = its / sg;

    const int64_t n_tiles = (rows + PBS_TILE - 1) / PBS_TILE;
    float4 greg[2][2];
    auto fetch = [&](int64_t tile) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = threadIdx.x + q * PBS_THREADS;                      // 8-channel chunk: row c >> 2, columns 8 (c & 3) ..
            const int64_t row = tile * PBS_TILE + (c >> 2);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (row < rows) {
                const float *src = gout + row * 32 + (c & 3) * 8;
                a = *reinterpret_cast<const float4 *>(src);
                b = *reinterpret_cast<const float4 *>(src + 4);
            }
            greg[q][0] = a;
            greg[q][1] = b;
        }
    };
    float xmax = 0.f, dmax_all = 0.f;
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) fetch(tile);
    const int myrow = wave * 32 + lp;
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                                      // the previous tile's output tiles have left region A
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = threadIdx.x + q * PBS_THREADS;
            uint4 hi, lo;
            pbs_split8(greg[q][0], greg[q][1], sg, hi, lo);
            uint16_t *dst = gs + (c >> 2) * GS + (c & 3) * 8;
            *reinterpret_cast<uint4 *>(dst) = hi;
            *reinterpret_cast<uint4 *>(dst + GPL) = lo;
        }
        const int64_t row = tile * PBS_TILE + myrow;
        const uint64_t xb = (row < rows ? xmask[row] : 0ull) >> (4 * lh);     // this lane's channels at the constant positions 8g + q
        const uint32_t hb = (row < rows ? hmask[row] : 0u) >> (4 * lh);
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);

        // A. d(h) = (d(out) W1) where h > 0
        pbs_f16x8 gh[2], gl[2];
        pbs_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            gh[kc] = *reinterpret_cast<const pbs_f16x8 *>(gs + myrow * GS + lh * 8 + kc * 16);
            gl[kc] = *reinterpret_cast<const pbs_f16x8 *>(gs + GPL + myrow * GS + lh * 8 + kc * 16);
            const pbs_f16x8 ah = *reinterpret_cast<const pbs_f16x8 *>(w1t + lp * GS + lh * 8 + kc * 16);
            const pbs_f16x8 al = *reinterpret_cast<const pbs_f16x8 *>(w1t + W32 + lp * GS + lh * 8 + kc * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl[kc], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh[kc], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh[kc], acc, 0, 0, 0);
        }
        float dv[16], dm = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = ((hb >> (8 * g + q)) & 1u) ? acc[4 * g + q] * k1 : 0.f;
                dv[4 * g + q] = v;
                dm = fmaxf(dm, fabsf(v));
                if (v != v) dm = __builtin_inff();
            }
        dm = pbs_wave_max(dm);
        dmax_all = fmaxf(dmax_all, dm);
        const float sd = pbs_scale_of(dm), k0 = it0 / sd;
        uint16_t *drow = ds + myrow * GS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 ph, pl;
            pbs_split2(dv[4 * g] * sd, dv[4 * g + 1] * sd, ph.x, pl.x);
            pbs_split2(dv[4 * g + 2] * sd, dv[4 * g + 3] * sd, ph.y, pl.y);
            *reinterpret_cast<uint2 *>(drow + 8 * g + 4 * lh) = ph;
            *reinterpret_cast<uint2 *>(drow + GPL + 8 * g + 4 * lh) = pl;
        }
        __syncthreads();                                                      // d(h) planes visible to the wave's other lanes

        // B. d(x) = (d(h) W0) where x > 0, + d(out) Ws: the two products carry different scales -> two accumulators per 32-column tile
        float4 ox[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            pbs_f32x16 a1, a2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a1[r] = 0.f; a2[r] = 0.f; }
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const pbs_f16x8 dhh = *reinterpret_cast<const pbs_f16x8 *>(drow + lh * 8 + kc * 16);
                const pbs_f16x8 dhl = *reinterpret_cast<const pbs_f16x8 *>(drow + GPL + lh * 8 + kc * 16);
                const pbs_f16x8 a0h = *reinterpret_cast<const pbs_f16x8 *>(w0t + (nt * 32 + lp) * GS + lh * 8 + kc * 16);
                const pbs_f16x8 a0l = *reinterpret_cast<const pbs_f16x8 *>(w0t + W64 + (nt * 32 + lp) * GS + lh * 8 + kc * 16);
                const pbs_f16x8 ash = *reinterpret_cast<const pbs_f16x8 *>(wst + (nt * 32 + lp) * GS + lh * 8 + kc * 16);
                const pbs_f16x8 asl = *reinterpret_cast<const pbs_f16x8 *>(wst + W64 + (nt * 32 + lp) * GS + lh * 8 + kc * 16);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, dhl, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ash, gl[kc], a2, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, dhh, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asl, gh[kc], a2, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, dhh, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ash, gh[kc], a2, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = (((xb >> (nt * 32 + 8 * g + q)) & 1ull) ? a1[4 * g + q] * k0 : 0.f) + a2[4 * g + q] * ks;
                }
                ox[nt][g] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();                                                      // every wave is done with the d(out) / d(h) planes (region A)
        float *xo = reinterpret_cast<float *>(smem) + myrow * XO;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4 *>(xo + nt * 32 + 8 * g + 4 * lh) = ox[nt][g];
        __syncthreads();
        {
            const int col = (threadIdx.x & 15) * 4;                           // 16-byte pieces: rows (threadIdx.x >> 4) + 16 q, this thread's 4 columns
            float *gbase = !SPLIT_OUT ? gxa + col : (col < 32 ? gxa + col : gxb + (col - 32));
            constexpr int gstride = SPLIT_OUT ? 32 : 64;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int lr = (threadIdx.x >> 4) + 16 * q;
                const int64_t r = tile * PBS_TILE + lr;
                if (r >= rows) continue;
                const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(smem) + lr * XO + col);
                xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) xmax = __builtin_inff();
                *reinterpret_cast<float4 *>(gbase + r * gstride) = v;
            }
        }
        __syncthreads();                                                      // the d(x) tile has left region A
        float *ho = reinterpret_cast<float *>(smem) + myrow * HO;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<float4 *>(ho + 8 * g + 4 * lh) = make_float4(dv[4 * g], dv[4 * g + 1], dv[4 * g + 2], dv[4 * g + 3]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * PBS_THREADS;
            const int64_t r = tile * PBS_TILE + (c >> 3);
            if (r < rows) *reinterpret_cast<float4 *>(dh + r * 32 + (c & 7) * 4) =
                              *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(smem) + (c >> 3) * HO + (c & 7) * 4);
        }
    }
    xmax = pbs_wave_max(xmax);
    if (lane == 0) {
        if (gx_amax) atomicMax(reinterpret_cast<unsigned *>(gx_amax) + (blockIdx.x & 255), __float_as_uint(xmax));
        if (dh_amax) atomicMax(reinterpret_cast<unsigned *>(dh_amax) + (blockIdx.x & 255), __float_as_uint(dmax_all));
    }
}

static int pbs_grid(int64_t rows, int per_cu)
{
    const int64_t n_tiles = (rows + PBS_TILE - 1) / PBS_TILE;
    const int64_t grid = (int64_t)PCACC_CUS * per_cu;
    return (int)(grid > n_tiles ? n_tiles : grid);
}

static constexpr size_t PBS_FWD_LDS = (size_t)(2 * PBS_TILE * 72 + 2 * PBS_TILE * 40 + 4 * 32 * 72 + 2 * 32 * 40) * 2 + 64 * 4 + 16;
static constexpr size_t PBS_DG_LDS = (size_t)(4 * PBS_TILE * 40 + 2 * 32 * 40 + 4 * 64 * 40) * 2 + 16;

// out [rows,32], relu_h [rows,32] f32; xmask [rows] u64 / hmask [rows] u32: x > 0 / h > 0 per channel; out_amax / hr_amax: 256 zeroed slots each
static int pfn_block_split_forward_impl(const float *xa, const float *xa_amax, const float *pooled, const float *pooled_amax, const int32_t *p2v,
                                        const float *w0, const float *b0, const float *ws, const float *w1, const float *b1, float *out,
                                        float *relu_h, uint64_t *xmask, uint32_t *hmask, float *out_amax, float *hr_amax, int64_t rows,
                                        uint16_t *out16, uint16_t *hr16, void *stream)
{
    if (rows < 0 || (pooled && (!p2v || !pooled_amax))) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!xa || !xa_amax || !w0 || !ws || !w1 || !out || (!out16 && (!relu_h || !xmask || !hmask)) || (xmask && !hmask)) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    const int grid = pbs_grid(rows, 2);
    const PbsPieces xp{pooled, p2v};
    if (pooled) {
        auto kern = pfn_block_split_fwd_kernel<true>;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PBS_FWD_LDS) != hipSuccess) return PCACC_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PBS_THREADS), PBS_FWD_LDS, st, xa, xp, xa_amax, pooled_amax, w0, b0, ws, w1, b1, out, relu_h, xmask, hmask,
                           out_amax, hr_amax, rows, out16, hr16);
    } else {
        auto kern = pfn_block_split_fwd_kernel<false>;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PBS_FWD_LDS) != hipSuccess) return PCACC_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PBS_THREADS), PBS_FWD_LDS, st, xa, xp, xa_amax, nullptr, w0, b0, ws, w1, b1, out, relu_h, xmask, hmask,
                           out_amax, hr_amax, rows, out16, hr16);
    }
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_pfn_block_split_forward(const float *xa, const float *xa_amax, const float *pooled, const float *pooled_amax, const int32_t *p2v,
                                             const float *w0, const float *b0, const float *ws, const float *w1, const float *b1, float *out,
                                             float *relu_h, uint64_t *xmask, uint32_t *hmask, float *out_amax, float *hr_amax, int64_t rows,
                                             void *stream)
{
    return pfn_block_split_forward_impl(xa, xa_amax, pooled, pooled_amax, p2v, w0, b0, ws, w1, b1, out, relu_h, xmask, hmask, out_amax, hr_amax, rows,
                                        nullptr, nullptr, stream);
}

// 'mixed' compute mode: the fp32 result plus what the bf16 backward kernel (pcacc_pfn_block_backward) reads -- out16 = bf16(out), hr16 = bf16(relu(h)) --
// from the same epilogue; no fp32 relu(h), no sign masks (the fp32x3 data-gradient kernel's inputs)
extern "C" int pcacc_pfn_block_split_forward_dual(const float *xa, const float *xa_amax, const float *pooled, const float *pooled_amax, const int32_t *p2v,
                                                  const float *w0, const float *b0, const float *ws, const float *w1, const float *b1, float *out,
                                                  uint16_t *out16, uint16_t *hr16, float *out_amax, int64_t rows, void *stream)
{
    if (!out16 || !hr16) return PCACC_E_ARG;
    return pfn_block_split_forward_impl(xa, xa_amax, pooled, pooled_amax, p2v, w0, b0, ws, w1, b1, out, nullptr, nullptr, nullptr, out_amax, nullptr, rows,
                                        out16, hr16, stream);
}

// grad_xa [rows,64] (grad_xb NULL) or grad_xa [rows,32] + grad_xb [rows,32]; grad_h [rows,32]: d(h), the operand of fc_0's weight gradient
extern "C" int pcacc_pfn_block_split_dgrad(const float *grad_out, const float *grad_out_amax, const uint64_t *xmask, const uint32_t *hmask,
                                           const float *w0, const float *ws, const float *w1, float *grad_xa, float *grad_xb, float *grad_h,
                                           float *gx_amax, float *dh_amax, int64_t rows, void *stream)
{
    if (rows < 0) return PCACC_E_ARG;
    if (rows == 0) return PCACC_OK;
    if (!grad_out || !grad_out_amax || !xmask || !hmask || !w0 || !ws || !w1 || !grad_xa || !grad_h) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    const int grid = pbs_grid(rows, 2);
    if (grad_xb) {
        auto kern = pfn_block_split_dgrad_kernel<true>;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PBS_DG_LDS) != hipSuccess) return PCACC_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PBS_THREADS), PBS_DG_LDS, st, grad_out, grad_out_amax, xmask, hmask, w0, ws, w1, grad_xa, grad_xb, grad_h,
                           gx_amax, dh_amax, rows);
    } else {
        auto kern = pfn_block_split_dgrad_kernel<false>;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PBS_DG_LDS) != hipSuccess) return PCACC_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PBS_THREADS), PBS_DG_LDS, st, grad_out, grad_out_amax, xmask, hmask, w0, ws, w1, grad_xa, nullptr, grad_h,
                           gx_amax, dh_amax, rows);
    }
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
