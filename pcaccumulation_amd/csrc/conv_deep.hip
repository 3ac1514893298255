// 3x3 convolution for the deep layers (c_in >= 128, images of 18^2 .. 144^2 pixels): the 128 -> 128 ... 512 -> 512 layers of
// models/unet.py:45-113 and of the STPN backbone (models/stpn.py:24-43), forward and -- on mirrored / transposed weights -- data
// gradient.  Same implicit GEMM as conv.hip (v_mfma_f32_32x32x16_bf16, A = 32 output channels x 16 k, B = 16 k x 32 pixels,
// channels-last bf16 in / out, fp32 accumulate, bias + ReLU in the epilogue), re-tiled for small images and deep K:
//
//   * M-tiles are 32 CONSECUTIVE PIXELS OF A STRIP (rows x W pixels of one image, row-major), not 32 pixels of one image row: an
//     18 x 18 image is 324 pixels = 10.1 tiles instead of 3 x 1 tiles of 8 x 32 with 58 % of the lanes outside the image.  Every lane
//     keeps the LDS offset of its pixel's 3x3 window; a tap only adds a constant.
//   * The strip's input patch ((rows + 2) x (W + 2) pixels x CS channels) sits in LDS for the 9 taps of a channel slice; the next
//     slice's patch travels in registers during the MFMAs.  Weight tiles ([64 NG output channels] x [CS] per tap) are double
//     buffered in LDS: one barrier per tap.
//   * 8 waves: NG = 2: 4 pixel groups x 2 channel groups, 3 x 2 accumulator tiles per wave (strips up to 384 pixels, 128 output
//     channels per workgroup); NG = 1: 8 pixel groups, 2 x 2 tiles (512 pixels, 64 output channels).
//   * Block order: the output-channel group is the fastest index, so an XCD (block % 8) keeps touching the same slice of the
//     weights (1.2 MB at 512 input channels) in its 4 MB L2.
#include "common.h"

#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CD_THREADS 512
#define CD_PCH 10                              // patch pieces (16 B) a thread carries per channel slice

template <int CS, int NG, int MT, bool MASKED>          // MASKED: in_mask != NULL as a compile-time fact (a run-time branch behind the loads makes the compiler copy -- and wait for -- every loaded register)
__global__ __launch_bounds__(CD_THREADS) void conv3x3_strip_kernel(const uint16_t *__restrict__ in, const uint16_t *__restrict__ in_mask,
                                                                   const uint16_t *__restrict__ wp,
                                                                   const float *__restrict__ bias, uint16_t *__restrict__ out, int n_img,
                                                                   int h, int w, int c_in, int c_out, int relu, int rows, int strips,
                                                                   int co_groups, int xcd)
{
    constexpr int PS = CS + 8;                                 // padded LDS row (elements): conflict-free 16-byte fragment reads
    constexpr int MG = NG == 2 ? 4 : 8;                        // waves along the pixel dimension; MT = 32-pixel tiles per wave
    constexpr int WROWS = NG * 64;                             // weight rows (output channels) per workgroup
    constexpr int C8 = CS / 8;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int pw = w + 2, pp = (rows + 2) * pw;                // patch width / pixels
    uint16_t *patch = lds;                                     // [pp][PS]
    uint16_t *wbuf = lds + (size_t)pp * PS;                    // [2][WROWS][PS]

    // xcd: the XCD-contiguous walk (common.h) -- the channel groups of a strip on one XCD share its input patch in that L2; an XCD then sees every group's
    // weight slice instead of the few that block % 8 gave it (the block order of round 2), which the strips of a launch re-read from the L2 all the same
    int bid = xcd ? pcacc_xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const int cog = bid % co_groups; bid /= co_groups;
    const int strip = bid % strips;
    const int img = bid / strips;
    const int y0 = strip * rows;
    const int co0 = cog * WROWS;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int mg = wave % MG, ng = wave / MG;
    const int n_px = rows * w;

    // the lane's pixels: LDS offset of the top-left tap of their 3x3 windows, image position for the store
    int poff[MT], pyx[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int q = (mg + MG * j) * 32 + lp;
        const int y = q / w, x = q - y * w;
        const bool ok = q < n_px && y0 + y < h;
        poff[j] = ok ? (y * pw + x) * PS : 0;
        pyx[j] = ok ? ((y0 + y) << 16 | x) : -1;
    }
    f32x16_t acc[MT][2];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

    // patch pieces of this thread: position packed as py << 20 | px << 8 | c8 (0x7ff rows never pass the bounds test)
    const int n_chunks = pp * C8;
    int pinfo[CD_PCH];
    uint4 preg[CD_PCH];
#pragma unroll
    for (int q = 0; q < CD_PCH; ++q) {
        const int c = threadIdx.x + q * CD_THREADS;
        const int px = c / C8, c8 = c - px * C8;
        const int py = px / pw, pxx = px - py * pw;
        pinfo[q] = c < n_chunks ? (py << 20 | pxx << 8 | c8) : (0x7ff << 20);
    }
    const uint16_t *img_in = in + (int64_t)img * h * w * c_in;
    int pok = 0;                                               // bit q: piece q of preg lies inside the image (else it is written to LDS as zeros)
    // [r5] addresses first, then every load of the slice back to back, masking behind them, the zero select in write_patch (a slice later).  A select (and,
    // with a mask, a branch) behind each load had made the fetch a chain of `global_load, s_waitcnt vmcnt(0)`: ten memory round trips in series per slice.
    auto piece_off = [&](int q, int cs, bool *ok) __attribute__((always_inline)) {   // element offset of piece q inside the image (clamped: always valid)
        const int py = pinfo[q] >> 20, pxx = (pinfo[q] >> 8) & 0xfff, c8 = pinfo[q] & 0xff;
        const int y = y0 - 1 + py, x = pxx - 1;
        *ok = (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w;
        return (min(max(y, 0), h - 1) * w + min(max(x, 0), w - 1)) * c_in + cs * CS + c8 * 8;      // h * w * c_in < 2^31: checked by the planner
    };
    auto fetch_patch = [&](int cs) {
        int okb = 0;
#pragma unroll
        for (int q = 0; q < CD_PCH; ++q) {                     // straight-line: no select, no branch between the loads
            bool ok;
            preg[q] = *reinterpret_cast<const uint4 *>(img_in + piece_off(q, cs, &ok));
            okb |= ok ? (1 << q) : 0;
        }
        if constexpr (MASKED) {                                // the ReLU backward of the layer whose gradient `in` is
            const uint16_t *mk = in_mask + (int64_t)img * h * w * c_in;
            constexpr int ROUNDS = MT >= 3 ? 5 : 2, HALF = CD_PCH / ROUNDS;   // rounds of mask loads: the registers for all ten at once are not there (three-tile waves: two at a time)
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                uint4 m[HALF];
                bool ok;
#pragma unroll
                for (int q = 0; q < HALF; ++q) m[q] = *reinterpret_cast<const uint4 *>(mk + piece_off(r * HALF + q, cs, &ok));
#pragma unroll
                for (int q = 0; q < HALF; ++q) preg[r * HALF + q] = pcacc_relu_mask8(preg[r * HALF + q], m[q]);
            }
        }
        pok = okb;
    };
    auto write_patch = [&]() {
#pragma unroll
        for (int q = 0; q < CD_PCH; ++q) {
            const int c = threadIdx.x + q * CD_THREADS;
            if (c < n_chunks) *reinterpret_cast<uint4 *>(patch + (c / C8) * PS + (c % C8) * 8) = (pok >> q) & 1 ? preg[q] : make_uint4(0, 0, 0, 0);
        }
    };
    // weight tile of (tap, slice): WROWS rows of CS elements = WROWS * C8 pieces, <= 2 per thread.  Tiles are requested TWO taps
    // ahead into alternating register sets (plain loads stay in flight across the barriers) and written to the other LDS buffer
    // one tap ahead: a tap of few MFMAs (small MT) does not wait for an L2 round trip.
    constexpr int W_CHUNKS = WROWS * C8, W_PER = (W_CHUNKS + CD_THREADS - 1) / CD_THREADS;
    uint4 wreg[2][W_PER];
    const int n_slices = c_in / CS, n_taps = n_slices * 9;
    auto fetch_w = [&](int set, int g) {                       // g = linear tap index: slice g / 9, tap g % 9
        const int cs = g / 9, tap = g - cs * 9;
        const uint16_t *src = wp + ((int64_t)tap * c_out + co0) * c_in + cs * CS;
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = (W_CHUNKS % CD_THREADS) ? min(threadIdx.x + q * CD_THREADS, W_CHUNKS - 1) : threadIdx.x + q * CD_THREADS;
            wreg[set][q] = *reinterpret_cast<const uint4 *>(src + (int64_t)(c / C8) * c_in + (c % C8) * 8);   // clamped, never masked
        }
    };
    auto write_w = [&](uint16_t *dst, int set) {
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = threadIdx.x + q * CD_THREADS;
            if (c < W_CHUNKS) *reinterpret_cast<uint4 *>(dst + (c / C8) * PS + (c % C8) * 8) = wreg[set][q];
        }
    };

    fetch_patch(0);
    fetch_w(0, 0);
    if (n_taps > 1) fetch_w(1, 1);
    uint16_t *wb0 = wbuf, *wb1 = wbuf + WROWS * PS;             // buffer of the even / odd taps of the current slice
    write_w(wb0, 0);                                           // tap 0 (nobody reads LDS yet)
    for (int cs = 0; cs < n_slices; ++cs) {
        __syncthreads();                                       // every wave is done with the previous slice's patch
        write_patch();
#ifndef CD_EXP_NOPATCHPF
        if (cs + 1 < n_slices) fetch_patch(cs + 1);            // in flight during the nine taps below
#endif
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int g = cs * 9 + tap;
            uint16_t *cur = (tap & 1) ? wb1 : wb0, *other = (tap & 1) ? wb0 : wb1;
#ifndef CD_EXP_NOBARRIER
            __syncthreads();                                   // buffer `cur` (and, at tap 0, the patch) is visible
#endif
#ifndef CD_EXP_NOWFETCH
            if (g + 2 < n_taps) fetch_w(tap & 1, g + 2);       // set (tap & 1) held tap g: already in LDS
#endif
            const uint16_t *wb = cur + (ng * 64 + lp) * PS + lh * 8;
            const int toff = ((tap / 3) * pw + tap % 3) * PS + lh * 8;
            constexpr int KC = CS / 16;
            constexpr int FB = MT >= 3 ? 1 : 2;                // fragment sets: the 3-tile waves have no registers for a second one
            bf16x8_t fa[FB][2], fb[FB][MT];
            auto load = [&](int slot, int kc) {
#pragma unroll
                for (int n = 0; n < 2; ++n) fa[slot][n] = *reinterpret_cast<const bf16x8_t *>(wb + n * 32 * PS + kc * 16);
#pragma unroll
                for (int j = 0; j < MT; ++j) fb[slot][j] = *reinterpret_cast<const bf16x8_t *>(patch + poff[j] + toff + kc * 16);
            };
            if (FB == 2) load(0, 0);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                if (FB == 1) load(0, kc);
                else if (kc + 1 < KC) load((kc + 1) & 1, kc + 1);   // fragments of the next step in flight under this step's MFMAs
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#ifndef CD_EXP_NOMFMA
                        acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kc & (FB - 1)][n], fb[kc & (FB - 1)][j], acc[j][n], 0, 0, 0);
#else
                        acc[j][n][0] += (float)fa[kc & (FB - 1)][n][0] * (float)fb[kc & (FB - 1)][j][0];
#endif
            }
#ifndef CD_EXP_NOWWRITE
            if (g + 1 < n_taps) write_w(other, (tap + 1) & 1);   // tap g + 1 (requested two taps ago) into the buffer tap g - 1 used
#endif
        }
        // nine taps per slice: the next slice's tap 0 sits in register set 1 / goes to the odd buffer -- swap the roles (a few moves)
        {
            uint16_t *t = wb0; wb0 = wb1; wb1 = t;
#pragma unroll
            for (int q = 0; q < W_PER; ++q) { const uint4 v = wreg[0][q]; wreg[0][q] = wreg[1][q]; wreg[1][q] = v; }
        }
    }

    // epilogue: lane = pixel, register quad g of tile n = channels n*32 + 8g + 4*lh .. +3 of this wave's 64
    const float *bptr = bias ? bias + co0 + ng * 64 : nullptr;
    // [r5] the wave's eight bias quads as ONE batch of loads (they were loaded, and waited for, one in front of every store): once for all tiles where the
    // registers are there, once per tile in the three-tile waves
    float4 bvs[2][4];
    auto load_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) bvs[n][g] = bptr ? *reinterpret_cast<const float4 *>(bptr + n * 32 + 8 * g + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (MT < 3) load_bias();
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        if (MT >= 3) load_bias();
        if (pyx[j] < 0) continue;
        uint16_t *dst = out + (((int64_t)img * h + (pyx[j] >> 16)) * w + (pyx[j] & 0xffff)) * c_out + co0 + ng * 64;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n * 32 + 8 * g + 4 * lh;
                const float4 bv = bvs[n][g];
                float v[4] = {acc[j][n][4 * g] + bv.x, acc[j][n][4 * g + 1] + bv.y, acc[j][n][4 * g + 2] + bv.z, acc[j][n][4 * g + 3] + bv.w};
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                *reinterpret_cast<uint2 *>(dst + c) = make_uint2(pcacc_pack_bf16x2(v[0], v[1]), pcacc_pack_bf16x2(v[2], v[3]));
            }
    }
}

// Tiling of a layer: NG (64-channel groups per workgroup: 2 -> 4 pixel groups of waves, 1 -> 8), MT (32-pixel tiles per wave) and the
// strip height.  Every wave runs MT x 2 accumulator tiles over the whole K, so a workgroup's time goes with MT; the launch takes
// ceil(blocks / 256 CUs) rounds of it (LDS leaves one workgroup per CU).  Pick the cheapest rounds x MT; among equals the larger MT
// (the weight tiles staged per tap are shared by more work), then fewer padded tiles.
struct ConvStripPlan { int cs, ng, mt, rows, strips, co_groups; size_t lds; int64_t blocks; };

static bool conv_strip_fits(int cs, int ng, int rows, int w, size_t *lds)
{
    const int pp = (rows + 2) * (w + 2);
    *lds = ((size_t)pp + 2 * ng * 64) * (cs + 8) * sizeof(uint16_t);
    return pp * (cs / 8) <= CD_THREADS * CD_PCH && *lds <= 160 * 1024;
}

static bool conv_strip_plan(int n_img, int h, int w, int c_in, int c_out, ConvStripPlan *best)
{
    if (c_in < 128 || c_in % 64 || c_out < 64 || c_out % 64 || h < 1 || w < 1 || (int64_t)h * w * c_in >= 0x7fffffffLL) return false;   // the kernel's in-image offsets are 32-bit
    bool found = false;
    int64_t best_cost = 0, best_waste = 0;
    for (int ng = 2; ng >= 1; --ng) {
        if (c_out % (64 * ng)) continue;
        const int mg = ng == 2 ? 4 : 8;
        for (int mt = 3; mt >= 1; --mt) {
            if (ng == 1 && mt > 2) continue;                   // instantiated: NG = 2 with MT 1..3, NG = 1 with MT 1..2
            int rows = mg * mt * 32 / w;
            if (rows > h) rows = h;
            for (int cs = 64; cs >= 32; cs -= 32) {
                size_t lds;
                int r = rows;
                while (r >= 1 && !conv_strip_fits(cs, ng, r, w, &lds)) --r;
                if (r < 1) continue;
                const int strips = (h + r - 1) / r;            // balance the strips: the same count with the least height
                r = (h + strips - 1) / strips;
                if (!conv_strip_fits(cs, ng, r, w, &lds)) continue;
                const int tiles = (r * w + 31) / 32;
                if (mt > 1 && tiles <= mg * (mt - 1)) break;   // a smaller MT covers this strip
                const int64_t blocks = (int64_t)n_img * strips * (c_out / (64 * ng));
                const int64_t rounds = (blocks + PCACC_CUS - 1) / PCACC_CUS;
                // a tap of one tile per wave is bound by its weight tile's arrival, not by its 8 MFMAs: price it like 1.5 tiles
                const int64_t cost = rounds * (mt == 1 ? 3 : 2 * mt) * (cs == 64 ? 16 : 17);   // 32-channel slices: twice the barriers
                const int64_t waste = (int64_t)mg * mt * 32 * strips - (int64_t)h * w;
                if (!found || cost < best_cost || (cost == best_cost && mt > best->mt) ||
                    (cost == best_cost && mt == best->mt && waste < best_waste)) {
                    found = true;
                    best_cost = cost;
                    best_waste = waste;
                    *best = ConvStripPlan{cs, ng, mt, r, strips, c_out / (64 * ng), lds, blocks};
                }
                break;                                         // the widest slice that fits is the one to use for this (ng, mt)
            }
        }
    }
    return found;
}

template <int CS, int NG, int MT>
static int conv_strip_launch(const ConvStripPlan &p, const uint16_t *in, const uint16_t *in_mask, const uint16_t *wp, const float *bias,
                             uint16_t *out, int n_img, int h, int w, int c_in, int c_out, int relu, hipStream_t st)
{
    auto kern = in_mask ? conv3x3_strip_kernel<CS, NG, MT, true> : conv3x3_strip_kernel<CS, NG, MT, false>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    if (p.blocks > 0x7fffffff) return PCACC_E_ARG;
    hipLaunchKernelGGL(kern, dim3((unsigned)p.blocks), dim3(CD_THREADS), p.lds, st, in, in_mask, wp, bias, out, n_img, h, w, c_in, c_out, relu,
                       p.rows, p.strips, p.co_groups, (p.co_groups > 1 && !pcacc_switches().xcd_off) ? 1 : 0);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// Which layers take this kernel: deep K (c_in >= 128 in steps of 64), output channels in groups of 64, images narrow enough for a
// strip of >= 1 row.  conv.hip keeps the c_in <= 64 layers (weights resident in LDS) and the 3x3x3 stack.
extern "C" int pcacc_conv3x3_deep_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out)
{
    ConvStripPlan p;
    return conv_strip_plan(1, h, w, c_in, c_out, &p) ? 1 : 0;
}

extern "C" int pcacc_conv3x3_deep_bf16(const uint16_t *in, const uint16_t *in_mask, const uint16_t *wp, const float *bias, uint16_t *out,
                                       int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t relu, void *stream)
{
    ConvStripPlan p;
    if (!in || !wp || !out || n_img < 1 || !conv_strip_plan(n_img, h, w, c_in, c_out, &p)) return PCACC_E_ARG;
    if (pcacc_switches().conv_plan) fprintf(stderr, "conv plan %dx%d %d->%d n=%d: cs=%d ng=%d mt=%d rows=%d blocks=%lld\n", h, w, c_in, c_out, n_img, p.cs, p.ng, p.mt, p.rows, (long long)p.blocks);
    hipStream_t st = pcacc_stream(stream);
#define CD_CASE(CSV, NGV, MTV) \
    if (p.cs == CSV && p.ng == NGV && p.mt == MTV) \
        return conv_strip_launch<CSV, NGV, MTV>(p, in, in_mask, wp, bias, out, n_img, h, w, c_in, c_out, relu, st)
    CD_CASE(64, 2, 1); CD_CASE(64, 2, 2); CD_CASE(64, 2, 3); CD_CASE(64, 1, 1); CD_CASE(64, 1, 2);
    CD_CASE(32, 2, 1); CD_CASE(32, 2, 2); CD_CASE(32, 2, 3); CD_CASE(32, 1, 1); CD_CASE(32, 1, 2);
#undef CD_CASE
    return PCACC_E_ARG;
}

// ---- weight gradient of the deep layers ---------------------------------------------------------------------------------------------
// dW[co][tap][ci] = sum over images and pixels of dY[px][co] * X[px + tap offset][ci]: M = co, N = (tap, ci), K = pixels.  The
// reduction runs over pixels, so both MFMA operands are "8 consecutive pixels of one channel" per lane: the dY rows and the X patch of
// a strip are staged channels-last as they come and the fragments are read with the hardware LDS transpose (ds_read_b64_tr_b16,
// the scheme of conv3x3_wgrad_kernel in conv.hip; row stride C + 4 elements).  Here a workgroup owns one 64 x 64 (co, ci) block
// of the weight tensor and every `slots`-th strip: 8 waves = 4 (co tile, ci tile) pairs x 2 tap groups (taps 0-4 / 5-8), up to 5
// accumulators per wave, so a dY fragment is read once per 4-5 MFMAs and no wave shares an output element with another.  Pixels of
// a strip are consecutive in row-major order; a per-strip LDS table maps a pixel to the patch row of its top-left tap (no
// divisions in the loop).  One partial slot per workgroup goes to the workspace, a second launch sums the slots.
typedef short cd_s16x4 __attribute__((ext_vector_type(4)));
union cd_frag { bf16x8_t v; cd_s16x4 h[2]; };
#define CDW_MAXP 256                                       // pixels per strip
#define CDW_PCH 11                                         // staged 16-byte pieces (dY rows + X patch) a thread carries

__global__ __launch_bounds__(CD_THREADS) void conv3x3_wgrad_strip_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ dy_mask,
                                                                         const uint16_t *__restrict__ x,
                                                                         float *__restrict__ partial, int n_img, int h, int w, int c_in,
                                                                         int c_out, int rows, int strips, int ci_blocks, int slots, int sw, int segs, int xcd)
{
    // sw / segs: a strip is `rows` image rows x `sw` columns, `segs` of them side by side cover a row (sw = w, segs = 1 whenever a full-width row
    // fits the staging; wider maps -- the 288-wide ego feature head -- are cut into column segments, each with its own one-pixel halo)
    constexpr int CO = 64, CI = 64, PAIRS = 4, TG = 5;        // TG = taps of the first tap group
    constexpr int YS = pcacc_tr_stride(CO), XS = pcacc_tr_stride(CI);
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int pw = sw + 2, pp = (rows + 2) * pw;
    const int n_px = rows * sw, n_steps = (n_px + 15) >> 4, py_rows = n_steps * 16;
    uint16_t *sdy = lds;                                       // [py_rows][YS]   (rows >= n_px are zero)
    uint16_t *sx = sdy + (size_t)py_rows * YS;                 // [pp][XS]
    uint16_t *ptab = sx + (size_t)pp * XS;                     // [py_rows] patch row of the pixel's top-left tap

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int pair = wave % PAIRS, grp = wave / PAIRS;
    const int ct = pair >> 1, it = pair & 1;
    // xcd: the (co, ci) blocks of a strip next to each other on one XCD (block fastest in the logical order of common.h's walk): the dY and X rows that two
    // blocks share are fetched into that XCD's L2 once (PMC round 4: 2.66 x the algorithmic bytes with block-major order -- every block on every XCD)
    const int n_blocks = gridDim.x / slots, lb = xcd ? pcacc_xcd_block(blockIdx.x, gridDim.x) : 0;
    const int block = xcd ? lb % n_blocks : blockIdx.x / slots, slot = xcd ? lb / n_blocks : blockIdx.x % slots;
    const int co0 = (block / ci_blocks) * CO, ci0 = (block % ci_blocks) * CI;

    for (int q = threadIdx.x; q < py_rows; q += CD_THREADS) ptab[q] = q < n_px ? (uint16_t)((q / sw) * pw + q % sw) : 0;

    const int tap0 = grp * TG, n_tap = grp ? 9 - TG : TG;      // this wave's taps (wave-uniform)
    f32x16_t acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    // staged pieces of this thread: first the dY rows (py_rows * 8 pieces), then the X patch (pp * 8 pieces)
    const int y_chunks = py_rows * (CO / 8), n_chunks = y_chunks + pp * (CI / 8);
    uint4 sreg[CDW_PCH];
    auto fetch = [&](int job) {
        const int img = job / (strips * segs), sj = job % (strips * segs), y0 = (sj / segs) * rows, x0 = (sj % segs) * sw;
        const uint16_t *ysrc = dy + (int64_t)img * h * w * c_out + co0;
        const uint16_t *xsrc = x + (int64_t)img * h * w * c_in + ci0;
#pragma unroll
        for (int q = 0; q < CDW_PCH; ++q) {
            const int c = threadIdx.x + q * CD_THREADS;
            // one unconditional load per piece from a clamped address, zero selected afterwards (no lane-masked loads: they cost a
            // branch each and make the compiler wait for the data on the spot)
            const bool is_y = c < y_chunks;
            const int e = is_y ? c : min(c, n_chunks - 1) - y_chunks;
            const int px = e >> 3, c8 = e & 7;
            const int yy = is_y ? y0 + px / sw : y0 - 1 + px / pw;
            const int xx = x0 + (is_y ? px % sw : px % pw - 1);
            const bool ok = c < n_chunks && (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && (!is_y || px < n_px);
            const int64_t pos = (int64_t)min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1);
            const uint16_t *src = is_y ? ysrc + pos * c_out + c8 * 8 : xsrc + pos * c_in + c8 * 8;
            uint4 v = *reinterpret_cast<const uint4 *>(src);
            if (dy_mask && is_y) v = pcacc_relu_mask8(v, *reinterpret_cast<const uint4 *>(dy_mask + (src - dy)));
            if (!ok) v = make_uint4(0, 0, 0, 0);
            sreg[q] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < CDW_PCH; ++q) {
            const int c = threadIdx.x + q * CD_THREADS;
            if (c >= n_chunks) continue;
            uint2 *dst = c < y_chunks ? reinterpret_cast<uint2 *>(sdy + (c >> 3) * YS + (c & 7) * 8)
                                      : reinterpret_cast<uint2 *>(sx + ((c - y_chunks) >> 3) * XS + ((c - y_chunks) & 7) * 8);
            dst[0] = make_uint2(sreg[q].x, sreg[q].y);
            dst[1] = make_uint2(sreg[q].z, sreg[q].w);
        }
    };

    const int n_jobs = n_img * strips * segs;
    const int tg = lane >> 4, tl = lane & 15;
    const int tr_row = (tg >> 1) * 8 + (tl >> 2), tr_col = (tg & 1) * 16 + (tl & 3) * 4;
    int job = slot;
    if (job < n_jobs) fetch(job);
    while (job < n_jobs) {
        __syncthreads();                                       // the previous strip's fragment reads are done
        stage();
        __syncthreads();
        const int next = job + slots;
        if (next < n_jobs) fetch(next);
        job = next;
        // software pipeline: a step's dY fragment and patch-row indices are read during the previous step's MFMAs, a tap's X fragment
        // during the previous tap's MFMA (as first written every MFMA waited for the LDS reads issued right before it: 16 % matrix-pipe
        // utilisation at 256 x 256 channels on 36^2 images)
        int toff[TG];                                          // wave-uniform tap offsets inside the patch
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const int tap = min(tap0 + t, 8);
            toff[t] = ((tap / 3) * pw + tap % 3) * XS;
        }
        auto load_a = [&](int s, cd_frag &af, int &p0, int &p1) {
            const int r0 = s * 16 + tr_row;
            const uint16_t *pa = sdy + r0 * YS + ct * 32 + tr_col;
            af.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_s16x4 __attribute__((address_space(3))) *)pa);
            af.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_s16x4 __attribute__((address_space(3))) *)(pa + 4 * YS));
            p0 = ptab[r0]; p1 = ptab[r0 + 4];
        };
        auto load_b = [&](cd_frag &bf, const uint16_t *pb0, const uint16_t *pb1, int t) {
            bf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_s16x4 __attribute__((address_space(3))) *)(pb0 + toff[t]));
            bf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_s16x4 __attribute__((address_space(3))) *)(pb1 + toff[t]));
        };
        cd_frag af_n;
        int p0_n, p1_n;
        load_a(0, af_n, p0_n, p1_n);
        for (int s = 0; s < n_steps; ++s) {
            const cd_frag af = af_n;
            const uint16_t *pb0 = sx + p0_n * XS + it * 32 + tr_col, *pb1 = sx + p1_n * XS + it * 32 + tr_col;
            cd_frag bf[2];
            load_b(bf[0], pb0, pb1, 0);
            load_b(bf[1], pb0, pb1, 1);
            if (s + 1 < n_steps) load_a(s + 1, af_n, p0_n, p1_n);
            if (it == 0 && grp == 0) {                         // bias gradient = column sums of dY: the fragment is at hand
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bsum += bf16_to_f32((uint16_t)af.h[j][q]);
            }
#pragma unroll
            for (int t = 0; t < TG; ++t) {                     // n_tap is 4 or 5
                if (t < TG - 1 || n_tap == TG) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.v, bf[t & 1].v, acc[t], 0, 0, 0);
                if (t + 2 < TG - 1 || (t + 2 == TG - 1 && n_tap == TG)) load_b(bf[t & 1], pb0, pb1, t + 2);
            }
        }
    }
    // slot of this workgroup: [CO][9][CI] then [CO] bias sums; D has lane = ci, register quads = co
    float *mine = partial + ((int64_t)block * slots + slot) * (CO * 9 * CI + CO);
#pragma unroll
    for (int t = 0; t < TG; ++t)
        if (t < n_tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                mine[((int64_t)co * 9 + tap0 + t) * CI + it * 32 + lp] = acc[t][r];
            }
    if (grp != 0) return;
    bsum += __shfl_xor(bsum, 32, 64);                          // the two half-waves hold pixels 0-7 / 8-15 of the same channel
    if (it == 0 && lh == 0) mine[CO * 9 * CI + ct * 32 + lp] = bsum;
}

// out[co][tap][ci] (full tensor) and db[co] from the per-workgroup slots: one thread per output element, its block's slots summed
__global__ __launch_bounds__(256) void conv_wgrad_strip_reduce_kernel(const float *__restrict__ partial, int slots, int c_in, int c_out,
                                                                      int ci_blocks, float *__restrict__ dw, float *__restrict__ db)
{
    constexpr int CO = 64, CI = 64, SLOT = CO * 9 * CI + CO;
    const int64_t n_w = (int64_t)c_out * 9 * c_in;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n_w) {
        const int ci = (int)(e % c_in), tap = (int)((e / c_in) % 9), co = (int)(e / ((int64_t)9 * c_in));
        const int block = (co / CO) * ci_blocks + ci / CI;
        const float *src = partial + (int64_t)block * slots * SLOT + ((int64_t)(co % CO) * 9 + tap) * CI + ci % CI;
        float s0 = 0.f, s1 = 0.f;
        int p = 0;
        for (; p + 2 <= slots; p += 2) {
            s0 += src[(int64_t)p * SLOT];
            s1 += src[(int64_t)(p + 1) * SLOT];
        }
        if (p < slots) s0 += src[(int64_t)p * SLOT];
        dw[e] = s0 + s1;
    } else if (e < n_w + c_out) {
        const int co = (int)(e - n_w);
        const float *src = partial + (int64_t)((co / CO) * ci_blocks) * slots * SLOT + CO * 9 * CI + co % CO;   // ci block 0 carries the bias sums
        float s = 0.f;
        for (int p = 0; p < slots; ++p) s += src[(int64_t)p * SLOT];
        db[co] = s;
    }
}

// strip shape for the weight gradient: full-width rows (segs = 1), the height with the least padded work (strips x 16-pixel steps) among those
// whose staging fits; maps wider than a strip's pixel budget are cut into the fewest equal column segments (width a multiple of 16, so no
// padded step), as many rows of them as fit
static int conv_wgrad_strip_rows(int h, int w, size_t *lds_bytes, int *sw_out = nullptr, int *segs_out = nullptr)
{
    auto fits = [&](int rows, int sw, size_t *lds) {
        const int pp = (rows + 2) * (sw + 2), py_rows = (rows * sw + 15) / 16 * 16;
        *lds = ((size_t)py_rows * pcacc_tr_stride(64) + (size_t)pp * pcacc_tr_stride(64) + py_rows) * sizeof(uint16_t);
        return (py_rows + pp) * 8 <= CD_THREADS * CDW_PCH && *lds <= 150 * 1024;
    };
    int best = 0, sw = w, segs = 1;
    int64_t best_cost = 0;
    for (int rows = 1; rows <= h && rows * w <= CDW_MAXP; ++rows) {
        size_t lds;
        if (!fits(rows, w, &lds)) continue;
        const int64_t cost = (int64_t)((h + rows - 1) / rows) * ((rows * w + 15) / 16 * 16);
        if (!best || cost <= best_cost) { best = rows; best_cost = cost; *lds_bytes = lds; }
    }
    if (!best && w > CDW_MAXP) {
        for (int sg = (w + CDW_MAXP - 1) / CDW_MAXP; sg <= 8 && !best; ++sg) {
            const int width = ((w + sg - 1) / sg + 15) / 16 * 16;
            for (int rows = 1; rows <= h && rows * width <= CDW_MAXP; ++rows) {
                size_t lds;
                if (fits(rows, width, &lds)) { best = rows; *lds_bytes = lds; sw = width; segs = sg; }
            }
        }
    }
    if (sw_out) *sw_out = sw;
    if (segs_out) *segs_out = segs;
    return best;
}

static int conv_wgrad_strip_slots(int n_img, int strips, int blocks)
{
    int slots = (PCACC_CUS + blocks - 1) / blocks;             // one workgroup per CU (the staging LDS allows no more)
    const int jobs = n_img * strips;
    if (slots > jobs) slots = jobs;
    return slots < 1 ? 1 : slots;
}

extern "C" int pcacc_conv3x3_wgrad_deep_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out)
{
    size_t lds;
    return c_in >= 64 && c_out >= 64 && c_in % 64 == 0 && c_out % 64 == 0 && (c_in > 64 || c_out > 64) && h >= 1 && w >= 1 &&
           conv_wgrad_strip_rows(h, w, &lds) >= 1;
}

extern "C" int pcacc_conv3x3_wgrad_deep_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out, size_t *bytes)
{
    size_t lds;
    if (!bytes || n_img < 1 || !pcacc_conv3x3_wgrad_deep_supported(h, w, c_in, c_out)) return PCACC_E_ARG;
    int sw, segs;
    const int rows = conv_wgrad_strip_rows(h, w, &lds, &sw, &segs);
    const int blocks = (c_out / 64) * (c_in / 64);
    *bytes = (size_t)blocks * conv_wgrad_strip_slots(n_img, (h + rows - 1) / rows * segs, blocks) * (64 * 9 * 64 + 64) * sizeof(float);
    return 0;
}

extern "C" int pcacc_conv3x3_wgrad_deep_bf16(const uint16_t *dy, const uint16_t *dy_mask, const uint16_t *x, float *dw, float *db,
                                             int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out, void *workspace,
                                             size_t workspace_bytes, void *stream)
{
    size_t lds = 0, need = 0;
    if (!dy || !x || !dw || !db || !workspace || pcacc_conv3x3_wgrad_deep_workspace_bytes(n_img, h, w, c_in, c_out, &need) != 0)
        return PCACC_E_ARG;
    if (workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t st = pcacc_stream(stream);
    int sw, segs;
    const int rows = conv_wgrad_strip_rows(h, w, &lds, &sw, &segs);
    const int strips = (h + rows - 1) / rows, ci_blocks = c_in / 64, blocks = (c_out / 64) * ci_blocks;
    const int slots = conv_wgrad_strip_slots(n_img, strips * segs, blocks);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wgrad_strip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(conv3x3_wgrad_strip_kernel, dim3(blocks * slots), dim3(CD_THREADS), lds, st, dy, dy_mask, x, partial, n_img, h, w, c_in, c_out,
                       rows, strips, ci_blocks, slots, sw, segs, (blocks > 1 && !pcacc_switches().xcd_off) ? 1 : 0);
    const int64_t elems = (int64_t)c_out * 9 * c_in + c_out;
    hipLaunchKernelGGL(conv_wgrad_strip_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, partial, slots, c_in, c_out,
                       ci_blocks, dw, db);
    PCACC_CHECK_LAUNCH();
    return 0;
}
