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

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CD_THREADS 512
#define CD_PCH 10                              // patch pieces (16 B) a thread carries per channel slice

template <int CS, int NG>
__global__ __launch_bounds__(CD_THREADS) void conv3x3_strip_kernel(const uint16_t *__restrict__ in, const uint16_t *__restrict__ wp,
                                                                   const float *__restrict__ bias, uint16_t *__restrict__ out, int n_img,
                                                                   int h, int w, int c_in, int c_out, int relu, int rows, int strips,
                                                                   int co_groups)
{
    constexpr int PS = CS + 8;                                 // padded LDS row (elements): conflict-free 16-byte fragment reads
    constexpr int MG = NG == 2 ? 4 : 8;                        // waves along the pixel dimension
    constexpr int MT = NG == 2 ? 3 : 2;                        // 32-pixel tiles per wave
    constexpr int WROWS = NG * 64;                             // weight rows (output channels) per workgroup
    constexpr int C8 = CS / 8;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int pw = w + 2, pp = (rows + 2) * pw;                // patch width / pixels
    uint16_t *patch = lds;                                     // [pp][PS]
    uint16_t *wbuf = lds + (size_t)pp * PS;                    // [2][WROWS][PS]

    int bid = blockIdx.x;
    const int cog = bid % co_groups; bid /= co_groups;
    const int strip = bid % strips;
    const int img = bid / strips;
    const int y0 = strip * rows;
    const int co0 = cog * WROWS;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int mg = wave % MG, ng = wave / MG;
    const int n_px = rows * w;
    const int n_mt = (n_px + 31) >> 5;

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
    int mt_count = 0;                                          // wave-uniform: tiles of this wave that exist in the strip
#pragma unroll
    for (int j = 0; j < MT; ++j) mt_count += (mg + MG * j) < n_mt;

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
    auto fetch_patch = [&](int cs) {
#pragma unroll
        for (int q = 0; q < CD_PCH; ++q) {
            const int py = pinfo[q] >> 20, pxx = (pinfo[q] >> 8) & 0xfff, c8 = pinfo[q] & 0xff;
            const int y = y0 - 1 + py, x = pxx - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w)
                v = *reinterpret_cast<const uint4 *>(img_in + ((int64_t)y * w + x) * c_in + cs * CS + c8 * 8);
            preg[q] = v;
        }
    };
    auto write_patch = [&]() {
#pragma unroll
        for (int q = 0; q < CD_PCH; ++q) {
            const int c = threadIdx.x + q * CD_THREADS;
            if (c < n_chunks) *reinterpret_cast<uint4 *>(patch + (c / C8) * PS + (c % C8) * 8) = preg[q];
        }
    };
    // weight tile of (tap, slice): WROWS rows of CS elements = WROWS * C8 pieces, <= 2 per thread
    constexpr int W_CHUNKS = WROWS * C8, W_PER = (W_CHUNKS + CD_THREADS - 1) / CD_THREADS;
    uint4 wreg[W_PER];
    auto fetch_w = [&](int tap, int cs) {
        const uint16_t *src = wp + ((int64_t)tap * c_out + co0) * c_in + cs * CS;
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = threadIdx.x + q * CD_THREADS;
            if (c < W_CHUNKS) wreg[q] = *reinterpret_cast<const uint4 *>(src + (int64_t)(c / C8) * c_in + (c % C8) * 8);
        }
    };
    auto write_w = [&](int buf) {
        uint16_t *dst = wbuf + buf * WROWS * PS;
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = threadIdx.x + q * CD_THREADS;
            if (c < W_CHUNKS) *reinterpret_cast<uint4 *>(dst + (c / C8) * PS + (c % C8) * 8) = wreg[q];
        }
    };

    const int n_slices = c_in / CS;
    fetch_patch(0);
    fetch_w(0, 0);
    int buf = 0;
    for (int cs = 0; cs < n_slices; ++cs) {
        __syncthreads();                                       // every wave is done with the previous slice's patch and weight tiles
        write_patch();
        write_w(buf);
        if (cs + 1 < n_slices) fetch_patch(cs + 1);            // in flight during the nine taps below
        for (int tap = 0; tap < 9; ++tap) {
            // the next weight tile: tap + 1 of this slice, or tap 0 of the next one
            const bool more = tap < 8 || cs + 1 < n_slices;
            if (more) fetch_w(tap < 8 ? tap + 1 : 0, tap < 8 ? cs : cs + 1);
            __syncthreads();                                   // buffer `buf` (and, at tap 0, the patch) is visible
            const uint16_t *wb = wbuf + buf * WROWS * PS + (ng * 64 + lp) * PS + lh * 8;
            const int toff = ((tap / 3) * pw + tap % 3) * PS + lh * 8;
#pragma unroll
            for (int kc = 0; kc < CS / 16; ++kc) {
                bf16x8_t a[2], b[MT];
#pragma unroll
                for (int n = 0; n < 2; ++n) a[n] = *reinterpret_cast<const bf16x8_t *>(wb + n * 32 * PS + kc * 16);
#pragma unroll
                for (int j = 0; j < MT; ++j) b[j] = *reinterpret_cast<const bf16x8_t *>(patch + poff[j] + toff + kc * 16);
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    if (j < mt_count) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n], b[j], acc[j][n], 0, 0, 0);
                    }
            }
            if (tap < 8) write_w(buf ^ 1);                     // the other buffer: its last readers passed the barrier above
            if (tap < 8) buf ^= 1;
        }
        // tap 0 of the next slice is in wreg; it is written after the barrier at the top of the loop (the patch changes there too)
        buf ^= 1;
    }

    // epilogue: lane = pixel, register quad g of tile n = channels n*32 + 8g + 4*lh .. +3 of this wave's 64
    const float *bptr = bias ? bias + co0 + ng * 64 : nullptr;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        if (pyx[j] < 0) continue;
        uint16_t *dst = out + (((int64_t)img * h + (pyx[j] >> 16)) * w + (pyx[j] & 0xffff)) * c_out + co0 + ng * 64;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n * 32 + 8 * g + 4 * lh;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bptr) bv = *reinterpret_cast<const float4 *>(bptr + c);
                float v[4] = {acc[j][n][4 * g] + bv.x, acc[j][n][4 * g + 1] + bv.y, acc[j][n][4 * g + 2] + bv.z, acc[j][n][4 * g + 3] + bv.w};
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                *reinterpret_cast<uint2 *>(dst + c) = make_uint2(pcacc_pack_bf16x2(v[0], v[1]), pcacc_pack_bf16x2(v[2], v[3]));
            }
    }
}

// rows of the strip for an image width: as many as the wave layout covers (32 * MT * MG pixels), the thread-carried patch pieces
// allow, and LDS holds next to the two weight buffers
template <int CS, int NG>
static int conv_strip_rows(int h, int w, size_t *lds_bytes)
{
    constexpr int PS = CS + 8;
    const int max_px = NG == 2 ? 384 : 512;
    int rows = max_px / w;
    if (rows > h) rows = h;
    while (rows >= 1) {
        const int pp = (rows + 2) * (w + 2);
        const size_t lds = ((size_t)pp + 2 * NG * 64) * PS * sizeof(uint16_t);
        if (pp * (CS / 8) <= CD_THREADS * CD_PCH && lds <= 160 * 1024) {
            *lds_bytes = lds;
            return rows;
        }
        --rows;
    }
    return 0;
}

template <int CS, int NG>
static int conv_strip_launch(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int n_img, int h, int w, int c_in,
                             int c_out, int relu, hipStream_t st)
{
    size_t lds = 0;
    const int rows = conv_strip_rows<CS, NG>(h, w, &lds);
    if (rows < 1) return PCACC_E_ARG;
    const int strips = (h + rows - 1) / rows, co_groups = c_out / (NG * 64);
    auto kern = conv3x3_strip_kernel<CS, NG>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    const int64_t blocks = (int64_t)n_img * strips * co_groups;
    if (blocks > 0x7fffffff) return PCACC_E_ARG;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(CD_THREADS), lds, st, in, wp, bias, out, n_img, h, w, c_in, c_out, relu, rows,
                       strips, co_groups);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// Which layers take this kernel: deep K (c_in >= 128 in steps of 64), output channels in groups of 64, images narrow enough for a
// strip of >= 1 row (W <= 510).  conv.hip keeps the c_in <= 64 layers (weights resident in LDS) and the 3x3x3 stack.
extern "C" int pcacc_conv3x3_deep_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out)
{
    size_t lds;
    if (c_in < 128 || c_in % 64 || c_out < 64 || c_out % 64 || h < 1 || w < 1) return 0;
    if (w <= 384 / 2) return (c_out % 128 == 0 ? conv_strip_rows<64, 2>(h, w, &lds) : conv_strip_rows<64, 1>(h, w, &lds)) >= 1;
    return (c_out % 128 == 0 ? conv_strip_rows<32, 2>(h, w, &lds) : conv_strip_rows<32, 1>(h, w, &lds)) >= 1;
}

extern "C" int pcacc_conv3x3_deep_bf16(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int32_t n_img, int32_t h,
                                       int32_t w, int32_t c_in, int32_t c_out, int32_t relu, void *stream)
{
    if (!in || !wp || !out || n_img < 1 || !pcacc_conv3x3_deep_supported(h, w, c_in, c_out)) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    const bool wide = w > 384 / 2;                             // one-row strips of wide images: 32-channel slices keep the patch in LDS
    if (c_out % 128 == 0) {
        // few, large workgroups leave CUs idle on the smallest layers: halve the channel group when that fills more of the chip
        size_t lds;
        const int rows = wide ? conv_strip_rows<32, 2>(h, w, &lds) : conv_strip_rows<64, 2>(h, w, &lds);
        const int64_t blocks = (int64_t)n_img * ((h + rows - 1) / rows) * (c_out / 128);
        if (blocks * 2 <= PCACC_CUS)
            return wide ? conv_strip_launch<32, 1>(in, wp, bias, out, n_img, h, w, c_in, c_out, relu, st)
                        : conv_strip_launch<64, 1>(in, wp, bias, out, n_img, h, w, c_in, c_out, relu, st);
        return wide ? conv_strip_launch<32, 2>(in, wp, bias, out, n_img, h, w, c_in, c_out, relu, st)
                    : conv_strip_launch<64, 2>(in, wp, bias, out, n_img, h, w, c_in, c_out, relu, st);
    }
    return wide ? conv_strip_launch<32, 1>(in, wp, bias, out, n_img, h, w, c_in, c_out, relu, st)
                : conv_strip_launch<64, 1>(in, wp, bias, out, n_img, h, w, c_in, c_out, relu, st);
}
