// 3x3 (and 3x3x3 over frames) convolution, stride 1, zero padding 1, as an implicit GEMM on the bf16 matrix cores:
// channels-last bf16 in, fp32 accumulate, bias + optional ReLU in the epilogue, channels-last bf16 out.
// Covers the conv + bias + ReLU layers of models/unet.py:15-27,45-113 (DownConv / UpConv / conv_final), the STPN
// backbone (models/stpn.py:24-43) and the four Conv3d(3x3x3) + ReLU of models/stpn.py:13-22, which are evaluated
// directly on the [B,T,H,W,C] rows (frame t reads frames t-1, t, t+1; no channel-stacked copy).
//
// GEMM view: D[c_out][pixel] = sum_k W[c_out][k] * X[k][pixel], k = (frame tap, 3x3 tap, c_in).
//   v_mfma_f32_32x32x16_bf16: A = 32 output channels x 16 k, B = 16 k x 32 pixels of one output row.
//   D lands as: lane&31 = pixel, 4 consecutive channels per register quad -> 8-byte channels-last stores.
// Workgroup = 4 waves = an 8 x 32 pixel tile of one image; wave w owns rows 2w, 2w+1 and all output channels of the
// launch group (<= 128).  The input patch (10 x 34 pixels x up to 128 channels) sits in LDS for all 9 taps; the
// weights of one tap are staged through LDS (prefetched into registers during the previous tap).  Rows are padded by
// 8 elements so that the 16-byte fragment reads of 16 consecutive pixels / channels hit 64 distinct banks.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CV_TH 8
#define CV_TW 32
#define CV_PW (CV_TW + 2)
#define CV_PH (CV_TH + 2)
#define CV_THREADS 256

// ---- weight preparation: fp32 [O][I][KT][3][3] (torch contiguous layout) -> bf16 [KT*9][O'][I'] -------------------------
// transpose = 0: forward weights, O' = O, I' = I.
// transpose = 1: weights of the data gradient, O' = I, I' = O, taps and frame taps mirrored.
__global__ __launch_bounds__(256) void conv_prepare_weights_kernel(const float *__restrict__ w, int o, int i, int kt, int transpose,
                                                                   uint16_t *__restrict__ out)
{
    const int taps = kt * 9;
    const int64_t total = (int64_t)taps * o * i;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int op = transpose ? i : o, ip = transpose ? o : i;
        const int ci = (int)(e % ip);
        const int co = (int)((e / ip) % op);
        const int tap = (int)(e / ((int64_t)ip * op));
        const int src_tap = transpose ? (taps - 1 - tap) : tap;
        const int so = transpose ? ci : co, si = transpose ? co : ci;
        out[e] = f32_to_bf16(w[((int64_t)so * i + si) * taps + src_tap]);
    }
}

extern "C" int pcacc_conv3x3_prepare_weights(const float *w, int32_t c_out, int32_t c_in, int32_t kt, int32_t transpose,
                                             uint16_t *out, void *stream)
{
    if (!w || !out || c_out < 1 || c_in < 1 || (kt != 1 && kt != 3)) return PCACC_E_ARG;
    const int64_t total = (int64_t)kt * 9 * c_out * c_in;
    hipLaunchKernelGGL(conv_prepare_weights_kernel, dim3(pcacc_grid(total, 256)), dim3(256), 0, pcacc_stream(stream), w, c_out, c_in,
                       kt, transpose, out);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// ---- the convolution ------------------------------------------------------------------------------------------------------------
// CT = output-channel tiles of 32 per workgroup (1, 2 or 4); CS = input channels resident in LDS at a time (32, 64, 128).
template <int CT, int CS>
__global__ __launch_bounds__(CV_THREADS) void conv3x3_mfma_kernel(const uint16_t *__restrict__ in, const uint16_t *__restrict__ wp,
                                                                  const float *__restrict__ bias, uint16_t *__restrict__ out,
                                                                  int n_img, int frames, int h, int w, int c_in, int c_out, int kt,
                                                                  int relu, int tiles_x, int tiles_y, int co_groups)
{
    constexpr int PS = CS + 8;                                 // padded row length (elements) of both LDS images
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *patch = lds;                                     // [CV_PH * CV_PW][PS]
    uint16_t *wl = lds + CV_PH * CV_PW * PS;                   // [CT * 32][PS]

    int bid = blockIdx.x;
    const int cog = bid % co_groups; bid /= co_groups;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int x0 = tx * CV_TW, y0 = ty * CV_TH;
    const int co0 = cog * CT * 32;
    const int t_frame = img % frames;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;

    f32x16_t acc[2][CT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][ct][r] = 0.f;

    constexpr int W_CHUNKS = CT * 32 * CS / 8;                 // 16-byte pieces of one tap's weights
    constexpr int W_PER_THREAD = (W_CHUNKS + CV_THREADS - 1) / CV_THREADS;
    constexpr int P_CHUNKS = CV_PH * CV_PW * CS / 8;
    uint4 wreg[W_PER_THREAD];

    const int n_slices = c_in / CS;
    for (int f = 0; f < kt; ++f) {
        const int dt = kt == 3 ? f - 1 : 0;
        if (t_frame + dt < 0 || t_frame + dt >= frames) continue;               // uniform: a missing frame contributes zeros
        const uint16_t *src = in + (int64_t)(img + dt) * h * w * c_in;
        for (int cs = 0; cs < n_slices; ++cs) {
            __syncthreads();                                                     // the previous pass is done with the patch
            for (int c = threadIdx.x; c < P_CHUNKS; c += CV_THREADS) {
                const int px = c / (CS / 8), c8 = c % (CS / 8);
                const int py = px / CV_PW, pxx = px % CV_PW;
                const int y = y0 - 1 + py, x = x0 - 1 + pxx;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (y >= 0 && y < h && x >= 0 && x < w)
                    v = *reinterpret_cast<const uint4 *>(src + ((int64_t)y * w + x) * c_in + cs * CS + c8 * 8);
                *reinterpret_cast<uint4 *>(patch + px * PS + c8 * 8) = v;
            }
            const uint16_t *wsrc = wp + ((int64_t)f * 9 * c_out + co0) * c_in + cs * CS;   // tap 0 of this frame tap
#pragma unroll
            for (int j = 0; j < W_PER_THREAD; ++j) {
                const int c = threadIdx.x + j * CV_THREADS;
                if (c < W_CHUNKS) wreg[j] = *reinterpret_cast<const uint4 *>(wsrc + (int64_t)(c / (CS / 8)) * c_in + (c % (CS / 8)) * 8);
            }
            for (int tap = 0; tap < 9; ++tap) {
                __syncthreads();                                                 // previous tap's fragment reads are done
#pragma unroll
                for (int j = 0; j < W_PER_THREAD; ++j) {
                    const int c = threadIdx.x + j * CV_THREADS;
                    if (c < W_CHUNKS) *reinterpret_cast<uint4 *>(wl + (c / (CS / 8)) * PS + (c % (CS / 8)) * 8) = wreg[j];
                }
                if (tap < 8) {
                    const uint16_t *wnext = wsrc + (int64_t)(tap + 1) * c_out * c_in;
#pragma unroll
                    for (int j = 0; j < W_PER_THREAD; ++j) {
                        const int c = threadIdx.x + j * CV_THREADS;
                        if (c < W_CHUNKS)
                            wreg[j] = *reinterpret_cast<const uint4 *>(wnext + (int64_t)(c / (CS / 8)) * c_in + (c % (CS / 8)) * 8);
                    }
                }
                __syncthreads();                                                 // weights (and, at tap 0, the patch) visible
                const int dy = tap / 3, dx = tap % 3;
                const uint16_t *prow0 = patch + ((2 * wave + dy) * CV_PW + lp + dx) * PS + lh * 8;
                const uint16_t *prow1 = prow0 + CV_PW * PS;
                const uint16_t *wrow = wl + lp * PS + lh * 8;
#pragma unroll
                for (int kc = 0; kc < CS / 16; ++kc) {
                    const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t *>(prow0 + kc * 16);
                    const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t *>(prow1 + kc * 16);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const bf16x8_t a = *reinterpret_cast<const bf16x8_t *>(wrow + ct * 32 * PS + kc * 16);
                        acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0][ct], 0, 0, 0);
                        acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1][ct], 0, 0, 0);
                    }
                }
            }
        }
    }

    // epilogue: lane = pixel lp of rows 2*wave + m; register quad g holds channels ct*32 + 8g + 4*lh .. +3
    const int x = x0 + lp;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int y = y0 + 2 * wave + m;
        if (y >= h || x >= w) continue;
        uint16_t *dst = out + (((int64_t)img * h + y) * w + x) * c_out + co0;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = ct * 32 + 8 * g + 4 * lh;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = acc[m][ct][4 * g + q] + (bias ? bias[co0 + c + q] : 0.f);
                    if (relu) v[q] = v[q] > 0.f ? v[q] : 0.f;
                }
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<uint2 *>(dst + c) = pk;
            }
        }
    }
}

template <int CT, int CS>
static int conv_launch(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int n_img, int frames, int h, int w,
                       int c_in, int c_out, int kt, int relu, hipStream_t st)
{
    const int tiles_x = (w + CV_TW - 1) / CV_TW, tiles_y = (h + CV_TH - 1) / CV_TH;
    const int co_groups = c_out / (CT * 32);
    const size_t lds = (size_t)(CV_PH * CV_PW + CT * 32) * (CS + 8) * sizeof(uint16_t);
    auto kern = conv3x3_mfma_kernel<CT, CS>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PCACC_E_LAUNCH;
    }
    const int64_t blocks = (int64_t)n_img * tiles_y * tiles_x * co_groups;
    if (blocks > 0x7fffffff) return PCACC_E_ARG;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(CV_THREADS), lds, st, in, wp, bias, out, n_img, frames, h, w, c_in, c_out, kt,
                       relu, tiles_x, tiles_y, co_groups);
    PCACC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcacc_conv3x3_bf16(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int32_t n_img,
                                  int32_t frames, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt, int32_t relu,
                                  void *stream)
{
    if (!in || !wp || !out || n_img < 1 || h < 1 || w < 1 || (kt != 1 && kt != 3) || frames < 1 || n_img % frames) return PCACC_E_ARG;
    if (c_in % 32 || c_out % 32 || c_in < 32 || c_out < 32) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    const int cs_sel = c_in % 128 == 0 ? 128 : (c_in % 64 == 0 ? 64 : 32);   // input channels per LDS pass
    const int ct = c_out % 128 == 0 ? 4 : (c_out % 64 == 0 ? 2 : 1);
#define CV_CASE(CTV, CSV) \
    if (ct == CTV && cs_sel == CSV) return conv_launch<CTV, CSV>(in, wp, bias, out, n_img, frames, h, w, c_in, c_out, kt, relu, st)
    CV_CASE(1, 32); CV_CASE(2, 32); CV_CASE(4, 32);
    CV_CASE(1, 64); CV_CASE(2, 64); CV_CASE(4, 64);
    CV_CASE(1, 128); CV_CASE(2, 128); CV_CASE(4, 128);
#undef CV_CASE
    return PCACC_E_ARG;
}
