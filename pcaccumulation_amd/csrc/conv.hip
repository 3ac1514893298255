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

// Both prepared forms (forward, and mirrored / transposed for the data gradient) of one fp32 weight tensor in ONE launch, read through its
// strides (contiguous or channels-last storage alike): a training step needs both, and the weight only changes at the optimizer step.
struct ConvWStrides { int64_t o, i, t, y, x; };

__global__ __launch_bounds__(256) void conv_prepare_weights_pair_kernel(const float *__restrict__ w, int o, int i, int kt, ConvWStrides st,
                                                                        uint16_t *__restrict__ out_fwd, uint16_t *__restrict__ out_bwd)
{
    const int taps = kt * 9;
    const int64_t total = (int64_t)taps * o * i;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < 2 * total; e += (int64_t)gridDim.x * 256) {
        const bool transpose = e >= total;
        const int64_t r = transpose ? e - total : e;
        const int op = transpose ? i : o, ip = transpose ? o : i;
        const int ci = (int)(r % ip);
        const int co = (int)((r / ip) % op);
        const int tap = (int)(r / ((int64_t)ip * op));
        const int src_tap = transpose ? (taps - 1 - tap) : tap;
        const int so = transpose ? ci : co, si = transpose ? co : ci;
        const int ft = src_tap / 9, fy = (src_tap % 9) / 3, fx = src_tap % 3;
        const uint16_t v = f32_to_bf16(w[so * st.o + si * st.i + ft * st.t + fy * st.y + fx * st.x]);
        (transpose ? out_bwd : out_fwd)[r] = v;
    }
}

extern "C" int pcacc_conv3x3_prepare_weights_pair(const float *w, int32_t c_out, int32_t c_in, int32_t kt, const int64_t *strides,
                                                  uint16_t *out_fwd, uint16_t *out_bwd, void *stream)
{
    if (!w || !out_fwd || !out_bwd || !strides || c_out < 1 || c_in < 1 || (kt != 1 && kt != 3)) return PCACC_E_ARG;
    const ConvWStrides st = {strides[0], strides[1], kt == 3 ? strides[2] : 0, strides[kt == 3 ? 3 : 2], strides[kt == 3 ? 4 : 3]};
    const int64_t total = 2 * (int64_t)kt * 9 * c_out * c_in;
    hipLaunchKernelGGL(conv_prepare_weights_pair_kernel, dim3(pcacc_grid(total, 256)), dim3(256), 0, pcacc_stream(stream), w, c_out, c_in, kt,
                       st, out_fwd, out_bwd);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// epilogue: lane = pixel lp of the wave's R rows; register quad g holds channels ct*32 + 8g + 4*lh .. +3.
// Split into pack (bias, ReLU, bf16) and store so that the persistent kernel can hold a finished tile in registers and
// issue its stores one pass later (loads and stores share the vmcnt counter: a wait for the prefetched patch would
// otherwise also wait for stores issued just before it).  `bias` points at the wave's first channel (or is NULL).
template <int R, int CT>
__device__ __forceinline__ void conv_pack_tile(const f32x16_t (&acc)[R][CT], const float *__restrict__ bias, int relu, int lh,
                                               uint2 (&pk)[R][CT][4])
{
#pragma unroll
    for (int m = 0; m < R; ++m)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = ct * 32 + 8 * g + 4 * lh;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias) bv = *reinterpret_cast<const float4 *>(bias + c);
                float v[4] = {acc[m][ct][4 * g] + bv.x, acc[m][ct][4 * g + 1] + bv.y, acc[m][ct][4 * g + 2] + bv.z,
                              acc[m][ct][4 * g + 3] + bv.w};
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                pk[m][ct][g].x = pcacc_pack_bf16x2(v[0], v[1]);
                pk[m][ct][g].y = pcacc_pack_bf16x2(v[2], v[3]);
            }
}

// `row0` = first of the wave's R rows inside the tile, `co` = its first output channel
// aten::threshold_backward(v, m, 0) on packed bf16 pairs: a half of v is dropped where the same half of m is <= 0 (zeros of either sign,
// negative numbers, -inf); NaN compares false and keeps it, as the library does
__device__ __forceinline__ uint32_t conv_mask2(uint32_t v, uint32_t m)
{
    auto drop = [](uint32_t h) { const uint32_t mag = h & 0x7fffu; return mag == 0u || ((h & 0x8000u) && mag <= 0x7f80u); };
    const uint32_t lo = drop(m & 0xffffu) ? 0u : 0xffffu;
    const uint32_t hi = drop(m >> 16) ? 0u : 0xffff0000u;
    return v & (lo | hi);
}

// omask (may be NULL): a map of the output's shape; results are stored as zero where it is <= 0 -- the data gradient of a layer whose input
// was a ReLU output leaves already masked for that ReLU (the producer then needs no threshold pass of its own)
// A lane holds 4 channels (8 bytes) of each of the four 8-channel groups of its pixel, lane + 32 the other 4: stored as they are, every
// global_store_dwordx2 of a wave writes 16 bytes into each of 32 different 64-byte lines, four instructions per row.  Lanes l and l + 32
// exchange halves first (v_permlane32_swap: the upper half-wave's group 2j against the lower half-wave's group 2j + 1), after which lane l
// holds channels 16j .. 16j + 7 and lane l + 32 channels 16j + 8 .. + 15: two dwordx4 stores per row, 32 contiguous bytes per pixel each.
__device__ __forceinline__ uint4 conv_pair16(uint2 g_even, uint2 g_odd)
{
    typedef unsigned cv_u2 __attribute__((ext_vector_type(2)));
    const cv_u2 x = __builtin_amdgcn_permlane32_swap(g_even.x, g_odd.x, false, false);
    const cv_u2 y = __builtin_amdgcn_permlane32_swap(g_even.y, g_odd.y, false, false);
    return make_uint4(x[0], y[0], x[1], y[1]);
}

template <int R, int CT>
__device__ __forceinline__ void conv_store_packed(const uint2 (&pk)[R][CT][4], uint16_t *__restrict__ out, int img, int y0, int x0, int h,
                                                  int w, int c_out, int co, int row0, int lp, int lh, const uint16_t *__restrict__ omask = nullptr)
{
    const int x = x0 + lp;
#pragma unroll
    for (int m = 0; m < R; ++m) {
        const int y = y0 + row0 + m;
        uint4 v[CT][2];                                        // the exchange involves both half-waves: before any lane drops out
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 2; ++j) v[ct][j] = conv_pair16(pk[m][ct][2 * j], pk[m][ct][2 * j + 1]);
        if (y >= h || x >= w) continue;
        const int64_t off = (((int64_t)img * h + y) * w + x) * c_out + co + 8 * lh;
        uint16_t *dst = out + off;
        if (omask) {
            uint4 mk[CT][2];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 2; ++j) mk[ct][j] = *reinterpret_cast<const uint4 *>(omask + off + ct * 32 + 16 * j);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<uint4 *>(dst + ct * 32 + 16 * j) =
                        make_uint4(conv_mask2(v[ct][j].x, mk[ct][j].x), conv_mask2(v[ct][j].y, mk[ct][j].y), conv_mask2(v[ct][j].z, mk[ct][j].z),
                                   conv_mask2(v[ct][j].w, mk[ct][j].w));
            continue;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<uint4 *>(dst + ct * 32 + 16 * j) = v[ct][j];
    }
}

template <int CT>
__device__ __forceinline__ void conv_store_tile(const f32x16_t (&acc)[2][CT], const float *__restrict__ bias, uint16_t *__restrict__ out,
                                                int img, int y0, int x0, int h, int w, int c_out, int co0, int relu, int wave, int lp,
                                                int lh, const uint16_t *__restrict__ omask = nullptr)
{
    uint2 pk[2][CT][4];
    conv_pack_tile<2, CT>(acc, bias ? bias + co0 : nullptr, relu, lh, pk);
    conv_store_packed<2, CT>(pk, out, img, y0, x0, h, w, c_out, co0, 2 * wave, lp, lh, omask);
}

// ---- the convolution ------------------------------------------------------------------------------------------------------------
// CT = output-channel tiles of 32 per workgroup (1, 2 or 4); CS = input channels resident in LDS at a time (32, 64, 128).
template <int CT, int CS>
__global__ __launch_bounds__(CV_THREADS) void conv3x3_mfma_kernel(const uint16_t *__restrict__ in, const uint16_t *__restrict__ wp,
                                                                  const float *__restrict__ bias, uint16_t *__restrict__ out,
                                                                  int n_img, int frames, int h, int w, int c_in, int c_out, int kt,
                                                                  int relu, int tiles_x, int tiles_y, int co_groups,
                                                                  const uint16_t *__restrict__ omask)
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

    conv_store_tile<CT>(acc, bias, out, img, y0, x0, h, w, c_out, co0, relu, wave, lp, lh, omask);
}

// All (tap, 16-channel) steps of one pass over a patch, as one straight line of code with the LDS fragment reads issued
// two steps ahead of the MFMAs that consume them.  `pbase` = patch + (row0*CV_PW + lp)*PS + lh*8; `wbase` = the wave's
// first weight row of tap 0 + lp*PS + lh*8; consecutive taps are TAP_ROWS rows apart.
template <int R, int CT, int CS, int TAP_ROWS>
__device__ __forceinline__ void conv_pass_mfma(f32x16_t (&acc)[R][CT], const uint16_t *pbase, const uint16_t *wbase)
{
    constexpr int PS = CS + 8;
    constexpr int KC = CS / 16;
    constexpr int STEPS = 9 * KC;
    constexpr int AHEAD = 2;
    bf16x8_t fb[AHEAD + 1][R], fa[AHEAD + 1][CT];
    auto load = [&](int slot, int s) {
        const int tap = s / KC, kc = s % KC;
        const uint16_t *p = pbase + ((tap / 3) * CV_PW + tap % 3) * PS + kc * 16;
#pragma unroll
        for (int m = 0; m < R; ++m) fb[slot][m] = *reinterpret_cast<const bf16x8_t *>(p + m * CV_PW * PS);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) fa[slot][ct] = *reinterpret_cast<const bf16x8_t *>(wbase + (tap * TAP_ROWS + ct * 32) * PS + kc * 16);
    };
#pragma unroll
    for (int s = 0; s < AHEAD && s < STEPS; ++s) load(s, s);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + AHEAD < STEPS) load((s + AHEAD) % (AHEAD + 1), s + AHEAD);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int m = 0; m < R; ++m)
                acc[m][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % (AHEAD + 1)][ct], fb[s % (AHEAD + 1)][m], acc[m][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same pass for 32-channel rows stored UNPADDED (64 B per pixel / weight row) with the 16-byte chunk index XOR-ed by (row >> 2) & 3:
// a ds_read_b128 lane group holds rows {o..o+3, o+12..o+15, o+20..o+27}; the four of them that share row % 4 (hence the 16-byte slot
// group row * 4 % 16) differ in (row >> 2) & 3 for every offset o, so the reads stay conflict-free like the 80-byte rows', and the 27-tap
// layer's weights + patch take 77 KB instead of 96: two workgroups per CU.  `lrow` = the lane's patch row for tap (0,0) of the wave's
// first pixel row; `wtap0` = the lane's weight row of tap 0; `wsw` = its swizzled element offset of channel chunk lh.
template <int R>
__device__ __forceinline__ void conv_pass_mfma_swz(f32x16_t (&acc)[R][1], const uint16_t *patch, int lrow, int lh, const uint16_t *wtap0, int wsw)
{
    constexpr int STEPS = 18;
    constexpr int AHEAD = 2;
    bf16x8_t fb[AHEAD + 1][R], fa[AHEAD + 1];
    auto load = [&](int slot, int s) {
        const int tap = s / 2, kc = s % 2;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int v = lrow + (tap / 3 + m) * CV_PW + tap % 3;
            fb[slot][m] = *reinterpret_cast<const bf16x8_t *>(patch + v * 32 + (((kc * 2 + lh) ^ ((v >> 2) & 3)) << 3));
        }
        fa[slot] = *reinterpret_cast<const bf16x8_t *>(wtap0 + tap * 32 * 32 + (wsw ^ (kc * 16)));
    };
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) load(s, s);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + AHEAD < STEPS) load((s + AHEAD) % (AHEAD + 1), s + AHEAD);
#pragma unroll
        for (int m = 0; m < R; ++m)
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % (AHEAD + 1)], fb[s % (AHEAD + 1)][m], acc[m][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- small layers (c_in <= 64): all taps' weights resident in LDS, persistent workgroups, patch prefetch -------------------
// One pass = one (tile, frame tap): the patch of the NEXT pass is fetched into registers while the matrix cores work on
// the current one, so global latency is paid once per workgroup instead of once per tap.  Workgroups are dealt to XCDs
// round-robin by the hardware; each XCD is given a contiguous range of tiles so that the halos are shared in its L2.
// CT = channel tiles of 32 per workgroup; a wave owns R rows of the 8 x 32 tile and CTW of the CT channel tiles, so the
// workgroup has (8/R) * (CT/CTW) waves: when LDS leaves room for one workgroup per CU only, 8 waves instead of 4 let one
// wave's epilogue / staging run under another's MFMAs.
// SWZ (CT = 1, CS = 32 only): unpadded swizzled rows, see conv_pass_mfma_swz.
template <int CT, int CS, int R, int CTW, bool SWZ = false>
__global__ __launch_bounds__(64 * (8 / R) * (CT / CTW)) __attribute__((amdgpu_waves_per_eu((CT == 1 && CS == 32 && R == 2) ? (SWZ ? 2 : 3) : 1, (CT == 1 && CS == 32 && R == 2) ? (SWZ ? 2 : 3) : 8)))
void conv3x3_resident_kernel(
    const uint16_t *__restrict__ in, const uint16_t *__restrict__ wp, const float *__restrict__ bias, uint16_t *__restrict__ out,
    int n_img, int frames, int h, int w, int c_out, int kt, int relu, int tiles_x, int tiles_y, int co_groups, const uint16_t *__restrict__ omask,
    int frame_order)
{
    static_assert(!SWZ || (CT == 1 && CS == 32 && CTW == 1), "swizzled rows: 32 channels, one channel tile");
    constexpr int THREADS = 64 * (8 / R) * (CT / CTW);
    constexpr int PS = SWZ ? CS : CS + 8;
    constexpr int P_CHUNKS = CV_PH * CV_PW * CS / 8;
    constexpr int P_PER_THREAD = (P_CHUNKS + THREADS - 1) / THREADS;
    auto chunk_at = [](int row, int c8) { return row * PS + ((SWZ ? c8 ^ ((row >> 2) & 3) : c8) << 3); };   // element offset of a 16-byte chunk
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *patch = lds;                                     // [CV_PH * CV_PW][PS]
    uint16_t *wl = lds + CV_PH * CV_PW * PS;                   // [kt * 9][CT * 32][PS]
    float *bias_l = reinterpret_cast<float *>(wl + kt * 9 * CT * 32 * PS);   // [CT * 32]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int row0 = (wave % (8 / R)) * R;                     // the wave's rows inside the tile
    const int cw0 = (wave / (8 / R)) * CTW * 32;               // its first channel inside the workgroup's group

    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int cog = j % co_groups, slot = j / co_groups;
    const int slots = (gridDim.x >> 3) / co_groups;
    const int n_tiles = n_img * tiles_y * tiles_x;
    const int lo = (int)((int64_t)n_tiles * xcd / 8), hi = (int)((int64_t)n_tiles * (xcd + 1) / 8);
    const int co0 = cog * CT * 32;

    // weights of this output-channel group, all taps
    const int w_rows = kt * 9 * CT * 32;
    for (int c = threadIdx.x; c < w_rows * (CS / 8); c += THREADS) {
        const int row = c / (CS / 8), c8 = c % (CS / 8);
        const int tap = row / (CT * 32), r = row % (CT * 32);
        *reinterpret_cast<uint4 *>(wl + chunk_at(row, c8)) =
            *reinterpret_cast<const uint4 *>(wp + ((int64_t)tap * c_out + co0 + r) * CS + c8 * 8);
    }
    if (threadIdx.x < CT * 32) bias_l[threadIdx.x] = bias ? bias[co0 + threadIdx.x] : 0.f;

    // per-thread constants of its patch pieces: position inside the patch and offset from the patch origin in the image
    uint4 preg[P_PER_THREAD];
    int p_off[P_PER_THREAD], p_yx[P_PER_THREAD];
#pragma unroll
    for (int q = 0; q < P_PER_THREAD; ++q) {
        const int c = threadIdx.x + q * THREADS;
        const int px = c / (CS / 8), c8 = c % (CS / 8);
        const int py = px / CV_PW, pxx = px % CV_PW;
        p_yx[q] = c < P_CHUNKS ? (py << 8 | pxx) : (0x7f << 8);                  // row 127 never passes the bounds test
        p_off[q] = ((py - 1) * w + pxx - 1) * CS + c8 * 8;
    }
    // with frame taps the FRAME runs fastest in the tile order, so that the tiles an XCD works on at one time are the same positions of
    // consecutive frames and the three reads of a frame's patch (as tap -1, 0, +1) meet in its L2 (FETCH_SIZE of the 27-tap layer at
    // 20 x 288^2: 148 -> 63 MB); frame-major order puts a whole frame (5.3 MB) between them
    ConvTileWalk walk;
    walk.init(lo, hi, slot, slots, tiles_y, tiles_x, frames, kt == 3 && frame_order, CV_TH, CV_TW);
    auto pass_valid = [&](const ConvTile &t, int f) {
        if (kt == 1) return true;
        const int t_frame = t.fr + f - 1;
        return t_frame >= 0 && t_frame < frames;
    };
    auto fetch = [&](const ConvTile &t, int f) {
        const int img = t.img + (kt == 3 ? f - 1 : 0);
        const uint16_t *src = in + ((int64_t)img * h * w + (int64_t)t.y0 * w + t.x0) * CS;
#pragma unroll
        for (int q = 0; q < P_PER_THREAD; ++q) {
            const int y = t.y0 - 1 + (p_yx[q] >> 8), x = t.x0 - 1 + (p_yx[q] & 0xff);
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) v = *reinterpret_cast<const uint4 *>(src + p_off[q]);
            preg[q] = v;
        }
    };
    auto store_pending = [&](const uint2 (&pk)[R][CTW][4], const ConvTile &t) {
        conv_store_packed<R, CTW>(pk, out, t.img, t.y0, t.x0, h, w, c_out, co0 + cw0, row0, lp, lh, omask);
    };

    const int n_mine = walk.count;
    int k = 0, f = 0;
    ConvTile cur = {0, 0, 0, 0}, nxt = cur, pnd = cur;
    if (n_mine > 0) {
        cur = walk.get(0);
        while (!pass_valid(cur, f)) ++f;                       // frame tap 1 (the frame itself) is always valid
        fetch(cur, f);
    }

    f32x16_t acc[R][CTW];
    uint2 pend[R][CTW][4];                                     // finished tile waiting for its stores
    bool have_pend = false, fresh = true;
    while (k < n_mine) {
        __syncthreads();                                       // the previous pass is done with the patch
#pragma unroll
        for (int q = 0; q < P_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * THREADS;
            if (c < P_CHUNKS) *reinterpret_cast<uint4 *>(patch + chunk_at(c / (CS / 8), c % (CS / 8))) = preg[q];
        }
        __syncthreads();
        if (have_pend) {
            store_pending(pend, pnd);
            have_pend = false;
        }
        int nk = k, nf = f + 1;
        nxt = cur;
        while (true) {
            if (nf >= kt) {
                nf = 0;
                if (++nk >= n_mine) break;
                nxt = walk.get(nk);
            }
            if (pass_valid(nxt, nf)) break;
            ++nf;
        }
        if (nk < n_mine) fetch(nxt, nf);                       // in flight during the MFMAs below

        if (fresh) {
#pragma unroll
            for (int m = 0; m < R; ++m)
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][ct][r] = 0.f;
            fresh = false;
        }
        if constexpr (SWZ)
            conv_pass_mfma_swz<R>(acc, patch, row0 * CV_PW + lp, lh, wl + (f * 9 * 32 + lp) * 32, (lh ^ ((lp >> 2) & 3)) << 3);
        else
            conv_pass_mfma<R, CTW, CS, CT * 32>(acc, patch + (row0 * CV_PW + lp) * PS + lh * 8,
                                                wl + ((f * 9) * CT * 32 + cw0 + lp) * PS + lh * 8);
        if (nk != k) {                                         // last pass of this tile
            conv_pack_tile<R, CTW>(acc, bias_l + cw0, relu, lh, pend);
            pnd = cur;
            have_pend = true;
            fresh = true;
        }
        k = nk;
        f = nf;
        cur = nxt;
    }
    if (have_pend) store_pending(pend, pnd);
}

template <int CT, int CS, int R, int CTW, bool SWZ = false>
static int conv_launch_resident(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int n_img, int frames, int h,
                                int w, int c_out, int kt, int relu, hipStream_t st, const uint16_t *omask = nullptr)
{
    constexpr int THREADS = 64 * (8 / R) * (CT / CTW);
    const int tiles_x = (w + CV_TW - 1) / CV_TW, tiles_y = (h + CV_TH - 1) / CV_TH;
    const int co_groups = c_out / (CT * 32);
    const size_t lds = (size_t)(CV_PH * CV_PW + kt * 9 * CT * 32) * (SWZ ? CS : CS + 8) * sizeof(uint16_t) + CT * 32 * sizeof(float);
    auto kern = conv3x3_resident_kernel<CT, CS, R, CTW, SWZ>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    int per_cu = (int)((160 * 1024) / lds);
    const int by_waves = 16 / (THREADS / 64);                                // keep <= 16 waves per CU: the kernels use ~128+ registers
    per_cu = per_cu > by_waves ? by_waves : per_cu;
    per_cu = per_cu < 1 ? 1 : per_cu;
    const int64_t n_tiles = (int64_t)n_img * tiles_y * tiles_x;
    if (n_tiles > 0x7fffffff || !PCACC_WALK_OK(n_img, frames, tiles_y, tiles_x)) return PCACC_E_ARG;
    int64_t slots = (int64_t)PCACC_CUS * per_cu / 8 / co_groups;            // per XCD and channel group
    const int64_t need = (n_tiles + 7) / 8;
    if (slots > need) slots = need;
    if (slots < 1) slots = 1;
    const unsigned grid = (unsigned)(8 * co_groups * slots);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, in, wp, bias, out, n_img, frames, h, w, c_out, kt, relu, tiles_x,
                       tiles_y, co_groups, omask, pcacc_switches().conv_frame_major ? 0 : 1);
    PCACC_CHECK_LAUNCH();
    return 0;
}

template <int CT, int CS>
static int conv_launch(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int n_img, int frames, int h, int w,
                       int c_in, int c_out, int kt, int relu, hipStream_t st, const uint16_t *omask = nullptr)
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
                       relu, tiles_x, tiles_y, co_groups, omask);
    PCACC_CHECK_LAUNCH();
    return 0;
}

static int conv3x3_bf16_any(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int32_t n_img,
                            int32_t frames, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt, int32_t relu,
                            void *stream, const uint16_t *omask)
{
    if (!in || !wp || !out || n_img < 1 || h < 1 || w < 1 || (kt != 1 && kt != 3) || frames < 1 || n_img % frames) return PCACC_E_ARG;
    if (c_in % 32 || c_out % 32 || c_in < 32 || c_out < 32) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    if (c_in == 32 || c_in == 64) {
        // all taps' weights + the patch within 150 KB of LDS: widest channel group that fits
        for (int ctr = 4; ctr >= 1; ctr >>= 1) {
            if (c_out % (ctr * 32)) continue;
            const size_t lds = (size_t)(CV_PH * CV_PW + kt * 9 * ctr * 32) * (c_in + 8) * sizeof(uint16_t) + ctr * 32 * sizeof(float);
            if (lds > 150 * 1024) continue;
            const bool alone = lds > 80 * 1024;                    // one workgroup per CU: run it with 8 waves
#define CV_RES(CTV, CSV, RV, CTWV) \
    return conv_launch_resident<CTV, CSV, RV, CTWV>(in, wp, bias, out, n_img, frames, h, w, c_out, kt, relu, st, omask)
            if (c_in == 32 && ctr == 1) {
                // 27 taps: 96 KB with padded rows = one 8-wave workgroup per CU, 77 KB with swizzled rows = two of 4 waves, 2 rows each
                if (alone && !pcacc_switches().conv_swz_off)                 // the switch is for A/B measurements and the equality test
                    return conv_launch_resident<1, 32, 1, 1, true>(in, wp, bias, out, n_img, frames, h, w, c_out, kt, relu, st, omask);
                if (alone) CV_RES(1, 32, 1, 1);
                CV_RES(1, 32, 2, 1);
            }
            if (c_in == 32 && ctr == 2) { if (alone) CV_RES(2, 32, 2, 1); CV_RES(2, 32, 2, 2); }
            if (c_in == 32 && ctr == 4) { CV_RES(4, 32, 2, 2); }
            if (c_in == 64 && ctr == 1) { if (alone) CV_RES(1, 64, 1, 1); CV_RES(1, 64, 2, 1); }
            if (c_in == 64 && ctr == 2) { CV_RES(2, 64, 2, 1); }
#undef CV_RES
        }
    }
    // deep layers on small images: strips of consecutive pixels, K-deep tiling (conv_deep.hip)
    if (kt == 1 && pcacc_conv3x3_deep_supported(h, w, c_in, c_out)) {
        if (omask) return PCACC_E_ARG;                             // the strip kernels take an input mask only (pcacc_conv3x3_outmask_supported)
        return pcacc_conv3x3_deep_bf16(in, nullptr, wp, bias, out, n_img, h, w, c_in, c_out, relu, stream);
    }
    const int cs_sel = c_in % 128 == 0 ? 128 : (c_in % 64 == 0 ? 64 : 32);   // input channels per LDS pass
    const int ct = c_out % 128 == 0 ? 4 : (c_out % 64 == 0 ? 2 : 1);
#define CV_CASE(CTV, CSV) \
    if (ct == CTV && cs_sel == CSV) return conv_launch<CTV, CSV>(in, wp, bias, out, n_img, frames, h, w, c_in, c_out, kt, relu, st, omask)
    CV_CASE(1, 32); CV_CASE(2, 32); CV_CASE(4, 32);
    CV_CASE(1, 64); CV_CASE(2, 64); CV_CASE(4, 64);
    CV_CASE(1, 128); CV_CASE(2, 128); CV_CASE(4, 128);
#undef CV_CASE
    return PCACC_E_ARG;
}

extern "C" int pcacc_conv3x3_bf16(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int32_t n_img,
                                  int32_t frames, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt, int32_t relu,
                                  void *stream)
{
    return conv3x3_bf16_any(in, wp, bias, out, n_img, frames, h, w, c_in, c_out, kt, relu, stream, nullptr);
}

// The same convolution with its result zeroed where out_mask [n_img, h, w, c_out] is <= 0: the data gradient of a layer whose forward input
// was a ReLU output (models/unet.py:45-71 conv -> ReLU -> conv), masked for that ReLU in the epilogue instead of by a threshold pass
// (three sweeps of the map).  Layers the strip kernels of conv_deep.hip would take are not supported (they mask on the input side).
extern "C" int pcacc_conv3x3_outmask_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt)
{
    if (c_in % 32 || c_out % 32 || c_in < 32 || c_out < 32 || (kt != 1 && kt != 3)) return 0;
    if (c_in == 32 || c_in == 64) return 1;                        // the resident kernels
    return !(kt == 1 && pcacc_conv3x3_deep_supported(h, w, c_in, c_out));
}

extern "C" int pcacc_conv3x3_outmask_bf16(const uint16_t *in, const uint16_t *wp, const uint16_t *out_mask, uint16_t *out, int32_t n_img,
                                          int32_t frames, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt, void *stream)
{
    if (!out_mask || !pcacc_conv3x3_outmask_supported(h, w, c_in, c_out, kt)) return PCACC_E_ARG;
    return conv3x3_bf16_any(in, wp, nullptr, out, n_img, frames, h, w, c_in, c_out, kt, 0, stream, out_mask);
}

// ReLU backward fused into the consumer of the gradient: only the deep (MFMA-bound) layers take it, where the second read is free.
// The c_in <= 64 layers are HBM-bound: reading the mask next to the gradient in both consumers costs what the separate pass costs.
extern "C" int pcacc_conv3x3_masked_bf16(const uint16_t *in, const uint16_t *in_mask, const uint16_t *wp, const float *bias, uint16_t *out,
                                         int32_t n_img, int32_t frames, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt,
                                         int32_t relu, void *stream)
{
    if (!in_mask) return pcacc_conv3x3_bf16(in, wp, bias, out, n_img, frames, h, w, c_in, c_out, kt, relu, stream);
    if (kt != 1 || !pcacc_conv3x3_deep_supported(h, w, c_in, c_out)) return PCACC_E_ARG;
    return pcacc_conv3x3_deep_bf16(in, in_mask, wp, bias, out, n_img, h, w, c_in, c_out, relu, stream);
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
// dW[co][tap][ci] = sum over images and pixels of dY[px][co] * X[px + tap offset][ci]  (zero outside the image), one frame tap per
// launch (dt: X is read from image n + dt of the same sample, or not at all when that frame does not exist).
// The reduction runs over pixels, so both MFMA operands are "8 consecutive pixels of one channel" per lane: the dY tile and
// the X patch are staged channels-last as they come and the fragments are read with the hardware LDS transpose
// (ds_read_b64_tr_b16, two per fragment; row stride pcacc_tr_stride(C) elements).  A wave owns one (co tile, ci tile) pair and one
// of TG = 4 / pairs TAP GROUPS (taps t with t % TG == group: 3 | 2 | 2 | 2 taps at 32 x 32 channels, 5 | 4 at 32 x 64, all 9 at
// 64 x 64) over ALL 16-pixel steps of the tile: 48 / 80 accumulator registers instead of 144, so two or three workgroups share a CU
// and one wave's LDS latency hides behind another's MFMAs (r02: with every wave holding all 9 taps of a quarter of the pixels the
// kernel ran one wave per SIMD -- 310 registers -- at 15 % matrix-pipe utilisation, and its accumulators had to be folded through
// LDS at the end).  Workgroups are persistent; one workspace slot per workgroup, a second launch sums the slots.
typedef short cv_s16x4 __attribute__((ext_vector_type(4)));
union cv_frag { bf16x8_t v; cv_s16x4 h[2]; };

template <int CO_T, int CI_T>
__global__ __launch_bounds__(CV_THREADS) void conv3x3_wgrad_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                                   float *__restrict__ partial, int n_img, int frames, int dt, int h,
                                                                   int w, int tiles_x, int tiles_y, float *__restrict__ dw_zero)
{
    constexpr int CO = CO_T * 32, CI = CI_T * 32, PAIRS = CO_T * CI_T, TG = 4 / PAIRS, NT = (9 + TG - 1) / TG;
    if (blockIdx.x == 0)                                       // the reduce launch that follows adds into dW: cleared here, not by a memset
        for (int e = threadIdx.x; e < CO * 9 * CI + CO; e += CV_THREADS) dw_zero[e] = 0.f;
    constexpr int YS = pcacc_tr_stride(CO), XS = pcacc_tr_stride(CI);
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *sdy = lds;                                       // [CV_TH * CV_TW][YS]
    uint16_t *sx = lds + CV_TH * CV_TW * YS;                   // [CV_PH * CV_PW][XS]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int pair = wave % PAIRS, grp = wave / PAIRS;
    const int ct = pair / CI_T, it = pair % CI_T;

    f32x16_t acc[NT];                                          // local tap j = tap grp + j * TG
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int n_tiles = n_img * tiles_y * tiles_x;
    const int lo = (int)((int64_t)n_tiles * xcd / 8), hi = (int)((int64_t)n_tiles * (xcd + 1) / 8);

    // the next tile's dY rows and X patch travel in registers while the current tile is multiplied
    constexpr int Y_CHUNKS = CV_TH * CV_TW * CO / 8, Y_PER_THREAD = Y_CHUNKS / CV_THREADS;
    constexpr int X_CHUNKS = CV_PH * CV_PW * CI / 8, X_PER_THREAD = (X_CHUNKS + CV_THREADS - 1) / CV_THREADS;
    uint4 yreg[Y_PER_THREAD], xreg[X_PER_THREAD];
    ConvTileWalk walk;                                         // tile coordinates without scalar divisions per tile
    walk.init(lo, hi, slot, slots, tiles_y, tiles_x, frames, 0, CV_TH, CV_TW);
    ConvTile cur = {0, 0, 0, 0};
    const int n_mine = walk.count;
    auto advance = [&](int from) {                             // the first of the workgroup's tiles >= from whose frame + dt exists
        int kk = from;
        while (kk < n_mine) {
            cur = walk.get(kk);
            const int t_frame = cur.fr + dt;
            if (t_frame >= 0 && t_frame < frames) break;       // a missing frame contributes nothing
            ++kk;
        }
        return kk;
    };
    auto fetch = [&]() {
        const int img = cur.img, y0 = cur.y0, x0 = cur.x0;
        const uint16_t *gsrc = dy + (int64_t)img * h * w * CO;
#pragma unroll
        for (int q = 0; q < Y_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * CV_THREADS;
            const int px = c / (CO / 8), c8 = c % (CO / 8);
            const int yy = y0 + px / CV_TW, xx = x0 + px % CV_TW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (yy < h && xx < w) v = *reinterpret_cast<const uint4 *>(gsrc + ((int64_t)yy * w + xx) * CO + c8 * 8);
            yreg[q] = v;
        }
        const uint16_t *xsrc = x + (int64_t)(img + dt) * h * w * CI;
#pragma unroll
        for (int q = 0; q < X_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * CV_THREADS;
            const int px = c / (CI / 8), c8 = c % (CI / 8);
            const int yy = y0 - 1 + px / CV_PW, xx = x0 - 1 + px % CV_PW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < X_CHUNKS && yy >= 0 && yy < h && xx >= 0 && xx < w)
                v = *reinterpret_cast<const uint4 *>(xsrc + ((int64_t)yy * w + xx) * CI + c8 * 8);
            xreg[q] = v;
        }
    };
    int k = advance(0);
    if (k < n_mine) fetch();
    while (k < n_mine) {
        __syncthreads();                                       // the previous tile's gathers are done
#pragma unroll
        for (int q = 0; q < Y_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * CV_THREADS;
            uint2 *dst = reinterpret_cast<uint2 *>(sdy + (c / (CO / 8)) * YS + (c % (CO / 8)) * 8);
            dst[0] = make_uint2(yreg[q].x, yreg[q].y);
            dst[1] = make_uint2(yreg[q].z, yreg[q].w);
        }
#pragma unroll
        for (int q = 0; q < X_PER_THREAD; ++q) {
            const int c = threadIdx.x + q * CV_THREADS;
            if (c < X_CHUNKS) {
                uint2 *dst = reinterpret_cast<uint2 *>(sx + (c / (CI / 8)) * XS + (c % (CI / 8)) * 8);
                dst[0] = make_uint2(xreg[q].x, xreg[q].y);
                dst[1] = make_uint2(xreg[q].z, xreg[q].w);
            }
        }
        __syncthreads();
        k = advance(k + 1);
        if (k < n_mine) fetch();
        // 16-pixel steps of the tile: step s = row s/2, columns (s%2)*16 ..; this wave's share is every GROUPS-th step
        // fragments through the LDS transpose read (see rows_wgrad_bf16_kernel in mlp_mfma.hip): two ds_read_b64_tr_b16 give a
        // lane its 8 consecutive pixels of one channel
        const int tg = lane >> 4, tl = lane & 15;
        const int tr_row = (tg >> 1) * 8 + (tl >> 2), tr_col = (tg & 1) * 16 + (tl & 3) * 4;
        for (int s = 0; s < CV_TH * 2; ++s) {
            const int ry = s >> 1, xb = (s & 1) * 16;
            const uint16_t *pa = sdy + (ry * CV_TW + xb + tr_row) * YS + ct * 32 + tr_col;
            cv_frag af;
            af.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cv_s16x4 __attribute__((address_space(3))) *)pa);
            af.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cv_s16x4 __attribute__((address_space(3))) *)(pa + 4 * YS));
            if (it == 0 && grp == 0) {                         // bias gradient = column sums of dY: the fragment is at hand
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bsum += bf16_to_f32((uint16_t)af.h[j][q]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int tap = grp + j * TG;                  // uniform per wave
                if (tap < 9) {
                    const uint16_t *pb = sx + ((ry + tap / 3) * CV_PW + xb + tap % 3 + tr_row) * XS + it * 32 + tr_col;
                    cv_frag bf;
                    bf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cv_s16x4 __attribute__((address_space(3))) *)pb);
                    bf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cv_s16x4 __attribute__((address_space(3))) *)(pb + 4 * XS));
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.v, bf.v, acc[j], 0, 0, 0);
                }
            }
        }
    }
    // slot of this workgroup: [CO][9][CI] then [CO] bias sums; D has lane = ci, register quads = co
    float *mine = partial + (int64_t)blockIdx.x * (CO * 9 * CI + CO);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tap = grp + j * TG;
        if (tap < 9)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                mine[((int64_t)co * 9 + tap) * CI + it * 32 + lp] = acc[j][r];
            }
    }
    if (grp != 0) return;
    bsum += __shfl_xor(bsum, 32, 64);                          // the two half-waves hold pixels 0-7 / 8-15 of the same channel
    if (it == 0 && lh == 0) mine[CO * 9 * CI + ct * 32 + lp] = bsum;
}

// out[e] = sum over the workgroup partials in a fixed order (common.h: pcacc_reduce_partials) -- run-to-run identical
template <int EL>
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(const float *__restrict__ partial, int n_parts, int elems, float *__restrict__ out)
{
    pcacc_reduce_partials<EL>(partial, n_parts, elems, [&](int e, float v) { out[e] = v; });
}

static int conv_wgrad_grid(int c_in, int c_out, int64_t n_tiles)
{
    const size_t lds = (size_t)(CV_TH * CV_TW * pcacc_tr_stride(c_out) + CV_PH * CV_PW * pcacc_tr_stride(c_in)) * sizeof(uint16_t);
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 3 ? 3 : (per_cu < 1 ? 1 : per_cu);
    int64_t slots = (int64_t)PCACC_CUS * per_cu / 8;
    const int64_t need = (n_tiles + 7) / 8;
    if (slots > need) slots = need;
    if (slots < 1) slots = 1;
    return (int)(8 * slots);
}

extern "C" int pcacc_conv3x3_wgrad_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out, size_t *bytes)
{
    if (!bytes || n_img < 1 || h < 1 || w < 1 || (c_in != 32 && c_in != 64) || (c_out != 32 && c_out != 64)) return PCACC_E_ARG;
    const int64_t n_tiles = (int64_t)n_img * ((h + CV_TH - 1) / CV_TH) * ((w + CV_TW - 1) / CV_TW);
    *bytes = (size_t)conv_wgrad_grid(c_in, c_out, n_tiles) * (c_out * 9 * c_in + c_out) * sizeof(float);
    return 0;
}

extern "C" int pcacc_conv3x3_wgrad_bf16(const uint16_t *dy, const uint16_t *x, float *dw, int32_t n_img, int32_t frames, int32_t dt,
                                        int32_t h, int32_t w, int32_t c_in, int32_t c_out, void *workspace, size_t workspace_bytes,
                                        void *stream)
{
    if (!dy || !x || !dw || !workspace || n_img < 1 || h < 1 || w < 1 || frames < 1 || n_img % frames || dt < -1 || dt > 1)
        return PCACC_E_ARG;
    if ((c_in != 32 && c_in != 64) || (c_out != 32 && c_out != 64)) return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    const int tiles_x = (w + CV_TW - 1) / CV_TW, tiles_y = (h + CV_TH - 1) / CV_TH;
    const int64_t n_tiles = (int64_t)n_img * tiles_y * tiles_x;
    if (n_tiles > 0x7fffffff || !PCACC_WALK_OK(n_img, frames, tiles_y, tiles_x)) return PCACC_E_ARG;
    const int grid = conv_wgrad_grid(c_in, c_out, n_tiles);
    const int elems = c_out * 9 * c_in + c_out;                 // weight gradient, then the bias gradient
    if (workspace_bytes < (size_t)grid * elems * sizeof(float)) return PCACC_E_WORKSPACE;
    const size_t lds = (size_t)(CV_TH * CV_TW * pcacc_tr_stride(c_out) + CV_PH * CV_PW * pcacc_tr_stride(c_in)) * sizeof(uint16_t);
    float *partial = reinterpret_cast<float *>(workspace);
#define CV_WG(COT, CIT)                                                                                                              \
    do {                                                                                                                             \
        auto kern = conv3x3_wgrad_kernel<COT, CIT>;                                                                                  \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   (int)lds) != hipSuccess)                                                          \
            return PCACC_E_LAUNCH;                                                                                                   \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(CV_THREADS), lds, st, dy, x, partial, n_img, frames, dt, h, w, tiles_x, tiles_y, dw); \
    } while (0)
    if (c_out == 32 && c_in == 32) CV_WG(1, 1);
    else if (c_out == 32 && c_in == 64) CV_WG(1, 2);
    else if (c_out == 64 && c_in == 32) CV_WG(2, 1);
    else CV_WG(2, 2);
#undef CV_WG
    if (pcacc_reduce_el(elems) == 64) conv_wgrad_reduce_kernel<64><<<(elems + 63) / 64, 1024, 0, st>>>(partial, grid, elems, dw);
    else conv_wgrad_reduce_kernel<16><<<(elems + 15) / 16, 1024, 0, st>>>(partial, grid, elems, dw);
    PCACC_CHECK_LAUNCH();
    return 0;
}
