// 3x3 (and 3x3x3 over frames) convolution at fp32 accuracy on the 16-bit matrix cores ("fp32x3" compute mode): fp32 channels-last in
// and out, every product formed from fp16 hi / lo halves of SCALED operands,
//
//     s x = x_hi + x_lo (+ 2^-22 s|x|),  t w = w_hi + w_lo:   (t w)(s x)  ~=  w_hi x_lo + w_lo x_hi + w_hi x_hi      (w_lo x_lo <= 2^-22 dropped)
//
// three v_mfma_f32_32x32x16_f16 per fragment pair, fp32 accumulation, result divided by s t.  fp16 carries 11 significant bits, so hi + lo
// keep 22 (fp32 has 24): relative error ~3e-7 per product -- the first version of this file split into bf16 halves (8 + 8 bits, 4e-6
// per product, 2e-5 after the U-Net) and left the c4 scene-flow EPE 1.04e-3 from the reference and the gradient norms of the
// ill-conditioned loss terms up to 6 % off (profiles/r03_gradnorm_sensitivity.txt).  fp16's narrow exponent range is handled by
// power-of-two scales: s per input TENSOR (from its absolute maximum, pcacc_absmax256: the scaled maximum lands in [2^13, 2^14)), t per
// output-channel row of the weights (fixed when the weights are prepared).  Elements far below the tensor's maximum lose relative, not
// absolute precision (their lo half becomes subnormal): errors stay below 2^-22 of the LARGEST operand, which is what a sum needs.
// Rate: a third of the fp16 / bf16 matrix rate = 5x the fp32 MFMA rate (v_mfma_f32_32x32x2_f32 runs at 1/16).  This is the mode in
// which the dense stacks (models/unet.py:11-20,45-113, models/stpn.py:13-43 -- fp32 convolutions in the reference) meet north_star's
// 1e-3 on hand-written kernels; the fp32 mode used the library's fp32 convolutions for that (98.8 ms per step, 43 ms of it MIOpen).
//
// One kernel family covers every layer (c_in, c_out multiples of 32; forward, and on mirrored / transposed weights the data
// gradient):
//   * tile = rows x bw pixels at (y0, x0) of one image; M-tiles are 32 consecutive pixels of the tile in row-major order (the strip
//     scheme of conv_deep.hip with a band width: a 288-wide image is cut into 32-wide bands, an 18-wide one is taken whole); every
//     lane keeps the LDS offset of its pixels' 3x3 windows, a tap adds a constant.
//   * K runs over (frame tap, CS-channel slice, 3x3 tap).  The slice's input patch is scaled and converted to hi / lo while it is
//     staged (two fp16 LDS planes, rows padded by 8 elements: conflict-free 16-byte fragment reads); the next slice's patch travels in
//     registers as fp32 during the MFMAs.  The ReLU backward of the layer whose gradient is being consumed is applied while staging
//     (in_mask = that layer's output).
//   * weight tiles ([hi | lo] x [WROWS output channels] x [CS]) per tap are double buffered in LDS and requested two taps ahead
//     (conv_deep.hip's scheme): one barrier per tap, 3 x the MFMAs of the bf16 kernel between two barriers.
//   * 8 waves = MG pixel groups x NGW channel groups; a wave owns MT pixel tiles x NW channel tiles.  LDS fragment reads per MFMA:
//     (2 NW + 2 MT) / (3 NW MT) = 0.67 at 2 x 2 -- the split turns the LDS-bound 32/64-channel layers of the bf16 kernels into
//     matrix-pipe-bound ones.
#include "common.h"

#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// [r5] experiment (CSP_SHADOW_NT=1): the bf16 shadow of a forward result ('mixed' mode) is read by the BACKWARD pass, a whole forward later; stored
// with the streaming policy it should not push the fp32 result out of the caches.  Measured the other way round: step 31.55 ms with streaming shadow
// stores against 30.78 ms with plain ones (three interleaved pairs, profiles/r05_cache_policy_ab.txt) -- the 8-byte shadow pieces of a lane are
// half a 16-byte fp32 piece's neighbours, and partial-line streaming writes cost more than they save.  Plain stores stay.
typedef uint32_t csp_u32x2 __attribute__((ext_vector_type(2)));
#ifndef CSP_SHADOW_NT
#define CSP_SHADOW_NT 0
#endif
// experiment (CSP_IN_NT=1): the fp32 input of a forward layer is read once per layer -- streaming loads mark its lines for early eviction
typedef float csp_f32x4v __attribute__((ext_vector_type(4)));
#ifndef CSP_IN_NT
#define CSP_IN_NT 0
#endif
__device__ __forceinline__ float4 csp_load_in4(const float *p)
{
    if (CSP_IN_NT) {
        const csp_f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const csp_f32x4v *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ void csp_store_shadow(uint16_t *p, uint32_t a, uint32_t b)
{
    const csp_u32x2 v = {a, b};
    if (CSP_SHADOW_NT) __builtin_nontemporal_store(v, reinterpret_cast<csp_u32x2 *>(p));
    else *reinterpret_cast<csp_u32x2 *>(p) = v;
}

#define CSP_THREADS 512
#define CSP_PCH 6                              // 8-channel patch chunks (two float4) a thread carries per slice
#define CSP_LDS_MAX (160 * 1024)
#define CSP_AMAX_PARTS 256                     // partial maxima pcacc_absmax256 leaves for its consumers

// ---- absolute maximum of a tensor: 256 partial maxima, every one always written (no initialisation, no second launch); the consumers
// reduce the 256 values themselves (csp_scale_from_parts) ---------------------------------------------------------------------------------
#define CSP_AMAX_THREADS 1024
__global__ __launch_bounds__(CSP_AMAX_THREADS) void absmax256_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ parts)
{
    // 256 workgroups x 1024 threads x 4 independent 16-byte loads per iteration = 16 MB in flight: an HBM stream (the first version,
    // one load per thread of 256 x 256, ran at 1.9 TB/s and cost 10 ms of a 58 ms step)
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    bool bad = false;
    const int64_t n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const int64_t stride = (int64_t)gridDim.x * CSP_AMAX_THREADS;
    int64_t i = (int64_t)blockIdx.x * CSP_AMAX_THREADS + threadIdx.x;
    auto fold = [&](float &m, const float4 &v) {
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));        // fmaxf drops NaN ...
        bad |= !(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w);                            // ... but a NaN must poison the result
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        fold(m0, a); fold(m1, b); fold(m2, c); fold(m3, d);
    }
    for (; i < n4; i += stride) fold(m0, x4[i]);
    float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = x[(n4 << 2) + threadIdx.x];
        m = fmaxf(m, fabsf(v));
        bad |= v != v;
    }
    if (bad) m = __builtin_inff();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    __shared__ float sm[CSP_AMAX_THREADS / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = sm[0];
#pragma unroll
        for (int w = 1; w < CSP_AMAX_THREADS / 64; ++w) r = fmaxf(r, sm[w]);
        parts[blockIdx.x] = r;
    }
}

extern "C" int pcacc_absmax256(const float *x, int64_t n, float *parts, void *stream)
{
    if (!x || !parts || n < 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return PCACC_E_ARG;
    hipLaunchKernelGGL(absmax256_kernel, dim3(CSP_AMAX_PARTS), dim3(CSP_AMAX_THREADS), 0, pcacc_stream(stream), x, n, parts);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// power-of-two scale that puts a tensor's absolute maximum into [2^13, 2^14) (fp16 overflows at 65504); 1 for an all-zero tensor.
// A non-finite maximum gives scale 1: the non-finite element then reaches the output as inf / NaN, as it would in fp32 arithmetic.
__device__ __forceinline__ float csp_scale_of(float amax)
{
    if (!(amax > 0.f) || !(amax < __builtin_inff())) return 1.f;
    int k;
    frexpf(amax, &k);                                         // amax = m 2^k, m in [0.5, 1)
    return ldexpf(1.f, 14 - k);
}

// every lane reduces the 256 partial maxima (wave-uniform result, no LDS, no barrier)
__device__ __forceinline__ float csp_scale_from_parts(const float *__restrict__ parts)
{
    const int lane = threadIdx.x & 63;
    float m = fmaxf(fmaxf(parts[lane], parts[lane + 64]), fmaxf(parts[lane + 128], parts[lane + 192]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    return csp_scale_of(m);
}

__device__ __forceinline__ uint32_t csp_pack_f16x2(float a, float b)
{
    const pcacc_f32x2 f = {a, b};
    const f16x2_t r = __builtin_convertvector(f, f16x2_t);    // round to nearest even
    return *reinterpret_cast<const uint32_t *>(&r);
}
__device__ __forceinline__ pcacc_f32x2 csp_unpack_f16x2(uint32_t v)
{
    return __builtin_convertvector(*reinterpret_cast<const f16x2_t *>(&v), pcacc_f32x2);
}
#ifdef PCACC_X3_EXPERIMENT
__device__ int csp_xword;                                    // common.h: precision-map experiment build
extern "C" int pcacc_x3_experiment_conv(int word, void *stream)
{
    if (hipStreamSynchronize(pcacc_stream(stream)) != hipSuccess) return PCACC_E_LAUNCH;     // kernels already queued keep the word they were launched under
    return hipMemcpyToSymbol(HIP_SYMBOL(csp_xword), &word, sizeof(int)) == hipSuccess ? PCACC_OK : PCACC_E_LAUNCH;
}
// a 16-byte piece of a prepared WEIGHT plane (plane 0 = hi, 1 = lo) as the experiment sees it
__device__ __forceinline__ uint4 csp_x_weight(uint4 v, int plane)
{
    const int f = PCACC_X_W(csp_xword);
    if (plane == 1) return (f & 5) ? make_uint4(0u, 0u, 0u, 0u) : v;
    if (f & 4) {
        uint32_t *u = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 h = *reinterpret_cast<const h2 *>(&u[i]);
            h[0] = (_Float16)pcacc_x_round_bf16((float)h[0]);
            h[1] = (_Float16)pcacc_x_round_bf16((float)h[1]);
            u[i] = *reinterpret_cast<const uint32_t *>(&h);
        }
    }
    return v;
}
#endif
__device__ __forceinline__ void csp_split2(float a, float b, uint32_t &hi, uint32_t &lo)
{
#ifdef PCACC_X3_EXPERIMENT
    const bool drop = pcacc_x_apply(PCACC_X_ACT(csp_xword), a, b);
#endif
    hi = csp_pack_f16x2(a, b);
    const pcacc_f32x2 back = csp_unpack_f16x2(hi);
    lo = csp_pack_f16x2(a - back[0], b - back[1]);           // exact differences (Sterbenz); an inf hi gives NaN here, as it should
#ifdef PCACC_X3_EXPERIMENT
    if (drop) lo = 0u;
#endif
}
// eight fp32, scaled by s -> eight fp16 hi + eight fp16 lo (round to nearest even both times)
__device__ __forceinline__ void csp_split8(const float4 &a, const float4 &b, float s, uint4 &hi, uint4 &lo)
{
    csp_split2(a.x * s, a.y * s, hi.x, lo.x);
    csp_split2(a.z * s, a.w * s, hi.y, lo.y);
    csp_split2(b.x * s, b.y * s, hi.z, lo.z);
    csp_split2(b.z * s, b.w * s, hi.w, lo.w);
}

__device__ __forceinline__ float4 csp_relu_mask4(float4 g, float4 y)
{
    return make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
}

// ---- weight preparation: fp32 [O][I][KT][3][3] read through its strides -> fp16 [2 = hi, lo][KT*9][O'][I'] + fp32 [O'] ----------------
// forward form (O' = O, I' = I) and data-gradient form (O' = I, I' = O, taps and frame taps mirrored) in one launch: one workgroup per
// output-channel row of either form -- the row's absolute maximum fixes its scale t (scaled maximum in [2^13, 2^14)), 1 / t goes to
// the scale vector the convolution's epilogue multiplies with.
struct CspWStrides { int64_t o, i, t, y, x; };

// `blk` = the row of the two forms (o forward rows, then i data-gradient rows) this workgroup prepares
__device__ __forceinline__ void csp_prepare_row(const float *__restrict__ w, int o, int i, int kt, CspWStrides st, uint16_t *__restrict__ out_fwd,
                                                float *__restrict__ inv_fwd, uint16_t *__restrict__ out_bwd, float *__restrict__ inv_bwd, int blk)
{
    const int taps = kt * 9;
    const bool transpose = blk >= o;
    const int row = transpose ? blk - o : blk;                   // output channel of this form
    const int op = transpose ? i : o, ip = transpose ? o : i;
    const int64_t total = (int64_t)taps * o * i;
    const int n = taps * ip;                                      // elements of the row: (tap, input channel of this form)
    auto src = [&](int e) {
        const int tap = e / ip, ci = e - tap * ip;
        const int src_tap = transpose ? (taps - 1 - tap) : tap;
        const int so = transpose ? ci : row, si = transpose ? row : ci;
        const int ft = src_tap / 9, fy = (src_tap % 9) / 3, fx = src_tap % 3;
        return w[so * st.o + si * st.i + ft * st.t + fy * st.y + fx * st.x];
    };
    // [r5] a row of up to CSP_PREP_KEEP * 256 elements (every layer of the model: 9 x 512 = 4 608) is read ONCE and kept in registers between the maximum
    // pass and the split pass -- the rows of the data-gradient form are gathers of single floats at a stride of 9 c_in, and reading them twice was most of
    // the 250 us the batched preparation takes at the head of every step
    constexpr int CSP_PREP_KEEP = 18;
    const bool keep = n <= CSP_PREP_KEEP * 256;
    float kept[CSP_PREP_KEEP];
    float m = 0.f;
    if (keep) {
#pragma unroll
        for (int j = 0; j < CSP_PREP_KEEP; ++j) {
            const int e = threadIdx.x + j * 256;
            const float v = e < n ? src(e) : 0.f;
            kept[j] = v;
            m = fmaxf(m, fabsf(v));
            if (v != v) m = __builtin_inff();
        }
    } else {
        for (int e = threadIdx.x; e < n; e += 256) {
            const float v = src(e);
            m = fmaxf(m, fabsf(v));
            if (v != v) m = __builtin_inff();
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    const float t = csp_scale_of(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])));
    if (threadIdx.x == 0) (transpose ? inv_bwd : inv_fwd)[row] = 1.f / t;
    uint16_t *dst = transpose ? out_bwd : out_fwd;
    auto put = [&](int e, float raw) {
        const int tap = e / ip, ci = e - tap * ip;
        const float v = raw * t;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int64_t r = ((int64_t)tap * op + row) * ip + ci;
        dst[r] = *reinterpret_cast<const uint16_t *>(&hi);
        dst[total + r] = *reinterpret_cast<const uint16_t *>(&lo);
    };
    if (keep) {
#pragma unroll
        for (int j = 0; j < CSP_PREP_KEEP; ++j) {
            const int e = threadIdx.x + j * 256;
            if (e < n) put(e, kept[j]);
        }
    } else {
        for (int e = threadIdx.x; e < n; e += 256) put(e, src(e));
    }
}

__global__ __launch_bounds__(256) void conv_split_prepare_kernel(const float *__restrict__ w, int o, int i, int kt, CspWStrides st,
                                                                 uint16_t *__restrict__ out_fwd, float *__restrict__ inv_fwd,
                                                                 uint16_t *__restrict__ out_bwd, float *__restrict__ inv_bwd)
{
    csp_prepare_row(w, o, i, kt, st, out_fwd, inv_fwd, out_bwd, inv_bwd, (int)blockIdx.x);
}

extern "C" int pcacc_conv3x3_split_prepare_weights(const float *w, int32_t c_out, int32_t c_in, int32_t kt, const int64_t *strides,
                                                   uint16_t *out_fwd, float *scale_fwd, uint16_t *out_bwd, float *scale_bwd, void *stream)
{
    if (!w || !out_fwd || !out_bwd || !scale_fwd || !scale_bwd || !strides || c_out < 1 || c_in < 1 || (kt != 1 && kt != 3)) return PCACC_E_ARG;
    const CspWStrides st = {strides[0], strides[1], kt == 3 ? strides[2] : 0, strides[kt == 3 ? 3 : 2], strides[kt == 3 ? 4 : 3]};
    hipLaunchKernelGGL(conv_split_prepare_kernel, dim3(c_out + c_in), dim3(256), 0, pcacc_stream(stream), w, c_out, c_in, kt, st, out_fwd,
                       scale_fwd, out_bwd, scale_bwd);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// ---- forward / data gradient --------------------------------------------------------------------------------------------------------
// TAPS = 9: the 3x3 (x kt frames) convolution.  TAPS = 1: a 1x1 product on the same tiles, the two halves of nn.ConvTranspose2d(kernel 2, stride 2)
// (models/unet.py:22-30,101-113):  mode CSP_UP  -- out[n, 2y+a, 2x+b, co] = bias[co] + sum_ci in[n,y,x,ci] W[ci,co,a,b]: prepared rows co' = (a, b, co),
// the epilogue scatters a pixel's 4 c_up results to its 2 x 2 output pixels;  mode CSP_S2D -- its data gradient: input channel k' = (a, b, co) of
// pixel (y, x) is dy[n, 2y+a, 2x+b, co] (read in place while staging), c_in' = 4 c_up.
#define CSP_PLAIN 0
#define CSP_UP 1
#define CSP_S2D 2

// relu == CSP_OUTMASK: no bias, no ReLU -- `bias` carries an fp32 map of the output's shape and the result is stored as zero where that map is
// <= 0 (aten::threshold_backward semantics: NaN keeps): the data gradient of conv -> ReLU -> conv masked for the first ReLU where it is stored
#define CSP_OUTMASK 2
__device__ __forceinline__ float4 csp_outmask4(float4 v, float4 m)
{
    return make_float4(m.x <= 0.f ? 0.f : v.x, m.y <= 0.f ? 0.f : v.y, m.z <= 0.f ? 0.f : v.z, m.w <= 0.f ? 0.f : v.w);
}

template <int CS, int NW, int NGW, int MT, int TAPS, bool MASKED>        // MASKED: in_mask != NULL, a compile-time fact (see conv3x3_split_res_kernel)
__global__ __launch_bounds__(CSP_THREADS) void conv3x3_split_kernel(const float *__restrict__ in, const float *__restrict__ in_amax,
                                                                    const float *__restrict__ in_mask, const uint16_t *__restrict__ wp,
                                                                    const float *__restrict__ wscale, const float *__restrict__ bias,
                                                                    float *__restrict__ out, float *__restrict__ out_amax, uint16_t *__restrict__ out16,
                                                                    int n_img, int frames,
                                                                    int h, int w, int c_in, int c_out, int kt, int relu, int rows, int bw,
                                                                    int tiles_y, int tiles_x, int co_groups, int mode, int c_up, int up_pitch,
                                                                    const float *__restrict__ in2, int c_a, int xcd)
{
    constexpr int PS = CS + 8;                                 // padded LDS row (elements)
    constexpr int MG = 8 / NGW;                                // waves along the pixel dimension
    constexpr int WROWS = 32 * NW * NGW;                       // output channels per workgroup
    constexpr int C8 = CS / 8;
    constexpr int WPL = WROWS * PS;                            // one weight plane of one buffer
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int pw = bw + 2, pp = (rows + 2) * pw;
    const int plane = pp * PS;
    uint16_t *patch = lds;                                     // [2 = hi, lo][pp][PS]
    uint16_t *wbuf = lds + 2 * (size_t)plane;                  // [2 buffers][2 = hi, lo][WROWS][PS]

    int bid = xcd ? pcacc_xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;   // the channel groups of a tile on one XCD: its input patch is fetched into one L2 once
    const int cog = bid % co_groups; bid /= co_groups;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int y0 = ty * rows, x0 = tx * bw;
    const int co0 = cog * WROWS;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int mg = wave % MG, ngw = wave / MG;
    const int n_px = rows * bw;
    const float sx = csp_scale_from_parts(in_amax);            // power-of-two scale of the input tensor

    // frame taps that exist for this image (uniform): a missing frame contributes zeros
    const int t_frame = img % frames;
    const int f_lo = (kt == 3 && t_frame == 0) ? 1 : 0;
    const int f_hi = kt == 3 ? (t_frame == frames - 1 ? 1 : 2) : 0;
    const int nc = c_in / CS;
    const int s0 = f_lo * nc, n_slices = (f_hi - f_lo + 1) * nc, n_taps = n_slices * TAPS;

    // the lane's pixels: LDS offset of the top-left tap of their 3x3 windows, image position for the store
    int poff[MT], pyx[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int q = (mg + MG * j) * 32 + lp;
        const int y = q / bw, x = q - y * bw;
        const bool ok = q < n_px && y0 + y < h && x0 + x < w;
        poff[j] = ok ? (y * pw + x) * PS : 0;
        pyx[j] = ok ? ((y0 + y) << 16 | (x0 + x)) : -1;
    }
    f32x16_t acc[MT][NW];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

    // patch chunks (8 channels) of this thread: position packed as py << 20 | px << 8 | c8 (0x7ff rows never pass the bounds test)
    const int n_chunks = pp * C8;
    int pinfo[CSP_PCH];
    float4 preg[CSP_PCH][2];
    int pok = 0;                                               // bit q: chunk q of preg lies inside the image
#pragma unroll
    for (int q = 0; q < CSP_PCH; ++q) {
        const int c = threadIdx.x + q * CSP_THREADS;
        const int px = c / C8, c8 = c - px * C8;
        const int py = px / pw, pxx = px - py * pw;
        pinfo[q] = c < n_chunks ? (py << 20 | pxx << 8 | c8) : (0x7ff << 20);
    }
    auto fetch_patch = [&](int sl) {                           // sl = index into the valid slices
        const int s = s0 + sl, f = s / nc, cs = s - f * nc;
        // two inputs (in2 != NULL: the decoder's cat(up, skip), models/unet.py:101-113, read in place): channels [0, c_a) of a pixel live in `in`,
        // [c_a, c_in) in `in2`; a slice never straddles the boundary (the planner keeps CS a divisor of c_a)
        const float *src = in;
        int pitch = c_in, ch0p = cs * CS;
        if (in2) {
            if (ch0p >= c_a) { src = in2; pitch = c_in - c_a; ch0p -= c_a; } else pitch = c_a;
        }
        const int64_t img_off = (int64_t)(img + (kt == 3 ? f - 1 : 0)) * h * w * pitch;
        // space-to-depth source (CSP_S2D): the slice's channels (a, b, ch ..) live in pixel (2y + a, 2x + b) of the [2h, 2w, c_up] map
        const int ab = mode == CSP_S2D ? (cs * CS) / c_up : 0, ch0 = mode == CSP_S2D ? (cs * CS) % c_up : 0;
        // [r5] every load of the slice is issued before anything reads a loaded value; the "inside the image" select happens in write_patch, a slice later
        // (pok carries the flags) -- see conv3x3_split_res_kernel: a select behind each pair of loads made the fetch a chain of memory round trips
        int64_t offs[CSP_PCH];
        int okb = 0;
#pragma unroll
        for (int q = 0; q < CSP_PCH; ++q) {
            const int py = pinfo[q] >> 20, pxx = (pinfo[q] >> 8) & 0xfff, c8 = pinfo[q] & 0xff;
            const int y = y0 - 1 + py, x = x0 - 1 + pxx;
            const bool ok = (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w;
            const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
            offs[q] = mode == CSP_S2D
                          ? (((int64_t)img * 2 * h + 2 * yc + (ab >> 1)) * 2 * w + 2 * xc + (ab & 1)) * c_up + ch0 + c8 * 8
                          : img_off + ((int64_t)yc * w + xc) * pitch + ch0p + c8 * 8;
            okb |= ok ? (1 << q) : 0;
        }
#pragma unroll
        for (int q = 0; q < CSP_PCH; ++q) {
            preg[q][0] = csp_load_in4(src + offs[q]);
            preg[q][1] = csp_load_in4(src + offs[q] + 4);
        }
        if constexpr (MASKED) {
#pragma unroll
            for (int q = 0; q < CSP_PCH; ++q) {
                preg[q][0] = csp_relu_mask4(preg[q][0], *reinterpret_cast<const float4 *>(in_mask + offs[q]));
                preg[q][1] = csp_relu_mask4(preg[q][1], *reinterpret_cast<const float4 *>(in_mask + offs[q] + 4));
            }
        }
        pok = okb;
    };
    auto write_patch = [&]() {
#pragma unroll
        for (int q = 0; q < CSP_PCH; ++q) {
            const int c = threadIdx.x + q * CSP_THREADS;
            if (c < n_chunks) {
                uint4 hi, lo;
                const bool ok = (pok >> q) & 1;                  // outside the image: zeros (the load came from a clamped position)
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                csp_split8(ok ? preg[q][0] : z, ok ? preg[q][1] : z, sx, hi, lo);
                uint16_t *dst = patch + (c / C8) * PS + (c % C8) * 8;
                *reinterpret_cast<uint4 *>(dst) = hi;
                *reinterpret_cast<uint4 *>(dst + plane) = lo;
            }
        }
    };
    // weight tile of (slice, tap): 2 planes x WROWS rows x CS elements.  Requested TWO taps ahead into alternating register sets, written
    // to the other LDS buffer one tap ahead.
    constexpr int W_CH_PLANE = WROWS * C8, W_CHUNKS = 2 * W_CH_PLANE, W_PER = (W_CHUNKS + CSP_THREADS - 1) / CSP_THREADS;
    const int64_t w_plane = (int64_t)kt * TAPS * c_out * c_in; // elements of one prepared plane
    uint4 wreg[2][W_PER];
    auto fetch_w = [&](int set, int g) {                       // g = linear tap index over the valid slices
        const int sl = g / TAPS, tap = g - sl * TAPS;
        const int s = s0 + sl, f = s / nc, cs = s - f * nc;
        const uint16_t *src = wp + ((int64_t)(f * TAPS + tap) * c_out + co0) * c_in + cs * CS;
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = (W_CHUNKS % CSP_THREADS) ? min((int)threadIdx.x + q * CSP_THREADS, W_CHUNKS - 1) : threadIdx.x + q * CSP_THREADS;
            const int p = c / W_CH_PLANE, r = c - p * W_CH_PLANE;
            wreg[set][q] = *reinterpret_cast<const uint4 *>(src + p * w_plane + (int64_t)(r / C8) * c_in + (r % C8) * 8);   // clamped, never masked
#ifdef PCACC_X3_EXPERIMENT
            wreg[set][q] = csp_x_weight(wreg[set][q], p);
#endif
        }
    };
    auto write_w = [&](uint16_t *dst, int set) {
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = threadIdx.x + q * CSP_THREADS;
            if (c < W_CHUNKS) {
                const int p = c / W_CH_PLANE, r = c - p * W_CH_PLANE;
                *reinterpret_cast<uint4 *>(dst + p * WPL + (r / C8) * PS + (r % C8) * 8) = wreg[set][q];
            }
        }
    };

    fetch_patch(0);
    fetch_w(0, 0);
    if (n_taps > 1) fetch_w(1, 1);
    uint16_t *wb0 = wbuf, *wb1 = wbuf + 2 * WPL;               // buffer of the even / odd taps of the current slice
    write_w(wb0, 0);                                           // tap 0 (nobody reads LDS yet)
    for (int sl = 0; sl < n_slices; ++sl) {
        __syncthreads();                                       // every wave is done with the previous slice's patch
        write_patch();
        if (sl + 1 < n_slices) fetch_patch(sl + 1);            // in flight during the nine taps below
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int g = sl * TAPS + tap;
            uint16_t *cur = (tap & 1) ? wb1 : wb0, *other = (tap & 1) ? wb0 : wb1;
            __syncthreads();                                   // buffer `cur` (and, at tap 0, the patch) is visible
            if (g + 2 < n_taps) fetch_w(tap & 1, g + 2);       // set (tap & 1) held tap g: already in LDS
            const uint16_t *wa = cur + (ngw * NW * 32 + lp) * PS + lh * 8;
            const int toff = (TAPS == 1 ? pw + 1 : (tap / 3) * pw + tap % 3) * PS + lh * 8;     // one tap: the pixel itself
            constexpr int KC = CS / 16;
            constexpr int FB = (MT * NW >= 6 || MT >= 3) ? 1 : 2;   // fragment sets: the widest waves have no registers for a second one
            f16x8_t ah[FB][NW], al[FB][NW], bh[FB][MT], bl[FB][MT];
            auto load = [&](int slot, int kc) {
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    ah[slot][n] = *reinterpret_cast<const f16x8_t *>(wa + n * 32 * PS + kc * 16);
                    al[slot][n] = *reinterpret_cast<const f16x8_t *>(wa + WPL + n * 32 * PS + kc * 16);
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    bh[slot][j] = *reinterpret_cast<const f16x8_t *>(patch + poff[j] + toff + kc * 16);
                    bl[slot][j] = *reinterpret_cast<const f16x8_t *>(patch + plane + poff[j] + toff + kc * 16);
                }
            };
            if (FB == 2) load(0, 0);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                if (FB == 1) load(0, kc);
                else if (kc + 1 < KC) load((kc + 1) & 1, kc + 1);   // fragments of the next step in flight under this step's MFMAs
                constexpr int M = FB - 1;
                // small terms first; consecutive MFMAs go to different accumulators
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kc & M][n], bl[kc & M][j], acc[j][n], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kc & M][n], bh[kc & M][j], acc[j][n], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kc & M][n], bh[kc & M][j], acc[j][n], 0, 0, 0);
            }
            if (g + 1 < n_taps) write_w(other, (tap + 1) & 1);   // tap g + 1 (requested two taps ago) into the buffer tap g - 1 used
        }
        // nine taps (or one) per slice -- an odd number: the next slice's tap 0 sits in register set 1 / goes to the odd buffer: swap the roles
        {
            uint16_t *t = wb0; wb0 = wb1; wb1 = t;
#pragma unroll
            for (int q = 0; q < W_PER; ++q) { const uint4 v = wreg[0][q]; wreg[0][q] = wreg[1][q]; wreg[1][q] = v; }
        }
    }

    // epilogue: lane = pixel, register quad g of tile n = channels n*32 + 8g + 4*lh .. +3 of this wave's NW * 32
    const int cw0 = co0 + ngw * NW * 32;
    const float inv_sx = 1.f / sx;
    float omax = 0.f;                                          // |output| maximum of this lane (NaN -> inf), for the consumer's scale
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        if (pyx[j] < 0) continue;
        float *dst = out + (((int64_t)img * h + (pyx[j] >> 16)) * w + (pyx[j] & 0xffff)) * c_out + cw0;
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n * 32 + 8 * g + 4 * lh;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                int cb = cw0 + c;                              // bias index; CSP_UP: channel (a, b, co) -> pixel (2y + a, 2x + b), channel co
                if (mode == CSP_UP) {
                    const int ab2 = cb / c_up;
                    cb -= ab2 * c_up;
                    // up_pitch: elements between consecutive pixels of the [2h, 2w] result (c_up, or the width of the decoder's concatenation
                    // buffer the result is written into, models/unet.py:101-113)
                    dst = out + (((int64_t)img * 2 * h + 2 * (pyx[j] >> 16) + (ab2 >> 1)) * 2 * w + 2 * (pyx[j] & 0xffff) + (ab2 & 1)) * up_pitch + cb - c;
                }
                if (bias && relu != CSP_OUTMASK) bv = *reinterpret_cast<const float4 *>(bias + cb);
                float4 sc = *reinterpret_cast<const float4 *>(wscale + cw0 + c);          // 1 / t per output channel
                sc = make_float4(sc.x * inv_sx, sc.y * inv_sx, sc.z * inv_sx, sc.w * inv_sx);
                float4 v = make_float4(acc[j][n][4 * g] * sc.x + bv.x, acc[j][n][4 * g + 1] * sc.y + bv.y, acc[j][n][4 * g + 2] * sc.z + bv.z,
                                       acc[j][n][4 * g + 3] * sc.w + bv.w);
                if (relu == CSP_OUTMASK) v = csp_outmask4(v, *reinterpret_cast<const float4 *>(bias + ((dst + c) - out)));
                else if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                *reinterpret_cast<float4 *>(dst + c) = v;
                if (out16)                                     // 'mixed' mode: the bf16 shadow of the result (same element offsets), for the bf16 backward
                    csp_store_shadow(out16 + ((dst + c) - out), pcacc_pack_bf16x2(v.x, v.y), pcacc_pack_bf16x2(v.z, v.w));
                omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) omax = __builtin_inff();
            }
    }
    if (out_amax) {                                            // uniform: one atomic per wave into one of 256 slots (zeroed by the caller)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) omax = fmaxf(omax, __shfl_xor(omax, d, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(out_amax) + (blockIdx.x & (CSP_AMAX_PARTS - 1)), __float_as_uint(omax));
    }
}

// ---- the narrow layers (c_in, c_out <= 64) on full-resolution maps: persistent workgroups, a slice's nine weight tiles resident ----------
// These layers move the most pixels (20 x 288^2) with the fewest MFMAs per pixel; in the kernel above they paid a barrier per tap for 12
// MFMAs per wave, and -- one slice per tile, one tile per workgroup -- nothing overlapped a tile's global loads with another's math.
// Here a workgroup walks tiles (stride = workgroups per channel group); a pass = (tile, slice): the NEXT pass's patch travels in
// registers during the current pass's MFMAs, the nine tap tiles of the pass's slice sit in LDS (re-staged only when the slice changes:
// never for a one-slice layer), and the 9 x CS/16 steps of a pass run without a barrier, fragment reads two steps ahead of their MFMAs.
// 8 waves along the pixels (NGW = 1); a wave owns MT pixel tiles x NW channel tiles.
#ifndef CSR_EXP
#define CSR_EXP 0                               // phase knock-outs for tools/exp_conv_res_phases.sh (1 MFMA loop, 2 patch staging, 4 stores, 8 patch loads, 16 weight staging)
#endif
#define CSR_WPT 9                               // 16-byte weight pieces a thread carries for the next slice (9 taps x 64 rows x 32 ch x 2 planes / 512)
// [r6] experiment CSR_LINES = 1: whole-line stores (VERDICT round 5, item 1's second half).  As first written the weights were the MFMA's A operand and the pixels its B operand: a lane ended up with ONE pixel's
// channels 8 g + 4 lh + 0..3 -- a store instruction wrote 32-byte pieces (two lanes) of 32 different lines, the shadow 16-byte pieces, four instructions to
// complete a line (4 096 write requests per 512-pixel tile).  With the operands exchanged (the accumulator holds the transposed tile: same products, same
// order of additions, bit-identical) a lane has one CHANNEL of pixels 8 g + 4 lh + 0..3; a 4 x 4 transpose inside each lane quad (two quad_perm exchanges)
// gives quad lane k the pixel 8 g + 4 lh + k with channels 4 Q .. 4 Q + 3: eight quads write one pixel's 128-byte row, a store instruction eight whole lines
// (the shadow: whole 64-byte rows, neighbours in memory) -- 1 024 requests per tile.
// MEASURED (profiles/r06_conv_line_stores_ab.txt, two interleaved rounds alone + three interleaved pairs of the whole step, one box): bit-identical results;
// alone 32 -> 32 182 vs 184 us, 64 -> 32 277 vs 284, 27 taps 316 vs 318, 32 -> 64 294 vs 320 us; in the step 30.12 / 30.05 / 30.00 ms against 29.87 / 29.87 /
// 29.94 ms with the partial-line stores.  The write-request count is NOT what holds these layers at 3.5 TB/s (the L2 merges the pieces before they reach HBM:
// counter traffic was 1.0 x before as well); the partial-line form stays the default, this one is kept behind the macro for the next reader of that idea.
#ifndef CSR_LINES
#define CSR_LINES 0
#endif
__device__ __forceinline__ float csr_quad_xchg1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }   // lane ^ 1
__device__ __forceinline__ float csr_quad_xchg2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); }   // lane ^ 2
// quad lane k holds column k of a 4 x 4 block (a0 .. a3 = its rows 0 .. 3) -> row k of the block
__device__ __forceinline__ float4 csr_quad_transpose(float a0, float a1, float a2, float a3, int k)
{
    const bool odd = k & 1, hi = k & 2;
    const float r01 = csr_quad_xchg1(odd ? a0 : a1), r23 = csr_quad_xchg1(odd ? a2 : a3);
    const float b0 = odd ? r01 : a0, b1 = odd ? a1 : r01;      // rows (k & 1) and 2 + (k & 1), columns (k & ~1), (k & ~1) + 1
    const float b2 = odd ? r23 : a2, b3 = odd ? a3 : r23;
    const float s0 = csr_quad_xchg2(hi ? b0 : b2), s1 = csr_quad_xchg2(hi ? b1 : b3);
    return hi ? make_float4(s0, s1, b2, b3) : make_float4(b0, b1, s0, s1);
}

template <int CS, int NW, int MT, int PCH, bool RESTAGE, bool MASKED>          // MASKED: in_mask != NULL (a compile-time fact: a run-time branch behind the loads made the compiler copy -- and so wait for -- every loaded register at once)
__global__ __launch_bounds__(CSP_THREADS) void conv3x3_split_res_kernel(const float *__restrict__ in, const float *__restrict__ in_amax,
                                                                        const float *__restrict__ in_mask, const uint16_t *__restrict__ wp,
                                                                        const float *__restrict__ wscale, const float *__restrict__ bias,
                                                                        float *__restrict__ out, float *__restrict__ out_amax, uint16_t *__restrict__ out16, int n_img, int frames,
                                                                        int h, int w, int c_in, int c_out, int kt, int relu, int rows, int bw,
                                                                        int tiles_y, int tiles_x, int co_groups, int slots,
                                                                        const float *__restrict__ in2, int c_a)
{
    constexpr int PS = CS + 8;
    constexpr int WROWS = 32 * NW;
    constexpr int C8 = CS / 8;
    constexpr int WPL = 9 * WROWS * PS;                        // one weight plane (nine taps)
    constexpr int KC = CS / 16, STEPS = 9 * KC, AHEAD = 1;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int pw = bw + 2, pp = (rows + 2) * pw;
    const int plane = pp * PS;
    uint16_t *patch = lds;                                     // [2 = hi, lo][pp][PS]
    uint16_t *wl = lds + 2 * (size_t)plane;                    // [2 = hi, lo][9][WROWS][PS]
    // [r5] per-channel scale (weight scale / input scale) and bias of this workgroup's WROWS output channels, staged ONCE: the epilogue read them from global
    // memory for every (tile, channel group) and each of those loads was followed by s_waitcnt vmcnt(0) -- which also waits for every store issued before
    // it (loads and stores retire in order on one counter): 16 memory round trips per pass in series, and the stores never overlapped anything
    float *sb = reinterpret_cast<float *>(wl + 2 * WPL);       // [2][WROWS]: scale, bias
    int *ptab = reinterpret_cast<int *>(sb + 2 * WROWS);       // [8 MT 32] (CSR_LINES): y << 16 | x of the tile's q-th pixel, -1 behind its last

    const int cog = blockIdx.x % co_groups, slot = blockIdx.x / co_groups;
    const int co0 = cog * WROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int n_px = rows * bw;
    const int nc = c_in / CS;
    const int tiles_img = tiles_y * tiles_x, n_tiles = n_img * tiles_img;
    const float sx = csp_scale_from_parts(in_amax);
    const float inv_sx = 1.f / sx;
    if ((int)threadIdx.x < WROWS) {                            // published by the first pass's barriers
        sb[threadIdx.x] = wscale[blockIdx.x % co_groups * WROWS + threadIdx.x] * inv_sx;
        sb[WROWS + threadIdx.x] = (bias && relu != CSP_OUTMASK) ? bias[blockIdx.x % co_groups * WROWS + threadIdx.x] : 0.f;
    }
    if (CSR_LINES)
        for (int q = threadIdx.x; q < 8 * MT * 32; q += CSP_THREADS) ptab[q] = q < n_px ? ((q / bw) << 16 | (q % bw)) : -1;

    // slices of a tile that exist: frame taps f_lo .. f_hi (a missing frame contributes zeros), all channel slices of each
    // (tile coordinates come from the walker, frame tap f and channel slice cs of a slice s = f * nc + cs are carried along: no divisions per pass)
    auto first_slice = [&](const ConvTile &t) { return (kt == 3 && t.fr == 0) ? nc : 0; };
    auto end_slice = [&](const ConvTile &t) { return kt == 3 ? (t.fr == frames - 1 ? 2 * nc : 3 * nc) : nc; };
    ConvTileWalk walk;                                         // with frame taps the frame runs fastest: a frame's three reads meet in the L2
    walk.init(0, n_tiles, slot, slots, tiles_y, tiles_x, frames, kt == 3, rows, bw);

    // patch chunks (8 channels) of this thread: position packed as py << 20 | px << 8 | c8
    const int n_chunks = pp * C8;
    int pinfo[PCH];                                            // PCH: patch chunks a thread carries (3 .. 6 by configuration: registers)
    float4 preg[PCH][2];
    int pok = 0;                                               // bit q: chunk q of preg lies inside the image
#pragma unroll
    for (int q = 0; q < PCH; ++q) {
        const int c = threadIdx.x + q * CSP_THREADS;
        const int px = c / C8, c8 = c - px * C8;
        const int py = px / pw, pxx = px - py * pw;
        pinfo[q] = c < n_chunks ? (py << 20 | pxx << 8 | c8) : (0x7ff << 20);
    }
    auto fetch_patch = [&](const ConvTile &t, int f, int cs) __attribute__((always_inline)) {
        const int img = t.img, y0 = t.y0, x0 = t.x0;
        const float *src = in;                                 // two inputs: see conv3x3_split_kernel
        int pitch = c_in, ch0p = cs * CS;
        if (in2) {
            if (ch0p >= c_a) { src = in2; pitch = c_in - c_a; ch0p -= c_a; } else pitch = c_a;
        }
        const int64_t img_off = (int64_t)(img + (kt == 3 ? f - 1 : 0)) * h * w * pitch;
        // [r5] ALL loads of the pass are issued before anything reads a loaded value: the "inside the image" select moved to write_patch (one pass later, pok
        // carries the flags).  As first written every chunk's two loads were followed by their selects -- and the uniform branches between the chunks kept the
        // compiler from hoisting the loads over them: issue 2 loads, s_waitcnt vmcnt(0), five times per pass, each a full memory round trip in series
        // (the ISA of the round-4 kernel; it is what the "loads + staging" phase of the knock-out measurements was made of).
        int64_t offs[PCH];
        int okb = 0;
#pragma unroll
        for (int q = 0; q < PCH; ++q) {
            const int py = pinfo[q] >> 20, pxx = (pinfo[q] >> 8) & 0xfff, c8 = pinfo[q] & 0xff;
            const int y = y0 - 1 + py, x = x0 - 1 + pxx;
            const bool ok = (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w;
            const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
            offs[q] = img_off + ((int64_t)yc * w + xc) * pitch + ch0p + c8 * 8;
            okb |= ok ? (1 << q) : 0;
        }
#pragma unroll
        for (int q = 0; q < PCH; ++q) {
            preg[q][0] = csp_load_in4(src + offs[q]);
            preg[q][1] = csp_load_in4(src + offs[q] + 4);
        }
        if constexpr (MASKED) {                                // the masked data gradient of the fp32x3 backward (not on the mixed mode's path)
#pragma unroll
            for (int q = 0; q < PCH; ++q) {
                preg[q][0] = csp_relu_mask4(preg[q][0], *reinterpret_cast<const float4 *>(in_mask + offs[q]));
                preg[q][1] = csp_relu_mask4(preg[q][1], *reinterpret_cast<const float4 *>(in_mask + offs[q] + 4));
            }
        }
        pok = okb;
    };
    auto write_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PCH; ++q) {
            const int c = threadIdx.x + q * CSP_THREADS;
            if (c < n_chunks) {
                uint4 hi, lo;
                const bool ok = (pok >> q) & 1;                  // outside the image: zeros (the load came from a clamped position)
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                csp_split8(ok ? preg[q][0] : z, ok ? preg[q][1] : z, sx, hi, lo);
                uint16_t *dst = patch + (c / C8) * PS + (c % C8) * 8;
                *reinterpret_cast<uint4 *>(dst) = hi;
                *reinterpret_cast<uint4 *>(dst + plane) = lo;
            }
        }
    };
    // the nine tap tiles of slice s, both planes: 2 x 9 x WROWS rows of CS elements
    constexpr int W_ROWS_ALL = 2 * 9 * WROWS, W_CHUNKS = W_ROWS_ALL * C8, W_PER = (W_CHUNKS + CSP_THREADS - 1) / CSP_THREADS;
    static_assert(W_PER <= CSR_WPT, "weight pieces per thread");
    const int64_t w_plane = (int64_t)kt * 9 * c_out * c_in;
    typedef uint32_t wvec_t __attribute__((ext_vector_type(4 * W_PER)));      // one SSA value, not an array: as `uint4 wreg[W_PER]` it stayed in scratch memory
    wvec_t wreg;
    auto fetch_w = [&](int f, int cs) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = min((int)threadIdx.x + q * CSP_THREADS, W_CHUNKS - 1);
            const int row = c / C8, c8 = c - row * C8;         // row = (plane, tap, r)
            const int pl = row / (9 * WROWS), tr = row - pl * 9 * WROWS;
            const int tap = tr / WROWS, r = tr - tap * WROWS;
            uint4 v = *reinterpret_cast<const uint4 *>(wp + pl * w_plane + ((int64_t)(f * 9 + tap) * c_out + co0 + r) * c_in + cs * CS + c8 * 8);
#ifdef PCACC_X3_EXPERIMENT
            v = csp_x_weight(v, pl);
#endif
            wreg[4 * q] = v.x; wreg[4 * q + 1] = v.y; wreg[4 * q + 2] = v.z; wreg[4 * q + 3] = v.w;
        }
    };
    auto write_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = threadIdx.x + q * CSP_THREADS;
            if (c < W_CHUNKS) *reinterpret_cast<uint4 *>(wl + (c / C8) * PS + (c % C8) * 8) = make_uint4(wreg[4 * q], wreg[4 * q + 1], wreg[4 * q + 2], wreg[4 * q + 3]);
        }
    };

    float omax = 0.f;
    const int n_mine = walk.count;
    if (n_mine == 0) return;
    int k = 0;
    ConvTile cur = walk.get(0), nxt = cur;
    int s = first_slice(cur), f = s ? 1 : 0, cs = 0;
    fetch_patch(cur, f, cs);
    fetch_w(f, cs);
    if (!RESTAGE) write_w();                                   // a one-slice layer (RESTAGE false): its nine tap tiles are staged once; the first pass's barriers publish them
    f32x16_t acc[MT][NW];
    // a finished tile's values wait in registers and leave one pass later, right after the next patch loads are issued: the wait for those
    // loads at the top of a pass (vmcnt counts stores too) then finds the stores a whole MFMA phase old instead of just issued -- as
    // written before, load burst, MFMA phase and store burst took turns (57 + 53 + 71 us of a 181 us layer, measured by knocking each out)
    constexpr bool DEFER = !RESTAGE;
    float4 pend[MT][NW][4];
    int pend_pyx[MT], pend_img = 0, pend_y0 = 0, pend_x0 = 0;
    bool have_pend = false;
    // the image position of the pixel this lane stores for pixel tile j, group g (CSR_LINES): -1 outside the tile or the image
    auto line_pyx = [&](int j, int g, int y0, int x0) __attribute__((always_inline)) {
        const int t = ptab[(wave + 8 * j) * 32 + 8 * g + 4 * lh + (lp & 3)];
        const int y = y0 + (t >> 16), x = x0 + (t & 0xffff);
        return (t >= 0 && y < h && x < w) ? (y << 16 | x) : -1;
    };
    auto flush = [&]() __attribute__((always_inline)) {
        if (CSR_LINES) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int yx = line_pyx(j, g, pend_y0, pend_x0);
                    if (yx < 0) continue;
                    const int64_t o = (((int64_t)pend_img * h + (yx >> 16)) * w + (yx & 0xffff)) * c_out + co0 + 4 * (lp >> 2);
#pragma unroll
                    for (int n = 0; n < NW; ++n) {
                        const float4 v = pend[j][n][g];
                        *reinterpret_cast<float4 *>(out + o + n * 32) = v;
                        if (out16) csp_store_shadow(out16 + o + n * 32, pcacc_pack_bf16x2(v.x, v.y), pcacc_pack_bf16x2(v.z, v.w));
                    }
                }
            return;
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            if (pend_pyx[j] < 0) continue;
            float *dst = out + (((int64_t)pend_img * h + (pend_pyx[j] >> 16)) * w + (pend_pyx[j] & 0xffff)) * c_out + co0;
#pragma unroll
            for (int n = 0; n < NW; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = pend[j][n][g];
                    *reinterpret_cast<float4 *>(dst + n * 32 + 8 * g + 4 * lh) = v;
                    if (out16)                                 // 'mixed' mode: the bf16 shadow of the result, same element offsets
                        csp_store_shadow(out16 + ((dst + n * 32 + 8 * g + 4 * lh) - out), pcacc_pack_bf16x2(v.x, v.y), pcacc_pack_bf16x2(v.z, v.w));
                }
        }
    };
    while (k < n_mine) {
        if (s == first_slice(cur)) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int n = 0; n < NW; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;
        }
        __syncthreads();                                       // the previous pass is done with the patch and the weight tiles
        if (!(CSR_EXP & 2)) write_patch();
        if (RESTAGE && !(CSR_EXP & 16)) write_w();             // unconditionally (46 - 92 KB of LDS writes against ~1 MB of fragment reads per pass):
        __syncthreads();                                       // under a condition the compiler keeps the pieces in scratch memory
        // the next pass: its patch and its weight tiles in flight during this pass's MFMAs
        int nk = k, ns = s + 1, nf = f, ncs = cs + 1;
        nxt = cur;
        if (ncs == nc) { ncs = 0; ++nf; }
        if (ns >= end_slice(cur)) {
            nk = k + 1;
            if (nk < n_mine) { nxt = walk.get(nk); ns = first_slice(nxt); nf = ns ? 1 : 0; ncs = 0; }
            else { ns = s; nf = f; ncs = cs; }
        }
        if (nk < n_mine && !(CSR_EXP & 8)) fetch_patch(nxt, nf, ncs);
        if (RESTAGE && !(CSR_EXP & 16)) fetch_w(nf, ncs);
        if (have_pend && !(CSR_EXP & 4)) flush();
        have_pend = false;

        // the lane's pixels of this tile
        const int img = cur.img, y0 = cur.y0, x0 = cur.x0;
        int poff[MT], pyx[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int q = (wave + 8 * j) * 32 + lp;
            const int y = q / bw, x = q - y * bw;
            const bool ok = q < n_px && y0 + y < h && x0 + x < w;
            poff[j] = ok ? (y * pw + x) * PS : 0;
            pyx[j] = ok ? ((y0 + y) << 16 | (x0 + x)) : -1;
        }
        if (!(CSR_EXP & 1)) {
            f16x8_t ah[AHEAD + 1][NW], al[AHEAD + 1][NW], bh[AHEAD + 1][MT], bl[AHEAD + 1][MT];
            const uint16_t *wa = wl + lp * PS + lh * 8;
            auto load = [&](int slt, int st) __attribute__((always_inline)) {
                const int tap = st / KC, kc = st - tap * KC;
                const int toff = ((tap / 3) * pw + tap % 3) * PS + lh * 8 + kc * 16;
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    ah[slt][n] = *reinterpret_cast<const f16x8_t *>(wa + (tap * WROWS + n * 32) * PS + kc * 16);
                    al[slt][n] = *reinterpret_cast<const f16x8_t *>(wa + WPL + (tap * WROWS + n * 32) * PS + kc * 16);
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    bh[slt][j] = *reinterpret_cast<const f16x8_t *>(patch + poff[j] + toff);
                    bl[slt][j] = *reinterpret_cast<const f16x8_t *>(patch + plane + poff[j] + toff);
                }
            };
#pragma unroll
            for (int st = 0; st < AHEAD && st < STEPS; ++st) load(st, st);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                if (st + AHEAD < STEPS) load((st + AHEAD) % (AHEAD + 1), st + AHEAD);
                const int c = st % (AHEAD + 1);
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[j][n] = CSR_LINES ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[c][j], ah[c][n], acc[j][n], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c][n], bl[c][j], acc[j][n], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[j][n] = CSR_LINES ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[c][j], al[c][n], acc[j][n], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c][n], bh[c][j], acc[j][n], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[j][n] = CSR_LINES ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[c][j], ah[c][n], acc[j][n], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c][n], bh[c][j], acc[j][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nk != k && CSR_LINES) {                            // last slice of this tile: a lane holds channel lp of pixels 8 g + 4 lh + 0..3 -- scales off, bias, ReLU, then the quad transpose
            pend_img = img;
            pend_y0 = y0;
            pend_x0 = x0;
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const float sc = sb[n * 32 + lp], bv = sb[WROWS + n * 32 + lp];
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float a[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            a[i] = acc[j][n][4 * g + i] * sc + bv;
                            if (relu && relu != CSP_OUTMASK) a[i] = fmaxf(a[i], 0.f);
                        }
                        float4 v = csr_quad_transpose(a[0], a[1], a[2], a[3], lp & 3);
                        bool ok = true;                        // a pixel outside the tile or the image repeats the tile's first pixel: it cannot raise the maximum ...
                        if (relu == CSP_OUTMASK) {             // ... unless that pixel's own result is masked away
                            const int yx = line_pyx(j, g, y0, x0);
                            ok = yx >= 0;
                            if (ok) v = csp_outmask4(v, *reinterpret_cast<const float4 *>(bias + (((int64_t)img * h + (yx >> 16)) * w + (yx & 0xffff)) * c_out + co0 + n * 32 + 4 * (lp >> 2)));
                        }
                        pend[j][n][g] = v;
                        if (ok) {
                            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                            if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) omax = __builtin_inff();
                        }
                    }
            }
            have_pend = true;
            if (!DEFER) {
                if (!(CSR_EXP & 4)) flush();
                have_pend = false;
            }
        }
        if (nk != k && !CSR_LINES) {                           // last slice of this tile: scales off, bias, ReLU -> the pending registers
            pend_img = img;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                pend_pyx[j] = pyx[j];
#pragma unroll
                for (int n = 0; n < NW; ++n)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = n * 32 + 8 * g + 4 * lh;
                        const float4 bv = *reinterpret_cast<const float4 *>(sb + WROWS + c);
                        const float4 sc = *reinterpret_cast<const float4 *>(sb + c);
                        float4 v = make_float4(acc[j][n][4 * g] * sc.x + bv.x, acc[j][n][4 * g + 1] * sc.y + bv.y, acc[j][n][4 * g + 2] * sc.z + bv.z,
                                               acc[j][n][4 * g + 3] * sc.w + bv.w);
                        if (relu == CSP_OUTMASK) {
                            if (pyx[j] >= 0)
                                v = csp_outmask4(v, *reinterpret_cast<const float4 *>(bias + (((int64_t)img * h + (pyx[j] >> 16)) * w + (pyx[j] & 0xffff)) * c_out + co0 + c));
                        } else if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                        pend[j][n][g] = v;
                        if (pyx[j] >= 0) {
                            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                            if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) omax = __builtin_inff();
                        }
                    }
            }
            have_pend = true;
            if (!DEFER) {                                      // the re-staging variants have no registers to spare for the delay
                if (!(CSR_EXP & 4)) flush();
                have_pend = false;
            }
        }
        k = nk;
        s = ns;
        f = nf;
        cs = ncs;
        cur = nxt;
    }
    if (have_pend && !(CSR_EXP & 4)) flush();
    if (out_amax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) omax = fmaxf(omax, __shfl_xor(omax, d, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(out_amax) + (blockIdx.x & (CSP_AMAX_PARTS - 1)), __float_as_uint(omax));
    }
}

struct ConvResPlan { int cs, nw, mt, rows, bw, tiles_y, tiles_x, co_groups, slots; size_t lds; };

// the four configurations that fit 160 KB of LDS with a slice's nine tap tiles resident: (CS, NW, MT) -> patch chunks per thread
static int conv_res_pch(int cs, int nw, int mt)
{
    if (cs == 32 && mt == 1) return 3;                         // 32 -> 32 / 64, 256-pixel tiles
    if (cs == 32 && nw == 1 && mt == 2) return 5;              // 32 -> 32, 512-pixel tiles
    if (cs == 64 && nw == 1 && mt == 1) return 6;              // 64 -> 32
    return 0;
}

static bool conv_res_fits(int cs, int nw, int mt, int rows, int bw, size_t *lds)
{
    const int64_t pp = (int64_t)(rows + 2) * (bw + 2);
    const int wrows = 32 * nw, pch = conv_res_pch(cs, nw, mt);
    *lds = (size_t)(2 * pp + 2 * 9 * wrows) * (cs + 8) * sizeof(uint16_t) + 2 * wrows * sizeof(float) +    // + scale / bias of the workgroup's channels
           (CSR_LINES ? 8 * mt * 32 * sizeof(int) : 0);                                                      // + the tile's pixel positions
    return pch > 0 && pp * (cs / 8) <= CSP_THREADS * pch && *lds <= CSP_LDS_MAX && 2 * 9 * wrows * (cs / 8) <= CSP_THREADS * CSR_WPT &&
           rows + 2 < 0x7ff && bw + 2 < 0xfff;
}

// c_in, c_out in {32, 64} on maps of at least a few thousand pixels (below that the launch is latency, not throughput)
static bool conv_res_plan(int n_img, int h, int w, int c_in, int c_out, ConvResPlan *best, int cs_must_divide = 0)
{
    const char e = pcacc_switches().conv_res;                 // '0': never, '2': whatever the size (tests)
    if ((c_in != 32 && c_in != 64) || (c_out != 32 && c_out != 64) || e == '0') return false;
    if ((int64_t)n_img * h * w < 200000 && e != '2') return false;
    const int nw = c_out / 32;
    bool found = false;
    int64_t best_cost = 0;
    for (int mt = 2; mt >= 1; --mt) {
        const int cap = 8 * mt * 32;
        for (int cs = 64; cs >= 32; cs -= 32) {
            if (c_in % cs || !conv_res_pch(cs, nw, mt) || (cs_must_divide && cs_must_divide % cs)) continue;
            for (int bi = 0; bi < 4; ++bi) {
                int bw = bi == 0 ? w : 32 * bi;
                if (bi > 0 && bw >= w) continue;
                if (bw > cap) continue;
                const int tiles_x = (w + bw - 1) / bw;
                bw = (w + tiles_x - 1) / tiles_x;
                int r = cap / bw;
                if (r > h) r = h;
                size_t lds;
                while (r >= 1 && !conv_res_fits(cs, nw, mt, r, bw, &lds)) --r;
                if (r < 1) continue;
                const int tiles_y = (h + r - 1) / r;
                r = (h + tiles_y - 1) / tiles_y;
                if (!conv_res_fits(cs, nw, mt, r, bw, &lds)) continue;
                const int tiles = (r * bw + 31) / 32;
                if (mt > 1 && tiles <= 8 * (mt - 1)) continue;
                // per tile: MFMAs of all slices (3 per fragment pair), one barrier pair + staging per slice; padded pixel tiles are wasted MFMAs
                const int64_t per_tile = (int64_t)(c_in / cs) * (9 * (cs / 16) * 3 * nw * mt + 12 + (int64_t)(r + 2) * (bw + 2) * cs / 512 +
                                                                 9 * 32 * nw * cs / 512);
                const int64_t cost = per_tile * tiles_y * tiles_x;
                if (!found || cost < best_cost) {
                    found = true;
                    best_cost = cost;
                    int64_t slots = PCACC_CUS;
                    const int64_t n_tiles = (int64_t)n_img * tiles_y * tiles_x;
                    if (slots > n_tiles) slots = n_tiles;
                    *best = ConvResPlan{cs, nw, mt, r, bw, tiles_y, tiles_x, 1, (int)slots, lds};
                }
            }
        }
    }
    return found;
}

template <int CS, int NW, int MT, int PCH, bool RESTAGE>
static int conv_res_launch(const ConvResPlan &p, const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp,
                           const float *wscale, const float *bias, float *out, float *out_amax, uint16_t *out16, int n_img, int frames, int h, int w,
                           int c_in, int c_out, int kt, int relu, hipStream_t st, const float *in2 = nullptr, int c_a = 0)
{
    auto kern = in_mask ? conv3x3_split_res_kernel<CS, NW, MT, PCH, RESTAGE, true> : conv3x3_split_res_kernel<CS, NW, MT, PCH, RESTAGE, false>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.co_groups * p.slots)), dim3(CSP_THREADS), p.lds, st, in, in_amax, in_mask, wp, wscale, bias, out, out_amax,
                       out16, n_img, frames, h, w, c_in, c_out, kt, relu, p.rows, p.bw, p.tiles_y, p.tiles_x, p.co_groups, p.slots, in2, c_a);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// Tiling of a layer: (NW, NGW) by the output width, MT, slice width, tile = rows x bw.  A workgroup's time goes with its MFMA count
// (3 per fragment pair) plus a per-tap cost for the barrier and the weight tile; the launch takes ceil(blocks / 256 CUs) rounds of it
// (LDS leaves one workgroup per CU).
struct ConvSplitPlan { int cs, nw, ngw, mt, rows, bw, tiles_y, tiles_x, co_groups; size_t lds; int64_t blocks; };

static bool conv_split_fits(int cs, int wrows, int rows, int bw, size_t *lds)
{
    const int64_t pp = (int64_t)(rows + 2) * (bw + 2);
    *lds = (size_t)(2 * pp + 4 * wrows) * (cs + 8) * sizeof(uint16_t);
    return pp * (cs / 8) <= CSP_THREADS * CSP_PCH && *lds <= CSP_LDS_MAX && rows + 2 < 0x7ff && bw + 2 < 0xfff;
}

static bool conv_split_plan(int n_img, int h, int w, int c_in, int c_out, int kt, ConvSplitPlan *best, int taps = 9, int cs_must_divide = 0)
{
    if (c_in < 32 || c_in % 32 || c_out < 32 || c_out % 32 || h < 1 || w < 1 || h > 32767 || w > 32767) return false;
    bool found = false;
    int64_t best_cost = 0, best_waste = 0;
    static const int shapes[3][2] = {{2, 2}, {2, 1}, {1, 1}};
    for (int si = 0; si < 3; ++si) {
        const int nw = shapes[si][0], ngw = shapes[si][1];
        const int wrows = 32 * nw * ngw, mg = 8 / ngw;
        if (c_out % wrows) continue;
        for (int mt = 3; mt >= 1; --mt) {
            if (nw == 2 && mt == 3) continue;                  // instantiated: (2,2) and (2,1) x MT 1..2 (MT = 3 spills), (1,1) x MT 1..3
            const int cap = mg * mt * 32;
            for (int cs = 64; cs >= 32; cs -= 32) {
                if (c_in % cs || (cs_must_divide && cs_must_divide % cs)) continue;
                // band widths: the whole row when it fits a tile, else bands of about 32 / 64 / 96 pixels (balanced over the row)
                for (int bi = 0; bi < 4; ++bi) {
                    int bw = bi == 0 ? w : 32 * bi;
                    if (bi > 0 && bw >= w) continue;
                    if (bw > cap) continue;
                    const int tiles_x = (w + bw - 1) / bw;
                    bw = (w + tiles_x - 1) / tiles_x;
                    int r = cap / bw;
                    if (r > h) r = h;
                    size_t lds;
                    while (r >= 1 && !conv_split_fits(cs, wrows, r, bw, &lds)) --r;
                    if (r < 1) continue;
                    const int tiles_y = (h + r - 1) / r;
                    r = (h + tiles_y - 1) / tiles_y;               // balance: the same count with the least height
                    if (!conv_split_fits(cs, wrows, r, bw, &lds)) continue;
                    const int tiles = (r * bw + 31) / 32;
                    if (mt > 1 && tiles <= mg * (mt - 1)) continue;   // a smaller MT covers this tile
                    const int64_t blocks = (int64_t)n_img * tiles_y * tiles_x * (c_out / wrows);
                    const int64_t rounds = (blocks + PCACC_CUS - 1) / PCACC_CUS;
                    // per (slice, tap): 3 MFMAs per fragment pair and 16 k, ~6 MFMA times for the barrier / weight tile; per slice: the patch
                    const int64_t per_slice = taps * ((int64_t)(cs / 16) * 3 * nw * mt + 6) + (int64_t)(r + 2) * (bw + 2) * cs / 1024;
                    const int64_t cost = rounds * per_slice * (c_in / cs);
                    const int64_t waste = (int64_t)mg * mt * 32 * tiles_y * tiles_x - (int64_t)h * w;
                    if (!found || cost < best_cost || (cost == best_cost && waste < best_waste)) {
                        found = true;
                        best_cost = cost;
                        best_waste = waste;
                        *best = ConvSplitPlan{cs, nw, ngw, mt, r, bw, tiles_y, tiles_x, c_out / wrows, lds, blocks};
                    }
                }
            }
        }
    }
    (void)kt;
    return found;
}

template <int CS, int NW, int NGW, int MT, int TAPS = 9>
static int conv_split_launch(const ConvSplitPlan &p, const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp,
                             const float *wscale, const float *bias, float *out, float *out_amax, uint16_t *out16, int n_img, int frames, int h, int w,
                             int c_in, int c_out, int kt, int relu, hipStream_t st, int mode = CSP_PLAIN, int c_up = 0, int up_pitch = 0,
                             const float *in2 = nullptr, int c_a = 0)
{
    auto kern = in_mask ? conv3x3_split_kernel<CS, NW, NGW, MT, TAPS, true> : conv3x3_split_kernel<CS, NW, NGW, MT, TAPS, false>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    if (p.blocks > 0x7fffffff) return PCACC_E_ARG;
    hipLaunchKernelGGL(kern, dim3((unsigned)p.blocks), dim3(CSP_THREADS), p.lds, st, in, in_amax, in_mask, wp, wscale, bias, out, out_amax, out16, n_img,
                       frames, h, w, c_in, c_out, kt, relu, p.rows, p.bw, p.tiles_y, p.tiles_x, p.co_groups, mode, c_up, up_pitch ? up_pitch : c_up, in2, c_a,
                       (p.co_groups > 1 && !pcacc_switches().xcd_off) ? 1 : 0);
    PCACC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcacc_conv3x3_split_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out)
{
    ConvSplitPlan p;
    return conv_split_plan(1, h, w, c_in, c_out, 1, &p) ? 1 : 0;
}

static int conv3x3_split_impl(const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp, const float *wscale,
                              const float *bias, float *out, float *out_amax, uint16_t *out16, int32_t n_img, int32_t frames, int32_t h, int32_t w,
                              int32_t c_in, int32_t c_out, int32_t kt, int32_t relu, void *stream, const float *in2 = nullptr, int32_t c_a = 0)
{
    ConvSplitPlan p;
    if (in2 && (c_a < 32 || c_a % 32 || c_a >= c_in || (c_in - c_a) % 32 || in_mask || kt != 1)) return PCACC_E_ARG;
    const int divides = in2 ? ((c_a % 64 == 0 && (c_in - c_a) % 64 == 0) ? 64 : 32) : 0;       // slice width the channel boundary allows
    if (!in || !in_amax || !wp || !wscale || !out || n_img < 1 || (kt != 1 && kt != 3) || frames < 1 || n_img % frames ||
        !conv_split_plan(n_img, h, w, c_in, c_out, kt, &p, 9, divides))
        return PCACC_E_ARG;
    hipStream_t st = pcacc_stream(stream);
    ConvResPlan rp;
    if (conv_res_plan(n_img, h, w, c_in, c_out, &rp, divides) && PCACC_WALK_OK(n_img, frames, rp.tiles_y, rp.tiles_x)) {
        if (pcacc_switches().conv_plan)
            fprintf(stderr, "split conv plan (resident) %dx%d %d->%d kt=%d n=%d: cs=%d nw=%d mt=%d rows=%d bw=%d slots=%d lds=%zu\n", h, w, c_in, c_out,
                    kt, n_img, rp.cs, rp.nw, rp.mt, rp.rows, rp.bw, rp.slots, rp.lds);
        const bool restage = kt == 3 || c_in != rp.cs;         // more than one (frame tap, channel slice) per tile
#define CSR_CASE(CSV, NWV, MTV, PCHV)                                          \
    if (rp.cs == CSV && rp.nw == NWV && rp.mt == MTV)                          \
        return restage ? conv_res_launch<CSV, NWV, MTV, PCHV, true>(rp, in, in_amax, in_mask, wp, wscale, bias, out, out_amax, out16, n_img, frames, h, w, c_in, c_out, kt, relu, st, in2, c_a) \
                       : conv_res_launch<CSV, NWV, MTV, PCHV, false>(rp, in, in_amax, in_mask, wp, wscale, bias, out, out_amax, out16, n_img, frames, h, w, c_in, c_out, kt, relu, st, in2, c_a)
        CSR_CASE(32, 1, 1, 3); CSR_CASE(32, 2, 1, 3); CSR_CASE(32, 1, 2, 5); CSR_CASE(64, 1, 1, 6);
#undef CSR_CASE
    }
    if (pcacc_switches().conv_plan)
        fprintf(stderr, "split conv plan %dx%d %d->%d kt=%d n=%d: cs=%d nw=%d ngw=%d mt=%d rows=%d bw=%d blocks=%lld lds=%zu\n", h, w, c_in, c_out,
                kt, n_img, p.cs, p.nw, p.ngw, p.mt, p.rows, p.bw, (long long)p.blocks, p.lds);
#define CSP_CASE(CSV, NWV, NGWV, MTV)                                          \
    if (p.cs == CSV && p.nw == NWV && p.ngw == NGWV && p.mt == MTV)            \
        return conv_split_launch<CSV, NWV, NGWV, MTV>(p, in, in_amax, in_mask, wp, wscale, bias, out, out_amax, out16, n_img, frames, h, w, c_in, c_out, kt, relu, st, CSP_PLAIN, 0, 0, in2, c_a)
    CSP_CASE(64, 2, 2, 1); CSP_CASE(64, 2, 2, 2); CSP_CASE(64, 2, 1, 1); CSP_CASE(64, 2, 1, 2);
    CSP_CASE(64, 1, 1, 1); CSP_CASE(64, 1, 1, 2); CSP_CASE(64, 1, 1, 3);
    CSP_CASE(32, 2, 2, 1); CSP_CASE(32, 2, 2, 2); CSP_CASE(32, 2, 1, 1); CSP_CASE(32, 2, 1, 2);
    CSP_CASE(32, 1, 1, 1); CSP_CASE(32, 1, 1, 2); CSP_CASE(32, 1, 1, 3);
#undef CSP_CASE
    return PCACC_E_ARG;
}

extern "C" int pcacc_conv3x3_split(const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp, const float *wscale,
                                   const float *bias, float *out, float *out_amax, int32_t n_img, int32_t frames, int32_t h, int32_t w,
                                   int32_t c_in, int32_t c_out, int32_t kt, int32_t relu, void *stream)
{
    return conv3x3_split_impl(in, in_amax, in_mask, wp, wscale, bias, out, out_amax, nullptr, n_img, frames, h, w, c_in, c_out, kt, relu, stream);
}

// the same convolution with a second result: out16 [n_img, h, w, c_out] bf16 = the fp32 result rounded to nearest even ('mixed' compute mode:
// the shadow the bf16 backward reads; one extra 2-byte store per element instead of a separate cast pass of 6 bytes per element)
extern "C" int pcacc_conv3x3_split_dual(const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp, const float *wscale,
                                        const float *bias, float *out, float *out_amax, uint16_t *out16, int32_t n_img, int32_t frames, int32_t h,
                                        int32_t w, int32_t c_in, int32_t c_out, int32_t kt, int32_t relu, void *stream)
{
    if (!out16 || relu == CSP_OUTMASK) return PCACC_E_ARG;
    return conv3x3_split_impl(in, in_amax, in_mask, wp, wscale, bias, out, out_amax, out16, n_img, frames, h, w, c_in, c_out, kt, relu, stream);
}

// the same convolution (3x3, forward) on the channel concatenation of TWO inputs read in place: in_a [n_img, h, w, c_a], in_b [n_img, h, w, c_in - c_a]
// (the decoder's cat(upconv(x), skip), models/unet.py:101-113 -- the fp32 concatenation is never written); in_amax bounds both (the element-wise
// maximum of their arrays); out16 may be NULL.  c_a and c_in - c_a multiples of 32.
extern "C" int pcacc_conv3x3_split_cat(const float *in_a, const float *in_b, int32_t c_a, const float *in_amax, const uint16_t *wp, const float *wscale,
                                       const float *bias, float *out, float *out_amax, uint16_t *out16, int32_t n_img, int32_t h, int32_t w,
                                       int32_t c_in, int32_t c_out, int32_t relu, void *stream)
{
    if (!in_b || relu == CSP_OUTMASK) return PCACC_E_ARG;
    return conv3x3_split_impl(in_a, in_amax, nullptr, wp, wscale, bias, out, out_amax, out16, n_img, 1, h, w, c_in, c_out, 1, relu, stream, in_b, c_a);
}

// the same convolution (no bias, no ReLU) with its result stored as zero where out_mask [n_img, h, w, c_out] f32 is <= 0 (see CSP_OUTMASK)
extern "C" int pcacc_conv3x3_split_outmask(const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp, const float *wscale,
                                           const float *out_mask, float *out, float *out_amax, int32_t n_img, int32_t frames, int32_t h, int32_t w,
                                           int32_t c_in, int32_t c_out, int32_t kt, void *stream)
{
    if (!out_mask) return PCACC_E_ARG;
    return conv3x3_split_impl(in, in_amax, in_mask, wp, wscale, out_mask, out, out_amax, nullptr, n_img, frames, h, w, c_in, c_out, kt, CSP_OUTMASK, stream);
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------------
// dW[co][tap][ci] = sum over images and pixels of dY[px][co] * X[px + tap offset][ci]: M = co, N = (tap, ci), K = pixels, one frame tap
// per launch (dt).  The scheme of conv3x3_wgrad_strip_kernel (conv_deep.hip) on split operands: a workgroup owns one (CO x CI) block of
// the weight tensor (CO, CI = 32 or 64) and every `slots`-th tile (rows x bw pixels); the fp32 dY rows and the X patch are split into
// scaled hi / lo planes while they are staged channels-last, the fragments ("8 consecutive pixels of one channel") come through the LDS
// transpose read; 8 waves = (co tile, ci tile) pairs x tap groups; three MFMAs per fragment pair.  The ReLU backward of dY is applied
// while staging (dy_mask = the layer's forward output).  One partial slot per workgroup, a second launch sums the slots.
typedef short csp_s16x4 __attribute__((ext_vector_type(4)));
union csp_frag { f16x8_t v; csp_s16x4 h[2]; };
#define CSW_PCH 6                                          // staged 8-channel chunks (dY rows + X patch) a thread carries

__device__ __forceinline__ f16x8_t csp_tr_frag(const uint16_t *p, int stride4)
{
    csp_frag f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((csp_s16x4 __attribute__((address_space(3))) *)p);
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((csp_s16x4 __attribute__((address_space(3))) *)(p + stride4));
    return f.v;
}

template <int CO_T, int CI_T, int TAPS>
__global__ __launch_bounds__(CSP_THREADS) void conv3x3_wgrad_split_kernel(const float *__restrict__ dy, const float *__restrict__ dy_amax,
                                                                          const float *__restrict__ dy_mask, const float *__restrict__ x,
                                                                          const float *__restrict__ x_amax, float *__restrict__ partial,
                                                                          int n_img, int frames, int dt, int h, int w, int c_in, int c_out,
                                                                          int rows, int bw, int tiles_y, int tiles_x, int ci_blocks, int slots, int c_up, int xcd)
{
    // 32 x 32 blocks: one (co, ci) pair -- the 8 waves are 2 halves of the 16-pixel steps x 4 tap groups (3 | 2 | 2 | 2 taps) and each half
    // keeps its own partial slot; with 8 tap groups (2 | 1 x 7 taps) wave 0 did twice the work of the others.
    // TAPS = 1 (the 2x2 transposed convolution: dW'[(a,b,co)][ci] = sum dy[2y+a,2x+b,co] x[y,x,ci], c_up > 0: dY read space-to-depth): no taps
    // to share out -- all waves of a pair split the pixel steps.
    constexpr int CO = CO_T * 32, CI = CI_T * 32, PAIRS = CO_T * CI_T, HALVES = TAPS == 1 ? 8 / PAIRS : (PAIRS == 1 ? 2 : 1), G = 8 / (PAIRS * HALVES),
                  NT = (TAPS + G - 1) / G;
    constexpr int YS = pcacc_tr_stride(CO), XS = pcacc_tr_stride(CI);
    constexpr int SLOT = CO * TAPS * CI + CO;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int pw = bw + 2, pp = (rows + 2) * pw;
    const int n_px = rows * bw, n_steps = (n_px + 15) >> 4, py_rows = n_steps * 16;
    const int yplane = py_rows * YS, xplane = pp * XS;
    uint16_t *sdy = lds;                                       // [2][py_rows][YS]   (rows >= n_px are zero)
    uint16_t *sx = sdy + 2 * (size_t)yplane;                   // [2][pp][XS]
    uint16_t *ptab = sx + 2 * (size_t)xplane;                  // [py_rows] patch row of the pixel's top-left tap

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    const int pair = wave % PAIRS, grp = (wave / PAIRS) % G, half = wave / (PAIRS * G);
    const int ct = pair / CI_T, it = pair % CI_T;
    // xcd: the (co, ci) blocks of a strip next to each other on one XCD (block fastest in the logical order): the dY / X rows they share are fetched once
    const int n_blocks = gridDim.x / slots, lb = xcd ? pcacc_xcd_block(blockIdx.x, gridDim.x) : 0;
    const int block = xcd ? lb % n_blocks : blockIdx.x / slots, slot = xcd ? lb / n_blocks : blockIdx.x % slots;
    const int co0 = (block / ci_blocks) * CO, ci0 = (block % ci_blocks) * CI;
    const float sy = csp_scale_from_parts(dy_amax), sxs = csp_scale_from_parts(x_amax);   // the reduce launch divides by sy * sxs

    for (int q = threadIdx.x; q < py_rows; q += CSP_THREADS) ptab[q] = q < n_px ? (uint16_t)((q / bw) * pw + q % bw) : 0;

    f32x16_t acc[NT];                                          // local tap j = tap grp + j * G
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    // staged chunks of this thread: first the dY rows (py_rows * CO / 8 chunks), then the X patch (pp * CI / 8 chunks)
    constexpr int YC8 = CO / 8, XC8 = CI / 8;
    const int y_chunks = py_rows * YC8, n_chunks = y_chunks + pp * XC8;
    float4 sreg[CSW_PCH][2];
    const int tiles = tiles_y * tiles_x;
    auto job_valid = [&](int job) {
        const int t_frame = (job / tiles) % frames + dt;
        return t_frame >= 0 && t_frame < frames;
    };
    auto fetch = [&](int job) {
        const int img = job / tiles, rem = job - img * tiles;
        const int y0 = (rem / tiles_x) * rows, x0 = (rem % tiles_x) * bw;
        const float *ysrc = dy + (int64_t)img * h * w * c_out + co0;
        const float *msrc = dy_mask ? dy_mask + (int64_t)img * h * w * c_out + co0 : nullptr;
        const int ab = c_up ? co0 / c_up : 0, ch0 = c_up ? co0 % c_up : 0;      // space-to-depth: this block's channels sit in pixel (2y+a, 2x+b)
        const float *xsrc = x + (int64_t)(img + dt) * h * w * c_in + ci0;
#pragma unroll
        for (int q = 0; q < CSW_PCH; ++q) {
            const int c = threadIdx.x + q * CSP_THREADS;
            // one unconditional load per chunk from a clamped address, zero selected afterwards
            const bool is_y = c < y_chunks;
            const int e = is_y ? c : min(c, n_chunks - 1) - y_chunks;
            const int px = is_y ? e / YC8 : e / XC8, c8 = is_y ? e % YC8 : e % XC8;
            const int yy = is_y ? y0 + px / bw : y0 - 1 + px / pw;
            const int xx = is_y ? x0 + px % bw : x0 - 1 + px % pw;
            const bool ok = c < n_chunks && (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && (!is_y || px < n_px);
            const int64_t pos = (int64_t)min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1);
            const float *src = is_y ? ysrc + pos * c_out + c8 * 8 : xsrc + pos * c_in + c8 * 8;
            if (c_up && is_y) {
                const int yc = min(max(yy, 0), h - 1), xc = min(max(xx, 0), w - 1);
                src = dy + (((int64_t)img * 2 * h + 2 * yc + (ab >> 1)) * 2 * w + 2 * xc + (ab & 1)) * c_up + ch0 + c8 * 8;
            }
            float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
            if (msrc && is_y) {
                const float *m = msrc + pos * c_out + c8 * 8;
                a = csp_relu_mask4(a, *reinterpret_cast<const float4 *>(m));
                b = csp_relu_mask4(b, *reinterpret_cast<const float4 *>(m + 4));
            }
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            sreg[q][0] = ok ? a : z;
            sreg[q][1] = ok ? b : z;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < CSW_PCH; ++q) {
            const int c = threadIdx.x + q * CSP_THREADS;
            if (c >= n_chunks) continue;
            uint4 hi, lo;
            const bool is_y = c < y_chunks;
            csp_split8(sreg[q][0], sreg[q][1], is_y ? sy : sxs, hi, lo);
            const int e = is_y ? c : c - y_chunks;
            uint16_t *dst = is_y ? sdy + (e / YC8) * YS + (e % YC8) * 8 : sx + (e / XC8) * XS + (e % XC8) * 8;
            const int pl = is_y ? yplane : xplane;
            // 8-byte stores: the transposed-read row strides are multiples of 8 bytes only
            reinterpret_cast<uint2 *>(dst)[0] = make_uint2(hi.x, hi.y);
            reinterpret_cast<uint2 *>(dst)[1] = make_uint2(hi.z, hi.w);
            reinterpret_cast<uint2 *>(dst + pl)[0] = make_uint2(lo.x, lo.y);
            reinterpret_cast<uint2 *>(dst + pl)[1] = make_uint2(lo.z, lo.w);
        }
    };

    const int n_jobs = n_img * tiles;
    const int tg = lane >> 4, tl = lane & 15;
    const int tr_row = (tg >> 1) * 8 + (tl >> 2), tr_col = (tg & 1) * 16 + (tl & 3) * 4;
    int job = slot;
    while (job < n_jobs && !job_valid(job)) job += slots;      // a missing frame contributes nothing
    if (job < n_jobs) fetch(job);
    while (job < n_jobs) {
        __syncthreads();                                       // the previous tile's fragment reads are done
        stage();
        __syncthreads();
        int next = job + slots;
        while (next < n_jobs && !job_valid(next)) next += slots;
        if (next < n_jobs) fetch(next);
        job = next;
        for (int s = half; s < n_steps; s += HALVES) {
            const int r0 = s * 16 + tr_row;
            const uint16_t *pa = sdy + r0 * YS + ct * 32 + tr_col;
            csp_frag ah, al;
            ah.v = csp_tr_frag(pa, 4 * YS);
            al.v = csp_tr_frag(pa + yplane, 4 * YS);
            if (it == 0 && grp == 0) {                         // bias gradient = column sums of dY: the fragments are at hand
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bsum += (float)ah.v[4 * j + q] + (float)al.v[4 * j + q];
            }
            const int p0 = ptab[r0], p1 = ptab[r0 + 4];
            const uint16_t *pb0 = sx + p0 * XS + it * 32 + tr_col, *pb1 = sx + p1 * XS + it * 32 + tr_col;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int tap = grp + j * G;                   // uniform per wave
                if (tap < TAPS) {
                    const int toff = (TAPS == 1 ? pw + 1 : (tap / 3) * pw + tap % 3) * XS;
                    csp_frag bh, bl;
                    bh.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((csp_s16x4 __attribute__((address_space(3))) *)(pb0 + toff));
                    bh.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((csp_s16x4 __attribute__((address_space(3))) *)(pb1 + toff));
                    bl.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((csp_s16x4 __attribute__((address_space(3))) *)(pb0 + xplane + toff));
                    bl.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((csp_s16x4 __attribute__((address_space(3))) *)(pb1 + xplane + toff));
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bl.v, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh.v, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh.v, acc[j], 0, 0, 0);
                }
            }
        }
    }
    // slot of this workgroup: [CO][9][CI] then [CO] bias sums; D has lane = ci, register quads = co
    float *mine = partial + (((int64_t)block * slots + slot) * HALVES + half) * SLOT;       // the reduce launch's order: a block's slots are consecutive
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tap = grp + j * G;
        if (tap < TAPS)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                mine[((int64_t)co * TAPS + tap) * CI + it * 32 + lp] = acc[j][r];
            }
    }
    if (grp != 0) return;
    bsum += __shfl_xor(bsum, 32, 64);                          // the two half-waves hold pixels 0-7 / 8-15 of the same channel
    if (it == 0 && lh == 0) mine[CO * TAPS * CI + ct * 32 + lp] = bsum;
}

// out[co][tap][ci] (full tensor) and db[co] from the per-workgroup slots, scales removed.  A workgroup = 64 output elements x 4 slot groups
// (wave g sums the slots p = g, g + 4, ...), combined through LDS in a fixed order: the 32 x 32 layers have 9 248 elements and 256 slots --
// with one thread per element the launch was 37 workgroups walking 256 partials each (40 us, 2 ms per step over 52 launches).
__global__ __launch_bounds__(256) void conv_wgrad_split_reduce_kernel(const float *__restrict__ partial, int slots, int c_in, int c_out,
                                                                      int cob, int cib, int taps, const float *__restrict__ dy_amax,
                                                                      const float *__restrict__ x_amax, float *__restrict__ dw,
                                                                      float *__restrict__ db)
{
    const float inv_y = 1.f / csp_scale_from_parts(dy_amax), inv_yx = inv_y / csp_scale_from_parts(x_amax);
    const int slot_elems = cob * taps * cib + cob, ci_blocks = c_in / cib;
    const int64_t n_w = (int64_t)c_out * taps * c_in;
    const int64_t e = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;
    __shared__ float sm[4][64];
    float s = 0.f;
    const float *src = nullptr;
    if (e < n_w) {
        const int ci = (int)(e % c_in), tap = (int)((e / c_in) % taps), co = (int)(e / ((int64_t)taps * c_in));
        const int block = (co / cob) * ci_blocks + ci / cib;
        src = partial + (int64_t)block * slots * slot_elems + ((int64_t)(co % cob) * taps + tap) * cib + ci % cib;
    } else if (e < n_w + c_out) {
        const int co = (int)(e - n_w);
        src = partial + (int64_t)((co / cob) * ci_blocks) * slots * slot_elems + cob * taps * cib + co % cob;   // ci block 0 carries the bias sums
    }
    if (src) {
        float s0 = 0.f, s1 = 0.f;
        int p = grp;
        for (; p + 4 < slots; p += 8) {
            s0 += src[(int64_t)p * slot_elems];
            s1 += src[(int64_t)(p + 4) * slot_elems];
        }
        if (p < slots) s0 += src[(int64_t)p * slot_elems];
        s = s0 + s1;
    }
    sm[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && src) {
        const float t = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
        if (e < n_w) dw[e] = t * inv_yx;
        else if (db) db[e - n_w] = t * inv_y;
    }
}

struct ConvSplitWPlan { int cob, cib, rows, bw, tiles_y, tiles_x, blocks, slots; size_t lds; };

static bool conv_wsplit_fits(int cob, int cib, int rows, int bw, size_t *lds)
{
    const int pp = (rows + 2) * (bw + 2), py_rows = (rows * bw + 15) / 16 * 16;
    *lds = ((size_t)2 * py_rows * pcacc_tr_stride(cob) + (size_t)2 * pp * pcacc_tr_stride(cib) + py_rows) * sizeof(uint16_t);
    return py_rows * (cob / 8) + pp * (cib / 8) <= CSP_THREADS * CSW_PCH && *lds <= 150 * 1024 && pp < 65536;
}

static bool conv_wsplit_plan(int n_img, int h, int w, int c_in, int c_out, ConvSplitWPlan *best, int co_group = 0)
{
    if (c_in < 32 || c_in % 32 || c_out < 32 || c_out % 32 || h < 1 || w < 1 || n_img < 1) return false;
    const int cob = (co_group ? co_group : c_out) % 64 == 0 ? 64 : 32, cib = c_in % 64 == 0 ? 64 : 32;     // co_group: channels per (a, b) group
    bool found = false;
    int64_t best_cost = 0;
    // band widths: the whole row, or bands of about 16 / 32 / 64 pixels; rows as many as fit.  Cost = padded 16-pixel steps (MFMA work)
    // plus the staged pixels (halo included) weighted by the share of a step they cost.
    for (int bi = 0; bi < 4; ++bi) {
        int bw = bi == 0 ? w : 16 << (bi - 1);
        if (bi > 0 && bw >= w) continue;
        const int tiles_x = (w + bw - 1) / bw;
        bw = (w + tiles_x - 1) / tiles_x;
        int best_r = 0;
        size_t lds = 0, l;
        for (int r = 1; r <= h; ++r) {
            if (!conv_wsplit_fits(cob, cib, r, bw, &l)) break;
            best_r = r;
            lds = l;
        }
        if (!best_r) continue;
        const int tiles_y = (h + best_r - 1) / best_r;
        const int r = (h + tiles_y - 1) / tiles_y;
        if (!conv_wsplit_fits(cob, cib, r, bw, &lds)) continue;
        const int64_t steps = (int64_t)tiles_y * tiles_x * ((r * bw + 15) / 16);
        const int64_t staged = (int64_t)tiles_y * tiles_x * ((int64_t)(r + 2) * (bw + 2) + r * bw);
        const int64_t cost = steps * 16 * 4 + staged;
        if (!found || cost < best_cost) {
            found = true;
            best_cost = cost;
            const int blocks = (c_out / cob) * (c_in / cib);
            int slots = (PCACC_CUS + blocks - 1) / blocks;         // one workgroup per CU (the staging LDS allows no more)
            const int64_t jobs = (int64_t)n_img * tiles_y * tiles_x;
            if (slots > jobs) slots = (int)jobs;
            *best = ConvSplitWPlan{cob, cib, r, bw, tiles_y, tiles_x, blocks, slots < 1 ? 1 : slots, lds};
        }
    }
    return found;
}

static int conv_wsplit_halves(const ConvSplitWPlan &p, int taps = 9)      // partial slots per workgroup
{
    const int pairs = (p.cob / 32) * (p.cib / 32);
    return taps == 1 ? 8 / pairs : (pairs == 1 ? 2 : 1);
}

extern "C" int pcacc_conv3x3_wgrad_split_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out, size_t *bytes)
{
    ConvSplitWPlan p;
    if (!bytes || !conv_wsplit_plan(n_img, h, w, c_in, c_out, &p)) return PCACC_E_ARG;
    *bytes = (size_t)p.blocks * p.slots * conv_wsplit_halves(p) * (p.cob * 9 * p.cib + p.cob) * sizeof(float);
    return 0;
}

extern "C" int pcacc_conv3x3_wgrad_split(const float *dy, const float *dy_amax, const float *dy_mask, const float *x, const float *x_amax,
                                         float *dw, float *db, int32_t n_img, int32_t frames, int32_t dt, int32_t h, int32_t w, int32_t c_in,
                                         int32_t c_out, void *workspace, size_t workspace_bytes, void *stream)
{
    ConvSplitWPlan p;
    if (!dy || !dy_amax || !x || !x_amax || !dw || !workspace || frames < 1 || n_img < 1 || n_img % frames || dt < -1 || dt > 1 ||
        !conv_wsplit_plan(n_img, h, w, c_in, c_out, &p))
        return PCACC_E_ARG;
    if ((int64_t)n_img * p.tiles_y * p.tiles_x > 0x7fffffff) return PCACC_E_ARG;
    const size_t need = (size_t)p.blocks * p.slots * conv_wsplit_halves(p) * (p.cob * 9 * p.cib + p.cob) * sizeof(float);
    if (workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t st = pcacc_stream(stream);
    float *partial = static_cast<float *>(workspace);
    if (pcacc_switches().conv_plan)
        fprintf(stderr, "split wgrad plan %dx%d %d->%d n=%d: block %dx%d rows=%d bw=%d blocks=%d slots=%d lds=%zu\n", h, w, c_in, c_out, n_img,
                p.cob, p.cib, p.rows, p.bw, p.blocks, p.slots, p.lds);
#define CSW_CASE(COT, CIT)                                                                                                              \
    if (p.cob == COT * 32 && p.cib == CIT * 32) {                                                                                       \
        auto kern = conv3x3_wgrad_split_kernel<COT, CIT, 9>;                                                                            \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess) \
            return PCACC_E_LAUNCH;                                                                                                      \
        hipLaunchKernelGGL(kern, dim3(p.blocks * p.slots), dim3(CSP_THREADS), p.lds, st, dy, dy_amax, dy_mask, x, x_amax, partial, n_img, frames, dt, h, w, \
                           c_in, c_out, p.rows, p.bw, p.tiles_y, p.tiles_x, c_in / p.cib, p.slots, 0, (p.blocks > 1 && !pcacc_switches().xcd_off) ? 1 : 0); \
    }
    CSW_CASE(1, 1) else CSW_CASE(1, 2) else CSW_CASE(2, 1) else CSW_CASE(2, 2) else return PCACC_E_ARG;
#undef CSW_CASE
    const int64_t elems = (int64_t)c_out * 9 * c_in + c_out;
    hipLaunchKernelGGL(conv_wgrad_split_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, st, partial, p.slots * conv_wsplit_halves(p), c_in,
                       c_out, p.cob, p.cib, 9, dy_amax, x_amax, dw, db);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// ---- nn.ConvTranspose2d(kernel 2, stride 2) of the two decoders (models/unet.py:22-30,101-113) on the kernels above ------------------
// weights: w f32 [c_in][c_up][2][2] (torch layout, read through `strides`: elements i, o, y, x) ->
//   forward form      fp16 [2][1][4 c_up][c_in]  rows co' = (a, b, co)      + 1 / row scale [4 c_up]
//   data-gradient form fp16 [2][1][c_in][4 c_up]  columns k' = (a, b, co)    + 1 / row scale [c_in]
__device__ __forceinline__ void upconv_prepare_row(const float *__restrict__ w, int c_in, int c_up, int64_t si, int64_t so, int64_t sy, int64_t sx,
                                                   uint16_t *__restrict__ out_fwd, float *__restrict__ inv_fwd, uint16_t *__restrict__ out_bwd,
                                                   float *__restrict__ inv_bwd, int blk)
{
    const int n4 = 4 * c_up;
    const bool bwd = blk >= n4;
    const int row = bwd ? blk - n4 : blk;
    const int n = bwd ? n4 : c_in;                             // elements of the row
    const int64_t total = (int64_t)n4 * c_in;
    auto src = [&](int e) {
        const int cop = bwd ? e : row, ci = bwd ? row : e;     // co' = (ab, co)
        const int ab = cop / c_up, co = cop - ab * c_up;
        return w[ci * si + co * so + (ab >> 1) * sy + (ab & 1) * sx];
    };
    float m = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {
        const float v = src(e);
        m = fmaxf(m, fabsf(v));
        if (v != v) m = __builtin_inff();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    const float t = csp_scale_of(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])));
    if (threadIdx.x == 0) (bwd ? inv_bwd : inv_fwd)[row] = 1.f / t;
    uint16_t *dst = (bwd ? out_bwd : out_fwd) + (int64_t)row * n;
    for (int e = threadIdx.x; e < n; e += 256) {
        const float v = src(e) * t;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        dst[e] = *reinterpret_cast<const uint16_t *>(&hi);
        dst[total + e] = *reinterpret_cast<const uint16_t *>(&lo);
    }
}

__global__ __launch_bounds__(256) void upconv_split_prepare_kernel(const float *__restrict__ w, int c_in, int c_up, int64_t si, int64_t so, int64_t sy,
                                                                   int64_t sx, uint16_t *__restrict__ out_fwd, float *__restrict__ inv_fwd,
                                                                   uint16_t *__restrict__ out_bwd, float *__restrict__ inv_bwd)
{
    upconv_prepare_row(w, c_in, c_up, si, so, sy, sx, out_fwd, inv_fwd, out_bwd, inv_bwd, (int)blockIdx.x);
}

// ---- every prepared form of every weight of a model in ONE launch ---------------------------------------------------------------------
// A training step prepares ~90 weight forms right after the optimizer wrote the parameters (43 split pairs + 8 transposed-convolution pairs
// + 39 bf16 pairs in the 'mixed' mode): 90 launches of 10-20 us kernels and 90 host calls.  `jobs` is a DEVICE table of 16 int64 per
// job (include/pcacc.h): the workgroup finds its job by bisection over the jobs' first-workgroup numbers and runs the body of the
// single-weight kernel on it.  kind 0: csp_prepare_row; 1: upconv_prepare_row; 2: the bf16 forms of conv.hip's conv_prepare_weights_pair_kernel;
// 3: those of upconv_bf16.hip's upconv_bf16_prepare_kernel.
#define PWB_FIELDS 16
__global__ __launch_bounds__(256) void prepare_weights_batch_kernel(const int64_t *__restrict__ jobs, int n_jobs)
{
    int lo = 0, hi = n_jobs - 1;                               // last job whose first workgroup <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[(int64_t)mid * PWB_FIELDS + 14] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t *j = jobs + (int64_t)lo * PWB_FIELDS;
    const float *w = reinterpret_cast<const float *>(j[0]);
    uint16_t *out_fwd = reinterpret_cast<uint16_t *>(j[1]);
    float *inv_fwd = reinterpret_cast<float *>(j[2]);
    uint16_t *out_bwd = reinterpret_cast<uint16_t *>(j[3]);
    float *inv_bwd = reinterpret_cast<float *>(j[4]);
    const int o = (int)j[10], i = (int)j[11], kt = (int)j[12], kind = (int)j[13];
    const int blk = (int)blockIdx.x - (int)j[14], n_blk = (int)j[15];
    if (blk >= n_blk) return;
    if (kind == 0) {
        const CspWStrides st = {j[5], j[6], j[7], j[8], j[9]};
        csp_prepare_row(w, o, i, kt, st, out_fwd, inv_fwd, out_bwd, inv_bwd, blk);
    } else if (kind == 1) {
        upconv_prepare_row(w, /*c_in*/ o, /*c_up*/ i, /*si*/ j[5], /*so*/ j[6], j[8], j[9], out_fwd, inv_fwd, out_bwd, inv_bwd, blk);
    } else if (kind == 3) {                                    // bf16 forms of a transposed 2 x 2 convolution (upconv_bf16.hip): o = c_in, i = c_up
        const int n4 = 4 * i;
        const int64_t total = (int64_t)n4 * o;
        for (int64_t e = (int64_t)blk * 256 + threadIdx.x; e < 2 * total; e += (int64_t)n_blk * 256) {
            const bool bwd = e >= total;
            const int64_t r = bwd ? e - total : e;
            const int ci = bwd ? (int)(r / n4) : (int)(r % o);
            const int cop = bwd ? (int)(r % n4) : (int)(r / o);
            const int ab = cop / i, co = cop - ab * i;
            (bwd ? out_bwd : out_fwd)[r] = f32_to_bf16(w[ci * j[5] + co * j[6] + (ab >> 1) * j[8] + (ab & 1) * j[9]]);
        }
    } else {
        const int taps = kt * 9;
        const int64_t total = (int64_t)taps * o * i;
        for (int64_t e = (int64_t)blk * 256 + threadIdx.x; e < 2 * total; e += (int64_t)n_blk * 256) {
            const bool transpose = e >= total;
            const int64_t r = transpose ? e - total : e;
            const int op = transpose ? i : o, ip = transpose ? o : i;
            const int ci = (int)(r % ip);
            const int co = (int)((r / ip) % op);
            const int tap = (int)(r / ((int64_t)ip * op));
            const int src_tap = transpose ? (taps - 1 - tap) : tap;
            const int so = transpose ? ci : co, si = transpose ? co : ci;
            const int ft = src_tap / 9, fy = (src_tap % 9) / 3, fx = src_tap % 3;
            (transpose ? out_bwd : out_fwd)[r] = f32_to_bf16(w[so * j[5] + si * j[6] + ft * j[7] + fy * j[8] + fx * j[9]]);
        }
    }
}

extern "C" int pcacc_prepare_weights_batch(const int64_t *jobs, int32_t n_jobs, int32_t total_blocks, void *stream)
{
    if (!jobs || n_jobs < 1 || total_blocks < 1) return PCACC_E_ARG;
    hipLaunchKernelGGL(prepare_weights_batch_kernel, dim3(total_blocks), dim3(256), 0, pcacc_stream(stream), jobs, n_jobs);
    PCACC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcacc_upconv2x2_split_prepare_weights(const float *w, int32_t c_in, int32_t c_up, const int64_t *strides, uint16_t *out_fwd,
                                                     float *scale_fwd, uint16_t *out_bwd, float *scale_bwd, void *stream)
{
    if (!w || !strides || !out_fwd || !scale_fwd || !out_bwd || !scale_bwd || c_in < 1 || c_up < 1) return PCACC_E_ARG;
    hipLaunchKernelGGL(upconv_split_prepare_kernel, dim3(4 * c_up + c_in), dim3(256), 0, pcacc_stream(stream), w, c_in, c_up, strides[0], strides[1],
                       strides[2], strides[3], out_fwd, scale_fwd, out_bwd, scale_bwd);
    PCACC_CHECK_LAUNCH();
    return 0;
}

extern "C" int pcacc_upconv2x2_split_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_up)
{
    ConvSplitPlan p;
    ConvSplitWPlan q;
    return c_up % 32 == 0 && conv_split_plan(1, h, w, c_in, 4 * c_up, 1, &p, 1) && conv_split_plan(1, h, w, 4 * c_up, c_in, 1, &p, 1, c_up) &&
           conv_wsplit_plan(1, h, w, c_in, 4 * c_up, &q, c_up);
}

// direction 0: out [n, 2h, 2w, c_up] = upconv(in [n, h, w, c_in]) + bias;  direction 1: out [n, h, w, c_in] = data gradient of in = dy [n, 2h, 2w, c_up]
// (h, w = the SMALL map's size in both directions; wp / wscale: the matching form of prepare_weights; bias NULL for direction 1)
static int upconv2x2_split_impl(const float *in, const float *in_amax, const uint16_t *wp, const float *wscale, const float *bias, float *out,
                                float *out_amax, uint16_t *out16, int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, int32_t direction,
                                void *stream, int32_t out_pitch = 0)
{
    if (out_pitch && (direction != 0 || out_pitch < c_up || out_pitch % 8)) return PCACC_E_ARG;
    if (!in || !in_amax || !wp || !wscale || !out || n_img < 1 || c_up < 32 || c_up % 32 || (direction != 0 && direction != 1)) return PCACC_E_ARG;
    const int k_in = direction ? 4 * c_up : c_in, k_out = direction ? c_in : 4 * c_up;
    ConvSplitPlan p;
    if (!conv_split_plan(n_img, h, w, k_in, k_out, 1, &p, 1, direction ? c_up : 0)) return PCACC_E_ARG;
    if (pcacc_switches().conv_plan)
        fprintf(stderr, "upconv plan dir %d %dx%d %d->%d n=%d: cs=%d nw=%d ngw=%d mt=%d rows=%d bw=%d blocks=%lld lds=%zu\n", direction, h, w, k_in, k_out,
                n_img, p.cs, p.nw, p.ngw, p.mt, p.rows, p.bw, (long long)p.blocks, p.lds);
    hipStream_t st = pcacc_stream(stream);
    const int mode = direction ? CSP_S2D : CSP_UP;
#define CSU_CASE(CSV, NWV, NGWV, MTV)                                          \
    if (p.cs == CSV && p.nw == NWV && p.ngw == NGWV && p.mt == MTV)            \
        return conv_split_launch<CSV, NWV, NGWV, MTV, 1>(p, in, in_amax, nullptr, wp, wscale, bias, out, out_amax, out16, n_img, 1, h, w, k_in, k_out, 1, 0, st, mode, c_up, out_pitch)
    CSU_CASE(64, 2, 2, 1); CSU_CASE(64, 2, 2, 2); CSU_CASE(64, 2, 1, 1); CSU_CASE(64, 2, 1, 2);
    CSU_CASE(64, 1, 1, 1); CSU_CASE(64, 1, 1, 2); CSU_CASE(64, 1, 1, 3);
    CSU_CASE(32, 2, 2, 1); CSU_CASE(32, 2, 2, 2); CSU_CASE(32, 2, 1, 1); CSU_CASE(32, 2, 1, 2);
    CSU_CASE(32, 1, 1, 1); CSU_CASE(32, 1, 1, 2); CSU_CASE(32, 1, 1, 3);
#undef CSU_CASE
    return PCACC_E_ARG;
}

extern "C" int pcacc_upconv2x2_split(const float *in, const float *in_amax, const uint16_t *wp, const float *wscale, const float *bias, float *out,
                                     float *out_amax, int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, int32_t direction,
                                     void *stream)
{
    return upconv2x2_split_impl(in, in_amax, wp, wscale, bias, out, out_amax, nullptr, n_img, h, w, c_in, c_up, direction, stream);
}

// direction 0 with a second result: out16 [n, 2h, 2w, c_up] bf16 = the fp32 result rounded to nearest even ('mixed' compute mode, see pcacc_conv3x3_split_dual)
// out_pitch: elements between consecutive pixels of out AND out16 (0 = c_up: dense results; 2 c_up: the result is the first half of the decoder's
// concatenation buffer [n, 2h, 2w, 2 c_up], models/unet.py:101-113 -- no copy of the up-sampled half into it)
extern "C" int pcacc_upconv2x2_split_dual(const float *in, const float *in_amax, const uint16_t *wp, const float *wscale, const float *bias, float *out,
                                          float *out_amax, uint16_t *out16, int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up,
                                          int32_t out_pitch, void *stream)
{
    if (!out16) return PCACC_E_ARG;
    return upconv2x2_split_impl(in, in_amax, wp, wscale, bias, out, out_amax, out16, n_img, h, w, c_in, c_up, 0, stream, out_pitch);
}

extern "C" int pcacc_upconv2x2_wgrad_split_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, size_t *bytes)
{
    ConvSplitWPlan p;
    if (!bytes || c_up < 32 || c_up % 32 || !conv_wsplit_plan(n_img, h, w, c_in, 4 * c_up, &p, c_up)) return PCACC_E_ARG;
    *bytes = (size_t)p.blocks * p.slots * conv_wsplit_halves(p, 1) * (p.cob * p.cib + p.cob) * sizeof(float);
    return 0;
}

// dw [4 c_up][c_in] f32 (row (a, b, co): the caller permutes to the module's [c_in][c_up][2][2]) and db4 [4 c_up] f32 (per (a, b, co) sums of dy: the
// caller adds the four groups) from dy [n, 2h, 2w, c_up] and x [n, h, w, c_in]
extern "C" int pcacc_upconv2x2_wgrad_split(const float *dy, const float *dy_amax, const float *x, const float *x_amax, float *dw, float *db4,
                                           int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, void *workspace, size_t workspace_bytes,
                                           void *stream)
{
    ConvSplitWPlan p;
    if (!dy || !dy_amax || !x || !x_amax || !dw || !workspace || c_up < 32 || c_up % 32 || !conv_wsplit_plan(n_img, h, w, c_in, 4 * c_up, &p, c_up))
        return PCACC_E_ARG;
    const int halves = conv_wsplit_halves(p, 1);
    const size_t need = (size_t)p.blocks * p.slots * halves * (p.cob * p.cib + p.cob) * sizeof(float);
    if (workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t st = pcacc_stream(stream);
    float *partial = static_cast<float *>(workspace);
    const int c_out = 4 * c_up;
#define CSUW_CASE(COT, CIT)                                                                                                             \
    if (p.cob == COT * 32 && p.cib == CIT * 32) {                                                                                       \
        auto kern = conv3x3_wgrad_split_kernel<COT, CIT, 1>;                                                                            \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess) \
            return PCACC_E_LAUNCH;                                                                                                      \
        hipLaunchKernelGGL(kern, dim3(p.blocks * p.slots), dim3(CSP_THREADS), p.lds, st, dy, dy_amax, nullptr, x, x_amax, partial, n_img, 1, 0, h, w, \
                           c_in, c_out, p.rows, p.bw, p.tiles_y, p.tiles_x, c_in / p.cib, p.slots, c_up, (p.blocks > 1 && !pcacc_switches().xcd_off) ? 1 : 0); \
    }
    CSUW_CASE(1, 1) else CSUW_CASE(1, 2) else CSUW_CASE(2, 1) else CSUW_CASE(2, 2) else return PCACC_E_ARG;
#undef CSUW_CASE
    const int64_t elems = (int64_t)c_out * c_in + c_out;
    hipLaunchKernelGGL(conv_wgrad_split_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, st, partial, p.slots * halves, c_in, c_out,
                       p.cob, p.cib, 1, dy_amax, x_amax, dw, db4);
    PCACC_CHECK_LAUNCH();
    return 0;
}
