// nn.ConvTranspose2d(kernel 2, stride 2) of the two decoders (models/unet.py:22-30,101-113) on bf16 channels-last rows: forward, data gradient
// and weight / bias gradient on the bf16 matrix cores (the bf16 compute mode's forward and backward, the 'mixed' mode's backward) -- the layer
// went through the library's transposed convolution before (layout copies, two library kernels and a column sum per backward).
//
//   out[n, 2y+a, 2x+b, co] = bias[co] + sum_ci in[n, y, x, ci] W[ci, co, a, b]
//
// is a row product with K = c_in and N' = 4 c_up columns n' = (a, b, co) whose result row scatters to the 2 x 2 output pixels (UP); its data
// gradient is the row product of the space-to-depth view of dy (row (n, y, x), columns (a, b, co)) with the transposed weights (S2D); the weight
// gradient is X^T dy_s2d summed over the pixels.  All three are HBM-bound (2 c_in flops per output byte at most): the kernels stream 128-row
// tiles with 16-byte loads and stores and read the (small, L2-resident) prepared bf16 weights per tile.  dy may be a channel slice of a wider
// map (the decoder's concatenation gradient): `pitch` = elements between pixels.
#include "common.h"

typedef __bf16 ub_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float ub_f32x16_t __attribute__((ext_vector_type(16)));
typedef short ub_s16x4 __attribute__((ext_vector_type(4)));
union ub_frag { ub_bf16x8_t v; ub_s16x4 h[2]; };

#define UB_ROWS 128
#define UB_KC 64
#define UB_XS (UB_KC + 8)
#define UB_UP 0
#define UB_S2D 1

// weights: w f32 [c_in][c_up][2][2] through strides (elements: i, o, y, x) -> fwd bf16 [4 c_up][c_in] (row (a, b, co)), bwd bf16 [c_in][4 c_up]
__global__ __launch_bounds__(256) void upconv_bf16_prepare_kernel(const float *__restrict__ w, int c_in, int c_up, int64_t si, int64_t so, int64_t sy,
                                                                  int64_t sx, uint16_t *__restrict__ out_fwd, uint16_t *__restrict__ out_bwd)
{
    const int n4 = 4 * c_up;
    const int64_t total = (int64_t)n4 * c_in;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < 2 * total; e += (int64_t)gridDim.x * 256) {
        const bool bwd = e >= total;
        const int64_t r = bwd ? e - total : e;
        const int ci = bwd ? (int)(r / n4) : (int)(r % c_in);
        const int cop = bwd ? (int)(r % n4) : (int)(r / c_in);
        const int ab = cop / c_up, co = cop - ab * c_up;
        (bwd ? out_bwd : out_fwd)[r] = f32_to_bf16(w[ci * si + co * so + (ab >> 1) * sy + (ab & 1) * sx]);
    }
}

extern "C" int pcacc_upconv2x2_bf16_prepare_weights(const float *w, int32_t c_in, int32_t c_up, const int64_t *strides, uint16_t *out_fwd,
                                                    uint16_t *out_bwd, void *stream)
{
    if (!w || !strides || !out_fwd || !out_bwd || c_in < 1 || c_up < 1) return PCACC_E_ARG;
    const int64_t total = 2 * (int64_t)4 * c_up * c_in;
    hipLaunchKernelGGL(upconv_bf16_prepare_kernel, dim3(pcacc_grid(total, 256)), dim3(256), 0, pcacc_stream(stream), w, c_in, c_up, strides[0],
                       strides[1], strides[2], strides[3], out_fwd, out_bwd);
    PCACC_CHECK_LAUNCH();
    return 0;
}

// C[P][N] = A[P][K] B[N][K]^T on 128-row tiles; blockIdx.y = group of NT = 32 CT output columns.
//   UB_UP : A = in rows [P][K = c_in] (pixel pitch in_pitch), N = 4 c_up, the result row (a, b, co) goes to out[n, 2y+a, 2x+b, co] (+ bias[co]),
//           pixel pitch out_pitch
//   UB_S2D: A = the space-to-depth view of in = dy [n, 2h, 2w, c_up] (pixel pitch in_pitch), K = 4 c_up, N = c_in, out rows [P][c_in]
// Wave = 32 rows x NT columns: v_mfma_f32_32x32x16_bf16 with A = 32 weight rows x 16 k, B = 16 k x 32 tile rows (lane = row in D).
template <int CT, int MODE>
__global__ __launch_bounds__(256) void upconv_bf16_kernel(const uint16_t *__restrict__ in, const uint16_t *__restrict__ wp, const float *__restrict__ bias,
                                                          uint16_t *__restrict__ out, int n_img, int h, int w, int c_in, int c_up, int in_pitch,
                                                          int out_pitch)
{
    constexpr int NT = 32 * CT, YS = NT + 8;
    constexpr int REGION = UB_ROWS * (UB_XS > YS ? UB_XS : YS);            // input tile, later the output tile
    constexpr int W_PER = NT * 8 / 256;                                    // 16-byte weight pieces per thread and k chunk
    extern __shared__ __attribute__((aligned(16))) uint16_t ub_lds[];
    uint16_t *xs = ub_lds;                                                 // [128][XS]
    uint16_t *ws = ub_lds + REGION;                                        // [NT][XS]
    int64_t *rowpix = reinterpret_cast<int64_t *>(ws + NT * UB_XS);       // [128]: top-left pixel of the row's 2 x 2 block in the large map, -1 past the end
    const int K = MODE == UB_UP ? c_in : 4 * c_up;
    const int64_t P = (int64_t)n_img * h * w;
    const int64_t row0 = (int64_t)blockIdx.x * UB_ROWS;
    const int n0 = blockIdx.y * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;

    if (threadIdx.x < UB_ROWS) {
        const int64_t p = row0 + threadIdx.x;
        int64_t v = -1;
        if (p < P) {
            const int64_t img = p / ((int64_t)h * w);
            const int rem = (int)(p - img * h * w);
            const int y = rem / w, x = rem - y * w;
            v = (img * 2 * h + 2 * y) * 2 * w + 2 * x;
        }
        rowpix[threadIdx.x] = v;
    }
    __syncthreads();

    uint4 xreg[4], wreg[W_PER];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * 256;
            const int r = c >> 3, kk = k0 + (c & 7) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row0 + r < P) {
                if (MODE == UB_UP) v = *reinterpret_cast<const uint4 *>(in + (row0 + r) * in_pitch + kk);
                else {
                    const int ab = kk / c_up, co = kk - ab * c_up;
                    v = *reinterpret_cast<const uint4 *>(in + (rowpix[r] + (ab >> 1) * 2 * w + (ab & 1)) * in_pitch + co);
                }
            }
            xreg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = threadIdx.x + q * 256;
            wreg[q] = *reinterpret_cast<const uint4 *>(wp + (int64_t)(n0 + (c >> 3)) * K + k0 + (c & 7) * 8);
        }
    };

    ub_f32x16_t acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += UB_KC) {
        __syncthreads();                                                   // the previous chunk's fragments are read
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * 256;
            *reinterpret_cast<uint4 *>(xs + (c >> 3) * UB_XS + (c & 7) * 8) = xreg[q];
        }
#pragma unroll
        for (int q = 0; q < W_PER; ++q) {
            const int c = threadIdx.x + q * 256;
            *reinterpret_cast<uint4 *>(ws + (c >> 3) * UB_XS + (c & 7) * 8) = wreg[q];
        }
        __syncthreads();
        if (k0 + UB_KC < K) fetch(k0 + UB_KC);                             // in flight during the MFMAs
        const uint16_t *xrow = xs + (wave * 32 + lp) * UB_XS + lh * 8;
        const uint16_t *wrow = ws + lp * UB_XS + lh * 8;
#pragma unroll
        for (int kc = 0; kc < UB_KC / 16; ++kc) {
            const ub_bf16x8_t b = *reinterpret_cast<const ub_bf16x8_t *>(xrow + kc * 16);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const ub_bf16x8_t a = *reinterpret_cast<const ub_bf16x8_t *>(wrow + ct * 32 * UB_XS + kc * 16);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ct], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                                       // every wave is done with the input tile
    uint16_t *yrow = xs + (wave * 32 + lp) * YS;                           // lane = row; quads of 4 consecutive columns
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = ct * 32 + 8 * g + 4 * lh;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == UB_UP && bias) bv = *reinterpret_cast<const float4 *>(bias + (n0 + c) % c_up);
            uint2 pk;
            pk.x = pcacc_pack_bf16x2(acc[ct][4 * g] + bv.x, acc[ct][4 * g + 1] + bv.y);
            pk.y = pcacc_pack_bf16x2(acc[ct][4 * g + 2] + bv.z, acc[ct][4 * g + 3] + bv.w);
            *reinterpret_cast<uint2 *>(yrow + c) = pk;
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NT * UB_ROWS / 8 / 256; ++q) {
        const int c = threadIdx.x + q * 256;
        const int r = c / (NT / 8), col = (c % (NT / 8)) * 8;
        if (row0 + r >= P) continue;
        const uint4 v = *reinterpret_cast<const uint4 *>(xs + r * YS + col);
        if (MODE == UB_S2D) *reinterpret_cast<uint4 *>(out + (row0 + r) * out_pitch + n0 + col) = v;
        else {
            const int n = n0 + col, ab = n / c_up, co = n - ab * c_up;
            *reinterpret_cast<uint4 *>(out + (rowpix[r] + (ab >> 1) * 2 * w + (ab & 1)) * out_pitch + co) = v;
        }
    }
}

static bool upconv_bf16_shape_ok(int c_in, int c_up) { return c_in >= 64 && c_in % 64 == 0 && c_up >= 32 && c_up % 32 == 0; }

extern "C" int pcacc_upconv2x2_bf16_supported(int32_t c_in, int32_t c_up) { return upconv_bf16_shape_ok(c_in, c_up) ? 1 : 0; }

extern "C" int pcacc_upconv2x2_bf16(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int32_t n_img, int32_t h, int32_t w,
                                    int32_t c_in, int32_t c_up, int32_t direction, int32_t in_pitch, int32_t out_pitch, void *stream)
{
    if (!in || !wp || !out || n_img < 1 || h < 1 || w < 1 || !upconv_bf16_shape_ok(c_in, c_up) || (direction != 0 && direction != 1)) return PCACC_E_ARG;
    const int in_c = direction == 0 ? c_in : c_up, out_c = direction == 0 ? c_up : c_in;
    if (in_pitch < in_c || out_pitch < out_c || in_pitch % 8 || out_pitch % 8) return PCACC_E_ARG;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)wp) & 15) return PCACC_E_ARG;
    const int64_t P = (int64_t)n_img * h * w;
    const int N = direction == 0 ? 4 * c_up : c_in;
    const int ct = N % 128 == 0 ? 4 : 2;
    const int nt = 32 * ct, ys = nt + 8;
    const size_t lds = (size_t)(UB_ROWS * (UB_XS > ys ? UB_XS : ys) + nt * UB_XS) * sizeof(uint16_t) + UB_ROWS * sizeof(int64_t);
    const dim3 grid((unsigned)((P + UB_ROWS - 1) / UB_ROWS), (unsigned)(N / nt));
    hipStream_t st = pcacc_stream(stream);
#define UB_LAUNCH(CT, MODE) hipLaunchKernelGGL((upconv_bf16_kernel<CT, MODE>), grid, dim3(256), lds, st, in, wp, bias, out, n_img, h, w, c_in, c_up, in_pitch, out_pitch)
    if (direction == 0) { if (ct == 4) UB_LAUNCH(4, UB_UP); else UB_LAUNCH(2, UB_UP); }
    else { if (ct == 4) UB_LAUNCH(4, UB_S2D); else UB_LAUNCH(2, UB_S2D); }
#undef UB_LAUNCH
    PCACC_CHECK_LAUNCH();
    return 0;
}

// ---- weight and bias gradient ---------------------------------------------------------------------------------------------------------
// partial[split][n' (4 c_up)][ci] += sum over the split's pixels of dy_s2d[p][n'] x[p][ci]; colsum[split][n'] = sum of dy_s2d[p][n'].
// Workgroup = (split, 64 input channels, 128 columns n'): 64-row tiles of x and of the gathered dy staged row-major, fragments (k = pixels)
// through the LDS transpose read (ds_read_b64_tr_b16, as csrc/mlp_mfma.hip's rows_wgrad_bf16_kernel); wave = 32 columns x 64 channels.
#define UW_ROWS 64
#define UW_CI 64
#define UW_N 128
__global__ __launch_bounds__(256) void upconv_bf16_wgrad_kernel(const uint16_t *__restrict__ dy, int dy_pitch, const uint16_t *__restrict__ x, int x_pitch,
                                                                float *__restrict__ partial, float *__restrict__ colsum, int n_img, int h, int w,
                                                                int c_in, int c_up, int64_t rows_per_split)
{
    constexpr int KS = pcacc_tr_stride(UW_CI), NS = pcacc_tr_stride(UW_N);
    __shared__ __attribute__((aligned(16))) uint16_t sx[UW_ROWS * KS];
    __shared__ __attribute__((aligned(16))) uint16_t sdy[UW_ROWS * NS];
    __shared__ float cs[256][8];
    const int n4 = 4 * c_up;
    const int64_t P = (int64_t)n_img * h * w;
    const int split = blockIdx.x, ci0 = blockIdx.y * UW_CI, nb0 = blockIdx.z * UW_N;
    const int64_t r_begin = (int64_t)split * rows_per_split, r_end = min(P, r_begin + rows_per_split);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lp = lane & 31, lh = lane >> 5;
    ub_f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // pieces (8 bf16) of a tile: x 64 rows x 8, dy 64 rows x 16.  A thread's dy pieces all sit in column group threadIdx.x % 16.
    uint4 xreg[2], yreg[4];
    auto fetch = [&](int64_t t0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = threadIdx.x + q * 256;
            const int64_t p = t0 + (c >> 3);
            xreg[q] = p < r_end ? *reinterpret_cast<const uint4 *>(x + p * x_pitch + ci0 + (c & 7) * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * 256;
            const int64_t p = t0 + (c >> 4);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (p < r_end) {
                const int pi = (int)p, img = pi / (h * w);                 // the launcher keeps n_img h w below 2^29
                const int rem = pi - img * h * w;
                const int y = rem / w, xx = rem - y * w;
                const int n = nb0 + (c & 15) * 8, ab = n / c_up, co = n - ab * c_up;
                v = *reinterpret_cast<const uint4 *>(dy + ((((int64_t)img * 2 * h + 2 * y + (ab >> 1)) * 2 * w) + 2 * xx + (ab & 1)) * dy_pitch + co);
            }
            yreg[q] = v;
        }
    };
    const int g = lane >> 4, li = lane & 15;
    const int tr_row = (g >> 1) * 8 + (li >> 2), tr_col = (g & 1) * 16 + (li & 3) * 4;
    if (r_begin < r_end) fetch(r_begin);
    for (int64_t t0 = r_begin; t0 < r_end; t0 += UW_ROWS) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = threadIdx.x + q * 256;
            uint2 *dst = reinterpret_cast<uint2 *>(sx + (c >> 3) * KS + (c & 7) * 8);
            dst[0] = make_uint2(xreg[q].x, xreg[q].y);
            dst[1] = make_uint2(xreg[q].z, xreg[q].w);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + q * 256;
            const uint4 v = yreg[q];
            uint2 *dst = reinterpret_cast<uint2 *>(sdy + (c >> 4) * NS + (c & 15) * 8);
            dst[0] = make_uint2(v.x, v.y);
            dst[1] = make_uint2(v.z, v.w);
            if (blockIdx.y == 0) {                                         // uniform: the bias gradient is collected once per column group
                csum[0] += pcacc_bf16_lo(v.x); csum[1] += pcacc_bf16_hi(v.x); csum[2] += pcacc_bf16_lo(v.y); csum[3] += pcacc_bf16_hi(v.y);
                csum[4] += pcacc_bf16_lo(v.z); csum[5] += pcacc_bf16_hi(v.z); csum[6] += pcacc_bf16_lo(v.w); csum[7] += pcacc_bf16_hi(v.w);
            }
        }
        __syncthreads();
        if (t0 + UW_ROWS < r_end) fetch(t0 + UW_ROWS);
        const uint16_t *pa = sdy + tr_row * NS + wave * 32 + tr_col;
        const uint16_t *pb = sx + tr_row * KS + tr_col;
#pragma unroll
        for (int r0 = 0; r0 < UW_ROWS; r0 += 16) {
            ub_frag a, b0, b1;
            a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ub_s16x4 __attribute__((address_space(3))) *)(pa + r0 * NS));
            a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ub_s16x4 __attribute__((address_space(3))) *)(pa + (r0 + 4) * NS));
            b0.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ub_s16x4 __attribute__((address_space(3))) *)(pb + r0 * KS));
            b0.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ub_s16x4 __attribute__((address_space(3))) *)(pb + (r0 + 4) * KS));
            b1.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ub_s16x4 __attribute__((address_space(3))) *)(pb + 32 + r0 * KS));
            b1.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ub_s16x4 __attribute__((address_space(3))) *)(pb + 32 + (r0 + 4) * KS));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b0.v, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b1.v, acc[1], 0, 0, 0);
        }
    }
    float *mine = partial + (int64_t)split * n4 * c_in;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nb0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            mine[(int64_t)n * c_in + ci0 + t * 32 + lp] = acc[t][r];
        }
    if (blockIdx.y == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[threadIdx.x][j] = csum[j];
        __syncthreads();
        if (threadIdx.x < UW_N) {
            const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) s += cs[grp + 16 * m][j];
            colsum[(int64_t)split * n4 + nb0 + threadIdx.x] = s;
        }
    }
}

// dw f32 [c_in][c_up][2][2] written through the element strides (si, so, sy, sx) of the weight it belongs to, db f32 [c_up] (may be NULL).
// Workgroup = 32 consecutive elements of the partial layout ([n'][ci]: coalesced reads) x 8 lanes over the splits, summed through LDS; one
// scattered store per element.  (As first written -- a thread per element of the torch layout looping over up to 1024 splits with reads
// c_up c_in floats apart -- the launch took 78 us per layer, more than the products.)
__global__ __launch_bounds__(256) void upconv_bf16_wgrad_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ colsum, int splits,
                                                                       int c_in, int c_up, float *__restrict__ dw, float *__restrict__ db, int64_t si,
                                                                       int64_t so, int64_t sy, int64_t sx)
{
    __shared__ float part[8][32];
    const int n4 = 4 * c_up;
    const int64_t total = (int64_t)n4 * c_in;
    const int64_t e = (int64_t)blockIdx.x * 32 + (threadIdx.x & 31);
    const int sl = threadIdx.x >> 5;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        const float *src = partial + e;
        int s = sl;
        for (; s + 8 < splits; s += 16) {
            s0 += src[(int64_t)s * total];
            s1 += src[(int64_t)(s + 8) * total];
        }
        if (s < splits) s0 += src[(int64_t)s * total];
    } else if (db && e < total + c_up) {                        // bias gradient: the four (a, b) column sums of every split
        const int co = (int)(e - total);
        for (int s = sl; s < splits; s += 8) {
            const float *c = colsum + (int64_t)s * n4 + co;
            s0 += c[0] + c[c_up];
            s1 += c[2 * c_up] + c[3 * c_up];
        }
    }
    part[sl][threadIdx.x & 31] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += part[k][threadIdx.x];
        if (e < total) {
            const int n = (int)(e / c_in), ci = (int)(e - (int64_t)n * c_in);
            const int ab = n / c_up, co = n - ab * c_up;
            dw[ci * si + co * so + (ab >> 1) * sy + (ab & 1) * sx] = v;
        } else if (db && e < total + c_up) db[e - total] = v;
    }
}

static int upconv_wgrad_splits(int64_t P, int c_in, int c_up, int64_t *rows_per_split)
{
    const int tiles = (c_in / UW_CI) * (4 * c_up / UW_N);
    int64_t s = (PCACC_CUS * 4 + tiles - 1) / tiles;
    const int64_t chunks = (P + UW_ROWS - 1) / UW_ROWS;
    if (s > chunks) s = chunks;
    if (s < 1) s = 1;
    int64_t per = ((chunks + s - 1) / s) * UW_ROWS;
    s = (P + per - 1) / per;
    *rows_per_split = per;
    return (int)s;
}

extern "C" int pcacc_upconv2x2_bf16_wgrad_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, size_t *bytes)
{
    if (!bytes || n_img < 1 || h < 1 || w < 1 || !upconv_bf16_shape_ok(c_in, c_up)) return PCACC_E_ARG;
    int64_t per;
    const int s = upconv_wgrad_splits((int64_t)n_img * h * w, c_in, c_up, &per);
    *bytes = (size_t)s * ((size_t)4 * c_up * c_in + 4 * c_up) * sizeof(float);
    return 0;
}

extern "C" int pcacc_upconv2x2_bf16_wgrad(const uint16_t *dy, int32_t dy_pitch, const uint16_t *x, int32_t x_pitch, float *dw, const int64_t *dw_strides,
                                          float *db, int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, void *workspace,
                                          size_t workspace_bytes, void *stream)
{
    if (!dy || !x || !dw || !workspace || n_img < 1 || h < 1 || w < 1 || !upconv_bf16_shape_ok(c_in, c_up)) return PCACC_E_ARG;
    if (dy_pitch < c_up || x_pitch < c_in || dy_pitch % 8 || x_pitch % 8 || (((uintptr_t)dy | (uintptr_t)x) & 15)) return PCACC_E_ARG;
    const int64_t P = (int64_t)n_img * h * w;
    if (P >= (1 << 29)) return PCACC_E_ARG;
    int64_t per;
    const int s = upconv_wgrad_splits(P, c_in, c_up, &per);
    const size_t n4 = (size_t)4 * c_up;
    if (workspace_bytes < (size_t)s * (n4 * c_in + n4) * sizeof(float)) return PCACC_E_WORKSPACE;
    float *partial = static_cast<float *>(workspace), *colsum = partial + (size_t)s * n4 * c_in;
    hipStream_t st = pcacc_stream(stream);
    hipLaunchKernelGGL(upconv_bf16_wgrad_kernel, dim3(s, c_in / UW_CI, (unsigned)(n4 / UW_N)), dim3(256), 0, st, dy, dy_pitch, x, x_pitch, partial, colsum,
                       n_img, h, w, c_in, c_up, per);
    const int64_t elems = (int64_t)n4 * c_in + c_up;
    hipLaunchKernelGGL(upconv_bf16_wgrad_reduce_kernel, dim3((unsigned)((elems + 31) / 32)), dim3(256), 0, st, partial, colsum, s, c_in, c_up, dw, db,
                       dw_strides ? dw_strides[0] : (int64_t)4 * c_up, dw_strides ? dw_strides[1] : 4, dw_strides ? dw_strides[2] : 2,
                       dw_strides ? dw_strides[3] : 1);
    PCACC_CHECK_LAUNCH();
    return 0;
}
