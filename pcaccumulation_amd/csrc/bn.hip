// A10 point heads: nn.BatchNorm1d over [rows, c] rows in training mode (models/unet.py:240-245, SegHead1D = Linear, BatchNorm1d,
// ReLU, Linear on the K foreground points; batch statistics over the rows, SURVEY.md appendix C trap 16).
//
// The library kernels walk this [rows, 128] layout at 0.5 TB/s (0.18 ms statistics + 0.05 ms transform forward, 0.21 + 0.08 ms
// backward per head at 380 k rows).  Here: four streaming passes, 16 bytes per lane, one lane per (row, 8 or 4 channels):
//   forward : per-workgroup partial sums of x and x^2 per channel -> finalize (mean, 1/sqrt(var + eps), running statistics with
//             the unbiased variance, as torch) -> y = (x - mean) * invstd * gamma + beta
//   backward: partial sums of dy and dy * xhat -> finalize (dbeta, dgamma) -> dx = gamma * invstd * (dy - dbeta/n - xhat * dgamma/n)
// Sums: fp32 per lane over its ~10-20 rows, fp32 per workgroup, float64 across workgroups.
#include "common.h"

#define BN_MAX_C 256

template <bool BF>
__device__ __forceinline__ void bn_load(const void *p, int64_t vec_index, float (&v)[BF ? 8 : 4])
{
    if (BF) {
        const uint4 r = reinterpret_cast<const uint4 *>(p)[vec_index];
        v[0] = pcacc_bf16_lo(r.x), v[1] = pcacc_bf16_hi(r.x), v[2] = pcacc_bf16_lo(r.y), v[3] = pcacc_bf16_hi(r.y);
        v[(BF ? 8 : 4) - 4] = pcacc_bf16_lo(r.z), v[(BF ? 8 : 4) - 3] = pcacc_bf16_hi(r.z);
        v[(BF ? 8 : 4) - 2] = pcacc_bf16_lo(r.w), v[(BF ? 8 : 4) - 1] = pcacc_bf16_hi(r.w);
    } else {
        const float4 r = reinterpret_cast<const float4 *>(p)[vec_index];
        v[0] = r.x, v[1] = r.y, v[2] = r.z, v[3] = r.w;
    }
}

template <bool BF>
__device__ __forceinline__ void bn_store(void *p, int64_t vec_index, const float (&v)[BF ? 8 : 4])
{
    if (BF)
        reinterpret_cast<uint4 *>(p)[vec_index] = make_uint4(pcacc_pack_bf16x2(v[0], v[1]), pcacc_pack_bf16x2(v[2], v[3]),
                                                             pcacc_pack_bf16x2(v[(BF ? 8 : 4) - 4], v[(BF ? 8 : 4) - 3]),
                                                             pcacc_pack_bf16x2(v[(BF ? 8 : 4) - 2], v[(BF ? 8 : 4) - 1]));
    else
        reinterpret_cast<float4 *>(p)[vec_index] = make_float4(v[0], v[1], v[2], v[3]);
}

// partial[block][2][c]: sums of a and a*b' per channel over the rows this workgroup visits.
//   forward  (B == NULL): a = x,  second sum = x^2
//   backward            : a = dy, second sum = dy * (x - mean) * invstd
// relu_gamma / relu_beta (backward of BatchNorm + ReLU): dy counts only where the forward output x * scale + shift was > 0
template <bool BF>
__global__ __launch_bounds__(256) void bn_sums_kernel(const void *__restrict__ A, const void *__restrict__ B, const float *__restrict__ mean,
                                                      const float *__restrict__ invstd, int64_t rows, int c, float *__restrict__ partial,
                                                      bool relu = false, const float *__restrict__ relu_gamma = nullptr,
                                                      const float *__restrict__ relu_beta = nullptr)
{
    constexpr int V = BF ? 8 : 4;
    __shared__ float red[2 * 256 * 8];
    const int lanes = c / V;                                        // lanes per row
    const int rows_per_pass = 256 / lanes;
    const int sub = threadIdx.x % lanes, rsub = threadIdx.x / lanes;
    float s0[V], s1[V], mu[V], is[V], rsc[V], rsh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        s0[k] = s1[k] = 0.f;
        mu[k] = B ? mean[sub * V + k] : 0.f;
        is[k] = B ? invstd[sub * V + k] : 0.f;
        const float g = (relu && relu_gamma) ? relu_gamma[sub * V + k] : 1.f;
        rsc[k] = is[k] * g;                                         // the forward's scale and shift, formed the same way (bn_apply_kernel)
        rsh[k] = ((relu && relu_beta) ? relu_beta[sub * V + k] : 0.f) - mu[k] * is[k] * g;
    }
    if (rsub < rows_per_pass)
        for (int64_t row = (int64_t)blockIdx.x * rows_per_pass + rsub; row < rows; row += (int64_t)gridDim.x * rows_per_pass) {
            float a[V], b[V];
            bn_load<BF>(A, row * lanes + sub, a);
            if (B) bn_load<BF>(B, row * lanes + sub, b);
            if (relu) {
#pragma unroll
                for (int k = 0; k < V; ++k) a[k] = (b[k] * rsc[k] + rsh[k] > 0.f) ? a[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < V; ++k) {
                s0[k] += a[k];
                s1[k] += B ? a[k] * ((b[k] - mu[k]) * is[k]) : a[k] * a[k];
            }
        }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        red[threadIdx.x * V + k] = s0[k];
        red[256 * 8 + threadIdx.x * V + k] = s1[k];
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < 2 * c; ch += 256) {
        const int which = ch / c, cc = ch % c;
        const int l = cc / V, k = cc % V;
        float t = 0.f;
        for (int r = 0; r < rows_per_pass; ++r) t += red[which * 256 * 8 + (r * lanes + l) * V + k];
        partial[(int64_t)blockIdx.x * 2 * c + ch] = t;
    }
}

// one workgroup per channel: the two sums of that channel over all partials (float64)
__device__ __forceinline__ void bn_channel_sums(const float *__restrict__ partial, int nb, int c, int ch, double *red, double &s, double &q)
{
    s = 0, q = 0;
    for (int b = threadIdx.x; b < nb; b += 256) {
        s += partial[(int64_t)b * 2 * c + ch];
        q += partial[(int64_t)b * 2 * c + c + ch];
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) {
        s += __shfl_down(s, d, 64);
        q += __shfl_down(q, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[(threadIdx.x >> 6) * 2] = s;
        red[(threadIdx.x >> 6) * 2 + 1] = q;
    }
    __syncthreads();
    s = red[0] + red[2] + red[4] + red[6];
    q = red[1] + red[3] + red[5] + red[7];
}

// forward statistics: mean, invstd, running statistics
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float *__restrict__ partial, int nb, int c, int64_t rows, float eps,
                                                              float momentum, float *__restrict__ running_mean,
                                                              float *__restrict__ running_var, float *__restrict__ save_mean,
                                                              float *__restrict__ save_invstd)
{
    __shared__ double red[8];
    const int ch = blockIdx.x;
    double s, q;
    bn_channel_sums(partial, nb, c, ch, red, s, q);
    if (threadIdx.x != 0) return;
    const double n = (double)rows, m = s / n;
    double var = q / n - m * m;
    if (var < 0) var = 0;
    save_mean[ch] = (float)m;
    save_invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * m);
    if (running_var) running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * (rows > 1 ? var * n / (n - 1.0) : var));
}

// backward sums -> dbeta = sum dy, dgamma = sum dy * xhat
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float *__restrict__ partial, int nb, int c, float *__restrict__ dbeta,
                                                              float *__restrict__ dgamma)
{
    __shared__ double red[8];
    const int ch = blockIdx.x;
    double s, q;
    bn_channel_sums(partial, nb, c, ch, red, s, q);
    if (threadIdx.x != 0) return;
    dbeta[ch] = (float)s;
    dgamma[ch] = (float)q;
}

// forward (DY == NULL): out = (x - mean) * invstd * gamma + beta
// backward            : out = gamma * invstd * (dy - dbeta / n - xhat * dgamma / n)
template <bool BF>
__global__ __launch_bounds__(256) void bn_apply_kernel(const void *__restrict__ X, const void *__restrict__ DY, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta_or_dbeta, const float *__restrict__ dgamma, int64_t rows,
                                                       int c, void *__restrict__ out, bool relu = false, const float *__restrict__ relu_beta = nullptr,
                                                       float *__restrict__ out_amax = nullptr, uint16_t *__restrict__ out16 = nullptr)
{
    constexpr int V = BF ? 8 : 4;                                   // out16 (f32 rows only): a bf16 copy of `out` from the same store phase ('mixed' mode shadow)
    float omax = 0.f;                                               // out_amax: 256 partial absolute maxima of `out` (the layout of pcacc_absmax256; zero-filled by the caller)
    const int lanes = c / V;
    const int64_t total = rows * lanes;
    const float inv_n = 1.f / (float)rows;
    // 256 % lanes == 0, so a lane keeps its channel group over the whole grid-stride loop: per-channel constants in registers
    const int sub = threadIdx.x % lanes;
    float scale[V], shift[V], kx[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int ch = sub * V + k;
        const float g = gamma ? gamma[ch] : 1.f, is = invstd[ch], mu = mean[ch];
        if (!DY) {                                                  // y = x * scale + shift
            scale[k] = is * g;
            shift[k] = (beta_or_dbeta ? beta_or_dbeta[ch] : 0.f) - mu * is * g;
            kx[k] = 0.f;
        } else {                                                    // dx = dy * scale + xhat * kx + shift,  xhat = x * is - mu * is
            scale[k] = g * is;
            shift[k] = -g * is * beta_or_dbeta[ch] * inv_n;
            kx[k] = -g * is * dgamma[ch] * inv_n;
        }
    }
    float is_[V], mis[V], rsc[V], rsh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        is_[k] = invstd[sub * V + k];
        mis[k] = mean[sub * V + k] * is_[k];
        const float g = gamma ? gamma[sub * V + k] : 1.f;
        rsc[k] = is_[k] * g;                                        // forward scale / shift (the ReLU mask of the backward)
        rsh[k] = ((relu && relu_beta) ? relu_beta[sub * V + k] : 0.f) - mean[sub * V + k] * is_[k] * g;
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        float x[V], o[V];
        bn_load<BF>(X, e, x);
        if (!DY) {
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = x[k] * scale[k] + shift[k];
            if (relu) {
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = fmaxf(o[k], 0.f);
            }
        } else {
            float g[V];
            bn_load<BF>(DY, e, g);
            if (relu) {
#pragma unroll
                for (int k = 0; k < V; ++k) g[k] = (x[k] * rsc[k] + rsh[k] > 0.f) ? g[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = g[k] * scale[k] + (x[k] * is_[k] - mis[k]) * kx[k] + shift[k];
        }
        bn_store<BF>(out, e, o);
        if (!BF && out16) reinterpret_cast<uint2 *>(out16)[e] = make_uint2(pcacc_pack_bf16x2(o[0], o[1]), pcacc_pack_bf16x2(o[2], o[3]));
        if (out_amax) {                                             // uniform
#pragma unroll
            for (int k = 0; k < V; ++k) {
                omax = fmaxf(omax, fabsf(o[k]));
                if (o[k] != o[k]) omax = __builtin_inff();
            }
        }
    }
    if (out_amax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) omax = fmaxf(omax, __shfl_xor(omax, d, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(out_amax) + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255), __float_as_uint(omax));
    }
}

static int bn_blocks(int64_t rows, int c, int v)
{
    const int rows_per_pass = 256 / (c / v);
    int64_t nb = (rows + (int64_t)rows_per_pass * 8 - 1) / ((int64_t)rows_per_pass * 8);      // >= 8 rows per lane
    if (nb > PCACC_CUS * 4) nb = PCACC_CUS * 4;
    return nb < 1 ? 1 : (int)nb;
}

static int bn_args(int dtype, int64_t rows, int c)
{
    if (dtype != PCACC_F32 && dtype != PCACC_BF16) return PCACC_E_ARG;
    const int v = dtype == PCACC_BF16 ? 8 : 4;
    if (rows < 1 || c < v || c > BN_MAX_C || c % v || 256 % (c / v)) return PCACC_E_ARG;
    return PCACC_OK;
}

extern "C" int pcacc_bn_rows_workspace_bytes(int64_t rows, int32_t c, size_t *bytes)
{
    if (!bytes || rows < 1 || c < 4 || c > BN_MAX_C) return PCACC_E_ARG;
    *bytes = (size_t)PCACC_CUS * 4 * 2 * c * sizeof(float);
    return PCACC_OK;
}

static int bn_forward_any(const void *x, int dtype, int64_t rows, int32_t c, const float *gamma, const float *beta, float eps,
                          float momentum, float *running_mean, float *running_var, void *y, float *save_mean,
                          float *save_invstd, void *workspace, size_t workspace_bytes, void *stream, bool relu, uint16_t *y16 = nullptr,
                          float *y_amax = nullptr)
{
    if ((y16 || y_amax) && dtype != PCACC_F32) return PCACC_E_ARG;
    if (bn_args(dtype, rows, c) != PCACC_OK || !x || !y || !save_mean || !save_invstd || !workspace) return PCACC_E_ARG;
    const bool bf = dtype == PCACC_BF16;
    const int nb = bn_blocks(rows, c, bf ? 8 : 4);
    if (workspace_bytes < (size_t)nb * 2 * c * sizeof(float)) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    float *partial = reinterpret_cast<float *>(workspace);
    if (bf) bn_sums_kernel<true><<<nb, 256, 0, s>>>(x, nullptr, nullptr, nullptr, rows, c, partial);
    else bn_sums_kernel<false><<<nb, 256, 0, s>>>(x, nullptr, nullptr, nullptr, rows, c, partial);
    bn_finalize_fwd_kernel<<<c, 256, 0, s>>>(partial, nb, c, rows, eps, momentum, running_mean, running_var, save_mean, save_invstd);
    const int grid = pcacc_grid(rows * (c / (bf ? 8 : 4)), 256, PCACC_CUS * 16);
    if (bf) bn_apply_kernel<true><<<grid, 256, 0, s>>>(x, nullptr, save_mean, save_invstd, gamma, beta, nullptr, rows, c, y, relu);
    else bn_apply_kernel<false><<<grid, 256, 0, s>>>(x, nullptr, save_mean, save_invstd, gamma, beta, nullptr, rows, c, y, relu, nullptr, y_amax, y16);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// pcacc_bn_rows_forward / pcacc_bn_relu_rows_forward (relu != 0) on f32 rows with two more outputs from the same store phase ('mixed' compute mode):
// y16 = y as bf16 (the shadow the bf16 backward reads), y_amax = 256 partial absolute maxima of y (zero-filled by the caller; either may be NULL)
extern "C" int pcacc_bn_rows_forward_dual(const float *x, int64_t rows, int32_t c, const float *gamma, const float *beta, float eps, float momentum,
                                          float *running_mean, float *running_var, int32_t relu, float *y, uint16_t *y16, float *y_amax,
                                          float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes, void *stream)
{
    return bn_forward_any(x, PCACC_F32, rows, c, gamma, beta, eps, momentum, running_mean, running_var, y, save_mean, save_invstd, workspace,
                          workspace_bytes, stream, relu != 0, y16, y_amax);
}

extern "C" int pcacc_bn_rows_forward(const void *x, int dtype, int64_t rows, int32_t c, const float *gamma, const float *beta, float eps,
                                     float momentum, float *running_mean, float *running_var, void *y, float *save_mean,
                                     float *save_invstd, void *workspace, size_t workspace_bytes, void *stream)
{
    return bn_forward_any(x, dtype, rows, c, gamma, beta, eps, momentum, running_mean, running_var, y, save_mean, save_invstd, workspace,
                          workspace_bytes, stream, false);
}

// y = max(BatchNorm(x), 0): the ReLU that follows the normalisation in SegHead2D (models/unet.py:264-268) in the same pass
extern "C" int pcacc_bn_relu_rows_forward(const void *x, int dtype, int64_t rows, int32_t c, const float *gamma, const float *beta, float eps,
                                          float momentum, float *running_mean, float *running_var, void *y, float *save_mean,
                                          float *save_invstd, void *workspace, size_t workspace_bytes, void *stream)
{
    return bn_forward_any(x, dtype, rows, c, gamma, beta, eps, momentum, running_mean, running_var, y, save_mean, save_invstd, workspace,
                          workspace_bytes, stream, true);
}

static int bn_backward_any(const void *grad_y, const void *x, int dtype, int64_t rows, int32_t c, const float *gamma,
                           const float *save_mean, const float *save_invstd, void *grad_x, float *grad_gamma, float *grad_beta,
                           void *workspace, size_t workspace_bytes, void *stream, bool relu, const float *beta, float *grad_x_amax = nullptr)
{
    if (bn_args(dtype, rows, c) != PCACC_OK || !grad_y || !x || !save_mean || !save_invstd || !grad_x || !grad_gamma || !grad_beta || !workspace)
        return PCACC_E_ARG;
    const bool bf = dtype == PCACC_BF16;
    const int nb = bn_blocks(rows, c, bf ? 8 : 4);
    if (workspace_bytes < (size_t)nb * 2 * c * sizeof(float)) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    float *partial = reinterpret_cast<float *>(workspace);
    if (bf) bn_sums_kernel<true><<<nb, 256, 0, s>>>(grad_y, x, save_mean, save_invstd, rows, c, partial, relu, gamma, beta);
    else bn_sums_kernel<false><<<nb, 256, 0, s>>>(grad_y, x, save_mean, save_invstd, rows, c, partial, relu, gamma, beta);
    bn_finalize_bwd_kernel<<<c, 256, 0, s>>>(partial, nb, c, grad_beta, grad_gamma);
    const int grid = pcacc_grid(rows * (c / (bf ? 8 : 4)), 256, PCACC_CUS * 16);
    if (bf) bn_apply_kernel<true><<<grid, 256, 0, s>>>(x, grad_y, save_mean, save_invstd, gamma, grad_beta, grad_gamma, rows, c, grad_x, relu, beta, grad_x_amax);
    else bn_apply_kernel<false><<<grid, 256, 0, s>>>(x, grad_y, save_mean, save_invstd, gamma, grad_beta, grad_gamma, rows, c, grad_x, relu, beta, grad_x_amax);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// pcacc_bn_rows_backward / pcacc_bn_relu_rows_backward (relu != 0: `beta` as there) with the 256 partial absolute maxima of grad_x from the same store
// phase (grad_x_amax zero-filled by the caller; layout of pcacc_absmax256): the fp32x3 layer in front of the normalisation scales its incoming gradient
// by them (one pass over grad_x less)
extern "C" int pcacc_bn_rows_backward_m(const void *grad_y, const void *x, int dtype, int64_t rows, int32_t c, const float *gamma, const float *beta,
                                        int32_t relu, const float *save_mean, const float *save_invstd, void *grad_x, float *grad_x_amax,
                                        float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!grad_x_amax) return PCACC_E_ARG;
    return bn_backward_any(grad_y, x, dtype, rows, c, gamma, save_mean, save_invstd, grad_x, grad_gamma, grad_beta, workspace, workspace_bytes, stream,
                           relu != 0, relu ? beta : nullptr, grad_x_amax);
}

extern "C" int pcacc_bn_rows_backward(const void *grad_y, const void *x, int dtype, int64_t rows, int32_t c, const float *gamma,
                                      const float *save_mean, const float *save_invstd, void *grad_x, float *grad_gamma, float *grad_beta,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    return bn_backward_any(grad_y, x, dtype, rows, c, gamma, save_mean, save_invstd, grad_x, grad_gamma, grad_beta, workspace, workspace_bytes, stream,
                           false, nullptr);
}

// backward of pcacc_bn_relu_rows_forward: grad_y counts where the forward output was > 0 (recomputed from x, the saved statistics, gamma, beta)
extern "C" int pcacc_bn_relu_rows_backward(const void *grad_y, const void *x, int dtype, int64_t rows, int32_t c, const float *gamma,
                                           const float *beta, const float *save_mean, const float *save_invstd, void *grad_x, float *grad_gamma,
                                           float *grad_beta, void *workspace, size_t workspace_bytes, void *stream)
{
    return bn_backward_any(grad_y, x, dtype, rows, c, gamma, save_mean, save_invstd, grad_x, grad_gamma, grad_beta, workspace, workspace_bytes, stream,
                           true, beta);
}
