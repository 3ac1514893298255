// L1/L2: the loss terms that read the path's own tensors.
//
// L1  two-class segmentation loss of libs/loss.py:110-137 on the selected rows of a logit tensor: class-frequency weighted cross
//     entropy (:90-108), Lovasz-Softmax (libs/lovasz_softmax.py:56-94) and the IoU counters of compute_iou (:17-50), in ~10
//     launches instead of ~150 element-wise ones.  The Lovasz term needs the errors |fg - p_c| of each class in descending
//     order: ONE radix sort of 2n (key, payload) pairs -- key = the error's float bits (errors lie in [0,1], so the bit
//     pattern is monotone and bit 31 is free for the class), payload = row << 1 | fg -- then a tiled scan of the fg bits
//     gives the Jaccard gradient of every rank, which is written back to the row it belongs to (the backward pass needs it
//     per row) and dotted with the errors.  All arithmetic on the ranks is float32 like the reference's cumsums (exact
//     integers below 2^24).  HBM-bound: ~40 B per row and class through the sort, 24 B per row elsewhere.
// L2  offset loss of libs/loss.py:194-250: ground-truth reconstruction of every point (ego pose, then instance motion),
//     instance centres (LDS-privatised sums), offsets of the selected rows to their centre, the three reductions.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

#define SL_ITEMS 8
#define SL_TILE (256 * SL_ITEMS)

// ---- block reductions (256 threads) ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

// sums NV doubles per thread over the block; the totals land in out[0..NV) (global memory) written by thread 0
template <int NV>
__device__ __forceinline__ void block_sum_to(double (&v)[NV], double *lds, double *out)
{
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum(v[k]);
        if (lane_id() == 0) lds[(threadIdx.x >> 6) * NV + k] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) out[k] = lds[k] + lds[NV + k] + lds[2 * NV + k] + lds[3 * NV + k];
    }
}

// ---- L1 -------------------------------------------------------------------------------------------------------------------
// logits of row i: [n,2] rows (plane == 0) or planes [n/plane][2][plane] (an NCHW head output read in place)
__device__ __forceinline__ int64_t seg_addr(int64_t i, int64_t plane)
{
    return plane > 0 ? (i / plane) * 2 * plane + (i % plane) : i * 2;
}
__device__ __forceinline__ int64_t seg_step(int64_t plane) { return plane > 0 ? plane : 1; }
__device__ __forceinline__ float seg_ld(const void *p, bool bf, int64_t a)
{
    return bf ? bf16_to_f32(reinterpret_cast<const uint16_t *>(p)[a]) : reinterpret_cast<const float *>(p)[a];
}
__device__ __forceinline__ void seg_st(void *p, bool bf, int64_t a, float v)
{
    if (bf) reinterpret_cast<uint16_t *>(p)[a] = f32_to_bf16(v);
    else reinterpret_cast<float *>(p)[a] = v;
}

struct SegProbs {
    float p0, p1, lp0, lp1;
};
__device__ __forceinline__ SegProbs seg_probs(float z0, float z1)
{
    const float m = fmaxf(z0, z1);
    const float e0 = expf(z0 - m), e1 = expf(z1 - m), s = e0 + e1, ls = logf(s);
    return SegProbs{e0 / s, e1 / s, (z0 - m) - ls, (z1 - m) - ls};
}

// per row: the two sort records and the counters.  part[block][7] = n(y=0), n(y=1), sum log p_y over y=0 / y=1, n(pred=1),
// n(pred=0 & y=0), n(pred=1 & y=1)
__global__ __launch_bounds__(256) void seg_rows_kernel(const void *__restrict__ logits, bool bf, int64_t plane, const int64_t *__restrict__ labels,
                                                       const int64_t *__restrict__ rows, int64_t n, uint32_t *__restrict__ key,
                                                       uint32_t *__restrict__ val, double *__restrict__ part)
{
    __shared__ double lds[4 * 7];
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const int64_t step = seg_step(plane);
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const int64_t i = rows ? rows[j] : j;
        const int64_t y = labels[i];
        const int64_t a = seg_addr(i, plane);
        const float z0 = seg_ld(logits, bf, a), z1 = seg_ld(logits, bf, a + step);
        const SegProbs q = seg_probs(z0, z1);
        const uint32_t fg0 = y == 0, fg1 = y == 1, pred = z1 > z0;             // argmax, ties -> class 0
        key[j] = __float_as_uint(fabsf((float)fg0 - q.p0));
        val[j] = ((uint32_t)j << 1) | fg0;
        key[n + j] = __float_as_uint(fabsf((float)fg1 - q.p1)) | 0x80000000u;
        val[n + j] = ((uint32_t)j << 1) | fg1;
        acc[0] += fg0;
        acc[1] += fg1;
        acc[2] += fg0 ? (double)q.lp0 : 0.0;
        acc[3] += fg1 ? (double)q.lp1 : 0.0;
        acc[4] += pred;
        acc[5] += (fg0 && !pred);
        acc[6] += (fg1 && pred);
    }
    block_sum_to<7>(acc, lds, part + (int64_t)blockIdx.x * 7);
}

// sorted order: class 1 occupies [0,n), class 0 [n,2n) (descending keys, class bit on top)
__device__ __forceinline__ int64_t seg_sorted_base(int c, int64_t n) { return c ? 0 : n; }

__global__ __launch_bounds__(256) void seg_tile_fg_kernel(const uint32_t *__restrict__ val, int64_t n, int nt, int *__restrict__ tile_fg)
{
    __shared__ int lds[4];
    const int c = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * SL_TILE;
    const uint32_t *v = val + seg_sorted_base(c, n);
    int s = 0;
    for (int k = threadIdx.x; k < SL_TILE; k += 256)
        if (r0 + k < n) s += v[r0 + k] & 1u;
    int total;
    block256_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) tile_fg[c * nt + blockIdx.x] = total;
}

// Jaccard gradient of every rank of one tile (libs/lovasz_softmax.py:56-68), scattered to the rows, and its dot product with
// the sorted errors
__global__ __launch_bounds__(256) void seg_lovasz_kernel(const uint32_t *__restrict__ key, const uint32_t *__restrict__ val, int64_t n, int nt,
                                                         const int *__restrict__ tile_fg, float *__restrict__ lov_grad, double *__restrict__ dot_part)
{
#pragma clang fp contract(off)
    __shared__ int lds_i[4];
    __shared__ double lds_d[4 * 2];
    const int c = blockIdx.y, tile = blockIdx.x;
    const int64_t base = seg_sorted_base(c, n);
    double pt[2] = {0, 0};                                                    // fg before this tile, fg in total
    for (int t = threadIdx.x; t < nt; t += 256) {
        const int f = tile_fg[c * nt + t];
        pt[1] += f;
        if (t < tile) pt[0] += f;
    }
    __shared__ double tot[2];
    block_sum_to<2>(pt, lds_d, tot);
    __syncthreads();
    const int before = (int)tot[0];
    const float gts = (float)tot[1];
    const int64_t r0 = (int64_t)tile * SL_TILE + (int64_t)threadIdx.x * SL_ITEMS;
    uint32_t k[SL_ITEMS], v[SL_ITEMS];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < SL_ITEMS; ++q) {
        const bool in = r0 + q < n;
        k[q] = in ? key[base + r0 + q] : 0u;
        v[q] = in ? val[base + r0 + q] : 0u;
        mine += v[q] & 1u;
    }
    int total;
    int cum = before + block256_exclusive_scan(mine, lds_i, &total);
    double acc[1] = {0};
#pragma unroll
    for (int q = 0; q < SL_ITEMS; ++q) {
        const int64_t r = r0 + q;
        if (r < n) {
            const int fg = v[q] & 1u;
            const int prev = cum;
            cum += fg;
            const float jac = 1.f - (gts - (float)cum) / (gts + (float)(r + 1 - cum));
            float g = jac;
            if (r > 0) g = jac - (1.f - (gts - (float)prev) / (gts + (float)(r - prev)));
            lov_grad[(int64_t)c * n + (v[q] >> 1)] = g;
            acc[0] += (double)__uint_as_float(k[q] & 0x7fffffffu) * (double)g;
        }
    }
    block_sum_to<1>(acc, lds_d, dot_part + (int64_t)c * nt + tile);
}

// out_loss = (cross entropy, Lovasz); out_metric[4][2] = intersection, union, predicted, labelled per class, / 1e3;
// saved = w0, w1, sum of the row weights, present_0 / n_present, present_1 / n_present
__global__ __launch_bounds__(256) void seg_final_kernel(const double *__restrict__ part, int nb, const double *__restrict__ dot_part, int nt, int64_t n,
                                                        float *__restrict__ out_loss, double *__restrict__ out_metric, float *__restrict__ saved)
{
#pragma clang fp contract(off)
    __shared__ double lds[4 * 9];
    __shared__ double tot[9];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nb; b += 256)
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[k] += part[(int64_t)b * 7 + k];
    for (int t = threadIdx.x; t < nt; t += 256) {
        acc[7] += dot_part[t];
        acc[8] += dot_part[nt + t];
    }
    block_sum_to<9>(acc, lds, tot);
    if (threadIdx.x != 0) return;
    const double n0 = tot[0], n1 = tot[1];
    // libs/loss.py:90-108: float32 counts (+1e-20), sqrt(total / count) clamped to [0, 50]
    const float c0 = n0 > 0 ? (float)n0 : 1e-20f, c1 = n1 > 0 ? (float)n1 : 1e-20f, sum = c0 + c1;
    const float w0 = fminf(fmaxf(sqrtf(sum / c0), 0.f), 50.f), w1 = fminf(fmaxf(sqrtf(sum / c1), 0.f), 50.f);
    const double wsum = (double)w0 * n0 + (double)w1 * n1;
    out_loss[0] = (float)(-((double)w0 * tot[2] + (double)w1 * tot[3]) / wsum);
    const double present = (n0 > 0) + (n1 > 0), div = present > 0 ? present : 1.0;
    out_loss[1] = (float)(((n0 > 0 ? tot[7] : 0.0) + (n1 > 0 ? tot[8] : 0.0)) / div);
    const double pred1 = tot[4], pred0 = (double)n - pred1;
    out_metric[0] = tot[5] / 1e3;
    out_metric[1] = tot[6] / 1e3;
    out_metric[2] = pred0 / 1e3 + n0 / 1e3 - tot[5] / 1e3;
    out_metric[3] = pred1 / 1e3 + n1 / 1e3 - tot[6] / 1e3;
    out_metric[4] = pred0 / 1e3;
    out_metric[5] = pred1 / 1e3;
    out_metric[6] = n0 / 1e3;
    out_metric[7] = n1 / 1e3;
    saved[0] = w0;
    saved[1] = w1;
    saved[2] = (float)wsum;
    saved[3] = (float)((n0 > 0) / div);
    saved[4] = (float)((n1 > 0) / div);
}

// d(g_bce * cross entropy + g_lov * Lovasz) / d logits of the selected rows (the rest of grad is zeroed by the caller)
__global__ __launch_bounds__(256) void seg_backward_kernel(const void *__restrict__ logits, bool bf, int64_t plane, const int64_t *__restrict__ labels,
                                                           const int64_t *__restrict__ rows, int64_t n, const float *__restrict__ lov_grad,
                                                           const float *__restrict__ saved, const float *__restrict__ g_bce,
                                                           const float *__restrict__ g_lov, void *__restrict__ grad)
{
    const float gb = g_bce ? *g_bce : 0.f, gl = g_lov ? *g_lov : 0.f;
    const float w0 = saved[0], w1 = saved[1], wsum = saved[2], h0 = saved[3] * gl, h1 = saved[4] * gl;
    const int64_t step = seg_step(plane);
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const int64_t i = rows ? rows[j] : j;
        const int64_t y = labels[i];
        const int64_t a = seg_addr(i, plane);
        const SegProbs q = seg_probs(seg_ld(logits, bf, a), seg_ld(logits, bf, a + step));
        const float fg0 = y == 0, fg1 = y == 1;
        float d0 = 0.f, d1 = 0.f;
        if (y == 0 || y == 1) {
            const float s = gb * (y == 0 ? w0 : w1) / wsum;
            d0 = s * (q.p0 - fg0);
            d1 = s * (q.p1 - fg1);
        }
        const float x0 = fg0 - q.p0, x1 = fg1 - q.p1;                        // d|x|/dp = -sign(x)
        const float dp0 = -h0 * lov_grad[j] * (float)((x0 > 0.f) - (x0 < 0.f));
        const float dp1 = -h1 * lov_grad[n + j] * (float)((x1 > 0.f) - (x1 < 0.f));
        const float dot = dp0 * q.p0 + dp1 * q.p1;
        d0 += q.p0 * (dp0 - dot);
        d1 += q.p1 * (dp1 - dot);
        seg_st(grad, bf, a, d0);
        seg_st(grad, bf, a + step, d1);
    }
}

struct SegWs {
    uint32_t *key_a, *key_b, *val_a, *val_b;
    double *part, *dot_part;
    int *tile_fg;
    void *sort_tmp;
    size_t sort_tmp_bytes, total;
    int nb, nt;
};

static int seg_ws_layout(int64_t n, char *base, SegWs *w)
{
    size_t sort_bytes = 0;
    if (rocprim::radix_sort_pairs_desc(nullptr, sort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                       (size_t)(2 * n), 0, 32, (hipStream_t)0) != hipSuccess)
        return PCACC_E_LAUNCH;
    w->nb = pcacc_grid(n, 256, PCACC_CUS * 4);
    w->nt = (int)((n + SL_TILE - 1) / SL_TILE);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += pcacc_align(bytes); return p; };
    w->key_a = (uint32_t *)take((size_t)2 * n * 4);
    w->key_b = (uint32_t *)take((size_t)2 * n * 4);
    w->val_a = (uint32_t *)take((size_t)2 * n * 4);
    w->val_b = (uint32_t *)take((size_t)2 * n * 4);
    w->part = (double *)take((size_t)w->nb * 7 * 8);
    w->dot_part = (double *)take((size_t)w->nt * 2 * 8);
    w->tile_fg = (int *)take((size_t)w->nt * 2 * 4);
    w->sort_tmp = take(sort_bytes);
    w->sort_tmp_bytes = sort_bytes;
    w->total = off;
    return PCACC_OK;
}

extern "C" int pcacc_seg_loss_workspace_bytes(int64_t n, size_t *bytes)
{
    SegWs w;
    if (n <= 0 || n >= (1ll << 30) || !bytes) return PCACC_E_ARG;
    if (seg_ws_layout(n, nullptr, &w) != PCACC_OK) return PCACC_E_LAUNCH;
    *bytes = w.total;
    return PCACC_OK;
}

extern "C" int pcacc_seg_loss_forward(const void *logits, int logits_dtype, int64_t plane, const int64_t *labels, const int64_t *rows, int64_t n,
                                      float *out_loss, double *out_metric, float *lovasz_grad, float *saved, void *ws, size_t ws_bytes, void *stream)
{
    if (n <= 0 || n >= (1ll << 30) || plane < 0 || !logits || !labels || !out_loss || !out_metric || !lovasz_grad || !saved || !ws) return PCACC_E_ARG;
    if (logits_dtype != PCACC_F32 && logits_dtype != PCACC_BF16) return PCACC_E_ARG;
    SegWs w;
    if (seg_ws_layout(n, (char *)ws, &w) != PCACC_OK) return PCACC_E_LAUNCH;
    if (ws_bytes < w.total) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    seg_rows_kernel<<<w.nb, 256, 0, s>>>(logits, logits_dtype == PCACC_BF16, plane, labels, rows, n, w.key_a, w.val_a, w.part);
    PCACC_CHECK_LAUNCH();
    if (rocprim::radix_sort_pairs_desc(w.sort_tmp, w.sort_tmp_bytes, w.key_a, w.key_b, w.val_a, w.val_b, (size_t)(2 * n), 0, 32, s) != hipSuccess)
        return PCACC_E_LAUNCH;
    seg_tile_fg_kernel<<<dim3(w.nt, 2), 256, 0, s>>>(w.val_b, n, w.nt, w.tile_fg);
    PCACC_CHECK_LAUNCH();
    seg_lovasz_kernel<<<dim3(w.nt, 2), 256, 0, s>>>(w.key_b, w.val_b, n, w.nt, w.tile_fg, lovasz_grad, w.dot_part);
    PCACC_CHECK_LAUNCH();
    seg_final_kernel<<<1, 256, 0, s>>>(w.part, w.nb, w.dot_part, w.nt, n, out_loss, out_metric, saved);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_seg_loss_backward(const void *logits, int logits_dtype, int64_t plane, const int64_t *labels, const int64_t *rows, int64_t n,
                                       int64_t n_total, const float *lovasz_grad, const float *saved, const float *grad_bce,
                                       const float *grad_lovasz, void *grad_logits, void *stream)
{
    if (n < 0 || n_total < n || plane < 0 || !grad_logits) return PCACC_E_ARG;
    if (logits_dtype != PCACC_F32 && logits_dtype != PCACC_BF16) return PCACC_E_ARG;
    hipStream_t s = pcacc_stream(stream);
    if (rows || n == 0) {
        const size_t bytes = (size_t)n_total * 2 * (logits_dtype == PCACC_BF16 ? 2 : 4);
        if (bytes && hipMemsetAsync(grad_logits, 0, bytes, s) != hipSuccess) return PCACC_E_LAUNCH;
    }
    if (n == 0) return PCACC_OK;
    if (!logits || !labels || !lovasz_grad || !saved) return PCACC_E_ARG;
    seg_backward_kernel<<<pcacc_grid(n, 256), 256, 0, s>>>(logits, logits_dtype == PCACC_BF16, plane, labels, rows, n, lovasz_grad, saved, grad_bce,
                                                         grad_lovasz, grad_logits);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- L2 -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void apply_pose(const float *__restrict__ m, float &x, float &y, float &z)
{
#pragma clang fp contract(off)
    const float a = m[0] * x + m[1] * y + m[2] * z + m[3];
    const float b = m[4] * x + m[5] * y + m[6] * z + m[7];
    const float c = m[8] * x + m[9] * y + m[10] * z + m[11];
    x = a, y = b, z = c;
}

// sums[k] = (sum x, sum y, count) of the reconstructed points of instance k (libs/loss.py:213-216).  LDS copy of the whole table
// per workgroup when it fits (few instances, many points each), global atomics otherwise.
// [r6] The sums are kept in 64-bit FIXED POINT (2^-24 m: finer than an fp32 ulp anywhere beyond 0.5 m; |x| < 2^15 m x 3.2 M points stays far inside 63 bits):
// integer additions commute, so the centres -- and with them the offset loss and its gradient -- are the same bits every run.  Rounds 1-5 added fp32 values
// with atomics in arrival order (as torch_scatter does); the fixed-point sum is exact, i.e. what those sums approximated.
#define OFF_FIX 16777216.0f
__global__ __launch_bounds__(256) void offset_centres_kernel(const float *__restrict__ points, const int64_t *__restrict__ time_indice,
                                                             const int64_t *__restrict__ inst, const int64_t *__restrict__ label_base,
                                                             const float *__restrict__ ego, const float *__restrict__ inst_tsfm, int n_frames,
                                                             int64_t n, int k3, bool use_lds, unsigned long long *__restrict__ sums)
{
    extern __shared__ unsigned long long tab[];
    if (use_lds) {
        for (int j = threadIdx.x; j < k3; j += 256) tab[j] = 0ull;
        __syncthreads();
    }
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per_block, hi = min(n, lo + per_block);
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int64_t b = time_indice[2 * i], t = time_indice[2 * i + 1];
        const int64_t lab = inst[i] + label_base[b];
        float x = points[3 * i], y = points[3 * i + 1], z = points[3 * i + 2];
        apply_pose(ego + (b * n_frames + t) * 16, x, y, z);
        apply_pose(inst_tsfm + (lab * n_frames + t) * 16, x, y, z);
        unsigned long long *dst = (use_lds ? tab : sums) + lab * 3;
        // non-finite coordinates (never produced by finite poses) would poison a float sum; here they are clamped into the representable range
        const float cx = fminf(fmaxf(x, -1.0e9f), 1.0e9f), cy = fminf(fmaxf(y, -1.0e9f), 1.0e9f);
        atomicAdd(dst, (unsigned long long)(long long)llrintf(cx * OFF_FIX));        // two's complement: signed sums through the unsigned add
        atomicAdd(dst + 1, (unsigned long long)(long long)llrintf(cy * OFF_FIX));
        atomicAdd(dst + 2, 1ull);
    }
    if (use_lds) {
        __syncthreads();
        for (int j = threadIdx.x; j < k3; j += 256) {
            const unsigned long long v = tab[j];
            if (v != 0ull) atomicAdd(&sums[j], v);
        }
    }
}

// per selected row: offset to the instance centre, and the partial sums |dx|, |dy|, ||d||, 1 - cos
__global__ __launch_bounds__(256) void offset_terms_kernel(const int64_t *__restrict__ time_indice, const int64_t *__restrict__ inst,
                                                           const int64_t *__restrict__ label_base, const unsigned long long *__restrict__ sums,
                                                           const float *__restrict__ tp, const float *__restrict__ est, const int64_t *__restrict__ rows,
                                                           int64_t m, float *__restrict__ offset_gt, double *__restrict__ part)
{
    __shared__ double lds[4 * 4];
    double acc[4] = {0, 0, 0, 0};
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (int64_t)gridDim.x * 256) {
        const int64_t i = rows ? rows[j] : j;
        const int64_t lab = inst[i] + label_base[time_indice[2 * i]];
        const float cnt = fmaxf((float)sums[lab * 3 + 2], 1.f);
        // the fixed-point sums back as fp32 (the correctly rounded sum), then the reference's fp32 division by the count (torch_scatter 'mean')
        const float sx = (float)((double)(long long)sums[lab * 3] * (1.0 / (double)OFF_FIX)), sy = (float)((double)(long long)sums[lab * 3 + 1] * (1.0 / (double)OFF_FIX));
        const float gx = sx / cnt - tp[3 * i], gy = sy / cnt - tp[3 * i + 1];
        const float ex = est[2 * i], ey = est[2 * i + 1];
        offset_gt[2 * j] = gx;
        offset_gt[2 * j + 1] = gy;
        const float dx = gx - ex, dy = gy - ey;
        const float gn = sqrtf(gx * gx + gy * gy) + 1e-20f, en = sqrtf(ex * ex + ey * ey) + 1e-20f;
        acc[0] += fabsf(dx);
        acc[1] += fabsf(dy);
        acc[2] += sqrtf(dx * dx + dy * dy);
        acc[3] += 1.f - ((gx / gn) * (ex / en) + (gy / gn) * (ey / en));
    }
    block_sum_to<4>(acc, lds, part + (int64_t)blockIdx.x * 4);
}

// out = (L1 term, direction term, mean L2 error)
__global__ __launch_bounds__(256) void offset_final_kernel(const double *__restrict__ part, int nb, int64_t m, float *__restrict__ out)
{
    __shared__ double lds[4 * 4];
    __shared__ double tot[4];
    double acc[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nb; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += part[(int64_t)b * 4 + k];
    block_sum_to<4>(acc, lds, tot);
    if (threadIdx.x != 0) return;
    out[0] = (float)(tot[0] / (double)m + tot[1] / (double)m);
    out[1] = (float)(tot[3] / (double)m);
    out[2] = (float)(tot[2] / (double)m);
}

__global__ __launch_bounds__(256) void offset_backward_kernel(const float *__restrict__ offset_gt, const float *__restrict__ est,
                                                              const int64_t *__restrict__ rows, int64_t m, const float *__restrict__ g_norm,
                                                              const float *__restrict__ g_dir, float *__restrict__ grad)
{
    const float gn_ = (g_norm ? *g_norm : 0.f) / (float)m, gd_ = (g_dir ? *g_dir : 0.f) / (float)m;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (int64_t)gridDim.x * 256) {
        const int64_t i = rows ? rows[j] : j;
        const float gx = offset_gt[2 * j], gy = offset_gt[2 * j + 1], ex = est[2 * i], ey = est[2 * i + 1];
        const float dx = gx - ex, dy = gy - ey;
        float rx = -gn_ * (float)((dx > 0.f) - (dx < 0.f)), ry = -gn_ * (float)((dy > 0.f) - (dy < 0.f));
        const float gnorm = sqrtf(gx * gx + gy * gy) + 1e-20f, e = sqrtf(ex * ex + ey * ey), en = e + 1e-20f;
        const float ux = gx / gnorm, uy = gy / gnorm;                         // d(-u . est/(|est|+eps)) / d est
        const float proj = (ux * ex + uy * ey) / (en * en);
        const float nx = e > 0.f ? ex / e : 0.f, ny = e > 0.f ? ey / e : 0.f;   // torch.norm's backward at 0 is 0
        rx -= gd_ * (ux / en - proj * nx);
        ry -= gd_ * (uy / en - proj * ny);
        grad[2 * i] = rx;
        grad[2 * i + 1] = ry;
    }
}

#define OFF_LDS_FLOATS 6144      // table entries (8 bytes each since [r6]: 48 KB of LDS)

static size_t offset_ws_bytes(int64_t m, int64_t k)
{
    return pcacc_align((size_t)k * 3 * 8) + pcacc_align((size_t)pcacc_grid(m, 256, PCACC_CUS * 4) * 4 * 8);
}

extern "C" int pcacc_offset_loss_workspace_bytes(int64_t m, int64_t k, size_t *bytes)
{
    if (m < 0 || k <= 0 || !bytes) return PCACC_E_ARG;
    *bytes = offset_ws_bytes(m, k);
    return PCACC_OK;
}

extern "C" int pcacc_offset_loss_forward(const float *points, const int64_t *time_indice, const int64_t *inst_labels, const int64_t *label_base,
                                         const float *ego_motion, const float *inst_motion, int32_t n_frames, int64_t n, int64_t k,
                                         const float *transformed_points, const float *offset_est, const int64_t *rows, int64_t m,
                                         float *out, float *offset_gt, void *ws, size_t ws_bytes, void *stream)
{
    if (n <= 0 || m <= 0 || k <= 0 || n_frames <= 0 || !points || !time_indice || !inst_labels || !label_base || !ego_motion || !inst_motion ||
        !transformed_points || !offset_est || !out || !offset_gt || !ws)
        return PCACC_E_ARG;
    if (ws_bytes < offset_ws_bytes(m, k)) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    unsigned long long *sums = (unsigned long long *)ws;
    double *part = (double *)((char *)ws + pcacc_align((size_t)k * 3 * 8));
    if (hipMemsetAsync(sums, 0, (size_t)k * 3 * 8, s) != hipSuccess) return PCACC_E_LAUNCH;
    const bool use_lds = k * 3 <= OFF_LDS_FLOATS;
    int grid = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (grid > PCACC_CUS * 4) grid = PCACC_CUS * 4;
    offset_centres_kernel<<<grid, 256, use_lds ? (size_t)k * 3 * 8 : 0, s>>>(points, time_indice, inst_labels, label_base, ego_motion, inst_motion, n_frames, n,
                                                                             (int)(k * 3), use_lds, sums);
    PCACC_CHECK_LAUNCH();
    const int nb = pcacc_grid(m, 256, PCACC_CUS * 4);
    offset_terms_kernel<<<nb, 256, 0, s>>>(time_indice, inst_labels, label_base, sums, transformed_points, offset_est, rows, m, offset_gt, part);
    PCACC_CHECK_LAUNCH();
    offset_final_kernel<<<1, 256, 0, s>>>(part, nb, m, out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_offset_loss_backward(const float *offset_gt, const float *offset_est, const int64_t *rows, int64_t m, int64_t n,
                                          const float *grad_norm, const float *grad_dir, float *grad_est, void *stream)
{
    if (m < 0 || n < m || !grad_est) return PCACC_E_ARG;
    hipStream_t s = pcacc_stream(stream);
    if (n && hipMemsetAsync(grad_est, 0, (size_t)n * 2 * 4, s) != hipSuccess) return PCACC_E_LAUNCH;
    if (m == 0) return PCACC_OK;
    if (!offset_gt || !offset_est) return PCACC_E_ARG;
    offset_backward_kernel<<<pcacc_grid(m, 256), 256, 0, s>>>(offset_gt, offset_est, rows, m, grad_norm, grad_dir, grad_est);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
