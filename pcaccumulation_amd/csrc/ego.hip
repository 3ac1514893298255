// A8: key-point matching for the ego-motion head, forward pass, fp32 -- models/egomotion.py:169-192 (pairwise_ego_motion_
// estimation after key-point choice), :100-137 (sinkhorn), toolbox/utils.py:125-144 (square_distance) and
// toolbox/register_utils.py:247-317 (kabsch_transformation_estimation), for P pairs at once.
//
// The reference issues ~60 small launches per pair (matmul, pad, 6 x (logsumexp, sub, cat), exp, mul, sums, diag_embed of a
// 1024x1024 weight matrix, svd, det ...).  Here: 1 affinity kernel (LDS-tiled 64x64 dot products), 2 kernels per Sinkhorn
// iteration working in place on the padded (k+1)x(k+1) log-matrix (4 MB per pair: L2 / Infinity-Cache resident), 1 row
// kernel (exp * support, row sums, soft targets) and 1 Kabsch kernel per pair (weighted means, 3x3 covariance by block
// reduction, 3x3 Jacobi SVD, reflection fix).  Used when no gradient is required (eval / val / test); training keeps the
// batched torch formulation for autograd.
#include "common.h"

#define EGO_TILE 64

// ---- 1. affinity = -(max(2 - 2 <fs_i, ft_j>, 1e-12) - softplus(alpha)) * inv_temp, written into the padded matrix ----------
__global__ __launch_bounds__(256) void ego_affinity_kernel(const float *__restrict__ fs, const float *__restrict__ ft, int k, int c,
                                                           const float *__restrict__ params, float *__restrict__ la, int kp)
{
    const float softplus_alpha = params[0], denom = params[1];      // device scalars: no host sync to read alpha / beta
    __shared__ float As[EGO_TILE][17], Bs[EGO_TILE][17];
    const int p = blockIdx.z;
    const int i0 = blockIdx.y * EGO_TILE, j0 = blockIdx.x * EGO_TILE;
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    fs += (int64_t)p * k * c;
    ft += (int64_t)p * k * c;
    la += (int64_t)p * kp * kp;                                 // kp = k + 1: the padded matrix of the eval pipeline; k: a dense [k, k] result
    float acc[4][4] = {};
    for (int c0 = 0; c0 < c; c0 += 16) {
        for (int e = threadIdx.x; e < EGO_TILE * 16; e += 256) {
            const int r = e / 16, cc = e % 16;
            As[r][cc] = (i0 + r < k && c0 + cc < c) ? fs[(int64_t)(i0 + r) * c + c0 + cc] : 0.f;
            Bs[r][cc] = (j0 + r < k && c0 + cc < c) ? ft[(int64_t)(j0 + r) * c + c0 + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = As[ty * 4 + u][cc]; b[u] = Bs[tx * 4 + u][cc]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(a[u], b[v], acc[u][v]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = i0 + ty * 4 + u, j = j0 + tx * 4 + v;
            if (i < k && j < k) {
                const float dist = fmaxf(-2.0f * acc[u][v] + 2.0f, 1e-12f);          // toolbox/utils.py:137-143
                la[(int64_t)i * kp + j] = -(dist - softplus_alpha) / denom;          // models/egomotion.py:180
            }
        }
}

// slack row / column of the padded matrix = 0 (nn.ZeroPad2d, models/egomotion.py:116-117)
__global__ __launch_bounds__(256) void ego_pad_kernel(int k, int n_pairs, float *la)
{
    const int kp = k + 1;
    const int64_t total = (int64_t)n_pairs * (2 * kp - 1);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int p = (int)(e / (2 * kp - 1));
        const int r = (int)(e % (2 * kp - 1));
        float *m = la + (int64_t)p * kp * kp;
        if (r < kp) m[(int64_t)k * kp + r] = 0.f;                // last row
        else m[(int64_t)(r - kp) * kp + k] = 0.f;               // last column (rows 0..k-1)
    }
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---- 2a. row normalisation: rows 0..k-1, logsumexp over ALL k+1 columns (egomotion.py:122-126) -----------------------------
__global__ __launch_bounds__(256) void ego_sinkhorn_rows_kernel(int k, int n_pairs, float *la, float *lse_out = nullptr)
{
    const int kp = k + 1;
    const int lane = threadIdx.x & 63;
    const int64_t n_rows = (int64_t)n_pairs * k;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const int p = (int)(row / k), i = (int)(row % k);
        float *r = la + (int64_t)p * kp * kp + (int64_t)i * kp;
        float mx = -__builtin_inff();
        for (int j = lane; j < kp; j += 64) mx = fmaxf(mx, r[j]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int j = lane; j < kp; j += 64) sm += expf(r[j] - mx);
        const float lse = mx + logf(wave_sum(sm));
        for (int j = lane; j < kp; j += 64) r[j] -= lse;
        if (lse_out && lane == 0) lse_out[row] = lse;            // [n_pairs][k]: what the training backward replays
    }
}

// ---- 2b. column normalisation: columns 0..k-1, logsumexp over ALL k+1 rows (egomotion.py:128-132) ---------------------------
// workgroup = 64 columns x EGO_CG row groups (16 waves: the walk down a column is latency-bound, 4 waves per workgroup left
// the 16 x P workgroups of this launch 5x slower than the row kernel); partial (max, sum) combined through LDS
#define EGO_CG 16
__global__ __launch_bounds__(64 * EGO_CG) void ego_sinkhorn_cols_kernel(int k, float *la, float *lse_out = nullptr)
{
    __shared__ float smax[EGO_CG][64], ssum[EGO_CG][64];
    const int kp = k + 1;
    const int p = blockIdx.y;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float *m = la + (int64_t)p * kp * kp;
    const int rows_per = (kp + EGO_CG - 1) / EGO_CG;
    const int r_lo = q * rows_per, r_hi = min(kp, r_lo + rows_per);
    float mx = -__builtin_inff(), sm = 0.f;
    if (col < k) {
        for (int i = r_lo; i < r_hi; ++i) mx = fmaxf(mx, m[(int64_t)i * kp + col]);
        for (int i = r_lo; i < r_hi; ++i) sm += expf(m[(int64_t)i * kp + col] - mx);
    }
    smax[q][threadIdx.x & 63] = mx;
    ssum[q][threadIdx.x & 63] = sm;
    __syncthreads();
    float gm = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < EGO_CG; ++u) gm = fmaxf(gm, smax[u][threadIdx.x & 63]);
    float gs = 0.f;
#pragma unroll
    for (int u = 0; u < EGO_CG; ++u) {
        const float pm = smax[u][threadIdx.x & 63];
        gs += (pm == -__builtin_inff()) ? 0.f : ssum[u][threadIdx.x & 63] * expf(pm - gm);
    }
    const float lse = gm + logf(gs);
    if (col < k) {
        for (int i = r_lo; i < r_hi; ++i) m[(int64_t)i * kp + col] -= lse;
        if (lse_out && q == 0) lse_out[(int64_t)p * k + col] = lse;
    }
}

// ---- 3. perm = exp(log_perm) * support; row sums; soft targets (egomotion.py:183-184) -----------------------------------------
__global__ __launch_bounds__(256) void ego_rows_finish_kernel(const float *__restrict__ la, const float *__restrict__ cs,
                                                              const float *__restrict__ ct, const float *__restrict__ thr2, int k,
                                                              int n_pairs, float *__restrict__ perm, float *__restrict__ rowsum,
                                                              float *__restrict__ wt, int kp)
{
    const int lane = threadIdx.x & 63;
    const int64_t n_rows = (int64_t)n_pairs * k;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const int p = (int)(row / k), i = (int)(row % k);
        const float *r = la + (int64_t)p * kp * kp + (int64_t)i * kp;
        const float *t = ct + (int64_t)p * k * 3;
        const float sx = cs[row * 3 + 0], sy = cs[row * 3 + 1], sz = cs[row * 3 + 2];
        const float s2 = sx * sx + sy * sy + sz * sz;
        const float th = thr2[p];
        float rs = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
        for (int j = lane; j < k; j += 64) {
            const float tx = t[j * 3 + 0], ty = t[j * 3 + 1], tz = t[j * 3 + 2];
            // square_distance(): -2 <s,t> + |s|^2 + |t|^2, clamped at 1e-12 (toolbox/utils.py:137-143)
            float d = -2.0f * (sx * tx + sy * ty + sz * tz);
            d += s2;
            d += tx * tx + ty * ty + tz * tz;
            d = fmaxf(d, 1e-12f);
            const float v = (d < th) ? expf(r[j]) : 0.f;
            perm[row * k + j] = v;
            rs += v; ax += v * tx; ay += v * ty; az += v * tz;
        }
        rs = wave_sum(rs); ax = wave_sum(ax); ay = wave_sum(ay); az = wave_sum(az);
        if (lane == 0) {
            rowsum[row] = rs;
            const float den = rs + 1e-20f;
            wt[row * 3 + 0] = ax / den; wt[row * 3 + 1] = ay / den; wt[row * 3 + 2] = az / den;
        }
    }
}

// ---- 4. weighted Kabsch (register_utils.py:268-313), one workgroup per pair -----------------------------------------------------
__device__ void jacobi_svd3(const double a[3][3], double u[3][3], double s[3], double v[3][3])
{
    // one-sided Jacobi on the columns of A: A V = U diag(s)
    double b[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { b[i][j] = a[i][j]; v[i][j] = (i == j); }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) { alpha += b[i][p] * b[i][p]; beta += b[i][q] * b[i][q]; gamma += b[i][p] * b[i][q]; }
                off += gamma * gamma;
                if (fabs(gamma) < 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double bp = b[i][p], bq = b[i][q];
                    b[i][p] = c * bp - sn * bq; b[i][q] = sn * bp + c * bq;
                    const double vp = v[i][p], vq = v[i][q];
                    v[i][p] = c * vp - sn * vq; v[i][q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-40) break;
    }
    for (int j = 0; j < 3; ++j) {
        double n = 0;
        for (int i = 0; i < 3; ++i) n += b[i][j] * b[i][j];
        s[j] = sqrt(n);
    }
    // order singular values descending (as LAPACK / torch.svd)
    for (int x = 0; x < 2; ++x)
        for (int y = x + 1; y < 3; ++y)
            if (s[y] > s[x]) {
                const double ts = s[x]; s[x] = s[y]; s[y] = ts;
                for (int i = 0; i < 3; ++i) {
                    const double tb = b[i][x]; b[i][x] = b[i][y]; b[i][y] = tb;
                    const double tv = v[i][x]; v[i][x] = v[i][y]; v[i][y] = tv;
                }
            }
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) u[i][j] = s[j] > 1e-300 ? b[i][j] / s[j] : (i == j);
    if (s[2] <= 1e-300 * 1.0 || s[2] < 1e-12 * s[0]) {                 // rank-deficient: complete U with a cross product
        u[0][2] = u[1][0] * u[2][1] - u[2][0] * u[1][1];
        u[1][2] = u[2][0] * u[0][1] - u[0][0] * u[2][1];
        u[2][2] = u[0][0] * u[1][1] - u[1][0] * u[0][1];
    }
}

__device__ __forceinline__ double block_sum256(double v, double *red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ego_kabsch_kernel(const float *__restrict__ cs, const float *__restrict__ wt,
                                                         const float *__restrict__ rowsum, int k, float *__restrict__ pose)
{
    __shared__ double red[4];
    const int p = blockIdx.x;
    const float *x1 = cs + (int64_t)p * k * 3, *x2 = wt + (int64_t)p * k * 3, *w0 = rowsum + (int64_t)p * k;
    const double eps = 1e-7;
    double sw = 0;
    for (int i = threadIdx.x; i < k; i += 256) sw += w0[i];
    sw = block_sum256(sw, red);
    const double norm = sw + eps;                                       // weights / (sum + eps), register_utils.py:269-270
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < k; i += 256) {
        const double w = w0[i] / norm;
        acc[6] += w;
        for (int d = 0; d < 3; ++d) { acc[d] += w * x1[i * 3 + d]; acc[3 + d] += w * x2[i * 3 + d]; }
    }
    double tot[7];
    for (int d = 0; d < 7; ++d) tot[d] = block_sum256(acc[d], red);
    double m1[3], m2[3];
    for (int d = 0; d < 3; ++d) { m1[d] = tot[d] / (tot[6] + eps); m2[d] = tot[3 + d] / (tot[6] + eps); }      // :284-285
    double cv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < k; i += 256) {
        const double w = w0[i] / norm;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) cv[a * 3 + b] += (x1[i * 3 + a] - m1[a]) * w * (x2[i * 3 + b] - m2[b]);     // x1c^T W x2c
    }
    double cov[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) cov[a][b] = block_sum256(cv[a * 3 + b], red);
    if (threadIdx.x == 0) {
        double u[3][3], s[3], v[3][3];
        jacobi_svd3(cov, u, s, v);
        // det(v^T u^T) decides the reflection fix; R = v diag(1,1,det) u^T; t = m2 - R m1   (register_utils.py:306-313)
        double vu[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) { vu[a][b] = 0; for (int c = 0; c < 3; ++c) vu[a][b] += v[c][a] * u[b][c]; }
        const double det = vu[0][0] * (vu[1][1] * vu[2][2] - vu[1][2] * vu[2][1]) - vu[0][1] * (vu[1][0] * vu[2][2] - vu[1][2] * vu[2][0]) +
                           vu[0][2] * (vu[1][0] * vu[2][1] - vu[1][1] * vu[2][0]);
        double R[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) R[a][b] = v[a][0] * u[b][0] + v[a][1] * u[b][1] + det * v[a][2] * u[b][2];
        float *o = pose + (int64_t)p * 16;
        for (int a = 0; a < 3; ++a) {
            double t = m2[a];
            for (int b = 0; b < 3; ++b) { o[a * 4 + b] = (float)R[a][b]; t -= R[a][b] * m1[b]; }
            o[a * 4 + 3] = (float)t;
        }
        o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
    }
}

extern "C" int pcacc_sinkhorn_kabsch_workspace_bytes(int n_pairs, int k, size_t *bytes)
{
    if (!bytes || n_pairs < 0 || k <= 0) return PCACC_E_ARG;
    *bytes = pcacc_align((size_t)n_pairs * (k + 1) * (k + 1) * 4) + pcacc_align((size_t)n_pairs * k * 4) + pcacc_align((size_t)n_pairs * k * 12);
    return PCACC_OK;
}

extern "C" int pcacc_sinkhorn_kabsch(const float *feats_s, const float *feats_t, const float *coor_s, const float *coor_t,
                                     const float *thr2, const float *params, int n_pairs, int k, int c, int n_iters,
                                     float *perm, float *pose, void *workspace, size_t workspace_bytes, void *stream)
{
    size_t need;
    if (pcacc_sinkhorn_kabsch_workspace_bytes(n_pairs, k, &need) != PCACC_OK || c <= 0 || n_iters < 0 || n_pairs > 65535) return PCACC_E_ARG;
    if (n_pairs == 0) return PCACC_OK;
    if (!feats_s || !feats_t || !coor_s || !coor_t || !thr2 || !params || !perm || !pose) return PCACC_E_ARG;
    if (!workspace || workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    char *ws = static_cast<char *>(workspace);
    float *la = reinterpret_cast<float *>(ws); ws += pcacc_align((size_t)n_pairs * (k + 1) * (k + 1) * 4);
    float *rowsum = reinterpret_cast<float *>(ws); ws += pcacc_align((size_t)n_pairs * k * 4);
    float *wt = reinterpret_cast<float *>(ws);
    const int tiles = (k + EGO_TILE - 1) / EGO_TILE;
    ego_affinity_kernel<<<dim3(tiles, tiles, n_pairs), 256, 0, s>>>(feats_s, feats_t, k, c, params, la, k + 1);
    ego_pad_kernel<<<pcacc_grid((int64_t)n_pairs * (2 * k + 1), 256), 256, 0, s>>>(k, n_pairs, la);
    const int row_grid = pcacc_grid((int64_t)n_pairs * k * 64, 256);
    for (int it = 0; it < n_iters; ++it) {
        ego_sinkhorn_rows_kernel<<<row_grid, 256, 0, s>>>(k, n_pairs, la);
        ego_sinkhorn_cols_kernel<<<dim3((k + 63) / 64, n_pairs), 64 * EGO_CG, 0, s>>>(k, la);
    }
    ego_rows_finish_kernel<<<row_grid, 256, 0, s>>>(la, coor_s, coor_t, thr2, k, n_pairs, perm, rowsum, wt, k + 1);
    ego_kabsch_kernel<<<n_pairs, 256, 0, s>>>(coor_s, wt, rowsum, k, pose);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Sinkhorn with a backward pass (training): models/egomotion.py:100-137 as a differentiable op of the [P,k,k] affinity.
// Forward = the in-place row / column kernels above on the padded matrix, recording the log-sum-exp each half-step
// subtracts.  Every intermediate matrix is then x0 - U[i] - V[j] (U, V = running sums of those vectors, 0 on the slack
// row / column), so the backward replays the 2*n_iters half-steps in reverse from x0 and the vectors alone:
//   y = x - lse(x)  =>  dx = dy - exp(y) * sum(dy)      (per normalised row / column; slack row / column: dx = dy)
// Two launches per iteration, each reading the gradient matrix and x0 and writing the gradient matrix once -- the torch
// formulation (pad, 12 slices, 6 logsumexp, 6 cats) keeps a dozen [P,1025,1025] intermediates and makes ~20 passes.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ego_copy_in_kernel(const float *__restrict__ x, int k, int n_pairs, float *__restrict__ la)
{
    const int kp = k + 1;
    const int64_t total = (int64_t)n_pairs * kp * kp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % kp), i = (int)((e / kp) % kp);
        const int64_t p = e / ((int64_t)kp * kp);
        la[e] = (i < k && j < k) ? x[(p * k + i) * k + j] : 0.f;
    }
}

__global__ __launch_bounds__(256) void ego_copy_out_kernel(const float *__restrict__ la, int k, int n_pairs, float *__restrict__ out)
{
    const int kp = k + 1;
    const int64_t total = (int64_t)n_pairs * k * k;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % k), i = (int)((e / k) % k);
        const int64_t p = e / ((int64_t)k * k);
        out[e] = la[(p * kp + i) * kp + j];
    }
}

// U[p][i] / V[p][j]: sums of the recorded vectors of the first `nu` row steps / `nv` column steps
__device__ __forceinline__ float ego_cum(const float *__restrict__ lse, int steps, int64_t stride, int64_t idx)
{
    float a = 0.f;
    for (int t = 0; t < steps; ++t) a += lse[t * stride + idx];
    return a;
}

// backward of a row step: rows 0..k-1, all k+1 columns; y[i][j] = x0[i][j] - U_nu[i] - V_nv[j]
__global__ __launch_bounds__(256) void ego_sinkhorn_rows_bwd_kernel(const float *__restrict__ x0, const float *__restrict__ lse_r,
                                                                    const float *__restrict__ lse_c, int nu, int nv, int k, int n_pairs,
                                                                    float *g)
{
    const int kp = k + 1;
    const int lane = threadIdx.x & 63;
    const int64_t n_rows = (int64_t)n_pairs * k, stride = n_rows;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const int64_t p = row / k;
        const int i = (int)(row % k);
        float *gr = g + (p * kp + i) * kp;
        const float *xr = x0 + (p * k + i) * k;
        float sm = 0.f;
        for (int j = lane; j < kp; j += 64) sm += gr[j];
        sm = wave_sum(sm);
        const float u = ego_cum(lse_r, nu, stride, row);
        for (int j = lane; j < kp; j += 64) {
            const float y = (j < k ? xr[j] - ego_cum(lse_c, nv, stride, p * k + j) : 0.f) - u;
            gr[j] -= expf(y) * sm;
        }
    }
}

// backward of a column step: columns 0..k-1, all k+1 rows; same tiling as the forward column kernel
__global__ __launch_bounds__(64 * EGO_CG) void ego_sinkhorn_cols_bwd_kernel(const float *__restrict__ x0, const float *__restrict__ lse_r,
                                                                    const float *__restrict__ lse_c, int nu, int nv, int k, int n_pairs,
                                                                    float *g)
{
    __shared__ float ssum[EGO_CG][64];
    const int kp = k + 1;
    const int64_t p = blockIdx.y, stride = (int64_t)n_pairs * k;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float *m = g + p * kp * kp;
    const int rows_per = (kp + EGO_CG - 1) / EGO_CG;
    const int r_lo = q * rows_per, r_hi = min(kp, r_lo + rows_per);
    float sm = 0.f;
    if (col < k)
        for (int i = r_lo; i < r_hi; ++i) sm += m[(int64_t)i * kp + col];
    ssum[q][threadIdx.x & 63] = sm;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int u = 0; u < EGO_CG; ++u) tot += ssum[u][threadIdx.x & 63];
    if (col < k) {
        const float v = ego_cum(lse_c, nv, stride, p * k + col);
        for (int i = r_lo; i < r_hi; ++i) {
            const float y = (i < k ? x0[(p * k + i) * k + col] - ego_cum(lse_r, nu, stride, p * k + i) : 0.f) - v;
            m[(int64_t)i * kp + col] -= expf(y) * tot;
        }
    }
}

// ---- read-only forward ----------------------------------------------------------------------------------------------------------
// After any number of half-steps the padded matrix is  y[i][j] = x0[i][j] - U[i] - V[j]  (x0 zero-padded; U / V = the log-sum-exps
// subtracted from row i / column j so far; the slack row carries -V[j], the slack column -U[i], the corner stays 0) -- the form the
// backward above replays.  Normalising row i replaces U[i] by  log( sum_j exp(x0[i][j] - V[j]) + 1 )  whatever it was, and likewise
// for a column.  So the forward never has to rewrite the matrix: a half-step reads x0 once and writes one vector (the kernels
// above read and write the padded matrix 2-3 times per half-step; 16 x 1024^2: 0.45 ms against 8 us per pass).  k % 4 == 0.
__global__ __launch_bounds__(256) void ego_sinkhorn_rows_ro_kernel(const float *__restrict__ x0, const float *__restrict__ V, int k,
                                                                   int n_pairs, float *__restrict__ U, float *__restrict__ lse_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t n_rows = (int64_t)n_pairs * k;
    const int k4 = k >> 2;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const int64_t p = row / k;
        const float4 *xr = reinterpret_cast<const float4 *>(x0 + row * k);
        const float4 *vr = reinterpret_cast<const float4 *>(V + p * k);
        float mx = 0.f;                                                      // the slack column: y = 0 - U[i], i.e. t = 0
        for (int j = lane; j < k4; j += 64) {
            const float4 a = xr[j], b = vr[j];
            mx = fmaxf(fmaxf(mx, fmaxf(a.x - b.x, a.y - b.y)), fmaxf(a.z - b.z, a.w - b.w));
        }
        mx = wave_max(mx);
        float sm = 0.f;
        for (int j = lane; j < k4; j += 64) {                                // second read of the row comes from L1 / L2
            const float4 a = xr[j], b = vr[j];
            sm += (expf(a.x - b.x - mx) + expf(a.y - b.y - mx)) + (expf(a.z - b.z - mx) + expf(a.w - b.w - mx));
        }
        const float u_new = mx + logf(wave_sum(sm) + expf(-mx));
        if (lane == 0) {
            lse_out[row] = u_new - U[row];                                   // what this half-step subtracted (the backward's record)
            U[row] = u_new;
        }
    }
}

// 64 columns x 16 row groups per workgroup; partial (max, sum) per column combined through LDS
__global__ __launch_bounds__(256) void ego_sinkhorn_cols_ro_kernel(const float *__restrict__ x0, const float *__restrict__ U, int k,
                                                                   float *__restrict__ V, float *__restrict__ lse_out)
{
    __shared__ float smax[16][64], ssum[16][64];
    const int p = blockIdx.y;
    const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;                  // 16 column quads, 16 row groups
    const int col = blockIdx.x * 64 + c4 * 4;
    const float *m = x0 + (int64_t)p * k * k;
    const float *u = U + (int64_t)p * k;
    float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()}, sm[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < k) {
        for (int i0 = rg; i0 < k; i0 += 16 * 8) {                            // 8 rows per round: one rescale per round and column
            float4 t[8];
            float cm[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = i0 + r * 16;
                if (i < k) {
                    const float4 a = *reinterpret_cast<const float4 *>(m + (int64_t)i * k + col);
                    const float ui = u[i];
                    t[r] = make_float4(a.x - ui, a.y - ui, a.z - ui, a.w - ui);
                    cm[0] = fmaxf(cm[0], t[r].x); cm[1] = fmaxf(cm[1], t[r].y); cm[2] = fmaxf(cm[2], t[r].z); cm[3] = fmaxf(cm[3], t[r].w);
                } else {
                    t[r] = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float nm = fmaxf(mx[q], cm[q]);
                if (nm > -__builtin_inff()) {
                    float acc = sm[q] * expf(mx[q] - nm);                    // exp(-inf) = 0 on the first round
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc += expf((q == 0 ? t[r].x : q == 1 ? t[r].y : q == 2 ? t[r].z : t[r].w) - nm);
                    sm[q] = acc;
                    mx[q] = nm;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { smax[rg][c4 * 4 + q] = mx[q]; ssum[rg][c4 * 4 + q] = sm[q]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < k) {
            float gm = 0.f;                                                  // the slack row: y = 0 - V[j], i.e. t = 0
#pragma unroll
            for (int g = 0; g < 16; ++g) gm = fmaxf(gm, smax[g][threadIdx.x]);
            float gs = expf(-gm);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float pm = smax[g][threadIdx.x];
                gs += (pm == -__builtin_inff()) ? 0.f : ssum[g][threadIdx.x] * expf(pm - gm);
            }
            const float v_new = gm + logf(gs);
            const int64_t o = (int64_t)p * k + c;
            lse_out[o] = v_new - V[o];
            V[o] = v_new;
        }
    }
}

__global__ __launch_bounds__(256) void ego_sinkhorn_finish_ro_kernel(const float4 *__restrict__ x0, const float *__restrict__ U,
                                                                     const float *__restrict__ V, int k, int64_t total4,
                                                                     float4 *__restrict__ out)
{
    const int k4 = k >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / k4;                                          // p * k + i
        const int j4 = (int)(e % k4);
        const int64_t p = row / k;
        const float4 a = x0[e];
        const float4 v = reinterpret_cast<const float4 *>(V + p * k)[j4];
        const float ui = U[row];
        out[e] = make_float4(a.x - ui - v.x, a.y - ui - v.y, a.z - ui - v.z, a.w - ui - v.w);
    }
}

// ---- backward in vector form --------------------------------------------------------------------------------------------------------
// Undoing a half-step subtracts a rank-one term weighted by that step's normalised matrix P_s = exp(x0 - U_s[i] - V_s[j]):
//   column step:  g -= P_s * c_s[j]   (c_s = current column sums of g over all k+1 rows),      row step:  g -= P_s * r_s[i].
// So the gradient matrix never has to be rewritten: dx0 = G - sum_s P_s (c_s[j] | r_s[i]) on the real block, and the sums obey vector
// recurrences -- a column of P_s (slack row included) sums to 1, so a column step zeroes the column sums and changes the row sums by
// -sum_j P_s[i][j] c_s[j]; a row step zeroes the row sums and changes the column sums by -sum_i P_s[i][j] r_s[i]: ONE read-only pass
// over x0 per half-step (a matrix-vector product), two passes for the sums of G, one final pass (read G and x0, write dx0) -- 10
// passes of 64 MB at 16 x 1024^2 instead of 26 read-modify-write passes over the padded gradient matrix.  k % 4 == 0.
__global__ __launch_bounds__(256) void sk_rowsum_kernel(const float *__restrict__ G, int k, int n_pairs, float *__restrict__ R)
{
    const int lane = threadIdx.x & 63, k4 = k >> 2;
    const int64_t n_rows = (int64_t)n_pairs * k;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const float4 *gr = reinterpret_cast<const float4 *>(G + row * k);
        float sm = 0.f;
        for (int j = lane; j < k4; j += 64) { const float4 a = gr[j]; sm += (a.x + a.y) + (a.z + a.w); }
        sm = wave_sum(sm);
        if (lane == 0) R[row] = sm;
    }
}

// column sums of G (W == NULL) or the column-side matrix-vector product of a row step:
//   out[j] = sum_i G[i][j]                                         |  out[j] = -exp(-V[j]) sum_i exp(x0[i][j] - U[i]) W[i]
// 64 columns x 16 row groups per workgroup, partial sums combined through LDS (the tiling of ego_sinkhorn_cols_ro_kernel)
__global__ __launch_bounds__(256) void sk_cols_kernel(const float *__restrict__ M, const float *__restrict__ lse_r, const float *__restrict__ lse_c,
                                                      int nu, int nv, const float *__restrict__ W, int k, int n_pairs, float *__restrict__ out)
{
    __shared__ float ssum[16][64];
    const int p = blockIdx.y;
    const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + c4 * 4;
    const float *m = M + (int64_t)p * k * k;
    const int64_t stride = (int64_t)n_pairs * k;
    float sm[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < k) {
        float4 vj = make_float4(0.f, 0.f, 0.f, 0.f);
        if (W) {
            vj.x = ego_cum(lse_c, nv, stride, (int64_t)p * k + col);     vj.y = ego_cum(lse_c, nv, stride, (int64_t)p * k + col + 1);
            vj.z = ego_cum(lse_c, nv, stride, (int64_t)p * k + col + 2); vj.w = ego_cum(lse_c, nv, stride, (int64_t)p * k + col + 3);
        }
        for (int i = rg; i < k; i += 16) {
            const float4 a = *reinterpret_cast<const float4 *>(m + (int64_t)i * k + col);
            if (W) {
                const float ui = ego_cum(lse_r, nu, stride, (int64_t)p * k + i), wi = W[(int64_t)p * k + i];
                sm[0] += wi * expf(a.x - ui - vj.x); sm[1] += wi * expf(a.y - ui - vj.y);
                sm[2] += wi * expf(a.z - ui - vj.z); sm[3] += wi * expf(a.w - ui - vj.w);
            } else {
                sm[0] += a.x; sm[1] += a.y; sm[2] += a.z; sm[3] += a.w;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[rg][c4 * 4 + q] = sm[q];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < k) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += ssum[g][threadIdx.x];
            out[(int64_t)p * k + c] = W ? -t : t;
        }
    }
}

// row-side matrix-vector product of a column step: out[i] = Rin[i] - exp(-U[i]) sum_j exp(x0[i][j] - V[j]) C[j]   (Rin == NULL: 0)
__global__ __launch_bounds__(256) void sk_rows_kernel(const float *__restrict__ x0, const float *__restrict__ lse_r, const float *__restrict__ lse_c,
                                                      int nu, int nv, const float *__restrict__ C, const float *__restrict__ Rin, int k, int n_pairs,
                                                      float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t n_rows = (int64_t)n_pairs * k, stride = n_rows;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const int64_t p = row / k;
        const float *xr = x0 + row * k;
        const float u = ego_cum(lse_r, nu, stride, row);
        float sm = 0.f;
        for (int j = lane; j < k; j += 64) sm += C[p * k + j] * expf(xr[j] - u - ego_cum(lse_c, nv, stride, p * k + j));
        sm = wave_sum(sm);
        if (lane == 0) out[row] = (Rin ? Rin[row] : 0.f) - sm;
    }
}

// dx0[i][j] = G[i][j] - sum_it ( Wc[it][j] exp(x0 - U_{it+1}[i] - V_{it+1}[j]) + Wr[it][i] exp(x0 - U_{it+1}[i] - V_it[j]) )
__global__ __launch_bounds__(256) void sk_final_kernel(const float4 *__restrict__ G, const float4 *__restrict__ x0, const float *__restrict__ lse_r,
                                                       const float *__restrict__ lse_c, const float *__restrict__ Wc, const float *__restrict__ Wr,
                                                       int n_iters, int k, int n_pairs, int64_t total4, float4 *__restrict__ out)
{
    const int k4 = k >> 2;
    const int64_t stride = (int64_t)n_pairs * k;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / k4;                                          // p * k + i
        const int j4 = (int)(e % k4);
        const int64_t col0 = (row / k) * k + (int64_t)j4 * 4;               // p * k + j
        const float4 a = x0[e];
        float4 acc = G[e];
        float u = 0.f;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                          // V_it
        for (int it = 0; it < n_iters; ++it) {
            u += lse_r[it * stride + row];                                   // U_{it+1}
            const float wr = Wr[it * stride + row];
            acc.x -= wr * expf(a.x - u - v.x); acc.y -= wr * expf(a.y - u - v.y); acc.z -= wr * expf(a.z - u - v.z); acc.w -= wr * expf(a.w - u - v.w);
            const float4 l = *reinterpret_cast<const float4 *>(lse_c + it * stride + col0);
            v = make_float4(v.x + l.x, v.y + l.y, v.z + l.z, v.w + l.w);    // V_{it+1}
            const float4 wc = *reinterpret_cast<const float4 *>(Wc + it * stride + col0);
            acc.x -= wc.x * expf(a.x - u - v.x); acc.y -= wc.y * expf(a.y - u - v.y); acc.z -= wc.z * expf(a.z - u - v.z); acc.w -= wc.w * expf(a.w - u - v.w);
        }
        out[e] = acc;
    }
}

extern "C" int pcacc_sinkhorn_train_workspace_bytes(int n_pairs, int k, size_t *bytes)
{
    if (!bytes || n_pairs < 1 || k < 1) return PCACC_E_ARG;
    *bytes = (size_t)n_pairs * (k + 1) * (k + 1) * sizeof(float);
    return PCACC_OK;
}

// log_alpha [P,k,k] -> log_perm [P,k,k]; lse_rows, lse_cols [n_iters][P][k] (saved for the backward)
extern "C" int pcacc_sinkhorn_forward(const float *log_alpha, int n_pairs, int k, int n_iters, float *log_perm, float *lse_rows,
                                      float *lse_cols, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!log_alpha || !log_perm || !lse_rows || !lse_cols || !workspace || n_pairs < 1 || k < 1 || n_iters < 0) return PCACC_E_ARG;
    if (workspace_bytes < (size_t)n_pairs * (k + 1) * (k + 1) * sizeof(float)) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    const int row_grid = pcacc_grid((int64_t)n_pairs * k * 64, 256);
    const int64_t stride = (int64_t)n_pairs * k;
    if (k % 4 == 0) {
        // read-only passes over log_alpha; the workspace (sized for the padded matrix) holds the two vectors U, V
        float *U = reinterpret_cast<float *>(workspace), *V = U + stride;
        if (hipMemsetAsync(U, 0, (size_t)2 * stride * sizeof(float), s) != hipSuccess) return PCACC_E_LAUNCH;
        for (int it = 0; it < n_iters; ++it) {
            ego_sinkhorn_rows_ro_kernel<<<row_grid, 256, 0, s>>>(log_alpha, V, k, n_pairs, U, lse_rows + it * stride);
            ego_sinkhorn_cols_ro_kernel<<<dim3((k + 63) / 64, n_pairs), 256, 0, s>>>(log_alpha, U, k, V, lse_cols + it * stride);
        }
        const int64_t total4 = stride * (k / 4);
        ego_sinkhorn_finish_ro_kernel<<<pcacc_grid(total4, 256), 256, 0, s>>>(reinterpret_cast<const float4 *>(log_alpha), U, V, k, total4,
                                                                              reinterpret_cast<float4 *>(log_perm));
        PCACC_CHECK_LAUNCH();
        return PCACC_OK;
    }
    float *la = reinterpret_cast<float *>(workspace);
    const int64_t padded = (int64_t)n_pairs * (k + 1) * (k + 1);
    ego_copy_in_kernel<<<pcacc_grid(padded, 256), 256, 0, s>>>(log_alpha, k, n_pairs, la);
    for (int it = 0; it < n_iters; ++it) {
        ego_sinkhorn_rows_kernel<<<row_grid, 256, 0, s>>>(k, n_pairs, la, lse_rows + it * stride);
        ego_sinkhorn_cols_kernel<<<dim3((k + 63) / 64, n_pairs), 64 * EGO_CG, 0, s>>>(k, la, lse_cols + it * stride);
    }
    ego_copy_out_kernel<<<pcacc_grid((int64_t)n_pairs * k * k, 256), 256, 0, s>>>(la, k, n_pairs, log_perm);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// grad_log_perm [P,k,k] -> grad_log_alpha [P,k,k]
extern "C" int pcacc_sinkhorn_backward(const float *grad_log_perm, const float *log_alpha, const float *lse_rows, const float *lse_cols,
                                       int n_pairs, int k, int n_iters, float *grad_log_alpha, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    if (!grad_log_perm || !log_alpha || !lse_rows || !lse_cols || !grad_log_alpha || !workspace || n_pairs < 1 || k < 1 || n_iters < 0)
        return PCACC_E_ARG;
    if (workspace_bytes < (size_t)n_pairs * (k + 1) * (k + 1) * sizeof(float)) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    float *g = reinterpret_cast<float *>(workspace);
    const int64_t padded = (int64_t)n_pairs * (k + 1) * (k + 1);
    const int row_grid = pcacc_grid((int64_t)n_pairs * k * 64, 256);
    const int64_t stride = (int64_t)n_pairs * k;
    if (k % 4 == 0 && n_iters >= 1 && (int64_t)(2 * n_iters + 1) * stride <= padded) {
        // vector form (see above): Wc[it] / Wr[it] = the column / row sums each half-step subtracts with, R0 = row sums of G
        float *Wc = g, *Wr = g + (int64_t)n_iters * stride, *R0 = g + (int64_t)2 * n_iters * stride;
        const dim3 cgrid((k + 63) / 64, n_pairs);
        sk_rowsum_kernel<<<row_grid, 256, 0, s>>>(grad_log_perm, k, n_pairs, R0);
        sk_cols_kernel<<<cgrid, 256, 0, s>>>(grad_log_perm, nullptr, nullptr, 0, 0, nullptr, k, n_pairs, Wc + (int64_t)(n_iters - 1) * stride);
        for (int it = n_iters - 1; it >= 0; --it) {
            // column step `it` (P = exp(x0 - U_{it+1} - V_{it+1})): the row sums it leaves = what row step `it` subtracts with
            sk_rows_kernel<<<row_grid, 256, 0, s>>>(log_alpha, lse_rows, lse_cols, it + 1, it + 1, Wc + (int64_t)it * stride,
                                                    it == n_iters - 1 ? R0 : nullptr, k, n_pairs, Wr + (int64_t)it * stride);
            // row step `it` (P = exp(x0 - U_{it+1} - V_it)): the column sums it leaves = what column step `it - 1` subtracts with
            if (it > 0)
                sk_cols_kernel<<<cgrid, 256, 0, s>>>(log_alpha, lse_rows, lse_cols, it + 1, it, Wr + (int64_t)it * stride, k, n_pairs,
                                                     Wc + (int64_t)(it - 1) * stride);
        }
        const int64_t total4 = stride * (k / 4);
        sk_final_kernel<<<pcacc_grid(total4, 256), 256, 0, s>>>(reinterpret_cast<const float4 *>(grad_log_perm), reinterpret_cast<const float4 *>(log_alpha),
                                                                lse_rows, lse_cols, Wc, Wr, n_iters, k, n_pairs, total4,
                                                                reinterpret_cast<float4 *>(grad_log_alpha));
        PCACC_CHECK_LAUNCH();
        return PCACC_OK;
    }
    ego_copy_in_kernel<<<pcacc_grid(padded, 256), 256, 0, s>>>(grad_log_perm, k, n_pairs, g);
    for (int it = n_iters - 1; it >= 0; --it) {
        // after column step `it`: U = rows 0..it, V = cols 0..it; after row step `it`: U = rows 0..it, V = cols 0..it-1
        ego_sinkhorn_cols_bwd_kernel<<<dim3((k + 63) / 64, n_pairs), 64 * EGO_CG, 0, s>>>(log_alpha, lse_rows, lse_cols, it + 1, it + 1, k, n_pairs, g);
        ego_sinkhorn_rows_bwd_kernel<<<row_grid, 256, 0, s>>>(log_alpha, lse_rows, lse_cols, it + 1, it, k, n_pairs, g);
    }
    ego_copy_out_kernel<<<pcacc_grid((int64_t)n_pairs * k * k, 256), 256, 0, s>>>(g, k, n_pairs, grad_log_alpha);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Key-point draws for the ego-motion head (models/egomotion.py:156-166): k distinct indices out of n, uniformly, for many
// (n, draw) pairs in ONE launch.  The reference shuffles all n indices on the host (torch.randperm) and keeps the first k;
// its device twin sorts n random keys, a dozen launches per draw and 32 draws per 4-sample step.  Here draw d evaluates
// a keyed pseudo-random permutation of [0, n_d) at the points 0..k-1: a 4-round Feistel network on the next even number
// of bits, cycle-walked back into range (a bijection, so the k outputs are distinct).  n <= k gives 0..n-1 followed by
// n-1 repeated, like the reference.  This is the 'device' sampler of the throughput runs; parity tests use the
// reference's host RNG stream.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t kp_mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void sample_subsets_kernel(const int32_t *__restrict__ counts, int k, uint64_t seed, int64_t *__restrict__ out)
{
    const int d = blockIdx.x;
    const uint32_t n = (uint32_t)counts[d];
    int64_t *dst = out + (int64_t)d * k;
    if (n <= (uint32_t)k) {
        for (int i = threadIdx.x; i < k; i += 256) dst[i] = n == 0 ? 0 : ((uint32_t)i < n ? i : n - 1);
        return;
    }
    int half = 1;
    while ((1u << (2 * half)) < n) ++half;                       // 2*half bits cover [0, n)
    const uint32_t hmask = (1u << half) - 1;
    uint32_t key[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) key[r] = kp_mix((uint32_t)(seed >> (8 * r)) ^ kp_mix((uint32_t)(seed >> 32) + 0x9e3779b9u * (uint32_t)(d * 4 + r + 1)));
    for (int i = threadIdx.x; i < k; i += 256) {
        uint32_t x = (uint32_t)i;
        do {
            uint32_t l = x >> half, r = x & hmask;
#pragma unroll
            for (int round = 0; round < 4; ++round) {
                const uint32_t f = kp_mix(r ^ key[round]) & hmask;
                const uint32_t nl = r;
                r = l ^ f;
                l = nl;
            }
            x = (l << half) | r;
        } while (x >= n);
        dst[i] = x;
    }
}

extern "C" int pcacc_sample_subsets(const int32_t *counts, int32_t n_draws, int32_t k, uint64_t seed, int64_t *out, void *stream)
{
    if (n_draws < 0 || k <= 0) return PCACC_E_ARG;
    if (n_draws == 0) return PCACC_OK;
    if (!counts || !out) return PCACC_E_ARG;
    sample_subsets_kernel<<<n_draws, 256, 0, pcacc_stream(stream)>>>(counts, k, seed, out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched 3x3 SVD as a differentiable op (toolbox/register_utils.py:293 `torch.svd(cov_mat)` inside the Kabsch solve of the
// training path).  The library call checks its status word on the host -- two queue drains per step in front of the small
// launches of the ego head.  One lane per matrix: one-sided Jacobi in float64 (the routine of the fused eval kernel above),
// a = u diag(s) v^T with s descending; the backward pass is the closed form for a square matrix with distinct singular values,
//   ga = u [ (skew(u^T gu) / E) diag(s) + diag(s) (skew(v^T gv) / E) + diag(gs) ] v^T,   E_ij = s_j^2 - s_i^2,  skew(x) = x - x^T.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void svd3_kernel(const float *__restrict__ a, int64_t n, float *__restrict__ u, float *__restrict__ s,
                                                  float *__restrict__ v)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double am[3][3], um[3][3], sm[3], vm[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) am[r][c] = a[i * 9 + r * 3 + c];
    jacobi_svd3(am, um, sm, vm);
    for (int r = 0; r < 3; ++r) {
        s[i * 3 + r] = (float)sm[r];
        for (int c = 0; c < 3; ++c) {
            u[i * 9 + r * 3 + c] = (float)um[r][c];
            v[i * 9 + r * 3 + c] = (float)vm[r][c];
        }
    }
}

__global__ __launch_bounds__(64) void svd3_bwd_kernel(const float *__restrict__ u, const float *__restrict__ s, const float *__restrict__ v,
                                                      const float *__restrict__ gu, const float *__restrict__ gs, const float *__restrict__ gv,
                                                      int64_t n, float *__restrict__ ga)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double U[3][3], V[3][3], S[3], GU[3][3], GV[3][3], inner[3][3];
    for (int r = 0; r < 3; ++r) {
        S[r] = s[i * 3 + r];
        for (int c = 0; c < 3; ++c) {
            U[r][c] = u[i * 9 + r * 3 + c];
            V[r][c] = v[i * 9 + r * 3 + c];
            GU[r][c] = gu ? gu[i * 9 + r * 3 + c] : 0.0;
            GV[r][c] = gv ? gv[i * 9 + r * 3 + c] : 0.0;
        }
    }
    double ku[3][3], kv[3][3];                                                 // u^T gu, v^T gv
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double x = 0, y = 0;
            for (int k = 0; k < 3; ++k) { x += U[k][r] * GU[k][c]; y += V[k][r] * GV[k][c]; }
            ku[r][c] = x, kv[r][c] = y;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            if (r == c) { inner[r][c] = gs ? gs[i * 3 + r] : 0.0; continue; }
            const double e = S[c] * S[c] - S[r] * S[r];
            inner[r][c] = (ku[r][c] - ku[c][r]) / e * S[c] + S[r] * (kv[r][c] - kv[c][r]) / e;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double x = 0;
            for (int k = 0; k < 3; ++k)
                for (int l = 0; l < 3; ++l) x += U[r][k] * inner[k][l] * V[c][l];
            ga[i * 9 + r * 3 + c] = (float)x;
        }
}

extern "C" int pcacc_svd3(const float *a, int64_t n, float *u, float *s, float *v, void *stream)
{
    if (n < 0) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (!a || !u || !s || !v) return PCACC_E_ARG;
    svd3_kernel<<<(unsigned)((n + 63) / 64), 64, 0, pcacc_stream(stream)>>>(a, n, u, s, v);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_svd3_backward(const float *u, const float *s, const float *v, const float *grad_u, const float *grad_s, const float *grad_v,
                                   int64_t n, float *grad_a, void *stream)
{
    if (n < 0) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (!u || !s || !v || !grad_a) return PCACC_E_ARG;
    svd3_bwd_kernel<<<(unsigned)((n + 63) / 64), 64, 0, pcacc_stream(stream)>>>(u, s, v, grad_u, grad_s, grad_v, n, grad_a);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- the same two stages for the TRAINING path (models/egomotion.py:169-184 under autograd) -------------------------------------------
// The batched torch formulation spent ~17 element-wise passes over the [P, k, k] matrices (64 MB each at 16 x 1024^2) forward and as many
// backward: square_distance's matmul / scale / shift / clamp, the affinity's sub / neg / div, exp, the support mask and its product, the
// row sums and the soft targets -- about 1 ms of a 29 ms step.  Here: the affinity comes from the eval pipeline's tiled kernel (one write),
// its backward is one pass producing d(dot) (the two feature gradients are then library GEMMs on it) plus the two scalar gradients; perm,
// row sums and soft targets are one pass over the Sinkhorn result, their backward one pass as well.

// d(dot) = 2 g / b where the clamp passed, partial sums of g and g * aff per workgroup (d softplus(alpha) = sum g / b, d denom = -sum g aff / b)
__global__ __launch_bounds__(256) void ego_affinity_bwd_kernel(const float4 *__restrict__ g_aff, const float4 *__restrict__ aff, int64_t n4,
                                                               const float *__restrict__ params, float4 *__restrict__ g_dot,
                                                               double *__restrict__ partial, int tail)
{
    const float a = params[0], b = params[1];
    const float two_over_b = 2.0f / b;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const float4 g = g_aff[e], v = aff[e];
        // aff = -(dist - a) / b  =>  dist = a - aff * b; the clamp at 1e-12 passes the gradient where the un-clamped distance was >= 1e-12
        float4 o;
        o.x = (a - v.x * b > 1e-12f) ? g.x * two_over_b : 0.f;
        o.y = (a - v.y * b > 1e-12f) ? g.y * two_over_b : 0.f;
        o.z = (a - v.z * b > 1e-12f) ? g.z * two_over_b : 0.f;
        o.w = (a - v.w * b > 1e-12f) ? g.w * two_over_b : 0.f;
        g_dot[e] = o;
        s1 += (double)g.x + (double)g.y + (double)g.z + (double)g.w;
        s2 += (double)g.x * v.x + (double)g.y * v.y + (double)g.z * v.z + (double)g.w * v.w;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) {            // the last n % 4 elements
        const int64_t e = n4 * 4 + threadIdx.x;
        const float g = reinterpret_cast<const float *>(g_aff)[e], v = reinterpret_cast<const float *>(aff)[e];
        reinterpret_cast<float *>(g_dot)[e] = (a - v * b > 1e-12f) ? g * two_over_b : 0.f;
        s1 += (double)g;
        s2 += (double)g * v;
    }
    __shared__ double red[8];
    s1 = block_sum256(s1, red);
    s2 = block_sum256(s2, red);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s1; partial[2 * blockIdx.x + 1] = s2; }
}

__global__ __launch_bounds__(256) void ego_affinity_bwd_final_kernel(const double *__restrict__ partial, int nb, const float *__restrict__ params,
                                                                     float *__restrict__ g_params)
{
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) { s1 += partial[2 * i]; s2 += partial[2 * i + 1]; }
    __shared__ double red[8];
    s1 = block_sum256(s1, red);
    s2 = block_sum256(s2, red);
    if (threadIdx.x == 0) {
        const double b = params[1];
        g_params[0] = (float)(s1 / b);
        g_params[1] = (float)(-s2 / b);
    }
}

// d(log_perm)[i, j] = perm[i, j] * (g_perm[i, j] + A_i + <B_i, ct_j>),  A_i = g_rowsum_i - <g_wt_i, wt_i> / (r_i + eps),  B_i = g_wt_i / (r_i + eps)
// (perm = exp(log_perm) * support: the support mask carries no gradient and zero entries of perm give zero)
__global__ __launch_bounds__(256) void ego_perm_bwd_kernel(const float *__restrict__ g_perm, const float *__restrict__ g_rowsum,
                                                           const float *__restrict__ g_wt, const float *__restrict__ perm,
                                                           const float *__restrict__ ct, const float *__restrict__ rowsum,
                                                           const float *__restrict__ wt, int k, int n_pairs, float *__restrict__ g_lp,
                                                           const float *__restrict__ g_colsum)
{
    const int lane = threadIdx.x & 63;
    const int64_t n_rows = (int64_t)n_pairs * k;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const int p = (int)(row / k);
        const float *t = ct + (int64_t)p * k * 3;
        const float den = rowsum[row] + 1e-20f;
        float bx = 0.f, by = 0.f, bz = 0.f, a = g_rowsum ? g_rowsum[row] : 0.f;
        if (g_wt) {
            bx = g_wt[row * 3 + 0] / den; by = g_wt[row * 3 + 1] / den; bz = g_wt[row * 3 + 2] / den;
            a -= bx * wt[row * 3 + 0] + by * wt[row * 3 + 1] + bz * wt[row * 3 + 2];
        }
        const float *gc = g_colsum ? g_colsum + (int64_t)p * k : nullptr;
        for (int j = lane; j < k; j += 64) {
            const float g = (g_perm ? g_perm[row * k + j] : 0.f) + (gc ? gc[j] : 0.f) + a + (bx * t[j * 3 + 0] + by * t[j * 3 + 1] + bz * t[j * 3 + 2]);
            g_lp[row * k + j] = g * perm[row * k + j];
        }
    }
}

// column sums of perm (the other half of the outlier loss, libs/outlier_loss.py): 64 columns x 4 row lanes per workgroup, fixed order
__global__ __launch_bounds__(256) void ego_colsum_kernel(const float *__restrict__ perm, int k, float *__restrict__ colsum)
{
    __shared__ float red[4][64];
    const int p = blockIdx.y, j = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const float *m = perm + (int64_t)p * k * k;
    float s0 = 0.f, s1 = 0.f;
    if (j < k) {
        int i = rl;
        for (; i + 4 < k; i += 8) { s0 += m[(int64_t)i * k + j]; s1 += m[(int64_t)(i + 4) * k + j]; }
        if (i < k) s0 += m[(int64_t)i * k + j];
    }
    red[rl][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (rl == 0 && j < k) colsum[(int64_t)p * k + j] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// affinity [P, k, k] = -(max(2 - 2 <fs_i, ft_j>, 1e-12) - params[0]) / params[1]   (params = softplus(alpha), exp(beta) + 0.02 on the device)
extern "C" int pcacc_ego_affinity_forward(const float *feats_s, const float *feats_t, const float *params, int n_pairs, int k, int c,
                                          float *affinity, void *stream)
{
    if (n_pairs < 1 || k < 1 || c < 1 || !feats_s || !feats_t || !params || !affinity) return PCACC_E_ARG;
    const int tiles = (k + EGO_TILE - 1) / EGO_TILE;
    ego_affinity_kernel<<<dim3(tiles, tiles, n_pairs), 256, 0, pcacc_stream(stream)>>>(feats_s, feats_t, k, c, params, affinity, k);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_ego_affinity_backward_workspace_bytes(size_t *bytes)
{
    if (!bytes) return PCACC_E_ARG;
    *bytes = (size_t)PCACC_CUS * 4 * 2 * sizeof(double);
    return PCACC_OK;
}

// grad_dot [P, k, k] = d loss / d <fs_i, ft_j> (the feature gradients are grad_dot @ ft and grad_dot^T @ fs); grad_params [2]
extern "C" int pcacc_ego_affinity_backward(const float *grad_affinity, const float *affinity, const float *params, int64_t n, float *grad_dot,
                                           float *grad_params, void *workspace, size_t workspace_bytes, void *stream)
{
    size_t need;
    pcacc_ego_affinity_backward_workspace_bytes(&need);
    if (n < 1 || !grad_affinity || !affinity || !params || !grad_dot || !grad_params || !workspace) return PCACC_E_ARG;
    if (workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    const int nb = pcacc_grid(n / 4 > 0 ? n / 4 : 1, 256, PCACC_CUS * 4);
    double *partial = static_cast<double *>(workspace);
    ego_affinity_bwd_kernel<<<nb, 256, 0, s>>>(reinterpret_cast<const float4 *>(grad_affinity), reinterpret_cast<const float4 *>(affinity), n / 4, params,
                                               reinterpret_cast<float4 *>(grad_dot), partial, (int)(n % 4));
    ego_affinity_bwd_final_kernel<<<1, 256, 0, s>>>(partial, nb, params, grad_params);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// perm = exp(log_perm) * [ |cs_i - ct_j|^2 < thr2 ], rowsum [P, k], weighted_t [P, k, 3] = perm @ ct / (rowsum + 1e-20)   (egomotion.py:173-184);
// colsum [P, k] (may be NULL): the column sums the outlier loss takes beside the row sums
extern "C" int pcacc_ego_perm_forward(const float *log_perm, const float *coor_s, const float *coor_t, const float *thr2, int n_pairs, int k,
                                      float *perm, float *rowsum, float *weighted_t, float *colsum, void *stream)
{
    if (n_pairs < 1 || k < 1 || !log_perm || !coor_s || !coor_t || !thr2 || !perm || !rowsum || !weighted_t) return PCACC_E_ARG;
    const int row_grid = pcacc_grid((int64_t)n_pairs * k, 4, PCACC_CUS * 16);
    ego_rows_finish_kernel<<<row_grid, 256, 0, pcacc_stream(stream)>>>(log_perm, coor_s, coor_t, thr2, k, n_pairs, perm, rowsum, weighted_t, k);
    if (colsum) ego_colsum_kernel<<<dim3((k + 63) / 64, n_pairs), 256, 0, pcacc_stream(stream)>>>(perm, k, colsum);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// grad_log_perm from the gradients of the three results (any of them NULL = 0)
extern "C" int pcacc_ego_perm_backward(const float *grad_perm, const float *grad_rowsum, const float *grad_weighted_t, const float *grad_colsum,
                                       const float *perm, const float *coor_t, const float *rowsum, const float *weighted_t, int n_pairs, int k,
                                       float *grad_log_perm, void *stream)
{
    if (n_pairs < 1 || k < 1 || !perm || !coor_t || !rowsum || !weighted_t || !grad_log_perm) return PCACC_E_ARG;
    const int row_grid = pcacc_grid((int64_t)n_pairs * k, 4, PCACC_CUS * 16);
    ego_perm_bwd_kernel<<<row_grid, 256, 0, pcacc_stream(stream)>>>(grad_perm, grad_rowsum, grad_weighted_t, perm, coor_t, rowsum, weighted_t, k,
                                                                    n_pairs, grad_log_perm, grad_colsum);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- weighted Kabsch, the part in front of the 3x3 SVD, under autograd (toolbox/register_utils.py:263-291) -----------------------------------
// weights w [P,k] -> normalised wn = w / (sum w + 1e-7), wsum = sum wn + 1e-7, weighted means m1, m2 of x1, x2 [P,k,3], covariance
// cov[a][b] = sum_i (x1_i - m1)[a] wn_i (x2_i - m2)[b].  One workgroup per pair, sums in float64.  The batched torch formulation is ~15 small
// launches forward and ~35 backward on [16, 1024, 3] tensors.
template <int N>
__device__ __forceinline__ void kb_block_sums(double (&v)[N], double *red /*[4 * N]*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v[i] += __shfl_xor(v[i], d, 64);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < N; ++i) red[w * N + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = red[i] + red[N + i] + red[2 * N + i] + red[3 * N + i];
}

__global__ __launch_bounds__(256) void kabsch_cov_fwd_kernel(const float *__restrict__ x1, const float *__restrict__ x2, const float *__restrict__ w,
                                                             int k, float *__restrict__ cov, float *__restrict__ m1o, float *__restrict__ m2o,
                                                             float *__restrict__ norm /*[P][2] = W, wsum*/)
{
    __shared__ double red[4 * 9];
    const int p = blockIdx.x;
    x1 += (int64_t)p * k * 3; x2 += (int64_t)p * k * 3; w += (int64_t)p * k;
    double a1[1] = {0.0};
    for (int i = threadIdx.x; i < k; i += 256) a1[0] += w[i];
    kb_block_sums<1>(a1, red);
    const double W = a1[0] + 1e-7;
    double a7[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < k; i += 256) {
        const double wn = w[i] / W;
        a7[0] += wn;
        a7[1] += wn * x1[i * 3 + 0]; a7[2] += wn * x1[i * 3 + 1]; a7[3] += wn * x1[i * 3 + 2];
        a7[4] += wn * x2[i * 3 + 0]; a7[5] += wn * x2[i * 3 + 1]; a7[6] += wn * x2[i * 3 + 2];
    }
    kb_block_sums<7>(a7, red);
    const double wsum = a7[0] + 1e-7;
    const double m1[3] = {a7[1] / wsum, a7[2] / wsum, a7[3] / wsum}, m2[3] = {a7[4] / wsum, a7[5] / wsum, a7[6] / wsum};
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < k; i += 256) {
        const double wn = w[i] / W;
        const double c1[3] = {x1[i * 3 + 0] - m1[0], x1[i * 3 + 1] - m1[1], x1[i * 3 + 2] - m1[2]};
        const double c2[3] = {wn * (x2[i * 3 + 0] - m2[0]), wn * (x2[i * 3 + 1] - m2[1]), wn * (x2[i * 3 + 2] - m2[2])};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) c[a * 3 + b] += c1[a] * c2[b];
    }
    kb_block_sums<9>(c, red);
    if (threadIdx.x < 9) cov[p * 9 + threadIdx.x] = (float)c[threadIdx.x];
    if (threadIdx.x < 3) { m1o[p * 3 + threadIdx.x] = (float)m1[threadIdx.x]; m2o[p * 3 + threadIdx.x] = (float)m2[threadIdx.x]; }
    if (threadIdx.x == 0) { norm[p * 2] = (float)W; norm[p * 2 + 1] = (float)wsum; }
}

// gradients of x2 and w from those of cov, m1, m2 (x1 carries none: pillar means)
__global__ __launch_bounds__(256) void kabsch_cov_bwd_kernel(const float *__restrict__ x1, const float *__restrict__ x2, const float *__restrict__ w,
                                                             const float *__restrict__ m1i, const float *__restrict__ m2i,
                                                             const float *__restrict__ norm, const float *__restrict__ g_cov,
                                                             const float *__restrict__ g_m1, const float *__restrict__ g_m2, int k,
                                                             float *__restrict__ g_x2, float *__restrict__ g_w)
{
    __shared__ double red[4 * 6];
    const int p = blockIdx.x;
    x1 += (int64_t)p * k * 3; x2 += (int64_t)p * k * 3; w += (int64_t)p * k;
    g_x2 += (int64_t)p * k * 3; g_w += (int64_t)p * k;
    const double W = norm[p * 2], wsum = norm[p * 2 + 1];
    double m1[3], m2[3], gc[9], gm1[3], gm2[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        m1[d] = m1i[p * 3 + d]; m2[d] = m2i[p * 3 + d];
        gm1[d] = g_m1 ? g_m1[p * 3 + d] : 0.f; gm2[d] = g_m2 ? g_m2[p * 3 + d] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) gc[e] = g_cov ? g_cov[p * 9 + e] : 0.f;
    // phase 1: sum_i g_c1_i, sum_i g_c2_i -> totals on the means
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < k; i += 256) {
        const double wn = w[i] / W;
        const double c1[3] = {x1[i * 3 + 0] - m1[0], x1[i * 3 + 1] - m1[1], x1[i * 3 + 2] - m1[2]};
        const double c2[3] = {x2[i * 3 + 0] - m2[0], x2[i * 3 + 1] - m2[1], x2[i * 3 + 2] - m2[2]};
#pragma unroll
        for (int a = 0; a < 3; ++a) s[a] += wn * (gc[a * 3 + 0] * c2[0] + gc[a * 3 + 1] * c2[1] + gc[a * 3 + 2] * c2[2]);      // g_c1_i[a]
#pragma unroll
        for (int b = 0; b < 3; ++b) s[3 + b] += wn * (gc[0 * 3 + b] * c1[0] + gc[1 * 3 + b] * c1[1] + gc[2 * 3 + b] * c1[2]);  // g_c2_i[b]
    }
    kb_block_sums<6>(s, red);
    const double G1[3] = {gm1[0] - s[0], gm1[1] - s[1], gm1[2] - s[2]}, G2[3] = {gm2[0] - s[3], gm2[1] - s[4], gm2[2] - s[5]};
    const double g_wsum = -((m1[0] * G1[0] + m1[1] * G1[1] + m1[2] * G1[2]) + (m2[0] * G2[0] + m2[1] * G2[1] + m2[2] * G2[2])) / wsum;
    auto g_wn = [&](int i) {
        const double c1[3] = {x1[i * 3 + 0] - m1[0], x1[i * 3 + 1] - m1[1], x1[i * 3 + 2] - m1[2]};
        const double c2[3] = {x2[i * 3 + 0] - m2[0], x2[i * 3 + 1] - m2[1], x2[i * 3 + 2] - m2[2]};
        double g = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g += gc[a * 3 + b] * c1[a] * c2[b];
        g += ((double)x2[i * 3 + 0] * G2[0] + (double)x2[i * 3 + 1] * G2[1] + (double)x2[i * 3 + 2] * G2[2]) / wsum;
        g += ((double)x1[i * 3 + 0] * G1[0] + (double)x1[i * 3 + 1] * G1[1] + (double)x1[i * 3 + 2] * G1[2]) / wsum;
        return g + g_wsum;
    };
    // phase 2: sum_j g_wn_j w_j (the normalisation w / (sum w + eps))
    double t[1] = {0.0};
    for (int i = threadIdx.x; i < k; i += 256) t[0] += g_wn(i) * w[i];
    kb_block_sums<1>(t, red);
    // phase 3: the results
    for (int i = threadIdx.x; i < k; i += 256) {
        const double wn = w[i] / W;
        const double c1[3] = {x1[i * 3 + 0] - m1[0], x1[i * 3 + 1] - m1[1], x1[i * 3 + 2] - m1[2]};
        g_w[i] = (float)(g_wn(i) / W - t[0] / (W * W));
#pragma unroll
        for (int b = 0; b < 3; ++b)
            g_x2[i * 3 + b] = (float)(wn * (gc[0 * 3 + b] * c1[0] + gc[1 * 3 + b] * c1[1] + gc[2 * 3 + b] * c1[2]) + wn * G2[b] / wsum);
    }
}

// cov [P,3,3], m1 / m2 [P,3] (the weighted means), norm [P,2] (kept for the backward) from x1, x2 [P,k,3] and the weights w [P,k]
extern "C" int pcacc_kabsch_cov_forward(const float *x1, const float *x2, const float *w, int n_pairs, int k, float *cov, float *m1, float *m2,
                                        float *norm, void *stream)
{
    if (n_pairs < 1 || k < 1 || !x1 || !x2 || !w || !cov || !m1 || !m2 || !norm) return PCACC_E_ARG;
    kabsch_cov_fwd_kernel<<<n_pairs, 256, 0, pcacc_stream(stream)>>>(x1, x2, w, k, cov, m1, m2, norm);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// grad_x2 [P,k,3], grad_w [P,k] from the gradients of cov / m1 / m2 (any of them NULL = 0)
extern "C" int pcacc_kabsch_cov_backward(const float *x1, const float *x2, const float *w, const float *m1, const float *m2, const float *norm,
                                         const float *grad_cov, const float *grad_m1, const float *grad_m2, int n_pairs, int k, float *grad_x2,
                                         float *grad_w, void *stream)
{
    if (n_pairs < 1 || k < 1 || !x1 || !x2 || !w || !m1 || !m2 || !norm || !grad_x2 || !grad_w) return PCACC_E_ARG;
    kabsch_cov_bwd_kernel<<<n_pairs, 256, 0, pcacc_stream(stream)>>>(x1, x2, w, m1, m2, norm, grad_cov, grad_m1, grad_m2, k, grad_x2, grad_w);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- ... and the part behind the SVD: R = V diag(1, 1, det(V U^T)) U^T, t = m2 - R m1 (toolbox/register_utils.py:305-313), one thread per pair.
// The torch formulation runs det() through an LU factorisation (rocsolver launches, forward and backward) for a 3x3 matrix.
__device__ __forceinline__ void kb_cofactor(const double (&m)[3][3], double (&c)[3][3])
{
    c[0][0] = m[1][1] * m[2][2] - m[1][2] * m[2][1]; c[0][1] = m[1][2] * m[2][0] - m[1][0] * m[2][2]; c[0][2] = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    c[1][0] = m[0][2] * m[2][1] - m[0][1] * m[2][2]; c[1][1] = m[0][0] * m[2][2] - m[0][2] * m[2][0]; c[1][2] = m[0][1] * m[2][0] - m[0][0] * m[2][1];
    c[2][0] = m[0][1] * m[1][2] - m[0][2] * m[1][1]; c[2][1] = m[0][2] * m[1][0] - m[0][0] * m[1][2]; c[2][2] = m[0][0] * m[1][1] - m[0][1] * m[1][0];
}

__global__ __launch_bounds__(64) void kabsch_rt_fwd_kernel(const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ m1,
                                                           const float *__restrict__ m2, int n, float *__restrict__ rot, float *__restrict__ trans)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n) return;
    double U[3][3], V[3][3], M[3][3], C[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { U[i][j] = u[p * 9 + i * 3 + j]; V[i][j] = v[p * 9 + i * 3 + j]; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + V[i][2] * U[j][2];
    kb_cofactor(M, C);
    const double d = M[0][0] * C[0][0] + M[0][1] * C[0][1] + M[0][2] * C[0][2];
    for (int i = 0; i < 3; ++i) {
        double R[3];
        for (int j = 0; j < 3; ++j) {
            R[j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + d * V[i][2] * U[j][2];
            rot[p * 9 + i * 3 + j] = (float)R[j];
        }
        trans[p * 3 + i] = (float)((double)m2[p * 3 + i] - (R[0] * m1[p * 3 + 0] + R[1] * m1[p * 3 + 1] + R[2] * m1[p * 3 + 2]));
    }
}

__global__ __launch_bounds__(64) void kabsch_rt_bwd_kernel(const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ m1,
                                                           const float *__restrict__ g_rot, const float *__restrict__ g_trans, int n,
                                                           float *__restrict__ g_u, float *__restrict__ g_v, float *__restrict__ g_m1,
                                                           float *__restrict__ g_m2)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n) return;
    double U[3][3], V[3][3], M[3][3], C[3][3], R[3][3], G[3][3], gt[3], a[3];
    for (int i = 0; i < 3; ++i) {
        a[i] = m1[p * 3 + i];
        gt[i] = g_trans ? g_trans[p * 3 + i] : 0.f;
        for (int j = 0; j < 3; ++j) { U[i][j] = u[p * 9 + i * 3 + j]; V[i][j] = v[p * 9 + i * 3 + j]; }
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + V[i][2] * U[j][2];
    kb_cofactor(M, C);
    const double d = M[0][0] * C[0][0] + M[0][1] * C[0][1] + M[0][2] * C[0][2];
    const double D[3] = {1.0, 1.0, d};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            R[i][j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + d * V[i][2] * U[j][2];
            G[i][j] = (g_rot ? (double)g_rot[p * 9 + i * 3 + j] : 0.0) - gt[i] * a[j];      // t = m2 - R m1
        }
    for (int j = 0; j < 3; ++j) {
        g_m2[p * 3 + j] = (float)gt[j];
        g_m1[p * 3 + j] = (float)(-(R[0][j] * gt[0] + R[1][j] * gt[1] + R[2][j] * gt[2]));
    }
    double gd = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) gd += G[i][j] * V[i][2] * U[j][2];
    // d = det(M), M = V U^T: d(det) / dM = cofactor matrix
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            double gv = 0.0, gu = 0.0;
            for (int j = 0; j < 3; ++j) {
                gv += (G[i][j] * D[k] + gd * C[i][j]) * U[j][k];      // g_V[i][k]
                gu += (G[j][i] * D[k] + gd * C[j][i]) * V[j][k];      // g_U[i][k] (row i of U pairs with column i of G / C)
            }
            g_v[p * 9 + i * 3 + k] = (float)gv;
            g_u[p * 9 + i * 3 + k] = (float)gu;
        }
}

extern "C" int pcacc_kabsch_rt_forward(const float *u, const float *v, const float *m1, const float *m2, int n, float *rot, float *trans,
                                       void *stream)
{
    if (n < 1 || !u || !v || !m1 || !m2 || !rot || !trans) return PCACC_E_ARG;
    kabsch_rt_fwd_kernel<<<(n + 63) / 64, 64, 0, pcacc_stream(stream)>>>(u, v, m1, m2, n, rot, trans);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_kabsch_rt_backward(const float *u, const float *v, const float *m1, const float *grad_rot, const float *grad_trans, int n,
                                        float *grad_u, float *grad_v, float *grad_m1, float *grad_m2, void *stream)
{
    if (n < 1 || !u || !v || !m1 || !grad_u || !grad_v || !grad_m1 || !grad_m2) return PCACC_E_ARG;
    kabsch_rt_bwd_kernel<<<(n + 63) / 64, 64, 0, pcacc_stream(stream)>>>(u, v, m1, grad_rot, grad_trans, n, grad_u, grad_v, grad_m1, grad_m2);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

