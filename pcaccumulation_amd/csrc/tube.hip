// TubeNet slot algebra: the per-(instance, frame) "slot" arithmetic of models/tpointnet.py:249-305 and of the refinement loop of
// models/alignnet.py:236-263 as a handful of kernels instead of ~250 element-wise launches per iteration and direction.
//
// A slot s = k * T + t holds one 7-vector (quaternion xyzw + translation) regressed for instance k in frame t.  Per iteration:
//   tube_rows       point rows of the positional embedding: (xyz - centre of the instance's anchor frame, t / T)
//   tube_code       [geometry | motion | frame | anchor-frame] code rows of the regressor (gather; backward = per-instance sums)
//   tube_pose_fwd   quaternion -> pose, ground truth re-expressed for the centred cloud (with its quaternion, scipy's branch
//                   scheme, in f64), rotation / translation losses, un-centred pose with frame 0 pinned, the loop's
//                   "remaining" and "total" updates
//   tube_gap_fwd    per point |est - gt|_2 and |est - gt|_1 on the centred cloud   (-> per-slot means by the segment kernels)
//   tube_fin        weighted mean of the slot means
//   tube_gap_bwd    per point d(loss)/d(pose of its slot), 12 numbers              (-> per-slot sums by the segment kernels)
//   tube_pose_bwd   chain through quaternion -> matrix and the normalisation, plus the rotation / translation loss terms
// Slot-level kernels run in ONE workgroup (K*T is a few hundred to a few thousand; the reductions stay deterministic).
#include "common.h"

#define TUBE_BLOCK 1024

__device__ __forceinline__ float tube_block_sum_f32(float v, float *lds)
{
    const int tid = threadIdx.x;
    lds[tid] = v;
    __syncthreads();
    for (int s = TUBE_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) lds[tid] += lds[tid + s];
        __syncthreads();
    }
    const float r = lds[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double tube_block_sum_f64(double v, double *lds)
{
    const int tid = threadIdx.x;
    lds[tid] = v;
    __syncthreads();
    for (int s = TUBE_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) lds[tid] += lds[tid + s];
        __syncthreads();
    }
    const double r = lds[0];
    __syncthreads();
    return r;
}

// rotation of a (unit) quaternion (x, y, z, w): toolbox/se3_utils.py:44-64
__device__ __forceinline__ void tube_quat_rot(float x, float y, float z, float w, float *r)
{
    r[0] = w * w + x * x - y * y - z * z;  r[1] = 2.f * (x * y - w * z);            r[2] = 2.f * (w * y + x * z);
    r[3] = 2.f * (w * z + x * y);          r[4] = w * w - x * x + y * y - z * z;    r[5] = 2.f * (y * z - w * x);
    r[6] = 2.f * (x * z - w * y);          r[7] = 2.f * (w * x + y * z);            r[8] = w * w - x * x - y * y + z * z;
}

// scipy's Rotation.from_matrix(m).as_quat() branch scheme (models/tpointnet.py:62-66), f64, normalised
__device__ __forceinline__ void tube_mat2quat(const double *m, double *q)
{
    const double tr = m[0] + m[4] + m[8];
    int choice = 0;
    double best = m[0];
    if (m[4] > best) { best = m[4]; choice = 1; }
    if (m[8] > best) { best = m[8]; choice = 2; }
    if (tr > best) { choice = 3; }
    if (choice == 3) {
        q[0] = m[7] - m[5];
        q[1] = m[2] - m[6];
        q[2] = m[3] - m[1];
        q[3] = 1.0 + tr;
    } else {
        const int i = choice, j = (i + 1) % 3, k = (i + 2) % 3;
        q[i] = 1.0 - tr + 2.0 * m[i * 3 + i];
        q[j] = m[j * 3 + i] + m[i * 3 + j];
        q[k] = m[k * 3 + i] + m[i * 3 + k];
        q[3] = m[k * 3 + j] - m[j * 3 + k];
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ void tube_rows_kernel(const float *__restrict__ xyz, const int *__restrict__ slot, const float *__restrict__ slot_centre,
                                 int64_t n, int T, float *__restrict__ rows)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int s = slot[p];
    const int t = s % T;
    const float *c = slot_centre + (int64_t)(s - t) * 3;                     // centroid of the instance in frame 0
    float4 r;
    r.x = xyz[p * 3 + 0] - c[0];
    r.y = xyz[p * 3 + 1] - c[1];
    r.z = xyz[p * 3 + 2] - c[2];
    r.w = (float)t / (float)T;
    reinterpret_cast<float4 *>(rows)[p] = r;
}

// code[s] = (geo[k], motion[k], frame[s], frame[k*T]); c4 = channels / 4
__global__ void tube_code_kernel(const float4 *__restrict__ geo, const float4 *__restrict__ motion, const float4 *__restrict__ frame,
                                 int64_t n_slots, int T, int c4, float4 *__restrict__ code)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots * 4 * c4) return;
    const int64_t s = i / (4 * c4);
    const int j = (int)(i % (4 * c4));
    const int part = j / c4, c = j % c4;
    const int64_t k = s / T;
    float4 v;
    if (part == 0) v = geo[k * c4 + c];
    else if (part == 1) v = motion[k * c4 + c];
    else if (part == 2) v = frame[s * c4 + c];
    else v = frame[k * T * c4 + c];
    code[i] = v;
}

// one thread per (instance, channel quad): sums over the T slots of the instance
__global__ void tube_code_bwd_kernel(const float4 *__restrict__ g_code, int64_t n_inst, int T, int c4, float4 *__restrict__ g_geo,
                                     float4 *__restrict__ g_motion, float4 *__restrict__ g_frame)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inst * c4) return;
    const int64_t k = i / c4;
    const int c = (int)(i % c4);
    float4 sg = make_float4(0, 0, 0, 0), sm = sg, sa = sg;
    for (int t = 0; t < T; ++t) {
        const float4 *row = g_code + (k * T + t) * 4 * c4;
        const float4 a = row[c], b = row[c4 + c], d = row[3 * c4 + c];
        sg.x += a.x; sg.y += a.y; sg.z += a.z; sg.w += a.w;
        sm.x += b.x; sm.y += b.y; sm.z += b.z; sm.w += b.w;
        sa.x += d.x; sa.y += d.y; sa.z += d.z; sa.w += d.w;
    }
    g_geo[i] = sg;
    g_motion[i] = sm;
    for (int t = 0; t < T; ++t) {
        float4 f = g_code[(k * T + t) * 4 * c4 + 2 * c4 + c];
        if (t == 0) { f.x += sa.x; f.y += sa.y; f.z += sa.z; f.w += sa.w; }
        g_frame[(k * T + t) * c4 + c] = f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct TubeSlot {
    float qh[4];        // normalised quaternion
    float inv_n;        // 1 / max(|q|, 1e-12)
    bool clamped;       // |q| < 1e-12
    float r[9], t[3];   // estimated pose for the centred cloud
    float rg[9], tg[3]; // ground truth for the centred cloud
    double dq[4], dt[3], rn, tn;   // gt - est of the 7-vector, their norms
};

__device__ __forceinline__ void tube_slot_terms(const float *pose_vec, const float *gt, const float *centre, TubeSlot &o)
{
    const float x = pose_vec[0], y = pose_vec[1], z = pose_vec[2], w = pose_vec[3];
    const float n = sqrtf(x * x + y * y + z * z + w * w);
    o.clamped = !(n >= 1e-12f);
    o.inv_n = 1.f / fmaxf(n, 1e-12f);                                      // F.normalize(p=2, eps=1e-12)
    o.qh[0] = x * o.inv_n; o.qh[1] = y * o.inv_n; o.qh[2] = z * o.inv_n; o.qh[3] = w * o.inv_n;
    tube_quat_rot(o.qh[0], o.qh[1], o.qh[2], o.qh[3], o.r);
    o.t[0] = pose_vec[4]; o.t[1] = pose_vec[5]; o.t[2] = pose_vec[6];
    // ground truth re-expressed for the cloud centred on `centre`: t += (R - I) c     (models/tpointnet.py:52-57)
    double m[9], q[4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            o.rg[a * 3 + b] = gt[a * 4 + b];
            m[a * 3 + b] = (double)gt[a * 4 + b];
        }
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += (gt[a * 4 + b] - (a == b ? 1.f : 0.f)) * centre[b];
        o.tg[a] = gt[a * 4 + 3] + acc;
    }
    tube_mat2quat(m, q);
    double rs = 0, ts = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) { o.dq[a] = q[a] - (double)o.qh[a]; rs += o.dq[a] * o.dq[a]; }
#pragma unroll
    for (int a = 0; a < 3; ++a) { o.dt[a] = (double)o.tg[a] - (double)o.t[a]; ts += o.dt[a] * o.dt[a]; }
    o.rn = sqrt(rs);
    o.tn = sqrt(ts);
}

// One workgroup.  pose_c / gt_c [S,12] = (R row-major, t) for the centred cloud; step [S,16] the un-centred pose with frame 0 pinned
// to the identity (models/tpointnet.py:291-296); rem_out = rem_in @ step^-1, total_out = step @ total_in (models/alignnet.py:257-263,
// total_in NULL = identity); loss_rt [2] f64 = weighted means of |dq| and |dt| (models/tpointnet.py:76-94); wsum [1] = sum(w) + 1e-20.
__global__ void __launch_bounds__(TUBE_BLOCK)
tube_pose_fwd_kernel(const float *__restrict__ pose_vec, const float *__restrict__ rem_in, const float *__restrict__ total_in,
                     const float *__restrict__ slot_centre, const float *__restrict__ weights, int n_slots, int T,
                     float *__restrict__ pose_c, float *__restrict__ gt_c, float *__restrict__ step, float *__restrict__ rem_out,
                     float *__restrict__ total_out, double *__restrict__ loss_rt, float *__restrict__ wsum_out)
{
    __shared__ double lds[TUBE_BLOCK];
    float wpart = 0.f;
    for (int s = threadIdx.x; s < n_slots; s += TUBE_BLOCK) wpart += weights[s];
    const float wsum = tube_block_sum_f32(wpart, reinterpret_cast<float *>(lds)) + 1e-20f;
    double rpart = 0, tpart = 0;
    for (int s = threadIdx.x; s < n_slots; s += TUBE_BLOCK) {
        const int t = s % T;
        const float *c = slot_centre + (int64_t)(s - t) * 3;
        const float *g = rem_in + (int64_t)s * 16;
        TubeSlot o;
        tube_slot_terms(pose_vec + (int64_t)s * 7, g, c, o);
        const double w = (double)weights[s];
        rpart += o.rn * w;
        tpart += o.tn * w;
        float *pc = pose_c + (int64_t)s * 12, *gc = gt_c + (int64_t)s * 12;
#pragma unroll
        for (int a = 0; a < 9; ++a) { pc[a] = o.r[a]; gc[a] = o.rg[a]; }
#pragma unroll
        for (int a = 0; a < 3; ++a) { pc[9 + a] = o.t[a]; gc[9 + a] = o.tg[a]; }
        // un-centre: t += (I - R) c; frame 0 = identity
        float sr[9], st[3];
        if (t == 0) {
#pragma unroll
            for (int a = 0; a < 9; ++a) sr[a] = (a % 4 == 0) ? 1.f : 0.f;
            st[0] = st[1] = st[2] = 0.f;
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float acc = 0.f;
#pragma unroll
                for (int b = 0; b < 3; ++b) { sr[a * 3 + b] = o.r[a * 3 + b]; acc += ((a == b ? 1.f : 0.f) - o.r[a * 3 + b]) * c[b]; }
                st[a] = o.t[a] + acc;
            }
        }
        float *so = step + (int64_t)s * 16;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) so[a * 4 + b] = sr[a * 3 + b];
            so[a * 4 + 3] = st[a];
        }
        so[12] = 0.f; so[13] = 0.f; so[14] = 0.f; so[15] = 1.f;
        // remaining <- remaining @ step^-1:  R' = Rg Rs^T,  t' = tg - R' ts
        float *ro = rem_out + (int64_t)s * 16;
        float rn[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc += g[a * 4 + k] * sr[b * 3 + k];
                rn[a * 3 + b] = acc;
            }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float acc = 0.f;
#pragma unroll
            for (int b = 0; b < 3; ++b) { ro[a * 4 + b] = rn[a * 3 + b]; acc += rn[a * 3 + b] * st[b]; }
            ro[a * 4 + 3] = g[a * 4 + 3] - acc;
        }
        ro[12] = g[12]; ro[13] = g[13]; ro[14] = g[14]; ro[15] = g[15];
        // total <- step @ total
        float *to = total_out + (int64_t)s * 16;
        if (total_in == nullptr) {
#pragma unroll
            for (int a = 0; a < 16; ++a) to[a] = so[a];
        } else {
            const float *ti = total_in + (int64_t)s * 16;
            float tv[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) tv[a] = ti[a];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sv = (a < 3) ? (k < 3 ? sr[a * 3 + k] : st[a]) : (k == 3 ? 1.f : 0.f);
                        acc += sv * tv[k * 4 + b];
                    }
                    to[a * 4 + b] = acc;
                }
        }
    }
    const double rsum = tube_block_sum_f64(rpart, lds);
    const double tsum = tube_block_sum_f64(tpart, lds);
    if (threadIdx.x == 0) {
        loss_rt[0] = rsum / (double)wsum;
        loss_rt[1] = tsum / (double)wsum;
        wsum_out[0] = wsum;
    }
}

// per point: pp[p] = (|gap|_2, |gap|_1, 0, 0), gap = (R_e l + t_e) - (R_g l + t_g) with l = rows[p].xyz
__global__ void tube_gap_fwd_kernel(const float4 *__restrict__ rows, const int *__restrict__ slot, const float *__restrict__ pose_c,
                                    const float *__restrict__ gt_c, int64_t n, float4 *__restrict__ pp)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 l = rows[p];
    const float *e = pose_c + (int64_t)slot[p] * 12, *g = gt_c + (int64_t)slot[p] * 12;
    float n2 = 0.f, n1 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float est = e[a * 3] * l.x + e[a * 3 + 1] * l.y + e[a * 3 + 2] * l.z + e[9 + a];
        const float gtv = g[a * 3] * l.x + g[a * 3 + 1] * l.y + g[a * 3 + 2] * l.z + g[9 + a];
        const float d = est - gtv;
        n2 += d * d;
        n1 += fabsf(d);
    }
    pp[p] = make_float4(sqrtf(n2), n1, 0.f, 0.f);
}

// l12[j] = sum_s w[s] * sums[s][j] / max(cnt[s], 1) / wsum      (one workgroup)
__global__ void __launch_bounds__(TUBE_BLOCK)
tube_fin_kernel(const float *__restrict__ sums, int stride, const float *__restrict__ cnt, const float *__restrict__ weights,
                const float *__restrict__ wsum, int n_slots, float *__restrict__ l12)
{
    __shared__ float lds[TUBE_BLOCK];
    float a = 0.f, b = 0.f;
    for (int s = threadIdx.x; s < n_slots; s += TUBE_BLOCK) {
        const float inv = weights[s] / fmaxf(cnt[s], 1.f);
        a += sums[(int64_t)s * stride] * inv;
        b += sums[(int64_t)s * stride + 1] * inv;
    }
    const float sa = tube_block_sum_f32(a, lds);
    const float sb = tube_block_sum_f32(b, lds);
    if (threadIdx.x == 0) {
        l12[0] = sa / wsum[0];
        l12[1] = sb / wsum[0];
    }
}

// per point: g[p] = d(l1 g_l1 + l2 g_l2) / d(pose_c of its slot): 9 (R, row-major) + 3 (t) + 4 zeros
__global__ void tube_gap_bwd_kernel(const float4 *__restrict__ rows, const int *__restrict__ slot, const float *__restrict__ pose_c,
                                    const float *__restrict__ gt_c, const float *__restrict__ weights, const float *__restrict__ cnt,
                                    const float *__restrict__ wsum, const float *__restrict__ g_l1, const float *__restrict__ g_l2, int64_t n,
                                    float4 *__restrict__ g16)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int s = slot[p];
    const float4 l = rows[p];
    const float *e = pose_c + (int64_t)s * 12, *g = gt_c + (int64_t)s * 12;
    float d[3], n2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float est = e[a * 3] * l.x + e[a * 3 + 1] * l.y + e[a * 3 + 2] * l.z + e[9 + a];
        const float gtv = g[a * 3] * l.x + g[a * 3 + 1] * l.y + g[a * 3 + 2] * l.z + g[9 + a];
        d[a] = est - gtv;
        n2 += d[a] * d[a];
    }
    const float base = weights[s] / (fmaxf(cnt[s], 1.f) * wsum[0]);
    const float c2 = (g_l1 ? g_l1[0] : 0.f) * base, c1 = (g_l2 ? g_l2[0] : 0.f) * base;     // l1 = mean |gap|_2, l2 = mean |gap|_1
    const float nrm = sqrtf(n2);
    float ge[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float sgn = d[a] > 0.f ? 1.f : (d[a] < 0.f ? -1.f : 0.f);
        ge[a] = (nrm > 0.f ? c2 * d[a] / nrm : 0.f) + c1 * sgn;            // torch: norm backward is 0 at 0
    }
    float4 *o = g16 + p * 4;
    o[0] = make_float4(ge[0] * l.x, ge[0] * l.y, ge[0] * l.z, ge[1] * l.x);
    o[1] = make_float4(ge[1] * l.y, ge[1] * l.z, ge[2] * l.x, ge[2] * l.y);
    o[2] = make_float4(ge[2] * l.z, ge[0], ge[1], ge[2]);
    o[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// per slot: gradient of the 7-vector from the gradient of its centred pose (g_pose [S, stride], first 12 used) and from the
// rotation / translation loss terms (g_rot, g_trans: f64 gradients of the two weighted means, NULL = 0)
__global__ void tube_pose_bwd_kernel(const float *__restrict__ pose_vec, const float *__restrict__ rem_in, const float *__restrict__ slot_centre,
                                     const float *__restrict__ weights, const float *__restrict__ wsum, const float *__restrict__ g_pose,
                                     int stride, const double *__restrict__ g_rot, const double *__restrict__ g_trans, int n_slots, int T,
                                     float *__restrict__ g_vec)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int t = s % T;
    TubeSlot o;
    tube_slot_terms(pose_vec + (int64_t)s * 7, rem_in + (int64_t)s * 16, slot_centre + (int64_t)(s - t) * 3, o);
    const float *G = g_pose + (int64_t)s * stride;
    const double wn = (double)weights[s] / (double)wsum[0];
    const float x = o.qh[0], y = o.qh[1], z = o.qh[2], w = o.qh[3];
    double gq[4];
    gq[0] = 2.0 * (double)(x * G[0] + y * G[1] + z * G[2] + y * G[3] - x * G[4] - w * G[5] + z * G[6] + w * G[7] - x * G[8]);
    gq[1] = 2.0 * (double)(-y * G[0] + x * G[1] + w * G[2] + x * G[3] + y * G[4] + z * G[5] - w * G[6] + z * G[7] - y * G[8]);
    gq[2] = 2.0 * (double)(-z * G[0] - w * G[1] + x * G[2] + w * G[3] - z * G[4] + y * G[5] + x * G[6] + y * G[7] + z * G[8]);
    gq[3] = 2.0 * (double)(w * G[0] - z * G[1] + y * G[2] + z * G[3] + w * G[4] - x * G[5] - y * G[6] + x * G[7] + w * G[8]);
    if (g_rot && o.rn > 0.0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) gq[a] -= g_rot[0] * wn * o.dq[a] / o.rn;
    }
    float *out = g_vec + (int64_t)s * 7;
    // through q / max(|q|, eps)
    if (o.clamped) {
#pragma unroll
        for (int a = 0; a < 4; ++a) out[a] = (float)(gq[a] * (double)o.inv_n);
    } else {
        const double dot = gq[0] * x + gq[1] * y + gq[2] * z + gq[3] * w;
#pragma unroll
        for (int a = 0; a < 4; ++a) out[a] = (float)((gq[a] - (double)o.qh[a] * dot) * (double)o.inv_n);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double gt = (double)G[9 + a];
        if (g_trans && o.tn > 0.0) gt -= g_trans[0] * wn * o.dt[a] / o.tn;
        out[4 + a] = (float)gt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int pcacc_tube_rows(const float *xyz, const int32_t *slot, const float *slot_centre, int64_t n, int32_t n_frames, float *rows,
                               void *stream)
{
    if (n < 0 || n_frames <= 0) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (!xyz || !slot || !slot_centre || !rows) return PCACC_E_ARG;
    tube_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, pcacc_stream(stream)>>>(xyz, slot, slot_centre, n, n_frames, rows);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_tube_code(const float *geo, const float *motion, const float *frame, int64_t n_inst, int32_t n_frames, int32_t c,
                               float *code, void *stream)
{
    if (n_inst < 0 || n_frames <= 0 || c <= 0 || c % 4) return PCACC_E_ARG;
    if (n_inst == 0) return PCACC_OK;
    if (!geo || !motion || !frame || !code) return PCACC_E_ARG;
    const int64_t total = n_inst * n_frames * c;                             // float4 elements of the [S, 4c] output
    tube_code_kernel<<<(unsigned)((total + 255) / 256), 256, 0, pcacc_stream(stream)>>>(
        reinterpret_cast<const float4 *>(geo), reinterpret_cast<const float4 *>(motion), reinterpret_cast<const float4 *>(frame),
        n_inst * n_frames, n_frames, c / 4, reinterpret_cast<float4 *>(code));
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_tube_code_backward(const float *grad_code, int64_t n_inst, int32_t n_frames, int32_t c, float *grad_geo,
                                        float *grad_motion, float *grad_frame, void *stream)
{
    if (n_inst < 0 || n_frames <= 0 || c <= 0 || c % 4) return PCACC_E_ARG;
    if (n_inst == 0) return PCACC_OK;
    if (!grad_code || !grad_geo || !grad_motion || !grad_frame) return PCACC_E_ARG;
    const int64_t total = n_inst * (c / 4);
    tube_code_bwd_kernel<<<(unsigned)((total + 127) / 128), 128, 0, pcacc_stream(stream)>>>(
        reinterpret_cast<const float4 *>(grad_code), n_inst, n_frames, c / 4, reinterpret_cast<float4 *>(grad_geo),
        reinterpret_cast<float4 *>(grad_motion), reinterpret_cast<float4 *>(grad_frame));
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_tube_pose_forward(const float *pose_vec, const float *remaining, const float *total_in, const float *slot_centre,
                                       const float *weights, int32_t n_slots, int32_t n_frames, float *pose_c, float *gt_c, float *step,
                                       float *remaining_out, float *total_out, double *loss_rt, float *wsum, void *stream)
{
    if (n_slots < 0 || n_frames <= 0 || (n_slots % n_frames)) return PCACC_E_ARG;
    if (!loss_rt || !wsum) return PCACC_E_ARG;
    if (n_slots > 0 && (!pose_vec || !remaining || !slot_centre || !weights || !pose_c || !gt_c || !step || !remaining_out || !total_out))
        return PCACC_E_ARG;
    tube_pose_fwd_kernel<<<1, TUBE_BLOCK, 0, pcacc_stream(stream)>>>(pose_vec, remaining, total_in, slot_centre, weights, n_slots, n_frames,
                                                                    pose_c, gt_c, step, remaining_out, total_out, loss_rt, wsum);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_tube_gap_forward(const float *rows, const int32_t *slot, const float *pose_c, const float *gt_c, int64_t n, float *pp,
                                      void *stream)
{
    if (n < 0) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (!rows || !slot || !pose_c || !gt_c || !pp) return PCACC_E_ARG;
    tube_gap_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, pcacc_stream(stream)>>>(reinterpret_cast<const float4 *>(rows), slot, pose_c,
                                                                                      gt_c, n, reinterpret_cast<float4 *>(pp));
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_tube_finish(const float *slot_sums, int32_t stride, const float *count, const float *weights, const float *wsum,
                                 int32_t n_slots, float *l12, void *stream)
{
    if (n_slots < 0 || stride < 2 || !wsum || !l12) return PCACC_E_ARG;
    if (n_slots > 0 && (!slot_sums || !count || !weights)) return PCACC_E_ARG;
    tube_fin_kernel<<<1, TUBE_BLOCK, 0, pcacc_stream(stream)>>>(slot_sums, stride, count, weights, wsum, n_slots, l12);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_tube_gap_backward(const float *rows, const int32_t *slot, const float *pose_c, const float *gt_c, const float *weights,
                                       const float *count, const float *wsum, const float *grad_l1, const float *grad_l2, int64_t n,
                                       float *grad_rows16, void *stream)
{
    if (n < 0) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (!rows || !slot || !pose_c || !gt_c || !weights || !count || !wsum || !grad_rows16) return PCACC_E_ARG;
    tube_gap_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, pcacc_stream(stream)>>>(
        reinterpret_cast<const float4 *>(rows), slot, pose_c, gt_c, weights, count, wsum, grad_l1, grad_l2, n, reinterpret_cast<float4 *>(grad_rows16));
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_tube_pose_backward(const float *pose_vec, const float *remaining, const float *slot_centre, const float *weights,
                                        const float *wsum, const float *grad_pose, int32_t stride, const double *grad_rot,
                                        const double *grad_trans, int32_t n_slots, int32_t n_frames, float *grad_vec, void *stream)
{
    if (n_slots < 0 || n_frames <= 0 || stride < 12) return PCACC_E_ARG;
    if (n_slots == 0) return PCACC_OK;
    if (!pose_vec || !remaining || !slot_centre || !weights || !wsum || !grad_pose || !grad_vec) return PCACC_E_ARG;
    tube_pose_bwd_kernel<<<(unsigned)((n_slots + 63) / 64), 64, 0, pcacc_stream(stream)>>>(pose_vec, remaining, slot_centre, weights, wsum,
                                                                                          grad_pose, stride, grad_rot, grad_trans, n_slots, n_frames, grad_vec);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Batched inverse of 4x4 matrices (torch.linalg.inv of the pose tables: models/motionnet.py:100, models/alignnet.py:33), one lane
// per matrix: Gauss-Jordan with partial pivoting in registers.  The library call is a dozen launches (LU factorisation, pivot
// swaps, triangular solves) for a [B*T] batch of 20.
__global__ void inv4x4_kernel(const float *__restrict__ m, int64_t n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[r][c] = m[i * 16 + r * 4 + c]; a[r][4 + c] = r == c ? 1.f : 0.f; }
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        float best = fabsf(a[col][col]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r > col && fabsf(a[r][col]) > best) { best = fabsf(a[r][col]); piv = r; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r > col && r == piv) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float t = a[col][c]; a[col][c] = a[r][c]; a[r][c] = t; }
            }
        const float inv = 1.f / a[col][col];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[col][c] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r != col) {
                const float f = a[r][col];
#pragma unroll
                for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[i * 16 + r * 4 + c] = a[r][4 + c];
}

extern "C" int pcacc_inv4x4(const float *m, int64_t n, float *out, void *stream)
{
    if (n < 0) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (!m || !out) return PCACC_E_ARG;
    inv4x4_kernel<<<(unsigned)((n + 63) / 64), 64, 0, pcacc_stream(stream)>>>(m, n, out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
