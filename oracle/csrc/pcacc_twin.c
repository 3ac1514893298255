/*
 * pcacc_twin.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU twin of the C ABI in include/pcacc.h (last sentence): for the irregular entry points of the hot path the same
 * symbol with the prefix `orc_` instead of `pcacc_`, the same argument meaning and layouts (host pointers; no stream, no
 * workspace), so that a test can diff CPU against HIP on raw buffers without torch, and so that bench.py's `cpu_baseline` leg
 * (kind "port") runs these stages multi-threaded instead of through per-row numpy.  OpenMP over pillars / points / cells; every
 * output element is produced by exactly one thread in a fixed order, so results do not depend on the thread count.
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).  Arithmetic follows the numpy /
 * ATen code it stands for: fp32 unless stated, no FMA contraction (the file is built with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_ARG (-1)

/* A2'. models/pillar_encoder.py:153-158, 188-197: cell = ((b*nt + t)*ny + y)*nx + x from rows (b,z,y,x,t); the highest pillar id
 * wins a duplicate cell ("later writes win", :163). */
int orc_cell_index(const void *coords, int coords_is_f64, int64_t m, int nx, int ny, int nt, int n_batch, int32_t *cell,
                   int32_t *cell2pillar)
{
    const int64_t n_cells = (int64_t)n_batch * nt * ny * nx;
    if (m < 0 || nx <= 0 || ny <= 0 || nt <= 0 || n_batch <= 0 || !cell2pillar) return ORC_E_ARG;
    for (int64_t c = 0; c < n_cells; ++c) cell2pillar[c] = -1;
    for (int64_t i = 0; i < m; ++i) {
        int64_t r[5];
        for (int k = 0; k < 5; ++k)
            r[k] = coords_is_f64 ? (int64_t)((const double *)coords)[i * 5 + k] : (int64_t)((const int32_t *)coords)[i * 5 + k];
        const int64_t b = r[0], y = r[2], x = r[3], t = r[4];
        const int64_t c = ((b * nt + t) * ny + y) * nx + x;
        const int ok = c >= 0 && c < n_cells && x >= 0 && x < nx && y >= 0 && y < ny && t >= 0 && t < nt;
        cell[i] = ok ? (int32_t)c : -1;
        if (ok && (int32_t)i > cell2pillar[c]) cell2pillar[c] = (int32_t)i;
    }
    return ORC_OK;
}

/* Point -> pillar CSR: counting sort, ascending point index inside every pillar. */
int orc_csr_build(const int32_t *p2v, int64_t n, int64_t m, int32_t *seg_offsets, int32_t *order)
{
    if (n < 0 || m < 0) return ORC_E_ARG;
    for (int64_t s = 0; s <= m; ++s) seg_offsets[s] = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (p2v[i] < 0 || p2v[i] >= m) return ORC_E_ARG;
        seg_offsets[p2v[i] + 1] += 1;
    }
    for (int64_t s = 0; s < m; ++s) seg_offsets[s + 1] += seg_offsets[s];
    int32_t *cur = (int32_t *)malloc((size_t)(m > 0 ? m : 1) * sizeof(int32_t));
    memcpy(cur, seg_offsets, (size_t)m * sizeof(int32_t));
    for (int64_t i = 0; i < n; ++i) order[cur[p2v[i]]++] = (int32_t)i;
    free(cur);
    return ORC_OK;
}

/* A3. models/motionnet.py:159-160: scatter(points, p2v, 'mean') (fp32 sum in point order / count), scatter(labels, p2v, 'max'). */
int orc_segment_mean3_maxlabel(const float *points, const int64_t *labels, const int32_t *seg_offsets, const int32_t *order,
                               int64_t m, float *mean, int64_t *max_label)
{
#pragma omp parallel for schedule(static, 1024)
    for (int64_t s = 0; s < m; ++s) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        int64_t best = 0;
        const int32_t lo = seg_offsets[s], hi = seg_offsets[s + 1];
        for (int32_t j = lo; j < hi; ++j) {
            const int64_t i = order[j];
            a0 += points[i * 3]; a1 += points[i * 3 + 1]; a2 += points[i * 3 + 2];
            if (labels && (j == lo || labels[i] > best)) best = labels[i];
        }
        const float cnt = (float)(hi - lo);
        mean[s * 3] = hi > lo ? a0 / cnt : 0.f;
        mean[s * 3 + 1] = hi > lo ? a1 / cnt : 0.f;
        mean[s * 3 + 2] = hi > lo ? a2 / cnt : 0.f;
        if (labels) max_label[s] = best;
    }
    return ORC_OK;
}

/* A4 pooling. models/pillar_encoder.py:116,120: scatter(net, p2v, dim=0, reduce='max'); arg = lowest point index attaining the
 * maximum, -1 (value 0) for an empty segment. */
int orc_segment_max(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m, float *out,
                    int32_t *arg)
{
    (void)n;
#pragma omp parallel for schedule(static, 256)
    for (int64_t s = 0; s < m; ++s) {
        float *o = out + s * c;
        int32_t *a = arg + s * c;
        for (int k = 0; k < c; ++k) { o[k] = 0.f; a[k] = -1; }
        for (int32_t j = seg_offsets[s]; j < seg_offsets[s + 1]; ++j) {
            const int32_t i = order[j];
            const float *row = src + (int64_t)i * c;
            for (int k = 0; k < c; ++k)
                if (a[k] < 0 || row[k] > o[k] || (row[k] == o[k] && i < a[k])) { o[k] = row[k]; a[k] = i; }
        }
    }
    return ORC_OK;
}

int orc_segment_max_backward(const float *grad_out, const int32_t *arg, const int32_t *p2v, int64_t n, int c, float *grad_src)
{
#pragma omp parallel for schedule(static, 1024)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = p2v[i];
        for (int k = 0; k < c; ++k) grad_src[i * c + k] = arg[s * c + k] == (int32_t)i ? grad_out[s * c + k] : 0.f;
    }
    return ORC_OK;
}

/* Backward of the [point_to_voxel_map] broadcast (models/pillar_encoder.py:116): per-pillar sum in point order. */
int orc_segment_sum(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m, float *out)
{
    (void)n;
#pragma omp parallel for schedule(static, 256)
    for (int64_t s = 0; s < m; ++s) {
        float *o = out + s * c;
        for (int k = 0; k < c; ++k) o[k] = 0.f;
        for (int32_t j = seg_offsets[s]; j < seg_offsets[s + 1]; ++j) {
            const float *row = src + (int64_t)order[j] * c;
            for (int k = 0; k < c; ++k) o[k] += row[k];
        }
    }
    return ORC_OK;
}

/* A4 feature build. models/pillar_encoder.py:98-110: [xyz, xyz - pillar_mean, xy - pillar_centre, t]; the pillar centre is
 * formed in float64 from the float64 coordinate and the difference rounded to fp32 on store; fp32 divisions. */
int orc_pfn_features(const float *points, const int32_t *p2v, const float *pillar_mean, const void *coords, int coords_is_f64,
                     const double *time_col, int64_t time_stride, int64_t n, double vx, double vy, double x_offset, double y_offset,
                     float scale, float n_frames, float *out)
{
#pragma omp parallel for schedule(static, 4096)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = p2v[i];
        const float x = points[i * 3], y = points[i * 3 + 1], z = points[i * 3 + 2];
        const double cy = coords_is_f64 ? ((const double *)coords)[s * 5 + 2] : (double)((const int32_t *)coords)[s * 5 + 2];
        const double cx = coords_is_f64 ? ((const double *)coords)[s * 5 + 3] : (double)((const int32_t *)coords)[s * 5 + 3];
        float f[9];
        f[0] = x; f[1] = y; f[2] = z;
        f[3] = x - pillar_mean[s * 3]; f[4] = y - pillar_mean[s * 3 + 1]; f[5] = z - pillar_mean[s * 3 + 2];
        f[6] = (float)((double)x - (cx * vx + x_offset));
        f[7] = (float)((double)y - (cy * vy + y_offset));
        f[8] = (float)time_col[i * time_stride];
        for (int k = 0; k < 8; ++k) out[i * 9 + k] = f[k] / scale;
        out[i * 9 + 8] = f[8] / n_frames;
    }
    return ORC_OK;
}

/* A5. models/pillar_encoder.py:125-174 into a channels-last canvas: canvas[cell,:] = cell2pillar[cell] >= 0 ? feats[pillar,:] : 0. */
int orc_pillar_scatter(const float *feats, const int32_t *cell2pillar, int64_t n_cells, int c, float *canvas)
{
#pragma omp parallel for schedule(static, 4096)
    for (int64_t cell = 0; cell < n_cells; ++cell) {
        const int32_t p = cell2pillar[cell];
        if (p >= 0) memcpy(canvas + cell * c, feats + (int64_t)p * c, (size_t)c * sizeof(float));
        else memset(canvas + cell * c, 0, (size_t)c * sizeof(float));
    }
    return ORC_OK;
}

/* A6 / backward of A5: out[i,:] = src[idx[i],:] (rows of row_bytes bytes), zeros for idx < 0.
 * models/pillar_encoder.py:177-204 and the [p2v] gathers at models/motionnet.py:192-193. */
int orc_gather_rows(const void *src, int row_bytes, const int32_t *idx, int64_t n_idx, void *out)
{
#pragma omp parallel for schedule(static, 4096)
    for (int64_t i = 0; i < n_idx; ++i) {
        if (idx[i] >= 0) memcpy((char *)out + i * row_bytes, (const char *)src + (int64_t)idx[i] * row_bytes, (size_t)row_bytes);
        else memset((char *)out + i * row_bytes, 0, (size_t)row_bytes);
    }
    return ORC_OK;
}

/* A11. models/pillar_encoder.py:231-267 (ungrid) / :206-228 (temporal_ungrid): F.grid_sample(bilinear, border, align_corners=False)
 * of map `map_idx[i]` at (x / x_scale, y / y_scale), on a channels-last map [n_maps,h,w,c].  ATen grid_sampler_2d arithmetic:
 * ix = ((g + 1) * W - 1) / 2 clamped to [0, W-1], corners floor / floor + 1, weights (x1-x)(y1-y) ..., out-of-range corners dropped. */
int orc_bilinear_gather(const float *fmap, int n_maps, int h, int w, int c, const float *points, const int32_t *map_idx, int64_t k,
                        float x_scale, float y_scale, float *out)
{
    (void)n_maps;
#pragma omp parallel for schedule(static, 1024)
    for (int64_t i = 0; i < k; ++i) {
        const float gx = points[i * 3] / x_scale, gy = points[i * 3 + 1] / y_scale;
        float x = ((gx + 1.f) * (float)w - 1.f) / 2.f, y = ((gy + 1.f) * (float)h - 1.f) / 2.f;
        x = fminf(fmaxf(x, 0.f), (float)(w - 1));
        y = fminf(fmaxf(y, 0.f), (float)(h - 1));
        const float x0 = floorf(x), y0 = floorf(y), x1 = x0 + 1.f, y1 = y0 + 1.f;
        const float wt[4] = {(x1 - x) * (y1 - y), (x - x0) * (y1 - y), (x1 - x) * (y - y0), (x - x0) * (y - y0)};
        const float xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
        float *o = out + i * c;
        for (int q = 0; q < c; ++q) o[q] = 0.f;
        const float *base = fmap + (int64_t)map_idx[i] * h * w * c;
        for (int t = 0; t < 4; ++t) {
            if (!(xs[t] >= 0.f && xs[t] <= (float)(w - 1) && ys[t] >= 0.f && ys[t] <= (float)(h - 1))) continue;
            const float *row = base + ((int64_t)ys[t] * w + (int64_t)xs[t]) * c;
            for (int q = 0; q < c; ++q) o[q] += wt[t] * row[q];
        }
    }
    return ORC_OK;
}

/* A9. models/motionnet.py:45-114: frames 1..T-1 resampled (bilinear, zeros padding, align_corners=False) on the grid
 * inv_pose[:2,:2] @ (pixel centre in metres) + inv_pose[:2,3], normalised by |min|; slot 0 = frame T-1 un-warped (the loop variable
 * leaks, :100,111).  bev / out [B,T,H,W,C] channels-last f32; inv_pose [B,T,4,4]. */
int orc_bev_warp(const float *bev, const float *inv_pose, int n_batch, int nt, int h, int w, int c, float x_reso, float y_reso,
                 float x_min, float y_min, float *out)
{
    const int64_t plane = (int64_t)h * w * c;
    for (int b = 0; b < n_batch; ++b) {
        memcpy(out + ((int64_t)b * nt) * plane, bev + ((int64_t)b * nt + nt - 1) * plane, (size_t)plane * sizeof(float));
        for (int t = 1; t < nt; ++t) {
            const float *p = inv_pose + ((int64_t)b * nt + t) * 16;
            const float *src = bev + ((int64_t)b * nt + t) * plane;
            float *dst = out + ((int64_t)b * nt + t) * plane;
#pragma omp parallel for schedule(static, 16)
            for (int yy = 0; yy < h; ++yy)
                for (int xx = 0; xx < w; ++xx) {
                    const float gx = ((float)xx + 0.5f) * x_reso + x_min, gy = ((float)yy + 0.5f) * y_reso + y_min;
                    const float tx = (p[0] * gx + p[1] * gy + p[3]) / fabsf(x_min);
                    const float ty = (p[4] * gx + p[5] * gy + p[7]) / fabsf(y_min);
                    const float x = ((tx + 1.f) * (float)w - 1.f) / 2.f, y = ((ty + 1.f) * (float)h - 1.f) / 2.f;
                    const float x0 = floorf(x), y0 = floorf(y), x1 = x0 + 1.f, y1 = y0 + 1.f;
                    const float wt[4] = {(x1 - x) * (y1 - y), (x - x0) * (y1 - y), (x1 - x) * (y - y0), (x - x0) * (y - y0)};
                    const float xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
                    float *o = dst + ((int64_t)yy * w + xx) * c;
                    for (int q = 0; q < c; ++q) o[q] = 0.f;
                    for (int k = 0; k < 4; ++k) {
                        if (!(xs[k] >= 0.f && xs[k] <= (float)(w - 1) && ys[k] >= 0.f && ys[k] <= (float)(h - 1))) continue;
                        const float *row = src + ((int64_t)ys[k] * w + (int64_t)xs[k]) * c;
                        for (int q = 0; q < c; ++q) o[q] += wt[k] * row[q];
                    }
                }
        }
    }
    return ORC_OK;
}

/* A9 / A13. models/motionnet.py:117-135, toolbox/register_utils.py:59-93: p' = R[idx] p + t[idx], un-fused fp32, products summed
 * left to right.  tsfm [n_tsfm,16] row-major 4x4. */
int orc_rigid_transform(const float *points, const int32_t *frame_idx, const float *tsfm, int64_t n, float *out)
{
#pragma omp parallel for schedule(static, 4096)
    for (int64_t i = 0; i < n; ++i) {
        const float *m = tsfm + (int64_t)frame_idx[i] * 16;
        const float x = points[i * 3], y = points[i * 3 + 1], z = points[i * 3 + 2];
        out[i * 3] = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
        out[i * 3 + 1] = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
        out[i * 3 + 2] = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
    }
    return ORC_OK;
}

/* A9 frame max. models/stpn.py:83: torch.max(x, dim=2) over the T frames of [S,T,P] rows; lowest frame wins ties. */
int orc_frames_max(const float *x, int64_t n_seq, int32_t frames, int64_t plane, float *out, uint8_t *arg)
{
#pragma omp parallel for schedule(static, 1)
    for (int64_t s = 0; s < n_seq; ++s)
        for (int64_t p = 0; p < plane; ++p) {
            float best = x[(s * frames) * plane + p];
            uint8_t who = 0;
            for (int t = 1; t < frames; ++t) {
                const float v = x[(s * frames + t) * plane + p];
                if (v > best || (v != v && best == best)) { best = v; who = (uint8_t)t; }
            }
            out[s * plane + p] = best;
            arg[s * plane + p] = who;
        }
    return ORC_OK;
}
