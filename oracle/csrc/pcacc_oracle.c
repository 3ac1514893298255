/*
 * pcacc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded restatement of the loop-shaped pieces of the PCAccumulation hot path.
 * It is the checker the HIP kernels are compared against (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py); nothing under pcaccumulation_amd/ may link or call it.
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 * Build: `make -C oracle` (gcc -O2 -ffp-contract=off: no FMA contraction, so fp32 expressions round
 * exactly like the numpy / scalar-C++ reference code).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * 4-D pillar voxelisation, first-touch numbering.
 * Restates libs/voxel_generator.py:4-61 (_points_to_voxel_reverse_kernel):
 *   c_j = floor((p_j - range_j) / voxel_size_j) in fp32 for j = x,y,z; the point is dropped when any
 *   c_j is outside [0, grid_j); coor = (z, y, x, t) with t = int(p[3]) (truncation); a cell gets the
 *   next free pillar id the first time a point touches it, unless max_voxels ids are already used.
 * table: int32[nz*ny*nx*nt] pre-filled with -1 (libs/voxel_generator.py:99).
 * Returns the number of pillars; coors[m*4] = (z,y,x,t); p2v[i] = pillar id or -1.
 * Deviation: t outside [0, nt) drops the point (the reference would index out of bounds / wrap).
 * ------------------------------------------------------------------------------------------- */
int orc_voxelize(const float *points, int64_t n, const float *voxel_size, const float *range,
                 const int32_t *grid /* nx,ny,nz */, int32_t nt, int32_t max_voxels,
                 int32_t *table, int32_t *coors, int32_t *num_points_per_voxel, int32_t *p2v)
{
    int32_t voxel_num = 0;
    const int32_t nx = grid[0], ny = grid[1];
    for (int64_t i = 0; i < n; ++i) {
        p2v[i] = -1;
        int32_t c3[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            volatile float d = points[i * 4 + j] - range[j];
            volatile float q = d / voxel_size[j];
            float c = floorf(q);
            if (c < 0.0f || c >= (float)grid[j]) { failed = 1; break; }
            c3[j] = (int32_t)c;
        }
        if (failed) continue;
        int32_t t = (int32_t)points[i * 4 + 3];
        if (t < 0 || t >= nt) continue;
        const int32_t z = c3[2], y = c3[1], x = c3[0];
        const int64_t cell = (((int64_t)z * ny + y) * nx + x) * nt + t;
        int32_t vid = table[cell];
        if (vid == -1) {
            vid = voxel_num;
            if (voxel_num >= max_voxels) continue;
            voxel_num += 1;
            table[cell] = vid;
            coors[vid * 4 + 0] = z; coors[vid * 4 + 1] = y; coors[vid * 4 + 2] = x; coors[vid * 4 + 3] = t;
        }
        num_points_per_voxel[vid] += 1;
        p2v[i] = vid;
    }
    return voxel_num;
}

/* ---------------------------------------------------------------------------------------------
 * Per-pillar mean of xyz and per-pillar max of an integer label.
 * Restates models/motionnet.py:159-160: scatter(points, p2v, 'mean'), scatter(labels, p2v, 'max')
 * (torch_scatter semantics: sum in fp32, divide by the count; empty segments stay 0).
 * ------------------------------------------------------------------------------------------- */
void orc_segment_mean_f32(const float *src, const int64_t *seg, int64_t n, int64_t m, int c, float *out)
{
    float *cnt = (float *)calloc((size_t)m, sizeof(float));
    memset(out, 0, (size_t)m * c * sizeof(float));
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = seg[i];
        for (int k = 0; k < c; ++k) out[s * c + k] += src[i * c + k];
        cnt[s] += 1.0f;
    }
    for (int64_t s = 0; s < m; ++s)
        if (cnt[s] > 0.0f)
            for (int k = 0; k < c; ++k) out[s * c + k] /= cnt[s];
    free(cnt);
}

void orc_segment_max_i64(const int64_t *src, const int64_t *seg, int64_t n, int64_t m, int64_t *out)
{
    uint8_t *seen = (uint8_t *)calloc((size_t)m, 1);
    memset(out, 0, (size_t)m * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = seg[i];
        if (!seen[s] || src[i] > out[s]) { out[s] = src[i]; seen[s] = 1; }
    }
    free(seen);
}

/* Restates the PFN pooling models/pillar_encoder.py:116,120: scatter(net, p2v, dim=0, reduce='max').
 * arg[s*c+k] = lowest point index attaining the maximum (the element the gradient is routed to),
 * -1 for an empty segment (whose value is 0). */
void orc_segment_max_f32(const float *src, const int64_t *seg, int64_t n, int64_t m, int c,
                         float *out, int32_t *arg)
{
    for (int64_t j = 0; j < m * c; ++j) { out[j] = 0.0f; arg[j] = -1; }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = seg[i];
        for (int k = 0; k < c; ++k) {
            const float v = src[i * c + k];
            if (arg[s * c + k] < 0 || v > out[s * c + k]) { out[s * c + k] = v; arg[s * c + k] = (int32_t)i; }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Chamfer nearest neighbour.  Restates chamfer_distance/chamfer_distance.cpp:59-87 (nnsearch):
 * squared distance accumulated in fp32 as (x*x + y*y) + z*z with x = target - query, widened to
 * double for the compare, strict '<' so the lowest index wins among equal minima; and
 * chamfer_distance.cpp:114-177 (backward): g = 2*grad_dist; +g*(p-q) on the query, -g*(p-q) on
 * its nearest target, both directions accumulated sequentially.
 * ------------------------------------------------------------------------------------------- */
void orc_nnsearch(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int32_t *idx)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float x1 = xyz1[(i * (int64_t)n + j) * 3 + 0];
            const float y1 = xyz1[(i * (int64_t)n + j) * 3 + 1];
            const float z1 = xyz1[(i * (int64_t)n + j) * 3 + 2];
            double best = 0; int besti = 0;
            for (int k = 0; k < m; ++k) {
                const float x2 = xyz2[(i * (int64_t)m + k) * 3 + 0] - x1;
                const float y2 = xyz2[(i * (int64_t)m + k) * 3 + 1] - y1;
                const float z2 = xyz2[(i * (int64_t)m + k) * 3 + 2] - z1;
                const float df = x2 * x2 + y2 * y2 + z2 * z2;
                const double d = df;
                if (k == 0 || d < best) { best = d; besti = k; }
            }
            dist[i * (int64_t)n + j] = (float)best;
            idx[i * (int64_t)n + j] = besti;
        }
}

static void chamfer_bwd_dir(int b, int n, int m, const float *xyz1, const float *xyz2,
                            const float *gd1, const int32_t *idx1, float *g1, float *g2)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const int64_t a = (i * (int64_t)n + j) * 3;
            const int64_t c = (i * (int64_t)m + idx1[i * (int64_t)n + j]) * 3;
            const float g = gd1[i * (int64_t)n + j] * 2;
            for (int k = 0; k < 3; ++k) {
                const float d = g * (xyz1[a + k] - xyz2[c + k]);
                g1[a + k] += d;
                g2[c + k] -= d;
            }
        }
}

void orc_chamfer_backward(int b, int n, int m, const float *xyz1, const float *xyz2,
                          const float *gd1, const float *gd2, const int32_t *idx1, const int32_t *idx2,
                          float *g1, float *g2)
{
    memset(g1, 0, (size_t)b * n * 3 * sizeof(float));
    memset(g2, 0, (size_t)b * m * 3 * sizeof(float));
    chamfer_bwd_dir(b, n, m, xyz1, xyz2, gd1, idx1, g1, g2);
    chamfer_bwd_dir(b, m, n, xyz2, xyz1, gd2, idx2, g2, g1);
}
