// A4 (feature build): the nine per-point inputs of the pillar encoder, models/pillar_encoder.py:98-110, in one pass.
//   [x, y, z,  x - mean_x, y - mean_y, z - mean_z,  x - pillar_centre_x, y - pillar_centre_y,  t]   (first 8 / |x_min|, t / n_sweeps)
// The reference builds them with two row gathers ([N,3] pillar means, [N,5] float64 coordinates = 40 B per point), a cat and
// several strided in-place divisions: ~12 launches over N-sized tensors.  The arithmetic order is kept: the pillar centre is
// formed in float64 from the float64 coordinate (`mapped_coords[:,3] * vx + x_offset`), subtracted from the float32 point in
// float64 and rounded to float32 when stored into the float32 tensor; the divisions are float32 divisions.
#include "common.h"

template <typename CT>
__global__ __launch_bounds__(256) void pfn_features_kernel(const float *__restrict__ pts, const int32_t *__restrict__ p2v,
                                                           const float *__restrict__ mean, const CT *__restrict__ coords,
                                                           const double *__restrict__ tcol, int64_t t_stride, int64_t n,
                                                           double vx, double vy, double x_off, double y_off, float scale, float n_frames,
                                                           const int32_t *__restrict__ order, float *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const int64_t i = order ? order[r] : r;                // [r6] row r of the result = point order[r]: rows in pillar order (see pcacc_pfn_features_ordered)
        const int64_t s = p2v[i];
        const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        const float dx = __fsub_rn(x, mean[s * 3 + 0]), dy = __fsub_rn(y, mean[s * 3 + 1]), dz = __fsub_rn(z, mean[s * 3 + 2]);
        const double cxi = (double)coords[s * 5 + 3], cyi = (double)coords[s * 5 + 2];      // (b,z,y,x,t): column 3 = x, 2 = y
        const float fx = (float)((double)x - (cxi * vx + x_off));
        const float fy = (float)((double)y - (cyi * vy + y_off));
        const float t = (float)tcol[i * t_stride];
        float *o = out + r * 9;
        o[0] = __fdiv_rn(x, scale); o[1] = __fdiv_rn(y, scale); o[2] = __fdiv_rn(z, scale);
        o[3] = __fdiv_rn(dx, scale); o[4] = __fdiv_rn(dy, scale); o[5] = __fdiv_rn(dz, scale);
        o[6] = __fdiv_rn(fx, scale); o[7] = __fdiv_rn(fy, scale);
        o[8] = __fdiv_rn(t, n_frames);
    }
}

// [r6] rows in the order of `order` (the point -> pillar CSR's point list: ascending pillar, ascending point inside a pillar).  The reference's own point
// tensor is pillar-major ([M, max_points, C], libs/voxel_generator.py:41-58); with the encoder's rows in that order every per-pillar reduction and every
// pillar -> point broadcast behind this call reads consecutive rows instead of gathering 128-byte rows at random.
extern "C" int pcacc_pfn_features_ordered(const float *points, const int32_t *p2v, const float *pillar_mean, const void *coords,
                                          int coords_is_f64, const double *time_col, int64_t time_stride, int64_t n,
                                          double vx, double vy, double x_offset, double y_offset, float scale, float n_frames,
                                          const int32_t *order, float *out, void *stream)
{
    if (n < 0 || (n > 0 && (!points || !p2v || !pillar_mean || !coords || !time_col || !out))) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    hipStream_t s = pcacc_stream(stream);
    if (coords_is_f64)
        pfn_features_kernel<double><<<pcacc_grid(n, 256), 256, 0, s>>>(points, p2v, pillar_mean, static_cast<const double *>(coords),
                                                                     time_col, time_stride, n, vx, vy, x_offset, y_offset, scale, n_frames, order, out);
    else
        pfn_features_kernel<int32_t><<<pcacc_grid(n, 256), 256, 0, s>>>(points, p2v, pillar_mean, static_cast<const int32_t *>(coords),
                                                                      time_col, time_stride, n, vx, vy, x_offset, y_offset, scale, n_frames, order, out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_pfn_features(const float *points, const int32_t *p2v, const float *pillar_mean, const void *coords,
                                  int coords_is_f64, const double *time_col, int64_t time_stride, int64_t n,
                                  double vx, double vy, double x_offset, double y_offset, float scale, float n_frames,
                                  float *out, void *stream)
{
    return pcacc_pfn_features_ordered(points, p2v, pillar_mean, coords, coords_is_f64, time_col, time_stride, n, vx, vy, x_offset, y_offset, scale, n_frames,
                                      nullptr, out, stream);
}
