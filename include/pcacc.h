/*
 * pcacc.h -- C ABI of libpcacc_hip.so, the MI355X (gfx950) implementation of the PCAccumulation
 * per-frame forward hot path.
 *
 * The reference has no FFI for this path: it is Python calling torch / torch_scatter / numba and one
 * JIT-built torch extension (chamfer_distance).  Each entry point below therefore names the Python
 * (or C++) interface of the reference it stands in for, `file:line` relative to the reference root.
 * The host mirror (the .py files in pcaccumulation_amd/) binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its comment says "host";
 *   - no allocation, no synchronisation, no global state: the caller owns all buffers, passes scratch
 *     through `workspace` (size from the matching *_workspace_bytes) and the HIP stream to launch on;
 *   - return value: 0 on success, a negative PCACC_E_* code otherwise (launch errors are reported,
 *     never printed-and-ignored as chamfer_distance.cu:152-154 does);
 *   - tensors are dense row-major; BEV canvases and feature maps are CHANNELS-LAST
 *     ([frame, y, x, channel]) -- the layout the scatter/gather kernels coalesce on;
 *   - `dtype` arguments: PCACC_F32 or PCACC_BF16 for the canvas / feature-map element type.
 */
#ifndef PCACC_H_
#define PCACC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCACC_OK 0
#define PCACC_E_ARG (-1)      /* bad argument (null pointer, negative size, unsupported channel count) */
#define PCACC_E_WORKSPACE (-2) /* workspace too small */
#define PCACC_E_LAUNCH (-3)   /* hipGetLastError() != hipSuccess after a launch */

#define PCACC_F32 0
#define PCACC_BF16 1

/* Library / build identification: returns the gfx target string the kernels were compiled for. */
const char *pcacc_target(void);

/* torch.cat((a, b), 1) of two channels-last maps as rows: out [rows, a_row_bytes + b_row_bytes] = a [rows, a_row_bytes] | b [rows, b_row_bytes]; row sizes
 * and pointers multiples of 16 bytes.  The decoder concatenations of models/unet.py:101-113. */
int pcacc_cat2_rows(const void *a, int32_t a_row_bytes, const void *b, int32_t b_row_bytes, int64_t rows, void *out, void *stream);

/* The launchers' A/B switches (PCACC_CONV_FRAME_MAJOR, PCACC_CONV_SWZ_OFF, PCACC_CONV_RES, PCACC_ROWS_FM_OFF, PCACC_CONV_PLAN: experiments and
 * the equality tests only) are read from the environment once per process; this reads them again.  Returns 0. */
int pcacc_reload_switches(void);

/* ------------------------------------------------------------------------------------------------
 * A1. 4-D pillar voxelisation, first-touch numbering, bit-exact.
 * Replaces libs/voxel_generator.py:4-61 (_points_to_voxel_reverse_kernel), :64-114 (points_to_voxel)
 * and the array part of Voxelization.__call__ (:131-154).
 *   points      [n,4] f32 (x,y,z,t)
 *   voxel_size  host float[3], range host float[6] (xyz min, xyz max)
 *   coords      [max_voxels,4] i32 (z,y,x,t); rows >= *num_voxels are left untouched
 *   p2v         [n] i32, pillar id or -1 (out of range, t outside [0,nt), or beyond max_voxels)
 *   num_voxels  [1] i32 (device)
 * Pillar ids are the order in which cells are first touched when the points are visited in index
 * order, exactly as the sequential reference numbers them.
 * ---------------------------------------------------------------------------------------------- */
int pcacc_voxelize_workspace_bytes(int64_t n, int nx, int ny, int nz, int nt, size_t *bytes /*host*/);
int pcacc_voxelize(const float *points, int64_t n, const float *voxel_size, const float *range,
                   int nx, int ny, int nz, int nt, int max_voxels,
                   int32_t *coords, int32_t *p2v, int32_t *num_voxels,
                   void *workspace, size_t workspace_bytes, void *stream);

/* A1 + A2 for a whole batch of samples that already live in HBM: libs/dataset.py:183-199 (the voxelisation at the end of prep_input, per sample) and
 * libs/dataloader.py:7-40 (collate_fn: concatenation, the batch column, pillar ids offset by the pillars of the samples before) in ONE set of launches.
 *   points[b] [n_b,3] f64, time[b] [n_b] i64 (frame index), sd / inst / fb [b] [n_b] i64 (each of the three arrays may be NULL): DEVICE pointers in HOST
 *   arrays of n_samples <= 16 entries; counts host int64 [n_samples], every n_b >= 1.
 *   points_out [N,3] f64, time_out [N,2] f64 (sample, frame), sd_out / inst_out / fb_out [N] i64, coords_out [capacity >= N, 5] f64 rows
 *   (sample, z, y, x, t) in first-touch order (rows >= the total pillar count untouched), p2v_out [N] i32 (row of coords_out; a point outside the grid or with t outside [0, nt) gets
 *   (pillars of the samples before its own) - 1, what collate_fn's blind offset makes of its -1), num_voxels [n_samples] i32 -- all device.  Bit-identical to pcacc_voxelize per sample + the reference's collate_fn. */
int pcacc_collate_voxelize_workspace_bytes(int64_t n_total, int32_t n_samples, int nx, int ny, int nz, int nt, size_t *bytes /*host*/);
int pcacc_collate_voxelize(const double *const *points /*host*/, const int64_t *const *time /*host*/, const int64_t *const *sd /*host*/,
                           const int64_t *const *inst /*host*/, const int64_t *const *fb /*host*/, const int64_t *counts /*host*/, int32_t n_samples,
                           const float *voxel_size, const float *range, int nx, int ny, int nz, int nt, double *points_out, double *time_out,
                           int64_t *sd_out, int64_t *inst_out, int64_t *fb_out, double *coords_out, int32_t *p2v_out, int32_t *num_voxels,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * A2'. Decode the collated `coordinates` tensor once per forward.
 * Replaces the index arithmetic repeated in models/pillar_encoder.py:153-158 and :188-197:
 * cell = b*nt*ny*nx + t*ny*nx + y*nx + x from rows (b,z,y,x,t).
 *   coords        [m,5] f64 (collate_fn layout, libs/dataloader.py:18-20) when coords_is_f64 != 0,
 *                 else [m,5] i32
 *   cell          [m] i32
 *   cell2pillar   [n_batch*nt*ny*nx] i32: pillar id occupying the cell, -1 if empty; on duplicate
 *                 cells the highest pillar id wins ("later writes win", pillar_encoder.py:163)
 * ---------------------------------------------------------------------------------------------- */
int pcacc_cell_index(const void *coords, int coords_is_f64, int64_t m, int nx, int ny, int nt, int n_batch,
                     int32_t *cell, int32_t *cell2pillar, void *stream);

/* Occupied pillars of every frame in ascending cell order -- the order in which
 * models/egomotion.py:419-424 enumerates them through the boolean occupancy mask.
 *   sorted_pillars [m] i32; frame_offsets [n_frames+1] i32 (n_frames = n_batch*nt) */
int pcacc_frame_pillars_workspace_bytes(int64_t n_cells, size_t *bytes /*host*/);
int pcacc_frame_pillars(const int32_t *cell2pillar, int64_t n_cells, int64_t cells_per_frame,
                        int32_t *sorted_pillars, int32_t *frame_offsets,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Indices of the non-zero entries of a byte mask, ascending: the boolean-mask indexing of models/motionnet.py:222-243 (`points[fb_mask]`,
 * `inst_labels[rec_mask]`) and models/egomotion.py:419-424 (background pillars), as ONE index list the caller gathers with -- the reference
 * re-evaluates the mask at every indexed tensor.  Wave ballot + popcount per 2048-entry chunk, a one-workgroup scan of the chunk sums, ranks by
 * ballot prefix inside a wave ([r5]: replaces three torch.nonzero_static calls = 0.3 ms of library select kernels behind the forward's host sync).
 *   mask [n] u8 (torch.bool); indices [capacity] i64: the first min(count, capacity) entries are written; count_out [1] i32 may be NULL */
int pcacc_compact_mask_workspace_bytes(int64_t n, size_t *bytes /*host*/);
int pcacc_compact_mask(const uint8_t *mask, int64_t n, int64_t *indices, int64_t capacity, int32_t *count_out,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Point -> pillar CSR (points grouped by pillar, ascending point index inside a pillar for
 * pillars of <= 64 points).  Internal layout shared by every per-pillar reduction below; it is what
 * removes the atomics torch_scatter uses (models/motionnet.py:159-160, models/pillar_encoder.py:116,120).
 *   p2v [n] i32 (all >= 0), seg_offsets [m+1] i32, order [n] i32
 * ---------------------------------------------------------------------------------------------- */
int pcacc_csr_workspace_bytes(int64_t n, int64_t m, size_t *bytes /*host*/);
int pcacc_csr_build(const int32_t *p2v, int64_t n, int64_t m, int32_t *seg_offsets, int32_t *order,
                    void *workspace, size_t workspace_bytes, void *stream);

/* A3. scatter(points, p2v, 'mean') and scatter(fb_labels, p2v, 'max') -- models/motionnet.py:159-160.
 *   points [n,3] f32; labels [n] i64 or NULL; mean [m,3] f32; max_label [m] i64 (ignored if labels NULL) */
int pcacc_segment_mean3_maxlabel(const float *points, const int64_t *labels, const int32_t *seg_offsets,
                                 const int32_t *order, int64_t m, float *mean, int64_t *max_label, void *stream);

/* A4 (pooling step). scatter(net, p2v, dim=0, reduce='max') -- models/pillar_encoder.py:116,120.
 *   src [n,c] f32, c % 4 == 0, c <= 256; out [m,c] f32; arg [m,c] i32 = lowest point index attaining
 *   the maximum (the element torch_scatter routes the gradient to), -1 for an empty segment (value 0).
 *   When n/m > 16 (the per-instance poolings of models/tpointnet.py:227-259, few rows with thousands of
 *   inputs each) the reduction runs in two levels over 64-row pieces and needs the workspace. */
int pcacc_segment_workspace_bytes(int64_t n, int64_t m, int c, size_t *bytes /*host*/);
int pcacc_segment_max(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                      float *out, int32_t *arg, void *workspace, size_t workspace_bytes, void *stream);
/* Backward of segment_max followed by the [p2v] gather is not needed separately: both reduce to
 * grad_src[i,k] += grad_out[p2v[i],k] * (arg[p2v[i],k] == i). */
int pcacc_segment_max_backward(const float *grad_out, const int32_t *arg, const int32_t *p2v, int64_t n, int c,
                               float *grad_src, void *stream);

/* Per-pillar sum of point rows: the backward of the `[point_to_voxel_map]` broadcast that follows each
 * pooling (models/pillar_encoder.py:116).  src [n,c] f32, out [m,c] f32, c % 4 == 0, c <= 256. */
int pcacc_segment_sum(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                      float *out, void *workspace, size_t workspace_bytes, void *stream);
/* The same three with the element type of the rows as a parameter (PCACC_F32 | PCACC_BF16; reductions in f32).  Short
 * segments (n/m <= 16, the pillar encoder): results in the rows' type.  Long segments (the per-instance poolings): rows in
 * either type, results always f32.  max_backward: grad_out and grad_src types are independent. */
int pcacc_segment_max_t(const void *src, int dtype, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n,
                        int64_t m, void *out, int32_t *arg, void *workspace, size_t workspace_bytes, void *stream);
/* segment_max on f32 rows of SHORT segments with a second output: out16 [m,c] = `out` rounded to bf16, from the same store (the 'mixed' mode's
 * shadow of the pooled rows: models/pillar_encoder.py:116,120 -- it was a conversion pass over [m,c] after each of the three poolings).
 * PCACC_E_ARG when the segments are long (n / m > 16: the two-level reduction has no second output). */
int pcacc_segment_max_dual(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m, float *out,
                           uint16_t *out16, int32_t *arg, void *workspace, size_t workspace_bytes, void *stream);
int pcacc_segment_max_backward_t(const void *grad_out, int dtype, const int32_t *arg, const int32_t *p2v, int64_t n, int c,
                                 void *grad_src, int out_dtype, void *stream);
int pcacc_segment_sum_t(const void *src, int dtype, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n,
                        int64_t m, void *out, void *workspace, size_t workspace_bytes, void *stream);
/* max_backward added into grad_src (in/out): the rows' gradient from their other consumer is already there -- the pooled PFN blocks,
 * models/pillar_encoder.py:116-118, where `net` feeds both the max-pool and the concatenation.  out_amax: 256 zeroed f32 slots that
 * receive the largest magnitude of the sums (pcacc_absmax256 layout), or NULL. */
int pcacc_segment_max_backward_acc(const void *grad_out, int dtype, const int32_t *arg, const int32_t *p2v, int64_t n, int c,
                                   void *grad_src, int out_dtype, float *out_amax, void *stream);

/* A4 (feature build). The nine per-point inputs of the pillar encoder -- models/pillar_encoder.py:98-110:
 * [xyz, xyz - pillar_mean, xy - pillar_centre, t], first eight divided by `scale` (= |x_min|), t by n_frames.
 *   points [n,3] f32; p2v [n] i32; pillar_mean [m,3] f32; coords [m,5] (b,z,y,x,t) f64 or i32;
 *   time_col: pointer to the t entry of row 0 of the collated `time_indice` (f64), time_stride = row stride in doubles (2);
 *   vx, vy = pillar size; x_offset = vx/2 + x_min, y_offset = vy/2 + y_min (pillar_encoder.py:91-94); out [n,9] f32. */
int pcacc_pfn_features(const float *points, const int32_t *p2v, const float *pillar_mean, const void *coords,
                       int coords_is_f64, const double *time_col, int64_t time_stride, int64_t n,
                       double vx, double vy, double x_offset, double y_offset, float scale, float n_frames,
                       float *out, void *stream);
/* The same rows in the order of `order` [n] i32 (row r = point order[r]; NULL = point order): with the point -> pillar CSR's point list the rows come out
 * pillar-major like the reference's own [M, max_points, C] tensor (libs/voxel_generator.py:41-58) and every per-pillar pass behind this call streams. */
int pcacc_pfn_features_ordered(const float *points, const int32_t *p2v, const float *pillar_mean, const void *coords,
                               int coords_is_f64, const double *time_col, int64_t time_stride, int64_t n,
                               double vx, double vy, double x_offset, double y_offset, float scale, float n_frames,
                               const int32_t *order, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * A4 (MLP part). Per-point linear layers: nn.Linear applied to 10^5..10^6 rows with <= 128 features, as the pillar
 * encoder (models/pillar_encoder.py:46-55, 112-121), the STPN point heads (models/stpn.py:94-102) and the TubeNet
 * embeddings (models/tpointnet.py:170-193) use them.  One HBM pass per layer.
 *   y[r,:] = post( pre(x[r,:]) @ w^T + bias + residual[r,:] )
 *   x [rows,k] f32, k in {2,3,4,9,32,64,128}; w [n,k] f32, n <= 128; bias [n] or NULL; residual [rows,n] or NULL
 *   flags: 1 = ReLU on x at load, 2 = ReLU on y before the store
 *   in_mask [rows,k] or NULL: x is zeroed where in_mask <= 0;  out_mask [rows,n] or NULL: y is zeroed where out_mask <= 0
 * The backward-data pass is the same entry point with w transposed, the forward output as in_mask (post-ReLU layers)
 * and the forward input as out_mask (pre-ReLU layers).
 * pcacc_rows_wgrad: dw_aug [n, k+1] f32 = dYeff^T @ [Xeff | 1] (column k is the bias gradient), fp32 MFMA;
 *   dy [rows,n], dy_mask [rows,n] or NULL (dY zeroed where dy_mask <= 0), x [rows,k], x_relu: ReLU on x at load.
 * ---------------------------------------------------------------------------------------------- */
int pcacc_rows_linear(const float *x, const float *in_mask, const float *w, const float *bias, const float *residual,
                      const float *out_mask, float *y, int64_t rows, int k, int n, int flags, void *stream);
int pcacc_rows_wgrad(const float *dy, const float *dy_mask, const float *x, int x_relu, int64_t rows, int k, int n,
                     float *dw_aug, void *stream);

/* 'mixed' mode, k <= 9 inputs (the position layer of the pillar encoder, models/pillar_encoder.py:100-108): pcacc_rows_linear on f32 rows in plain
 * fp32 arithmetic with two more outputs from the same store phase: y16 = y as bf16 [rows][n], y_amax = 256 partial absolute maxima of y
 * (zero-filled by the caller; layout of pcacc_absmax256).  n in {8,16,32,64,128}; flags as pcacc_rows_linear. */
int pcacc_rows_linear_few_dual(const float *x, const float *w, const float *bias, const float *residual, float *y, uint16_t *y16,
                               float *y_amax, int64_t rows, int k, int n, int flags, void *stream);

/* bf16 compute mode of the same layers (cfg misc.compute_dtype = bf16): rows stored as bf16, fp32 accumulation.
 *   pcacc_rows_linear_bf16: x, in_mask, residual, out_mask, y all bf16; k, n in {32, 64, 128}; w, bias f32 (rounded to
 *     bf16 once per launch); the product runs on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16).
 *   pcacc_rows_linear_mixed / pcacc_rows_wgrad_mixed: the fp32-arithmetic kernels above with per-tensor element types,
 *     for the layers at the edges of a bf16 chain (k < 32, n = 2) and for the weight gradients.
 *     rows_linear dtypes bits: 1 = x, 2 = in_mask, 4 = residual, 8 = out_mask, 16 = y are bf16 (clear = f32);
 *     rows_wgrad  dtypes bits: 1 = dy, 2 = dy_mask, 4 = x are bf16. */
int pcacc_rows_linear_bf16(const uint16_t *x, const uint16_t *in_mask, const float *w, const float *bias,
                           const uint16_t *residual, const uint16_t *out_mask, uint16_t *y, int64_t rows, int32_t k,
                           int32_t n, int32_t flags, void *stream);
int pcacc_rows_linear_mixed(const void *x, const void *in_mask, const float *w, const float *bias, const void *residual,
                            const void *out_mask, void *y, int64_t rows, int k, int n, int flags, int dtypes, void *stream);
int pcacc_rows_wgrad_mixed(const void *dy, const void *dy_mask, const void *x, int x_relu, int64_t rows, int k, int n,
                           float *dw_aug, int dtypes, void *stream);
/* all-bf16 rows (k, n multiples of 32, <= 128): the same product on the bf16 matrix cores, dw_aug f32.
 * x_relu is a flags word here, in pcacc_rows_wgrad_cat_bf16 and in pcacc_rows_wgrad_few: bit 0 = ReLU on X; bit 1 = split result layout:
 * dw_aug holds dW [n,k] followed by the bias gradients [n], each contiguous (what an optimizer wants), instead of [n,k+1] rows. */
int pcacc_rows_wgrad_bf16_workspace_bytes(int64_t rows, int32_t k, int32_t n, size_t *bytes /*host*/);
int pcacc_rows_wgrad_bf16(const uint16_t *dy, const uint16_t *dy_mask, const uint16_t *x, int32_t x_relu, int64_t rows,
                          int32_t k, int32_t n, float *dw_aug, void *workspace, size_t workspace_bytes, void *stream);
/* The two kernels above on rows made of two pieces -- models/pillar_encoder.py:116-118 feeds every PFN block
 * cat(point row, pooled row of the point's pillar): x[row] = cat(xa[row] (ka columns), xb[b_index[row]] (k - ka columns)), read
 * in place (no gather, no concatenation; b_index NULL: xb[row]).  Forward: pcacc_rows_linear_cat_bf16 with y2 = NULL.
 * Backward-data: xa = the output gradient (xb NULL), w = W^T; the [rows,n] result leaves as y [rows,na] and y2 [rows,n-na],
 * masked where the forward input cat(out_mask_a[row], out_mask_b[b_index[row]]) was <= 0 (out_mask_* NULL: no mask). */
int pcacc_rows_linear_cat_bf16(const uint16_t *xa, const uint16_t *xb, const int32_t *b_index, int32_t ka, const uint16_t *in_mask,
                               const float *w, const float *bias, const uint16_t *residual, const uint16_t *out_mask_a,
                               const uint16_t *out_mask_b, uint16_t *y, uint16_t *y2, int32_t na, int64_t rows, int32_t k,
                               int32_t n, int32_t flags, void *stream);
int pcacc_rows_wgrad_cat_bf16(const uint16_t *dy, const uint16_t *dy_mask, const uint16_t *xa, const uint16_t *xb,
                              const int32_t *b_index, int32_t ka, int32_t x_relu, int64_t rows, int32_t k, int32_t n,
                              float *dw_aug, void *workspace, size_t workspace_bytes, void *stream);

/* [r6] The encoder's last max-pooling and the pillar scatter as ONE pass -- models/pillar_encoder.py:119-122 (scatter(net, point_to_voxel_map, 'max')) followed by
 * models/pillar_encoder.py:125-174 (scatter_point_pillar): the kernel walks the canvas cells; an occupied cell (cell2pillar >= 0) reduces its pillar's point
 * rows and stores the maximum at the cell, an empty cell stores zeros.  No [m, c] pooled-row table in between.  'mixed' mode form: fp32 rows in, the fp32
 * canvas AND its bf16 shadow out; arg [m, c] i32 = winners (lowest point index attaining the maximum), as pcacc_segment_max.  Short segments only
 * (n / m <= 16; PCACC_E_ARG otherwise: pool and fill separately).  start_event / stop_event: optional hipEvents attached to the dispatch (both or neither).
 *   src [n,c] f32; seg_offsets [m+1], order [n] (pcacc_csr_build); cell2pillar [n_cells] i32; canvas32 [n_cells,c] f32; canvas16 [n_cells,c] bf16
 * Backward: grad_src[i,k] = grad_canvas[cell[p2v[i]],k] where point i won channel k of its pillar, else 0; cell [m] i32 = the pillars' cell numbers. */
int pcacc_segment_max_canvas(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                             const int32_t *cell2pillar, int64_t n_cells, float *canvas32, uint16_t *canvas16, int32_t *arg,
                             void *start_event, void *stop_event, void *stream);
int pcacc_segment_max_canvas_backward(const void *grad_canvas, int dtype, const int32_t *arg, const int32_t *p2v, const int32_t *cell, int64_t n,
                                      int c, void *grad_src, int out_dtype, void *stream);

/* Sum of rows per index for FEW output rows (m*c <= 8192), no CSR needed: LDS-privatised accumulation.
 * The per-instance 'sum' / 'mean' poolings of models/tpointnet.py:227,251,283-284 and libs/loss.py:216 (torch_scatter: fp32 atomics in arrival order).
 * [r6] run-to-run identical: every workgroup sums its slice in 64-bit fixed point (scaled by the slice's own maximum; integer additions commute) and the
 * workgroups' fp32 partials meet in a fixed order.  A non-finite input element makes the affected partial -- and so every output element -- NaN.
 *   src [n,c] f32; idx [n] i32 in [0,m) (negative = skip); out [m,c] f32 (every element written); workspace: the per-workgroup partials */
int pcacc_scatter_sum_small_workspace_bytes(int64_t n, int c, int m, size_t *bytes /*host*/);
int pcacc_scatter_sum_small(const float *src, const int32_t *idx, int64_t n, int c, int m, float *out, void *workspace, size_t workspace_bytes,
                            void *stream);

/* ------------------------------------------------------------------------------------------------
 * A5. Pillar scatter into the BEV canvas -- models/pillar_encoder.py:125-174 (scatter_point_pillar).
 * One pass writes every canvas element exactly once (feature row or zeros), so there is no separate
 * zero-fill: canvas[cell, :] = cell2pillar[cell] >= 0 ? feats[cell2pillar[cell], :] : 0.
 *   feats [m,c] f32, c % 4 == 0 or c in {1,2,3}; canvas [n_cells, c] of `dtype` (channels-last)
 * ---------------------------------------------------------------------------------------------- */
int pcacc_pillar_scatter(const float *feats, const int32_t *cell2pillar, int64_t n_cells, int c,
                         void *canvas, int dtype, void *stream);
/* The same launch with two events attached to the dispatch (bench.py's roofline object): after the stream has been
 * synchronised, pcacc_timer_elapsed_us(start, stop) is the kernel's own begin-to-end time, as a kernel trace reports it. */
int pcacc_pillar_scatter_timed(const float *feats, const int32_t *cell2pillar, int64_t n_cells, int c, void *canvas, int dtype,
                               void *start_event, void *stop_event, void *stream);
/* Typed form: feats [m,c] of `feats_dtype` (PCACC_F32, or PCACC_BF16 with a PCACC_BF16 canvas and c % 8 == 0: the bf16 compute mode,
 * where the pillar encoder's last pooling leaves bf16 rows and the kernel moves exactly SURVEY.md 8d's bytes with s = 2:
 * c*2*n_cells written, c*2*m + 4*n_cells read).  start_event / stop_event: both NULL, or the two events of pcacc_timer_create. */
int pcacc_pillar_scatter_t(const void *feats, int feats_dtype, const int32_t *cell2pillar, int64_t n_cells, int c, void *canvas,
                           int dtype, void *start_event, void *stop_event, void *stream);
int pcacc_timer_create(void **start_event, void **stop_event);
int pcacc_timer_elapsed_us(void *start_event, void *stop_event, float *us);
int pcacc_timer_destroy(void *start_event, void *stop_event);

/* A6 and the backward of A5: row gather out[i,:] = src[idx[i],:] for rows of row_bytes bytes
 * (row_bytes % 4 == 0).  Replaces models/pillar_encoder.py:177-204 (inverse_scatter_point_pillar)
 * and the [point_to_voxel_map] gathers at models/motionnet.py:192-193. */
int pcacc_gather_rows(const void *src, int row_bytes, const int32_t *idx, int64_t n_idx, void *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * A11. Per-point bilinear feature gather ("ungrid") -- models/pillar_encoder.py:231-267 and
 * temporal_ungrid :206-228.  Direct 4-tap gather; the reference's replication of the feature map
 * into fake H x W grids is not reproduced.  grid_sample semantics: bilinear, padding 'border',
 * align_corners=False; u = x / x_scale, v = y / y_scale (fp32 division as in pillar_encoder.py:247-249).
 *   fmap [n_maps,h,w,c] `dtype` channels-last; points [k,3] f32; map_idx [k] i32; out [k,c] f32; c % 4 == 0
 * Backward accumulates into grad_fmap [n_maps,h,w,c] f32 (caller zero-fills) with fp32 atomics.
 * ---------------------------------------------------------------------------------------------- */
int pcacc_bilinear_gather(const void *fmap, int dtype, int n_maps, int h, int w, int c,
                          const float *points, const int32_t *map_idx, int64_t k,
                          float x_scale, float y_scale, float *out, void *stream);
int pcacc_bilinear_gather_backward(const float *grad_out, int n_maps, int h, int w, int c,
                                   const float *points, const int32_t *map_idx, int64_t k,
                                   float x_scale, float y_scale, float *grad_fmap, void *stream);
/* The same gradient without atomics (index-ordered sums): pcacc_bilinear_base_cells writes, per point, the cell of its
 * upper-left tap (n_maps*h*w for points of no map); after pcacc_csr_build over those keys (m = n_maps*h*w + 1),
 * pcacc_bilinear_gather_backward_sorted sums, per map cell, the contributions of the four neighbouring base cells in
 * index order (two launches: tap weights and gradient rows laid out in CSR order, then the per-cell sums over consecutive
 * rows; workspace = those two tables).  grad_out [k,c] and grad_fmap [n_maps,h,w,c] may each be f32 or bf16. */
int pcacc_bilinear_base_cells(const float *points, const int32_t *map_idx, int64_t k, int n_maps, int h, int w,
                              float x_scale, float y_scale, int32_t *cell, void *stream);
int pcacc_bilinear_sorted_workspace_bytes(int64_t k, int c, int grad_dtype, size_t *bytes /*host*/);
int pcacc_bilinear_gather_backward_sorted(const void *grad_out, int grad_dtype, int n_maps, int h, int w, int c,
                                          const float *points, const int32_t *seg_offsets, const int32_t *order, int64_t k,
                                          float x_scale, float y_scale, void *grad_fmap, int out_dtype, void *workspace,
                                          size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * A9. Ego-motion BEV warp -- models/motionnet.py:45-80 (get_transformed_grid) + :82-114 (warp_feats).
 *   bev    [n_batch, nt, h, w, c] `dtype`; out same shape/dtype
 *   inv_pose [n_batch, nt, 16] f32 row-major 4x4 = inverse of the estimated pose of frame t
 * Frames 1..nt-1 are resampled bilinearly (zeros padding, align_corners=False) on the grid
 * inv_pose[:2,:2] @ (pixel centre in metres) + inv_pose[:2,3], divided by |x_min|, |y_min|.
 * Slot 0 receives frame nt-1 UNWARPED: bug-compatible with the leaked loop variable at
 * models/motionnet.py:100,111 (trap 1).
 * ---------------------------------------------------------------------------------------------- */
int pcacc_bev_warp(const void *bev, int dtype, int n_batch, int nt, int h, int w, int c,
                   const float *inv_pose, float x_reso, float y_reso, float x_min, float y_min,
                   void *out, void *stream);
/* [r6] 'mixed' mode form: fp32 map in, the fp32 result AND its bf16 copy (same element offsets) out of one pass. */
int pcacc_bev_warp_dual(const float *bev, int n_batch, int nt, int h, int w, int c, const float *inv_pose, float x_reso, float y_reso, float x_min,
                        float y_min, float *out, uint16_t *out16, void *stream);

/* A10. Per-point rigid transform -- models/motionnet.py:117-135 (transform_points).
 *   points [n,3] f32; frame_idx [n] i32 = b*nt + t; tsfm [n_frames,16] f32; out [n,3] f32 */
int pcacc_rigid_transform(const float *points, const int32_t *frame_idx, const float *tsfm, int64_t n,
                          float *out, void *stream);

/* Key-point draws of the ego-motion head -- models/egomotion.py:156-166 (torch.randperm(n)[:k], or arange clamped to n-1
 * when n <= k).  counts [n_draws] i32 (device): population of every draw; out [n_draws, k] i64: k distinct indices per
 * draw (keyed Feistel permutation of [0, n) evaluated at 0..k-1), all draws in one launch.  This is the 'device' sampler
 * (cfg pose_estimation.kpt_sampler); the default sampler reproduces the reference's host RNG stream instead. */
int pcacc_sample_subsets(const int32_t *counts, int32_t n_draws, int32_t k, uint64_t seed, int64_t *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * A8. Key-point matching of the ego-motion head, forward pass, fp32, for n_pairs (frame, anchor) pairs at once --
 * models/egomotion.py:169-192 after the key-point choice: square_distance (toolbox/utils.py:125-144), affinity,
 * sinkhorn with slack row/column (egomotion.py:100-137), exp * support, soft targets, weighted Kabsch with a 3x3 SVD
 * and reflection fix (toolbox/register_utils.py:247-317).
 *   feats_s, feats_t [n_pairs,k,c] f32 (L2-normalised rows); coor_s, coor_t [n_pairs,k,3] f32;
 *   thr2 [n_pairs] f32 = (duration * max_speed)^2;  params [2] f32 (device) = {softplus(alpha), exp(beta) + 0.02}
 *   perm [n_pairs,k,k] f32 out;  pose [n_pairs,16] f32 out (row-major 4x4)
 * Used when no gradient is needed (eval / val / test); training keeps the batched torch formulation for autograd.
 * ---------------------------------------------------------------------------------------------- */
/* Sinkhorn normalisation as a differentiable op (training) -- models/egomotion.py:100-137 with add_slack: the [P,k,k]
 * log-affinity is padded with a zero slack row / column, n_iters x (row, column) log-sum-exp normalisations that leave the
 * slack row / column themselves un-normalised, result = the un-padded block.  forward records the subtracted
 * log-sum-exps (lse_rows, lse_cols [n_iters][P][k]); backward replays the half-steps in reverse from log_alpha and those
 * vectors.  workspace: one padded [P,k+1,k+1] f32 matrix (pcacc_sinkhorn_train_workspace_bytes). */
int pcacc_sinkhorn_train_workspace_bytes(int n_pairs, int k, size_t *bytes /*host*/);
int pcacc_sinkhorn_forward(const float *log_alpha, int n_pairs, int k, int n_iters, float *log_perm, float *lse_rows,
                           float *lse_cols, void *workspace, size_t workspace_bytes, void *stream);
int pcacc_sinkhorn_backward(const float *grad_log_perm, const float *log_alpha, const float *lse_rows, const float *lse_cols,
                            int n_pairs, int k, int n_iters, float *grad_log_alpha, void *workspace, size_t workspace_bytes,
                            void *stream);
/* The matching stage around the Sinkhorn iterations in the TRAINING path (models/egomotion.py:169-184 under autograd), P pairs at once:
 *  ego_affinity_forward   affinity [P,k,k] = -(max(2 - 2 <fs_i, ft_j>, 1e-12) - params[0]) / params[1]; feats [P,k,c] f32 (L2-normalised rows),
 *                         params [2] f32 on the device = (softplus(alpha), exp(beta) + 0.02)
 *  ego_affinity_backward  grad_dot [P,k,k] = d loss / d <fs_i, ft_j> (feature gradients: grad_dot @ ft, grad_dot^T @ fs), grad_params [2]
 *  ego_perm_forward       perm = exp(log_perm) * [ |cs_i - ct_j|^2 < thr2[p] ], rowsum [P,k], weighted_t [P,k,3] = perm @ ct / (rowsum + 1e-20),
 *                         colsum [P,k] (NULL = not wanted): with rowsum all the outlier loss (libs/outlier_loss.py) reads of perm
 *  ego_perm_backward      grad_log_perm from the gradients of perm / rowsum / weighted_t / colsum (any of them NULL = 0) */
int pcacc_ego_affinity_forward(const float *feats_s, const float *feats_t, const float *params, int n_pairs, int k, int c, float *affinity,
                               void *stream);
int pcacc_ego_affinity_backward_workspace_bytes(size_t *bytes /*host*/);
int pcacc_ego_affinity_backward(const float *grad_affinity, const float *affinity, const float *params, int64_t n, float *grad_dot,
                                float *grad_params, void *workspace, size_t workspace_bytes, void *stream);
int pcacc_ego_perm_forward(const float *log_perm, const float *coor_s, const float *coor_t, const float *thr2, int n_pairs, int k, float *perm,
                           float *rowsum, float *weighted_t, float *colsum, void *stream);
int pcacc_ego_perm_backward(const float *grad_perm, const float *grad_rowsum, const float *grad_weighted_t, const float *grad_colsum,
                            const float *perm, const float *coor_t, const float *rowsum, const float *weighted_t, int n_pairs, int k,
                            float *grad_log_perm, void *stream);
/* Weighted Kabsch in front of its 3x3 SVD, under autograd (toolbox/register_utils.py:263-291), P pairs at once: weights w [P,k] normalised by
 * (sum w + 1e-7), weighted means m1, m2 [P,3] of x1, x2 [P,k,3] (divided by sum wn + 1e-7), covariance cov [P,3,3] of the centred clouds;
 * norm [P,2] = the two normalisers, kept for the backward.  backward: grad_x2, grad_w (x1 -- pillar means -- carries no gradient). */
int pcacc_kabsch_cov_forward(const float *x1, const float *x2, const float *w, int n_pairs, int k, float *cov, float *m1, float *m2, float *norm,
                             void *stream);
int pcacc_kabsch_cov_backward(const float *x1, const float *x2, const float *w, const float *m1, const float *m2, const float *norm,
                              const float *grad_cov, const float *grad_m1, const float *grad_m2, int n_pairs, int k, float *grad_x2,
                              float *grad_w, void *stream);
/* ... and behind the SVD (toolbox/register_utils.py:305-313): rot [n,3,3] = V diag(1, 1, det(V U^T)) U^T, trans [n,3] = m2 - rot m1, from the
 * singular vectors u, v [n,3,3] and the weighted means m1, m2 [n,3]; backward from grad_rot / grad_trans (NULL = 0) incl. the path through
 * the determinant (its derivative is the cofactor matrix: no LU factorisation of a 3x3 matrix). */
int pcacc_kabsch_rt_forward(const float *u, const float *v, const float *m1, const float *m2, int n, float *rot, float *trans, void *stream);
int pcacc_kabsch_rt_backward(const float *u, const float *v, const float *m1, const float *grad_rot, const float *grad_trans, int n,
                             float *grad_u, float *grad_v, float *grad_m1, float *grad_m2, void *stream);
/* Batched 3x3 SVD of the Kabsch solve in the training path -- toolbox/register_utils.py:293 (`torch.svd(cov_mat)`):
 * a [n,3,3] f32 = u diag(s) v^T, s descending, no status word read back on the host.  backward: grad_a from the gradients of
 * u, s, v (any of them NULL = 0), closed form for distinct singular values. */
int pcacc_svd3(const float *a, int64_t n, float *u, float *s, float *v, void *stream);
int pcacc_svd3_backward(const float *u, const float *s, const float *v, const float *grad_u, const float *grad_s,
                        const float *grad_v, int64_t n, float *grad_a, void *stream);
int pcacc_sinkhorn_kabsch_workspace_bytes(int n_pairs, int k, size_t *bytes /*host*/);
int pcacc_sinkhorn_kabsch(const float *feats_s, const float *feats_t, const float *coor_s, const float *coor_t,
                          const float *thr2, const float *params, int n_pairs, int k, int c, int n_iters,
                          float *perm, float *pose, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * A12. Chamfer nearest neighbour -- chamfer_distance/chamfer_distance.cu:6-155 (forward kernel and
 * launcher), chamfer_distance.cpp:59-111 (CPU twin), called through
 * chamfer_distance/chamfer_distance.py:9-31.  dist = squared fp32 distance (x*x + y*y) + z*z of
 * (target - query), idx = LOWEST target index attaining it (chamfer_distance.cu:39,49,129).
 *   xyz1 [b,n,3], xyz2 [b,m,3] f32; dist1 [b,n], dist2 [b,m] f32; idx1 [b,n], idx2 [b,m] i32
 * Backward -- chamfer_distance.cu:158-209 / chamfer_distance.cpp:114-177; grad_xyz* are zero-filled
 * by the call.
 * ---------------------------------------------------------------------------------------------- */
int pcacc_chamfer_workspace_bytes(int b, int n, int m, size_t *bytes /*host*/);
int pcacc_chamfer_forward(const float *xyz1, const float *xyz2, int b, int n, int m,
                          float *dist1, int32_t *idx1, float *dist2, int32_t *idx2,
                          void *workspace, size_t workspace_bytes, void *stream);
int pcacc_chamfer_backward(const float *xyz1, const float *xyz2, int b, int n, int m,
                           const float *grad_dist1, const int32_t *idx1, const float *grad_dist2, const int32_t *idx2,
                           float *grad_xyz1, float *grad_xyz2, void *stream);

/* ------------------------------------------------------------------------------------------------
 * C1. Test-mode instance clustering -- models/cluster.py:86-111 (Cluster.forward with fb_labels=None),
 * :52-84 (cluster_per_batch), :23-49 (cluster), :9-13 (voxel_downsample); called from
 * models/motionnet.py:238.  Replaces the host round trip through torchsparse v1.4.0 sparse_quantize
 * (README.md:27) and sklearn.cluster.DBSCAN(metric='euclidean') with one set of launches for the whole
 * batch; same labels, bit for bit (DESIGN.md section 9 has the argument).
 *   points [n,3] f32   ego-motion-compensated points (results['transformed_points'])
 *   offset [n,2] f32   predicted centre offsets, added to x,y before clustering (use_offset=True), or NULL
 *   sel    [n]   u8    1 = predicted moving (mos.argmax(1) == 1)
 *   batch  [n]   i32   sample index of every point, 0 <= batch < n_batches <= 64
 *   voxel_size         0.05 with offsets, 0.15 without (cluster.py:76,80)
 *   eps, min_samples   DBSCAN parameters (cfg.cluster.eps_dbscan, min_samples_dbscan); z is ignored
 *   min_p_cluster      clusters with fewer down-sampled points are dropped, and a sample with
 *                      <= min_p_cluster selected points is not clustered at all (cluster.py:38-41,66)
 *   labels [n]   i64   0 = no instance, 1..K per sample in canonical order (toolbox/utils.py:237-250)
 * Voxel indices are packed in 19 bits per axis: |coordinate / voxel_size| must stay below 2^18.
 * ---------------------------------------------------------------------------------------------- */
int pcacc_cluster_workspace_bytes(int64_t n, size_t *bytes /*host*/);
int pcacc_cluster(const float *points, const float *offset, const uint8_t *sel, const int32_t *batch, int64_t n,
                  int32_t n_batches, float voxel_size, double eps, int32_t min_samples, int32_t min_p_cluster,
                  int64_t *labels, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * A6/A9. 3x3 convolution + bias + ReLU on the bf16 matrix cores -- the nn.Conv2d(3x3, stride 1, padding 1)
 * layers of models/unet.py:15-27 (conv3x3), :45-71 (DownConv), :74-113 (UpConv), :196-199 (conv_final),
 * the STPN backbone models/stpn.py:24-43, and with kt = 3 the Conv3d(3x3x3, padding 1) + ReLU stack of
 * models/stpn.py:13-22 evaluated on [B*T] frames (frame t reads frames t-1, t, t+1 of its own sample).
 *   in   [n_img, h, w, c_in]  bf16 channels-last;  out [n_img, h, w, c_out] bf16
 *   wp   prepared weights, bf16 [kt*9][c_out][c_in] (pcacc_conv3x3_prepare_weights)
 *   bias [c_out] f32 or NULL;  relu != 0 clamps at 0 before the store
 *   frames: images per sample (T) when kt = 3, else 1;  n_img % frames == 0
 *   c_in, c_out multiples of 32.  fp32 accumulation; the result is rounded to bf16 once.
 * prepare_weights: w f32 [c_out][c_in][kt][3][3] (torch layout) -> wp.  transpose != 0 builds the weights
 * of the data gradient instead (bf16 [kt*9][c_in][c_out], taps mirrored): dX = conv(dY, wp_T) with the
 * same entry point and c_in / c_out exchanged.
 * ---------------------------------------------------------------------------------------------- */
int pcacc_conv3x3_prepare_weights(const float *w, int32_t c_out, int32_t c_in, int32_t kt, int32_t transpose,
                                  uint16_t *out, void *stream);
/* Both prepared forms of one weight tensor in one launch (out_fwd = transpose 0, out_bwd = transpose 1 of the entry point above), the fp32
 * weight read through `strides` (host array, in elements: o, i, [t,] y, x -- contiguous or channels-last storage). */
int pcacc_conv3x3_prepare_weights_pair(const float *w, int32_t c_out, int32_t c_in, int32_t kt, const int64_t *strides /*host*/,
                                       uint16_t *out_fwd, uint16_t *out_bwd, void *stream);
int pcacc_conv3x3_bf16(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int32_t n_img,
                       int32_t frames, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt, int32_t relu,
                       void *stream);
/* Weight gradient of the same layers (c_in, c_out in {32, 64}): dw = [c_out][9][c_in] f32 followed by [c_out] f32 (bias
 * gradient = sum of dy over the images and pixels this launch visits; complete for dt = 0);  dw[co][tap][ci] = sum of
 * dy[n][px][co] * x[n + dt][px + tap offset][ci]; dt in {-1, 0, 1} selects the frame tap of a kt = 3 layer (0 for kt = 1),
 * frames as above.  bf16 MFMA with fp32 accumulation; per-workgroup partials in `workspace`, summed by a second launch. */
/* The deep layers (c_in >= 128 in steps of 64, c_out in steps of 64, kt = 1: the 128..512-channel layers of models/unet.py:45-113 and
 * models/stpn.py:24-43 on 18^2 .. 144^2 images): tiles of 32 consecutive pixels of a strip instead of 8 x 32 image tiles, the strip's
 * input patch resident in LDS per channel slice, weight tiles double buffered.  pcacc_conv3x3_bf16 routes such shapes here itself;
 * pcacc_conv3x3_deep_supported tells whether a shape qualifies (1) or not (0). */
int pcacc_conv3x3_deep_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out);
int pcacc_conv3x3_deep_bf16(const uint16_t *in, const uint16_t *in_mask, const uint16_t *wp, const float *bias, uint16_t *out, int32_t n_img,
                            int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t relu, void *stream);
/* ReLU backward fused into the consumers of the gradient: `in_mask` / `dy_mask` (NULL = none) is the forward OUTPUT of the ReLU layer
 * whose gradient `in` / `dy` is, same shape; elements where it is not > 0 are read as zero (aten::threshold_backward on the fly, no
 * separate pass over the gradient map).  Deep layers only (pcacc_conv3x3_deep_supported / pcacc_conv3x3_wgrad_deep_supported): they
 * are MFMA-bound, the second read is free; PCACC_E_ARG for a mask on any other shape. */
int pcacc_conv3x3_masked_bf16(const uint16_t *in, const uint16_t *in_mask, const uint16_t *wp, const float *bias, uint16_t *out,
                              int32_t n_img, int32_t frames, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt,
                              int32_t relu, void *stream);
/* ReLU backward on the OUTPUT side of a data gradient: the convolution's result is stored as zero where out_mask [n_img,h,w,c_out] (the
 * forward input of the layer, itself a ReLU output: models/unet.py:45-71 conv -> ReLU -> conv) is <= 0, so the producing layer needs no
 * threshold pass.  No bias, no ReLU.  The layers the strip kernels would take are not supported (pcacc_conv3x3_outmask_supported = 0). */
int pcacc_conv3x3_outmask_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt);
int pcacc_conv3x3_outmask_bf16(const uint16_t *in, const uint16_t *wp, const uint16_t *out_mask, uint16_t *out, int32_t n_img, int32_t frames,
                               int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t kt, void *stream);
/* Weight gradient of the same deep layers (c_in, c_out multiples of 64, at least one of them > 64; kt = 1): dw [c_out][9][c_in] f32 and
 * db [c_out] f32 = bias gradient, from dy [n_img,h,w,c_out] and x [n_img,h,w,c_in] (bf16, channels-last).  64 x 64 blocks of the weight
 * tensor per workgroup, strips of consecutive pixels, per-workgroup partial slots in the workspace + a reduce launch. */
int pcacc_conv3x3_wgrad_deep_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out);
int pcacc_conv3x3_wgrad_deep_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out, size_t *bytes /*host*/);
int pcacc_conv3x3_wgrad_deep_bf16(const uint16_t *dy, const uint16_t *dy_mask, const uint16_t *x, float *dw, float *db, int32_t n_img,
                                  int32_t h, int32_t w, int32_t c_in, int32_t c_out, void *workspace, size_t workspace_bytes, void *stream);
/* ------------------------------------------------------------------------------------------------
 * A6/A9 at fp32 accuracy ("fp32x3" compute mode): the same layers -- fp32 convolutions in the reference, models/unet.py:11-20,45-113,
 * models/stpn.py:13-43 -- on fp32 channels-last maps, every product formed on the 16-bit matrix cores from fp16 hi / lo halves of
 * power-of-two scaled operands (w_hi x_lo + w_lo x_hi + w_hi x_hi, fp32 accumulation; 22 significant bits per factor, relative error
 * ~3e-7 per product).  One kernel family for all layers (c_in, c_out multiples of 32; kt = 1 or 3): tiles of rows x band-width pixels,
 * hi / lo planes of the patch and of the per-tap weight tiles in LDS (csrc/conv_split.hip).
 *   absmax256: parts[256] f32 <- partial maxima of |x| over n f32 elements (x 16-byte aligned; a NaN gives +inf).  Every consumer below
 *       takes such an array for each fp32 tensor it splits and derives the tensor's scale from it (no host round trip).
 *   prepare_weights: w f32 [c_out][c_in][kt][3][3] read through `strides` (host, elements: o, i, [t,] y, x) ->
 *       out_fwd fp16 [2 = hi, lo][kt*9][c_out][c_in] + scale_fwd f32 [c_out]; out_bwd fp16 [2][kt*9][c_in][c_out] (taps mirrored:
 *       data gradient) + scale_bwd f32 [c_in]; scale = 1 / (power-of-two scale of that output-channel row)
 *   split:  in [n_img,h,w,c_in] f32 (+ in_amax) -> out [n_img,h,w,c_out] f32; in_mask (NULL = none): same shape as `in`, elements of `in`
 *       are read as zero where in_mask <= 0 (ReLU backward of the layer whose gradient `in` is, fused into the staging); wp / wscale from
 *       prepare_weights; bias / relu / frames / kt as pcacc_conv3x3_bf16; out_amax (NULL = not wanted): 256 f32 slots, ZEROED by the caller,
 *       receive partial maxima of |out| -- an absmax256 array of the result for the kernel that consumes it
 *   wgrad_split: dw [c_out][9][c_in] f32, db [c_out] f32 (NULL = not wanted; complete for dt = 0) from dy [n_img,h,w,c_out] f32
 *       (dy_mask as in_mask) and x [n_img,h,w,c_in] f32 with their absmax256 arrays; dt / frames as pcacc_conv3x3_wgrad_bf16
 * ---------------------------------------------------------------------------------------------- */
int pcacc_absmax256(const float *x, int64_t n, float *parts, void *stream);
int pcacc_conv3x3_split_prepare_weights(const float *w, int32_t c_out, int32_t c_in, int32_t kt, const int64_t *strides /*host*/,
                                        uint16_t *out_fwd, float *scale_fwd, uint16_t *out_bwd, float *scale_bwd, void *stream);
int pcacc_conv3x3_split_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_out);
int pcacc_conv3x3_split(const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp, const float *wscale,
                        const float *bias, float *out, float *out_amax, int32_t n_img, int32_t frames, int32_t h, int32_t w, int32_t c_in,
                        int32_t c_out, int32_t kt, int32_t relu, void *stream);
/* pcacc_conv3x3_split with a second result, out16 [n_img,h,w,c_out] bf16 = the fp32 result rounded to nearest even, written by the same epilogue
 * ('mixed' compute mode: fp32x3 forward values, bf16 gradient graph -- the shadow the bf16 data / weight gradient kernels read; 2 extra bytes per
 * element stored instead of a 6-byte cast pass).  models/unet.py:11-20,45-113, models/stpn.py:13-43. */
int pcacc_conv3x3_split_dual(const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp, const float *wscale,
                             const float *bias, float *out, float *out_amax, uint16_t *out16, int32_t n_img, int32_t frames, int32_t h,
                             int32_t w, int32_t c_in, int32_t c_out, int32_t kt, int32_t relu, void *stream);

/* The same convolution (no bias, no ReLU; in_mask as above) with its result stored as zero where out_mask [n_img,h,w,c_out] f32 is <= 0: the
 * data gradient of conv -> ReLU -> conv masked for the first ReLU in the epilogue (the fp32x3 twin of pcacc_conv3x3_outmask_bf16). */
/* pcacc_conv3x3_split_dual (kt = 1, no input mask) on the channel concatenation of TWO inputs read in place: in_a [n_img][h][w][c_a], in_b
 * [n_img][h][w][c_in - c_a] -- conv1(cat(upconv(x), skip)) of a decoder stage (models/unet.py:101-113) without writing the fp32 concatenation.
 * in_amax bounds both inputs (element-wise maximum of their pcacc_absmax256 arrays); out16 may be NULL; c_a and c_in - c_a multiples of 32. */
int pcacc_conv3x3_split_cat(const float *in_a, const float *in_b, int32_t c_a, const float *in_amax, const uint16_t *wp, const float *wscale,
                            const float *bias, float *out, float *out_amax, uint16_t *out16, int32_t n_img, int32_t h, int32_t w, int32_t c_in,
                            int32_t c_out, int32_t relu, void *stream);
int pcacc_conv3x3_split_outmask(const float *in, const float *in_amax, const float *in_mask, const uint16_t *wp, const float *wscale,
                                const float *out_mask, float *out, float *out_amax, int32_t n_img, int32_t frames, int32_t h, int32_t w,
                                int32_t c_in, int32_t c_out, int32_t kt, void *stream);
int pcacc_conv3x3_wgrad_split_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out, size_t *bytes /*host*/);
int pcacc_conv3x3_wgrad_split(const float *dy, const float *dy_amax, const float *dy_mask, const float *x, const float *x_amax, float *dw,
                              float *db, int32_t n_img, int32_t frames, int32_t dt, int32_t h, int32_t w, int32_t c_in, int32_t c_out,
                              void *workspace, size_t workspace_bytes, void *stream);
/* nn.ConvTranspose2d(kernel 2, stride 2) of the two decoders (models/unet.py:22-30,101-113) in the fp32x3 mode: a 1-tap product on the
 * tiles of pcacc_conv3x3_split whose epilogue scatters a pixel's 4 c_up results to its 2 x 2 output pixels (direction 0), its data
 * gradient reading dy space-to-depth (direction 1), and the weight gradient on pcacc_conv3x3_wgrad_split's kernel.  c_in, c_up
 * multiples of 32; h, w = the small map's size.
 *   prepare_weights: w f32 [c_in][c_up][2][2] through `strides` (host, elements: i, o, y, x) -> out_fwd fp16 [2][4 c_up][c_in] (rows
 *       (a, b, co)) + scale_fwd [4 c_up], out_bwd fp16 [2][c_in][4 c_up] + scale_bwd [c_in]
 *   upconv2x2_split: direction 0: in [n,h,w,c_in] -> out [n,2h,2w,c_up] (+ bias [c_up]); direction 1: in = dy [n,2h,2w,c_up] -> out
 *       [n,h,w,c_in]; in_amax / out_amax as pcacc_conv3x3_split
 *   wgrad: dw [4 c_up][c_in] f32 (row (a, b, co)), db4 [4 c_up] f32 (sums of dy per (a, b, co); the bias gradient is the sum of the four groups) */
int pcacc_upconv2x2_split_prepare_weights(const float *w, int32_t c_in, int32_t c_up, const int64_t *strides /*host*/, uint16_t *out_fwd,
                                          float *scale_fwd, uint16_t *out_bwd, float *scale_bwd, void *stream);
int pcacc_upconv2x2_split_supported(int32_t h, int32_t w, int32_t c_in, int32_t c_up);

/* The same layer (nn.ConvTranspose2d(kernel 2, stride 2), models/unet.py:22-30,101-113) on bf16 channels-last rows -- the bf16 compute mode's forward
 * and backward, the 'mixed' mode's backward -- on the bf16 matrix cores, fp32 accumulation.  c_in a multiple of 64, c_up a multiple of 32.
 *   prepare_weights: w f32 [c_in][c_up][2][2] through `strides` (host, elements: i, o, y, x) -> out_fwd bf16 [4 c_up][c_in] (rows (a, b, co)),
 *       out_bwd bf16 [c_in][4 c_up]
 *   upconv2x2_bf16: direction 0: in [n,h,w,c_in] (pixel pitch in_pitch >= c_in elements), wp = out_fwd -> out [n,2h,2w,c_up] (pixel pitch out_pitch),
 *       + bias[c_up] (f32, may be NULL); direction 1: in = dy [n,2h,2w,c_up] (pixel pitch in_pitch: dy may be a channel slice of a wider map),
 *       wp = out_bwd -> out [n,h,w,c_in] (pixel pitch out_pitch).  Pitches are multiples of 8 elements, pointers 16-byte aligned.
 *   wgrad: dy as above, x [n,h,w,c_in] (pitch x_pitch) -> dw f32 [c_in][c_up][2][2] written through `dw_strides` (host, elements: i, o, y, x -- the
 *       layout of the weight the gradient belongs to; NULL = contiguous), db f32 [c_up] (may be NULL); workspace from
 *       pcacc_upconv2x2_bf16_wgrad_workspace_bytes. */
int pcacc_upconv2x2_bf16_supported(int32_t c_in, int32_t c_up);
int pcacc_upconv2x2_bf16_prepare_weights(const float *w, int32_t c_in, int32_t c_up, const int64_t *strides /*host*/, uint16_t *out_fwd,
                                         uint16_t *out_bwd, void *stream);
int pcacc_upconv2x2_bf16(const uint16_t *in, const uint16_t *wp, const float *bias, uint16_t *out, int32_t n_img, int32_t h, int32_t w,
                         int32_t c_in, int32_t c_up, int32_t direction, int32_t in_pitch, int32_t out_pitch, void *stream);
int pcacc_upconv2x2_bf16_wgrad_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, size_t *bytes);
int pcacc_upconv2x2_bf16_wgrad(const uint16_t *dy, int32_t dy_pitch, const uint16_t *x, int32_t x_pitch, float *dw, const int64_t *dw_strides /*host*/,
                               float *db, int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, void *workspace, size_t workspace_bytes,
                               void *stream);

/* Every prepared form of every weight of a model in ONE launch (a training step re-prepares ~90 forms right after the optimizer wrote the
 * parameters -- the per-layer weight handling of the reference's nn.Conv2d / nn.Conv3d / nn.ConvTranspose2d, models/unet.py:22-113,
 * models/stpn.py:39-70).  `jobs` = DEVICE table, 16 int64 per job:
 *   [0] w (f32, device)  [1] out_fwd  [2] scale_fwd  [3] out_bwd  [4] scale_bwd  [5..9] strides of w in elements
 *   [10] a  [11] b  [12] kt  [13] kind  [14] first workgroup of the job (ascending, job 0 starts at 0)  [15] workgroups of the job
 * kind 0 = pcacc_conv3x3_split_prepare_weights   (a = c_out, b = c_in; strides o, i, t, y, x; a + b workgroups)
 * kind 1 = pcacc_upconv2x2_split_prepare_weights (a = c_in, b = c_up; strides i, o, -, y, x; 4 b + a workgroups)
 * kind 2 = pcacc_conv3x3_prepare_weights_pair    (a = c_out, b = c_in; strides o, i, t, y, x; any number of workgroups >= 1; no scales)
 * kind 3 = pcacc_upconv2x2_bf16_prepare_weights  (a = c_in, b = c_up; strides i, o, -, y, x; any number of workgroups >= 1; no scales)
 * with the outputs of those entry points; total_blocks = [14] + [15] of the last job. */
int pcacc_prepare_weights_batch(const int64_t *jobs, int32_t n_jobs, int32_t total_blocks, void *stream);
int pcacc_upconv2x2_split(const float *in, const float *in_amax, const uint16_t *wp, const float *wscale, const float *bias, float *out,
                          float *out_amax, int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, int32_t direction, void *stream);
/* pcacc_upconv2x2_split (direction 0) with the bf16 shadow of its result as a second output ('mixed' mode, see pcacc_conv3x3_split_dual).
 * out_pitch: elements between consecutive pixels of out and out16 (0 = c_up, dense; 2 c_up: both results are the first half of the decoder's
 * concatenation buffers [n,2h,2w,2 c_up], models/unet.py:101-113 -- the up-sampled half is never copied). */
int pcacc_upconv2x2_split_dual(const float *in, const float *in_amax, const uint16_t *wp, const float *wscale, const float *bias, float *out,
                               float *out_amax, uint16_t *out16, int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up,
                               int32_t out_pitch, void *stream);
int pcacc_upconv2x2_wgrad_split_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_up, size_t *bytes /*host*/);
int pcacc_upconv2x2_wgrad_split(const float *dy, const float *dy_amax, const float *x, const float *x_amax, float *dw, float *db4, int32_t n_img,
                                int32_t h, int32_t w, int32_t c_in, int32_t c_up, void *workspace, size_t workspace_bytes, void *stream);
/* The last convolution of a SegHead2D -- models/unet.py:259-277: Conv2d(mid, 2, 3, padding 1), the fg / bg logits -- c_in = 32 or 64,
 * c_out <= 4: exact fp32 arithmetic on f32 (dtype 0) or bf16 (1) channels-last rows, streamed (csrc/head_conv.hip; matrix-core tiles
 * would be 94 % padding).  w f32 [c_out][c_in][3][3] read through `w_strides` (host, elements: o, i, y, x).
 *   forward: y [n,h,w,c_out] f32 = conv(x) + bias;  dgrad: dx [n,h,w,c_in] (f32 / bf16) from dy [n,h,w,c_out] f32;
 *   wgrad: dw [c_out][c_in][3][3] f32 contiguous, db [c_out] f32 or NULL; per-workgroup partials in `workspace`, summed in a fixed order. */
int pcacc_head_conv3x3_supported(int32_t c_in, int32_t c_out);
int pcacc_head_conv3x3_forward(const void *x, int32_t x_dtype, const float *w, const int64_t *w_strides /*host*/, const float *bias, float *y,
                               int32_t n_img, int32_t h, int32_t wd, int32_t c_in, int32_t c_out, void *stream);
int pcacc_head_conv3x3_dgrad(const float *dy, const float *w, const int64_t *w_strides /*host*/, void *dx, int32_t dx_dtype, int32_t n_img,
                             int32_t h, int32_t wd, int32_t c_in, int32_t c_out, void *stream);
int pcacc_head_conv3x3_wgrad_workspace_bytes(int32_t n_img, int32_t h, int32_t wd, int32_t c_in, int32_t c_out, size_t *bytes /*host*/);
int pcacc_head_conv3x3_wgrad(const float *dy, const void *x, int32_t x_dtype, float *dw, float *db, int32_t n_img, int32_t h, int32_t wd,
                             int32_t c_in, int32_t c_out, void *workspace, size_t workspace_bytes, void *stream);
/* The per-point linear layers in the fp32x3 mode (csrc/mlp_split.hip): the contracts of pcacc_rows_linear_bf16 / _cat_bf16 /
 * pcacc_rows_wgrad_bf16 / _cat_bf16 above on fp32 rows (x, masks, residual, y all f32; k, n in {32, 64, 128}), products from scaled
 * fp16 hi / lo halves as in pcacc_conv3x3_split; every fp32 row tensor that is split comes with its pcacc_absmax256 array
 * (x_amax; two-piece rows: one per piece; dy_amax), the weight matrix is scaled per output row inside the kernel; y_amax (NULL = not
 * wanted; 256 f32 slots zeroed by the caller) receives partial maxima of |y| (of y and y2 together), as out_amax above.  nn.Linear of
 * models/pillar_encoder.py:13-55,113-122, models/stpn.py:94-102, models/tpointnet.py:176-196 (fp32 in the reference). */
int pcacc_rows_linear_split(const float *x, const float *x_amax, const float *in_mask, const float *w, const float *bias,
                            const float *residual, const float *out_mask, float *y, float *y_amax, int64_t rows, int32_t k, int32_t n,
                            int32_t flags, void *stream);
/* 'mixed' compute mode: pcacc_rows_linear_split with y16 [rows,n] = bf16(y) as a second output of the same epilogue (see pcacc_conv3x3_split_dual). */
int pcacc_rows_linear_split_dual(const float *x, const float *x_amax, const float *in_mask, const float *w, const float *bias,
                                 const float *residual, const float *out_mask, float *y, uint16_t *y16, float *y_amax, int64_t rows, int32_t k,
                                 int32_t n, int32_t flags, void *stream);
int pcacc_rows_linear_cat_split(const float *xa, const float *xa_amax, const float *xb, const float *xb_amax, const int32_t *b_index,
                                int32_t ka, const float *in_mask, const float *w, const float *bias, const float *residual,
                                const float *out_mask_a, const float *out_mask_b, float *y, float *y2, int32_t na, float *y_amax, int64_t rows,
                                int32_t k, int32_t n, int32_t flags, void *stream);
int pcacc_rows_wgrad_split_workspace_bytes(int64_t rows, int32_t k, int32_t n, size_t *bytes /*host*/);
int pcacc_rows_wgrad_split(const float *dy, const float *dy_amax, const float *dy_mask, const float *x, const float *x_amax, int32_t x_relu,
                           int64_t rows, int32_t k, int32_t n, float *dw_aug, void *workspace, size_t workspace_bytes, void *stream);
int pcacc_rows_wgrad_cat_split(const float *dy, const float *dy_amax, const float *dy_mask, const float *xa, const float *xa_amax,
                               const float *xb, const float *xb_amax, const int32_t *b_index, int32_t ka, int32_t x_relu, int64_t rows,
                               int32_t k, int32_t n, float *dw_aug, void *workspace, size_t workspace_bytes, void *stream);
int pcacc_conv3x3_wgrad_workspace_bytes(int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out, size_t *bytes /*host*/);
int pcacc_conv3x3_wgrad_bf16(const uint16_t *dy, const uint16_t *x, float *dw, int32_t n_img, int32_t frames, int32_t dt,
                             int32_t h, int32_t w, int32_t c_in, int32_t c_out, void *workspace, size_t workspace_bytes,
                             void *stream);

/* ------------------------------------------------------------------------------------------------
 * D1. Data step in front of the path -- libs/dataset.py:147-182 (BaseDataset.prep_input before the voxeliser),
 * :93-104 (apply_data_augmentation), toolbox/register_utils.py:199-206 (apply_tsfm).  One pass over the raw points:
 *   points [m,3] f64;  tsfm12 [12] f64 on the device = 3x3 rotation (row-major) then translation, or NULL;
 *   noise [m,3] f64 uniform(0,1) draws or NULL (added as (u - 0.5) * noise_scale);  scale: global factor applied when
 *   tsfm12 or noise is given;  crop_xy, z_min, z_max: crop box;  remove_ground / ground_z: keep only z > ground_z;
 *   out_points [m,3] f64: augmented points;  keep [m] u8: 1 where the point passes crop (and ground) tests.
 * The kept rows are compacted by the caller (stable order); libs/dataset.py:166-181.
 * ---------------------------------------------------------------------------------------------- */
int pcacc_prep_points(const double *points, const double *tsfm12, const double *noise, double noise_scale, double scale,
                      double crop_xy, double z_min, double z_max, int32_t remove_ground, double ground_z, int64_t m,
                      double *out_points, uint8_t *keep, void *stream);

/* ------------------------------------------------------------------------------------------------
 * L1. Two-class segmentation loss on selected rows -- libs/loss.py:110-137 (FuseLoss.get_seg_loss: weighted cross entropy
 * with the class weights of :90-108, Lovasz-Softmax of libs/lovasz_softmax.py:56-94, IoU counters of compute_iou :17-50),
 * as called on the occupied pillars (:167-191) and on the foreground points (:140-165).
 *   logits: f32 or bf16 (PCACC_F32 | PCACC_BF16), row i at [i][2] (plane = 0) or as planes [i / plane][2][i % plane]
 *           (an NCHW head output read in place, plane = H*W);  labels [n_total] i64 (0, 1, or -1 = ignored by the cross
 *           entropy and background of both classes in the Lovasz term, as in the reference);  rows [n] i64 = the selected
 *           rows (NULL: rows 0..n-1)
 *   out_loss [2] f32 = (cross entropy, Lovasz);  out_metric [4][2] f64 = intersection, union, predicted, labelled per
 *   class, each / 1e3;  lovasz_grad [2][n] f32 and saved [8] f32 are what backward needs (the Jaccard gradient of each
 *   row's rank per class; the class weights, their sum over the rows and the share of each present class)
 *   backward: grad_bce, grad_lovasz = device scalars (NULL = 0);  grad_logits has the layout and type of logits over all
 *   n_total rows (rows outside the selection get 0).  Equal errors are ranked by row index.
 * ---------------------------------------------------------------------------------------------- */
int pcacc_seg_loss_workspace_bytes(int64_t n, size_t *bytes /*host*/);
int pcacc_seg_loss_forward(const void *logits, int logits_dtype, int64_t plane, const int64_t *labels, const int64_t *rows,
                           int64_t n, float *out_loss, double *out_metric, float *lovasz_grad, float *saved,
                           void *workspace, size_t workspace_bytes, void *stream);
int pcacc_seg_loss_backward(const void *logits, int logits_dtype, int64_t plane, const int64_t *labels, const int64_t *rows,
                            int64_t n, int64_t n_total, const float *lovasz_grad, const float *saved, const float *grad_bce,
                            const float *grad_lovasz, void *grad_logits, void *stream);

/* ------------------------------------------------------------------------------------------------
 * L2. Offset loss -- libs/loss.py:194-250 (FuseLoss.get_offset_loss): every point is reconstructed with the ground truth
 * (toolbox/register_utils.py:59-93: ego pose of its frame, then the motion of its instance), the instance centres are the
 * means of the reconstructed points per instance, and the selected (foreground) rows compare the offset to their centre
 * with the estimate.
 *   points [n,3] f32;  time_indice [n,2] i64 (sample, frame);  inst_labels [n] i64 (per sample);  label_base [B] i64 =
 *   first row of each sample in inst_motion;  ego_motion [B,T,16] f32;  inst_motion [k,T,16] f32 (all samples
 *   concatenated);  transformed_points [n,3] f32;  offset_est [n,2] f32;  rows [m] i64 (NULL: all rows)
 *   out [3] f32 = (L1 term, direction term, mean L2 error);  offset_gt [m,2] f32 = predictions['offset_gt'] (:246)
 *   backward: grad_norm / grad_dir = device scalars (NULL = 0);  grad_est [n,2] f32 (0 outside the selection).
 * ---------------------------------------------------------------------------------------------- */
int pcacc_offset_loss_workspace_bytes(int64_t m, int64_t k, size_t *bytes /*host*/);
int pcacc_offset_loss_forward(const float *points, const int64_t *time_indice, const int64_t *inst_labels,
                              const int64_t *label_base, const float *ego_motion, const float *inst_motion, int32_t n_frames,
                              int64_t n, int64_t k, const float *transformed_points, const float *offset_est,
                              const int64_t *rows, int64_t m, float *out, float *offset_gt, void *workspace,
                              size_t workspace_bytes, void *stream);
int pcacc_offset_loss_backward(const float *offset_gt, const float *offset_est, const int64_t *rows, int64_t m, int64_t n,
                               const float *grad_norm, const float *grad_dir, float *grad_est, void *stream);

/* A9. Max over the frames of a map stack -- models/stpn.py:83 (`torch.max(x, dim=2)` between the temporal convolutions and the
 * U-Net): x [n_seq, frames, plane] f32 or bf16 (plane = H*W*C contiguous elements, a multiple of 4 / 8), out [n_seq, plane],
 * arg [n_seq, plane] u8 = the winning frame (lowest on ties; a NaN wins).  backward: grad_x[s,t,p] = arg[s,p] == t ? grad_out[s,p] : 0,
 * every element written. */
int pcacc_frames_max(const void *x, int dtype, int64_t n_seq, int32_t frames, int64_t plane, void *out, uint8_t *arg, void *stream);
int pcacc_frames_max_backward(const void *grad_out, const uint8_t *arg, int dtype, int64_t n_seq, int32_t frames, int64_t plane,
                              void *grad_x, void *stream);

/* A10 point heads: nn.BatchNorm1d in training mode over [rows, c] rows -- models/unet.py:240-245 (SegHead1D: Linear, BatchNorm1d,
 * ReLU, Linear on the K foreground points; batch statistics over the rows).  x, y: f32 or bf16 (PCACC_F32 | PCACC_BF16), c a
 * multiple of 4 / 8 with 256 % (c / 4 | 8) == 0, c <= 256;  gamma, beta [c] f32 or NULL;  running_mean / running_var [c] f32
 * updated in place when given (momentum; unbiased variance, as torch);  save_mean, save_invstd [c] f32 out (for backward).
 * backward: grad_x in the type of x, grad_gamma / grad_beta [c] f32. */
int pcacc_bn_rows_workspace_bytes(int64_t rows, int32_t c, size_t *bytes /*host*/);
int pcacc_bn_rows_forward(const void *x, int dtype, int64_t rows, int32_t c, const float *gamma, const float *beta, float eps,
                          float momentum, float *running_mean, float *running_var, void *y, float *save_mean,
                          float *save_invstd, void *workspace, size_t workspace_bytes, void *stream);
int pcacc_bn_rows_backward(const void *grad_y, const void *x, int dtype, int64_t rows, int32_t c, const float *gamma,
                           const float *save_mean, const float *save_invstd, void *grad_x, float *grad_gamma, float *grad_beta,
                           void *workspace, size_t workspace_bytes, void *stream);

/* BatchNorm + ReLU in the same passes (SegHead2D, models/unet.py:264-268: conv, BatchNorm2d, ReLU, conv): y = max(bn(x), 0); the backward
 * takes grad_y only where that output was > 0 -- recomputed from x, the saved statistics, gamma and beta (NULL = 1 / 0), no mask tensor. */
int pcacc_bn_relu_rows_forward(const void *x, int dtype, int64_t rows, int32_t c, const float *gamma, const float *beta, float eps,
                               float momentum, float *running_mean, float *running_var, void *y, float *save_mean, float *save_invstd,
                               void *workspace, size_t workspace_bytes, void *stream);
int pcacc_bn_relu_rows_backward(const void *grad_y, const void *x, int dtype, int64_t rows, int32_t c, const float *gamma, const float *beta,
                                const float *save_mean, const float *save_invstd, void *grad_x, float *grad_gamma, float *grad_beta,
                                void *workspace, size_t workspace_bytes, void *stream);
/* pcacc_bn_rows_forward / pcacc_bn_relu_rows_forward (relu != 0) on f32 rows with two more outputs from the same store phase ('mixed' compute mode):
 * y16 = y as bf16 [rows][c], y_amax = 256 partial absolute maxima of y (zero-filled by the caller; layout of pcacc_absmax256); either may be NULL */
int pcacc_bn_rows_forward_dual(const float *x, int64_t rows, int32_t c, const float *gamma, const float *beta, float eps, float momentum,
                               float *running_mean, float *running_var, int32_t relu, float *y, uint16_t *y16, float *y_amax, float *save_mean,
                               float *save_invstd, void *workspace, size_t workspace_bytes, void *stream);

/* the two backward entries above (relu != 0: pcacc_bn_relu_rows_backward with its `beta`) with a second output from the same store phase:
 * grad_x_amax = 256 partial absolute maxima of grad_x (zero-filled by the caller; layout of pcacc_absmax256) -- the fp32x3 layer in front of
 * the normalisation scales its incoming gradient by them */
int pcacc_bn_rows_backward_m(const void *grad_y, const void *x, int dtype, int64_t rows, int32_t c, const float *gamma, const float *beta,
                             int32_t relu, const float *save_mean, const float *save_invstd, void *grad_x, float *grad_x_amax, float *grad_gamma,
                             float *grad_beta, void *workspace, size_t workspace_bytes, void *stream);

/* A ResnetBlockFC of the pillar encoder -- models/pillar_encoder.py:13-55 with size_in 64, size_h 32, size_out 32 and the linear
 * shortcut (the blocks of PillarFeatureNet, :76-78) -- fused over bf16 point rows:
 *     h = relu(x) w0^T + b0,   out = relu(h) w1^T + b1 + x ws^T          w0, ws [32,64], w1 [32,32], b0, b1 [32] f32
 * x [rows,64] = `xa` when pooled == NULL, else cat(xa[row] [32], pooled[p2v[row]] [32]) (models/pillar_encoder.py:116-118).
 * forward: out [rows,32], relu_h [rows,32] (kept for the backward; may be NULL).
 * backward: grad_xa [rows,64] (pooled == NULL) or grad_xa [rows,32] + grad_xb [rows,32] (per point: the caller sums it over each
 * pillar's points); grad_params [5184] f32 = d w1 [32,32] | d b1 [32] | d ws [32,64] | d w0 [32,64] | d b0 [32]. */
int pcacc_pfn_block_forward(const uint16_t *xa, const uint16_t *pooled, const int32_t *p2v, const float *w0, const float *b0,
                            const float *ws, const float *w1, const float *b1, uint16_t *out, uint16_t *relu_h, int64_t rows, void *stream);
int pcacc_pfn_block_backward_workspace_bytes(int64_t rows, size_t *bytes /*host*/);
int pcacc_pfn_block_backward(const uint16_t *xa, const uint16_t *pooled, const int32_t *p2v, const uint16_t *relu_h,
                             const uint16_t *grad_out, const float *w0, const float *ws, const float *w1, uint16_t *grad_xa,
                             uint16_t *grad_xb, float *grad_params, int64_t rows, void *workspace, size_t workspace_bytes, void *stream);

/* The same block on fp32 point rows at fp32 accuracy (compute mode fp32x3: products on the matrix cores from scaled fp16 hi / lo
 * halves, csrc/pfn_block_split.hip).  *_amax arguments are pcacc_absmax256 arrays of the tensor they follow (inputs), or 256 zeroed
 * slots the kernel's store phase fills (outputs; may be NULL).
 * forward: out, relu_h [rows,32] f32; xmask [rows] u64 / hmask [rows] u32: bit c set where x[row][c] > 0 / h[row][c] > 0 -- all the
 * data-gradient kernel needs of x and h.
 * dgrad: grad_xa [rows,64] (grad_xb == NULL) or grad_xa [rows,32] + grad_xb [rows,32]; grad_h [rows,32] = d(h), the operand of fc_0's
 * weight gradient.  The three weight gradients are pcacc_rows_wgrad_split / _cat_split calls on (grad_out, relu_h), (grad_out, x)
 * and (grad_h, relu(x)). */
int pcacc_pfn_block_split_forward(const float *xa, const float *xa_amax, const float *pooled, const float *pooled_amax, const int32_t *p2v,
                                  const float *w0, const float *b0, const float *ws, const float *w1, const float *b1, float *out,
                                  float *relu_h, uint64_t *xmask, uint32_t *hmask, float *out_amax, float *hr_amax, int64_t rows,
                                  void *stream);
/* 'mixed' compute mode: pcacc_pfn_block_split_forward's fp32 result plus the two operands of the bf16 backward kernel pcacc_pfn_block_backward,
 * written by the same epilogue: out16 = bf16(out), hr16 = bf16(relu(h)) [rows,32]; no fp32 relu(h), no sign masks.  models/pillar_encoder.py:13-55. */
int pcacc_pfn_block_split_forward_dual(const float *xa, const float *xa_amax, const float *pooled, const float *pooled_amax, const int32_t *p2v,
                                       const float *w0, const float *b0, const float *ws, const float *w1, const float *b1, float *out,
                                       uint16_t *out16, uint16_t *hr16, float *out_amax, int64_t rows, void *stream);
int pcacc_pfn_block_split_dgrad(const float *grad_out, const float *grad_out_amax, const uint64_t *xmask, const uint32_t *hmask,
                                const float *w0, const float *ws, const float *w1, float *grad_xa, float *grad_xb, float *grad_h,
                                float *gx_amax, float *dh_amax, int64_t rows, void *stream);

/* Tail of a U-Net encoder stage -- models/unet.py:60-71 (DownConv: ... conv, ReLU, 2x2 max-pool; returns the pooled map and the map
 * before the pool).  Channels-last bf16 maps [n_img, h, w, c], c a multiple of 8.
 *  maxpool2x2                out [n_img, h/2, w/2, c]: nn.MaxPool2d(2, 2) (floor mode; NaN propagates), no index map.
 *  pool_skip_relu_backward   grad_y = (un-pool(grad_pooled) + grad_skip) where y > 0, else 0: the gradient of the stage's ReLU input
 *                            from the gradients of its two consumers in one pass (the window's first maximum in scan order takes
 *                            the pooled gradient, as the library's forward picks it); either gradient may be NULL (= 0). */
int pcacc_maxpool2x2_bf16(const uint16_t *x, int64_t n_img, int32_t h, int32_t w, int32_t c, uint16_t *out, void *stream);
int pcacc_pool_skip_relu_backward_bf16(const uint16_t *y, const uint16_t *grad_pooled, const uint16_t *grad_skip, int64_t n_img, int32_t h,
                                       int32_t w, int32_t c, uint16_t *grad_y, void *stream);
/* The same two passes on fp32 rows (c % 4 == 0; compute mode fp32x3).  out_amax: 256 zeroed f32 slots receiving the largest magnitude of
 * grad_y (pcacc_absmax256 layout) for the split kernels that read it, or NULL. */
int pcacc_maxpool2x2_f32(const float *x, int64_t n_img, int32_t h, int32_t w, int32_t c, float *out, void *stream);
int pcacc_pool_skip_relu_backward_f32(const float *y, const float *grad_pooled, const float *grad_skip, int64_t n_img, int32_t h, int32_t w,
                                      int32_t c, float *grad_y, float *out_amax, void *stream);
/* grad_skip read in place from a wider map (skip_pitch = elements between consecutive pixels, >= c, a multiple of 8 (bf16) / 4 (f32), the
 * pointer 16-byte aligned): the skip half of the decoder's concatenation gradient (models/unet.py:101-113) without a contiguous copy. */
int pcacc_pool_skip_relu_backward_strided_bf16(const uint16_t *y, const uint16_t *grad_pooled, const uint16_t *grad_skip, int64_t skip_pitch,
                                               int64_t n_img, int32_t h, int32_t w, int32_t c, uint16_t *grad_y, void *stream);
int pcacc_pool_skip_relu_backward_strided_f32(const float *y, const float *grad_pooled, const float *grad_skip, int64_t skip_pitch,
                                              int64_t n_img, int32_t h, int32_t w, int32_t c, float *grad_y, float *out_amax, void *stream);
/* 'mixed' compute mode: y f32 (the forward's own values decide the window's winner -- a bf16 copy of y ties values closer than 2^-8 and sends
 * ~1 % of the windows' gradient to the wrong pixel), gradients and result bf16; c a multiple of 8. */
int pcacc_pool_skip_relu_backward_strided_y32(const float *y, const uint16_t *grad_pooled, const uint16_t *grad_skip, int64_t skip_pitch,
                                              int64_t n_img, int32_t h, int32_t w, int32_t c, uint16_t *grad_y, void *stream);

/* Batched inverse of n 4x4 f32 matrices (the pose tables: torch.linalg.inv at models/motionnet.py:100 and models/alignnet.py:33),
 * Gauss-Jordan with partial pivoting, one launch; a singular matrix yields inf / nan entries (no status word). */
int pcacc_inv4x4(const float *m, int64_t n, float *out, void *stream);

/* Weight gradient of a row-linear layer with few inputs (k in {1,2,3,4,9}, n in {32,64,128}: first layers of the point chains) or few
 * outputs (n in {1,2,3,4,9}, k in {32,64,128}: the heads) -- same result and operand conventions as pcacc_rows_wgrad_mixed
 * (dw_aug [n, k+1] f32, bias gradient in the last column; dtypes bit 0 dY, bit 1 dy_mask, bit 2 X set = bf16), streamed at HBM speed
 * instead of padded into 32 x 32 matrix-core tiles; summed in a fixed order through workspace partials (no atomics). */
int pcacc_rows_wgrad_few_supported(int32_t k, int32_t n);
int pcacc_rows_wgrad_few_workspace_bytes(int64_t rows, int32_t k, int32_t n, size_t *bytes /*host*/);
int pcacc_rows_wgrad_few(const void *dy, const void *dy_mask, const void *x, int32_t x_relu, int64_t rows, int32_t k, int32_t n,
                         float *dw_aug, int32_t dtypes, void *workspace, size_t workspace_bytes, void *stream);

/* TubeNet slot algebra -- models/tpointnet.py:249-305 (TPointNet.forward after the pooled embeddings) and the refinement loop of
 * models/alignnet.py:236-263, per refinement iteration.  A slot s = k * n_frames + t is instance k in frame t; slot [n] i32 is the
 * slot of every foreground point; slot_centre [S,3] the per-slot centroids (the anchor frame's row k * n_frames is the one used).
 *  rows           rows [n,4] = (xyz - centre of the instance's anchor frame, t / n_frames): input of the positional embedding
 *                 (tpointnet.py:246-251).
 *  code           code [S,4c] = (geo[k], motion[k], frame[s], frame[k * n_frames]) (tpointnet.py:259-262); backward: the three gradients.
 *  pose_forward   pose_vec [S,7] (quaternion xyzw, translation) -> pose_c / gt_c [S,12] (R row-major, t): estimated pose and ground truth
 *                 `remaining` [S,4,4] re-expressed for the centred cloud (batch_quat2mat / batch_mat2quat, tpointnet.py:20-73);
 *                 loss_rt [2] f64 = weighted means of |q_gt - q| and |t_gt - t| (evaluate_pose, :76-94; q_gt by scipy's branch scheme
 *                 in f64); step [S,4,4] = the un-centred pose, frame 0 = identity (:291-296); remaining_out = remaining @ step^-1 and
 *                 total_out = step @ total_in (alignnet.py:257-263; total_in NULL = identity); wsum [1] = sum(weights) + 1e-20.
 *  gap_forward    pp [n,4] = (|est - gt|_2, |est - gt|_1, 0, 0) of the centred point under the two poses (tpointnet.py:276-281).
 *  finish         l12 [2] = sum_s weights[s] * slot_sums[s][0|1] / max(count[s], 1) / wsum (tpointnet.py:282-286).
 *  gap_backward   grad_rows16 [n,16]: per point gradient of (grad_l1 * l12[0] + grad_l2 * l12[1]) w.r.t. pose_c of its slot (12 + 4 zeros);
 *                 their per-slot sums are grad_pose [S,stride] of
 *  pose_backward  grad_vec [S,7], including the loss_rt terms (grad_rot, grad_trans f64 scalars; NULL = 0).
 * The slot-level kernels run in one workgroup each. */
int pcacc_tube_rows(const float *xyz, const int32_t *slot, const float *slot_centre, int64_t n, int32_t n_frames, float *rows, void *stream);
int pcacc_tube_code(const float *geo, const float *motion, const float *frame, int64_t n_inst, int32_t n_frames, int32_t c, float *code,
                    void *stream);
int pcacc_tube_code_backward(const float *grad_code, int64_t n_inst, int32_t n_frames, int32_t c, float *grad_geo, float *grad_motion,
                             float *grad_frame, void *stream);
int pcacc_tube_pose_forward(const float *pose_vec, const float *remaining, const float *total_in, const float *slot_centre,
                            const float *weights, int32_t n_slots, int32_t n_frames, float *pose_c, float *gt_c, float *step,
                            float *remaining_out, float *total_out, double *loss_rt, float *wsum, void *stream);
int pcacc_tube_gap_forward(const float *rows, const int32_t *slot, const float *pose_c, const float *gt_c, int64_t n, float *pp, void *stream);
int pcacc_tube_finish(const float *slot_sums, int32_t stride, const float *count, const float *weights, const float *wsum, int32_t n_slots,
                      float *l12, void *stream);
int pcacc_tube_gap_backward(const float *rows, const int32_t *slot, const float *pose_c, const float *gt_c, const float *weights,
                            const float *count, const float *wsum, const float *grad_l1, const float *grad_l2, int64_t n,
                            float *grad_rows16, void *stream);
int pcacc_tube_pose_backward(const float *pose_vec, const float *remaining, const float *slot_centre, const float *weights, const float *wsum,
                             const float *grad_pose, int32_t stride, const double *grad_rot, const double *grad_trans, int32_t n_slots,
                             int32_t n_frames, float *grad_vec, void *stream);

/* Host words -> device memory as kernel arguments (asynchronous, unlike a pageable hipMemcpy on the compute stream):
 * n 32-bit words from host_words to dst, 240 per launch.  For the per-step index tables a host loop of the reference
 * becomes (sample offsets, per-pair counts, thresholds). */
int pcacc_upload_words(const uint32_t *host_words, int64_t n, uint32_t *dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCACC_H_ */
