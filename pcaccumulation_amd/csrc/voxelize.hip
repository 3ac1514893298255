// A1: 4-D pillar voxelisation with the reference's sequential first-touch numbering, done in parallel.
//
// The reference (libs/voxel_generator.py:4-61) walks the points in index order and gives a cell the
// next free pillar id the first time it is touched.  Equivalent parallel statement:
//   first[cell] = min point index touching the cell            (atomicMin on a dense cell table)
//   a point is a "first" iff first[cell(i)] == i
//   pillar id of a cell = number of firsts with a smaller point index   (exclusive prefix sum)
// which reproduces coordinates[] and point_to_voxel_map[] bit for bit.
//
// HBM traffic per point: read 16 B once (k_keys) and keep a 4-byte cell key, so the three later
// passes read 4 B + one 4-byte table word instead of the 16-byte point again.
#include "scan.h"

#define VOX_INVALID 0xFFFFFFFFu
#define VOX_RANK_BIT 0x80000000u

struct VoxGeom {
    float r0, r1, r2;      // range min x,y,z
    float v0, v1, v2;      // voxel size
    int nx, ny, nz, nt;
};

// pass 1: cell key per point + first-touch table.  key layout = reference table [nz,ny,nx,nt].
__global__ __launch_bounds__(256) void vox_keys(const float4 *__restrict__ pts, int64_t n, VoxGeom g,
                                                uint32_t *__restrict__ keys, uint32_t *table)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float4 p = pts[i];
        // fp32 subtract, IEEE fp32 divide, floor -- the numpy float32 arithmetic of voxel_generator.py:41
        const float cx = floorf(__fdiv_rn(__fsub_rn(p.x, g.r0), g.v0));
        const float cy = floorf(__fdiv_rn(__fsub_rn(p.y, g.r1), g.v1));
        const float cz = floorf(__fdiv_rn(__fsub_rn(p.z, g.r2), g.v2));
        const int t = (int)p.w;                                             // int() truncation, :46
        const bool ok = cx >= 0.0f && cx < (float)g.nx && cy >= 0.0f && cy < (float)g.ny &&
                        cz >= 0.0f && cz < (float)g.nz && t >= 0 && t < g.nt;   // NaN fails every compare
        uint32_t key = VOX_INVALID;
        if (ok) {
            key = (uint32_t)((((int)cz * g.ny + (int)cy) * g.nx + (int)cx) * g.nt + t);
            atomicMin(&table[key], (uint32_t)i);
        }
        keys[i] = key;
    }
}

__device__ __forceinline__ int vox_is_first(const uint32_t *keys, const uint32_t *table, int64_t i, int64_t n,
                                            uint32_t *key_out)
{
    if (i >= n) { *key_out = VOX_INVALID; return 0; }
    const uint32_t key = keys[i];
    *key_out = key;
    return key != VOX_INVALID && table[key] == (uint32_t)i;
}

// pass 2: number of first-touch points per 2048-point chunk (wave ballot + popcount)
// first_bits (may be NULL): one 64-bit word per 64 points, bit = "this point is the first of its cell" -- the ranking pass of the batched voxeliser reads
// it back instead of gathering the table word of every point a second time ([r5]: 3.2 M random 4-byte reads = most of that pass's 4.7 x algorithmic bytes)
__global__ __launch_bounds__(256) void vox_count(const uint32_t *__restrict__ keys, const uint32_t *table,
                                                 int64_t n, int *chunk_sums, unsigned long long *__restrict__ first_bits = nullptr)
{
    __shared__ int lds[4];
    // [r6] chunk = the XCD-contiguous logical id of this workgroup (common.h): an XCD then walks one contiguous eighth of the points -- half a sample of a
    // four-sample batch -- and the 1.7 MB first-touch table of that sample stays in its 4 MB L2; in launch order every XCD saw every sample's table
    // (6.6 MB) and the random 4-byte table reads went out as 64-byte sector fetches: 4.99 x the algorithmic bytes (round 5's counters)
    const int chunk = pcacc_xcd_block(blockIdx.x, gridDim.x);
    const int64_t base = (int64_t)chunk * PCACC_CHUNK;
    int acc = 0;
#pragma unroll
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        uint32_t key;
        const int64_t i = base + r * 256 + threadIdx.x;
        const int f = vox_is_first(keys, table, i, n, &key);
        const unsigned long long m = __ballot(f);
        if (first_bits && lane_id() == 0 && i < n) first_bits[i >> 6] = m;      // lane 0 holds the lowest index of the wave's 64 (aligned: 256-thread rows of 2048-point chunks)
        acc += __popcll(m);                                 // same value in all 64 lanes
    }
    if (lane_id() == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[chunk] = lds[0] + lds[1] + lds[2] + lds[3];
}

// pass 3: rank the firsts, emit coordinates, and overwrite table[cell] with (RANK_BIT | pillar id).
// Safe while other lanes still test table[key] == j: only the first point i of a cell writes it, every
// other point j != i sees either i or a value >= 2^31, never j.
__global__ __launch_bounds__(256) void vox_assign(const uint32_t *__restrict__ keys, uint32_t *table, int64_t n,
                                                  const int *chunk_offsets, VoxGeom g, int max_voxels,
                                                  int32_t *coords)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * PCACC_CHUNK;
    int carry = chunk_offsets[blockIdx.x];
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        uint32_t key;
        const int f = vox_is_first(keys, table, base + r * 256 + threadIdx.x, n, &key);
        int tot;
        const int rank = carry + block256_exclusive_scan(f, lds, &tot);
        carry += tot;
        if (f) {
            if (rank < max_voxels) {
                const int t = key % g.nt;
                uint32_t q = key / g.nt;
                const int x = q % g.nx; q /= g.nx;
                const int y = q % g.ny;
                const int z = q / g.ny;
                reinterpret_cast<int4 *>(coords)[rank] = make_int4(z, y, x, t);
                table[key] = VOX_RANK_BIT | (uint32_t)rank;
            } else {
                table[key] = VOX_INVALID;                   // voxel_generator.py:52-53: cap reached, dropped
            }
        }
    }
}

// pass 4: point -> pillar id
__global__ __launch_bounds__(256) void vox_p2v(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ table,
                                               int64_t n, int32_t *__restrict__ p2v)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t key = keys[i];
        int32_t v = -1;
        if (key != VOX_INVALID) {
            const uint32_t e = table[key];
            if (e != VOX_INVALID) v = (int32_t)(e & ~VOX_RANK_BIT);
        }
        p2v[i] = v;
    }
}

extern "C" int pcacc_voxelize_workspace_bytes(int64_t n, int nx, int ny, int nz, int nt, size_t *bytes)
{
    if (!bytes || n < 0 || nx <= 0 || ny <= 0 || nz <= 0 || nt <= 0) return PCACC_E_ARG;
    const size_t cells = (size_t)nx * ny * nz * nt;
    *bytes = pcacc_align(cells * 4) + pcacc_align((size_t)n * 4) + pcacc_align((size_t)(pcacc_chunks(n) + 1) * 4);
    return PCACC_OK;
}

extern "C" int pcacc_voxelize(const float *points, int64_t n, const float *voxel_size, const float *range,
                              int nx, int ny, int nz, int nt, int max_voxels,
                              int32_t *coords, int32_t *p2v, int32_t *num_voxels,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    size_t need;
    if (pcacc_voxelize_workspace_bytes(n, nx, ny, nz, nt, &need) != PCACC_OK) return PCACC_E_ARG;
    if (!voxel_size || !range || !num_voxels || (n > 0 && (!points || !p2v || !coords))) return PCACC_E_ARG;
    if ((int64_t)nx * ny * nz * nt >= 0x7fffffffLL || n >= 0x7fffffffLL) return PCACC_E_ARG;
    if (!workspace || workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    const size_t cells = (size_t)nx * ny * nz * nt;
    char *ws = static_cast<char *>(workspace);
    uint32_t *table = reinterpret_cast<uint32_t *>(ws);
    uint32_t *keys = reinterpret_cast<uint32_t *>(ws + pcacc_align(cells * 4));
    int *sums = reinterpret_cast<int *>(ws + pcacc_align(cells * 4) + pcacc_align((size_t)n * 4));
    if (n == 0) {
        if (hipMemsetAsync(num_voxels, 0, 4, s) != hipSuccess) return PCACC_E_LAUNCH;
        return PCACC_OK;
    }
    VoxGeom g{range[0], range[1], range[2], voxel_size[0], voxel_size[1], voxel_size[2], nx, ny, nz, nt};
    if (hipMemsetAsync(table, 0xFF, cells * 4, s) != hipSuccess) return PCACC_E_LAUNCH;
    const int chunks = pcacc_chunks(n);
    vox_keys<<<pcacc_grid(n, 256), 256, 0, s>>>(reinterpret_cast<const float4 *>(points), n, g, keys, table);
    vox_count<<<chunks, 256, 0, s>>>(keys, table, n, sums);
    scan_chunk_sums<<<1, 1024, 0, s>>>(sums, chunks, num_voxels, max_voxels);
    vox_assign<<<chunks, 256, 0, s>>>(keys, table, n, sums, g, max_voxels, coords);
    vox_p2v<<<pcacc_grid(n, 256), 256, 0, s>>>(keys, table, n, p2v);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---- A1 + A2 for a whole batch: libs/dataset.py:183-199 per sample (voxelisation) and libs/dataloader.py:7-40 (collate_fn) in one pass -------------
// The reference voxelises every sample on a DataLoader worker and concatenates the per-sample arrays in collate_fn, giving every sample's pillar ids
// the offset of the pillars before it.  For samples that already live in HBM this used to be ~70 launches per 4-sequence batch (a float4 copy and six
// voxeliser launches per sample, then one torch.cat per key): 0.30 ms of voxeliser kernels and ~1 ms of small copies per step.  Here: ONE pass over the
// points of all samples copies them into the collated layout and enters them into a first-touch table with one table per sample; since the samples
// are concatenated in order, the global first-touch rank of a cell IS "rank inside its sample + pillars of the samples before it" -- the collated
// point_to_voxel_map -- and the collated `coordinates` rows (b, z, y, x, t) in float64 are written by the ranking pass itself.
#define VOX_MAX_SAMPLES 16
struct VoxBatch {
    const double *points[VOX_MAX_SAMPLES];      // [n_b, 3] f64
    const int64_t *time[VOX_MAX_SAMPLES];       // [n_b] i64 frame index
    const int64_t *sd[VOX_MAX_SAMPLES], *inst[VOX_MAX_SAMPLES], *fb[VOX_MAX_SAMPLES];   // [n_b] i64 labels (NULL: key absent)
    int64_t start[VOX_MAX_SAMPLES + 1];         // first collated row of sample b; start[n_samples] = N
    int n_samples;
};

__device__ __forceinline__ int vox_sample_of(const VoxBatch &b, int64_t i)
{
    int s = 0;
#pragma unroll
    for (int k = 1; k < VOX_MAX_SAMPLES; ++k) s += (k < b.n_samples && i >= b.start[k]) ? 1 : 0;
    return s;
}

__global__ __launch_bounds__(256) void vox_batch_keys(VoxBatch b, int64_t n, VoxGeom g, uint32_t cells, double *__restrict__ points_out,
                                                      double *__restrict__ time_out, int64_t *__restrict__ sd_out, int64_t *__restrict__ inst_out,
                                                      int64_t *__restrict__ fb_out, uint32_t *__restrict__ keys, uint32_t *table)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int s = vox_sample_of(b, i);
        const int64_t j = i - b.start[s];
        const double x = b.points[s][3 * j], y = b.points[s][3 * j + 1], z = b.points[s][3 * j + 2];
        const int64_t t64 = b.time[s][j];
        points_out[3 * i] = x; points_out[3 * i + 1] = y; points_out[3 * i + 2] = z;
        time_out[2 * i] = (double)s; time_out[2 * i + 1] = (double)t64;
        if (sd_out) sd_out[i] = b.sd[s][j];
        if (inst_out) inst_out[i] = b.inst[s][j];
        if (fb_out) fb_out[i] = b.fb[s][j];
        // the voxeliser's input row is (float) of the f64 point and of the frame index (libs/dataset.py:196 hands float32 points over)
        const float px = (float)x, py = (float)y, pz = (float)z, pw = (float)t64;
        const float cx = floorf(__fdiv_rn(__fsub_rn(px, g.r0), g.v0));
        const float cy = floorf(__fdiv_rn(__fsub_rn(py, g.r1), g.v1));
        const float cz = floorf(__fdiv_rn(__fsub_rn(pz, g.r2), g.v2));
        const int t = (int)pw;
        const bool ok = cx >= 0.0f && cx < (float)g.nx && cy >= 0.0f && cy < (float)g.ny && cz >= 0.0f && cz < (float)g.nz && t >= 0 && t < g.nt;
        uint32_t key = VOX_INVALID;
        if (ok) {
            key = (uint32_t)s * cells + (uint32_t)((((int)cz * g.ny + (int)cy) * g.nx + (int)cx) * g.nt + t);
            atomicMin(&table[key], (uint32_t)i);
        }
        keys[i] = key;
    }
}

// ranks the firsts like vox_assign; a row of `coordinates` is (sample, z, y, x, t) in float64; rank_at_start[s] = number of pillars before sample s
__global__ __launch_bounds__(256) void vox_batch_assign(const uint32_t *__restrict__ keys, uint32_t *table, int64_t n, const int *chunk_offsets, VoxGeom g,
                                                        uint32_t cells, VoxBatch b, double *__restrict__ coords, int *__restrict__ rank_at_start,
                                                        const unsigned long long *__restrict__ first_bits)
{
    __shared__ int lds[4];
    const int chunk = pcacc_xcd_block(blockIdx.x, gridDim.x);     // see vox_count: the table writes of a sample meet in one L2
    const int64_t base = (int64_t)chunk * PCACC_CHUNK;
    int carry = chunk_offsets[chunk];
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        // the first-touch flags of this wave's 64 points: one word written by vox_count (same value in every lane: a scalar load)
        const int f = i < n ? (int)((first_bits[i >> 6] >> (i & 63)) & 1ull) : 0;
        int tot;
        const int rank = carry + block256_exclusive_scan(f, lds, &tot);
        carry += tot;
        if (i < n) {
#pragma unroll
            for (int k = 0; k < VOX_MAX_SAMPLES; ++k)
                if (k < b.n_samples && i == b.start[k]) rank_at_start[k] = rank;      // pillars of the samples before this one
        }
        if (f) {
            const uint32_t key = keys[i];
            const uint32_t s = key / cells;
            uint32_t q = key - s * cells;
            const int t = q % g.nt; q /= g.nt;
            const int x = q % g.nx; q /= g.nx;
            const int y = q % g.ny;
            const int z = q / g.ny;
            double *row = coords + (int64_t)rank * 5;
            row[0] = (double)s; row[1] = (double)z; row[2] = (double)y; row[3] = (double)x; row[4] = (double)t;
            table[key] = VOX_RANK_BIT | (uint32_t)rank;
        }
    }
}

// point -> collated pillar id.  A point outside the grid has -1 in its own sample; collate_fn adds the sample's pillar offset to EVERY entry
// (libs/dataloader.py:29-31), so the collated value is offset - 1 -- reproduced here (the data step crops to the grid, such points do not occur in practice)
__global__ __launch_bounds__(256) void vox_batch_p2v(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ table, int64_t n, VoxBatch b,
                                                     const int *__restrict__ rank_at_start, int32_t *__restrict__ p2v)
{
    // [r6] a workgroup takes a CONTIGUOUS range of the points, the ranges in XCD-contiguous order (see vox_count): 2.63 x the algorithmic bytes with the grid-stride walk
    const int64_t per = (((n + gridDim.x - 1) / gridDim.x) + 255) / 256 * 256;
    const int64_t lo = (int64_t)pcacc_xcd_block(blockIdx.x, gridDim.x) * per, hi = min(n, lo + per);
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const uint32_t key = keys[i];
        int32_t v;
        if (key != VOX_INVALID) v = (int32_t)(table[key] & ~VOX_RANK_BIT);
        else v = rank_at_start[vox_sample_of(b, i)] - 1;
        p2v[i] = v;
    }
}

__global__ void vox_batch_counts(const int *rank_at_start, const int *total, int n_samples, int32_t *num_voxels)
{
    const int s = threadIdx.x;
    if (s < n_samples) num_voxels[s] = (s + 1 < n_samples ? rank_at_start[s + 1] : *total) - rank_at_start[s];
}

extern "C" int pcacc_collate_voxelize_workspace_bytes(int64_t n, int32_t n_samples, int nx, int ny, int nz, int nt, size_t *bytes)
{
    if (!bytes || n < 0 || n_samples < 1 || n_samples > VOX_MAX_SAMPLES || nx <= 0 || ny <= 0 || nz <= 0 || nt <= 0) return PCACC_E_ARG;
    const size_t cells = (size_t)nx * ny * nz * nt * n_samples;
    *bytes = pcacc_align(cells * 4) + pcacc_align((size_t)n * 4) + pcacc_align((size_t)(pcacc_chunks(n) + 1) * 4) + pcacc_align((VOX_MAX_SAMPLES + 1) * 4) +
             pcacc_align(((size_t)n / 64 + 1) * 8);
    return PCACC_OK;
}

// points[b] [n_b,3] f64, time[b] [n_b] i64, sd / inst / fb [b] [n_b] i64 (the three label arrays may be NULL as a whole): DEVICE pointers in HOST arrays;
// counts host int64 [n_samples] (every n_b >= 1).  Outputs (device): points_out [N,3] f64, time_out [N,2] f64 (sample, frame), sd_out / inst_out /
// fb_out [N] i64 (NULL with their inputs), coords_out [>= pillars, 5] f64 rows (sample, z, y, x, t) in first-touch order -- size it for N rows --,
// p2v_out [N] i32 (row of coords_out, -1 outside the grid), num_voxels [n_samples] i32.
extern "C" int pcacc_collate_voxelize(const double *const *points, const int64_t *const *time, const int64_t *const *sd, const int64_t *const *inst,
                                      const int64_t *const *fb, const int64_t *counts, int32_t n_samples, const float *voxel_size, const float *range,
                                      int nx, int ny, int nz, int nt, double *points_out, double *time_out, int64_t *sd_out, int64_t *inst_out,
                                      int64_t *fb_out, double *coords_out, int32_t *p2v_out, int32_t *num_voxels, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    if (!points || !time || !counts || !voxel_size || !range || !points_out || !time_out || !coords_out || !p2v_out || !num_voxels) return PCACC_E_ARG;
    if (n_samples < 1 || n_samples > VOX_MAX_SAMPLES) return PCACC_E_ARG;
    VoxBatch b;
    b.n_samples = n_samples;
    int64_t n = 0;
    for (int s = 0; s < VOX_MAX_SAMPLES; ++s) {
        const bool in = s < n_samples;
        if (in && (counts[s] < 1 || !points[s] || !time[s] || (sd && !sd[s]) || (inst && !inst[s]) || (fb && !fb[s]))) return PCACC_E_ARG;
        b.points[s] = in ? points[s] : nullptr;
        b.time[s] = in ? time[s] : nullptr;
        b.sd[s] = in && sd ? sd[s] : nullptr;
        b.inst[s] = in && inst ? inst[s] : nullptr;
        b.fb[s] = in && fb ? fb[s] : nullptr;
        b.start[s] = n;
        if (in) n += counts[s];
    }
    for (int s = n_samples; s <= VOX_MAX_SAMPLES; ++s) b.start[s] = n;
    b.start[n_samples] = n;
    size_t need;
    if (pcacc_collate_voxelize_workspace_bytes(n, n_samples, nx, ny, nz, nt, &need) != PCACC_OK) return PCACC_E_ARG;
    const size_t cells1 = (size_t)nx * ny * nz * nt, cells = cells1 * n_samples;
    if (cells >= 0x7fffffffULL || n >= 0x7fffffffLL) return PCACC_E_ARG;
    if ((sd && !sd_out) || (inst && !inst_out) || (fb && !fb_out)) return PCACC_E_ARG;
    if (!workspace || workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t st = pcacc_stream(stream);
    char *ws = static_cast<char *>(workspace);
    uint32_t *table = reinterpret_cast<uint32_t *>(ws);
    uint32_t *keys = reinterpret_cast<uint32_t *>(ws + pcacc_align(cells * 4));
    int *sums = reinterpret_cast<int *>(ws + pcacc_align(cells * 4) + pcacc_align((size_t)n * 4));
    int *rank_at_start = reinterpret_cast<int *>(ws + pcacc_align(cells * 4) + pcacc_align((size_t)n * 4) + pcacc_align((size_t)(pcacc_chunks(n) + 1) * 4));
    int *total = rank_at_start + VOX_MAX_SAMPLES;
    unsigned long long *first_bits = reinterpret_cast<unsigned long long *>(ws + pcacc_align(cells * 4) + pcacc_align((size_t)n * 4) +
                                                                           pcacc_align((size_t)(pcacc_chunks(n) + 1) * 4) + pcacc_align((VOX_MAX_SAMPLES + 1) * 4));
    VoxGeom g{range[0], range[1], range[2], voxel_size[0], voxel_size[1], voxel_size[2], nx, ny, nz, nt};
    if (hipMemsetAsync(table, 0xFF, cells * 4, st) != hipSuccess) return PCACC_E_LAUNCH;
    const int chunks = pcacc_chunks(n);
    vox_batch_keys<<<pcacc_grid(n, 256), 256, 0, st>>>(b, n, g, (uint32_t)cells1, points_out, time_out, sd ? sd_out : nullptr, inst ? inst_out : nullptr,
                                                        fb ? fb_out : nullptr, keys, table);
    vox_count<<<chunks, 256, 0, st>>>(keys, table, n, sums, first_bits);
    scan_chunk_sums<<<1, 1024, 0, st>>>(sums, chunks, total, -1);
    vox_batch_assign<<<chunks, 256, 0, st>>>(keys, table, n, sums, g, (uint32_t)cells1, b, coords_out, rank_at_start, first_bits);
    vox_batch_p2v<<<pcacc_grid(n, 256), 256, 0, st>>>(keys, table, n, b, rank_at_start, p2v_out);
    vox_batch_counts<<<1, 64, 0, st>>>(rank_at_start, total, n_samples, num_voxels);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
