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
__global__ __launch_bounds__(256) void vox_count(const uint32_t *__restrict__ keys, const uint32_t *table,
                                                 int64_t n, int *chunk_sums)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * PCACC_CHUNK;
    int acc = 0;
#pragma unroll
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        uint32_t key;
        const int f = vox_is_first(keys, table, base + r * 256 + threadIdx.x, n, &key);
        acc += __popcll(__ballot(f));                       // same value in all 64 lanes
    }
    if (lane_id() == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
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
