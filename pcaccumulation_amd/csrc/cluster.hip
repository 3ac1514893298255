// Test-mode instance clustering on the device (models/cluster.py:9-111): 5 cm voxel down-sampling of the points
// predicted moving, DBSCAN in the horizontal plane, small clusters dropped, labels canonicalised per sample.
// The reference does this on the host (torchsparse sparse_quantize + scikit-learn DBSCAN) behind a
// device->host->device round trip; here every sample of the batch is clustered in one set of launches and the
// result is bit-identical to that host path:
//
//   voxel down-sampling   key = (sample, vx, vy, vz) packed in 64 bits, stable radix sort -> the kept points come
//                         out in the order np.unique gives the ravel hash (lexicographic voxel order) and each voxel
//                         keeps its first (lowest-index) point, like sparse_quantize(return_index, return_inverse).
//   neighbourhoods        uniform grid with cells 0.7 eps wide, second radix sort by cell; a point scans the 5x5 cells
//                         around it (5 contiguous runs).  The distance test is the KD-tree's: float64
//                         dx*dx + dy*dy (summed in that order, no FMA) <= eps*eps, the point itself included.
//                         Points of one cell are always in range of each other, which bounds the work on dense
//                         objects: a full cell makes all its points core at once, and one edge joins two cells.
//   clusters              lock-free union-find over core-core edges (on grid positions, so the candidate loops read
//                         contiguous memory), then the lowest kept-point index among each component's core points.
//                         scikit-learn numbers clusters in the order of their lowest core index and finishes cluster k
//                         before starting k+1, hence a border point belongs to the lowest-numbered cluster with a core
//                         point in range = the smallest such index among its core neighbours.  None of this depends
//                         on visiting order.
//   labels                clusters with fewer than min_p_cluster (down-sampled) points -> 0; survivors are ranked per
//                         sample with a prefix sum (canonicalise_random_indice) and expanded through the inverse map.
//
// Sizes that the host path reads back (number of kept voxels, per-sample counts) stay on the device: every launch is
// sized by the input length and guards on the device-side count.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "scan.h"

#define CL_BLOCK 256
#define CL_INVALID 0xffffffffffffffffull
#define CL_VBITS 19                       // per-axis voxel field; |coordinate| < 2^18 voxels (13 km at 5 cm)
#define CL_VBIAS (1 << 18)
#define CL_GBITS 17                       // per-axis grid-cell field: the grid key is 6 + 2*17 = 40 bits = 5 radix passes
#define CL_GBIAS (1 << 16)
#define CL_GSHIFT (2 * CL_GBITS)          // sample index of the grid key
#define CL_GINVALID ((1ull << (CL_GSHIFT + 6)) - 1)
#define CL_BSHIFT 57                      // sample index in the top bits: n_batches <= 64

struct ClusterWs {
    uint64_t *key_a, *key_b, *gk_a, *gk_b;
    int *idx_a, *idx_b, *gi_a, *gi_b;
    int *head, *vox;                      // vox[n] = number of kept voxels M
    float *sx, *sy, *gpx, *gpy;           // kept points by voxel rank / by grid order
    int *cell_lo, *cell_hi;               // extent of a point's own cell, by grid position
    int *near_lo, *near_hi, *far_lo, *far_hi;   // [3][n] / [5][n] candidate ranges in grid order
    int *core, *parent, *root_min, *comp, *size, *flag, *rank;   // core/parent/root_min by grid position
    int *chunk;                           // chunk sums scratch
    int *base, *gate;                     // per sample
    void *sort_tmp;
    size_t sort_tmp_bytes;
};

static size_t cluster_sort_tmp_bytes(int64_t n)
{
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int *)nullptr, (int *)nullptr,
                                  (size_t)n, 0, 64, (hipStream_t)0) != hipSuccess)
        return 0;
    return bytes;
}

static size_t cluster_carve(ClusterWs *w, char *p, int64_t n)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char *q = p ? p + off : nullptr; off += pcacc_align(bytes); return q; };
    const size_t n1 = (size_t)n + 1;
    w->key_a = (uint64_t *)take(n1 * 8); w->key_b = (uint64_t *)take(n1 * 8);
    w->gk_a = (uint64_t *)take(n1 * 8);  w->gk_b = (uint64_t *)take(n1 * 8);
    w->idx_a = (int *)take(n1 * 4); w->idx_b = (int *)take(n1 * 4);
    w->gi_a = (int *)take(n1 * 4);  w->gi_b = (int *)take(n1 * 4);
    w->head = (int *)take(n1 * 4);  w->vox = (int *)take(n1 * 4);
    w->sx = (float *)take(n1 * 4);  w->sy = (float *)take(n1 * 4);
    w->gpx = (float *)take(n1 * 4); w->gpy = (float *)take(n1 * 4);
    w->cell_lo = (int *)take(n1 * 4); w->cell_hi = (int *)take(n1 * 4);
    w->near_lo = (int *)take(3 * n1 * 4); w->near_hi = (int *)take(3 * n1 * 4);
    w->far_lo = (int *)take(5 * n1 * 4); w->far_hi = (int *)take(5 * n1 * 4);
    w->core = (int *)take(n1 * 4); w->parent = (int *)take(n1 * 4); w->root_min = (int *)take(n1 * 4); w->comp = (int *)take(n1 * 4);
    w->size = (int *)take(n1 * 4); w->flag = (int *)take(n1 * 4); w->rank = (int *)take(n1 * 4);
    w->chunk = (int *)take(((size_t)pcacc_chunks(n) + 2) * 4);
    w->base = (int *)take(65 * 4); w->gate = (int *)take(65 * 4);
    w->sort_tmp_bytes = cluster_sort_tmp_bytes(n);
    w->sort_tmp = take(w->sort_tmp_bytes);
    return off;
}

extern "C" int pcacc_cluster_workspace_bytes(int64_t n, size_t *bytes)
{
    if (n < 0 || n >= (1ll << 31) - 1 || !bytes) return PCACC_E_ARG;
    ClusterWs w;
    *bytes = cluster_carve(&w, nullptr, n < 1 ? 1 : n);
    return w.sort_tmp_bytes ? 0 : PCACC_E_LAUNCH;       // the radix-sort size query needs a device
}

__device__ __forceinline__ uint64_t cl_field(int v, int bias, int bits)
{
    int64_t t = (int64_t)v + bias;
    const int64_t hi = (1ll << bits) - 2;     // the all-ones pattern is left to the invalid key
    t = t < 0 ? 0 : (t > hi ? hi : t);
    return (uint64_t)t;
}

// ---- 1. voxel keys (cluster.py:9-13,67-79: round(points / voxel_size), then floor->int32 inside sparse_quantize) ----
__global__ __launch_bounds__(CL_BLOCK) void cluster_voxel_keys(const float *__restrict__ pts, const float *__restrict__ off,
                                                               const uint8_t *__restrict__ sel, const int *__restrict__ batch,
                                                               int64_t n, int n_batches, float voxel, uint64_t *key, int *idx)
{
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * CL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CL_BLOCK) {
        uint64_t k = CL_INVALID;
        const int b = batch[i];
        if (sel[i] && b >= 0 && b < n_batches) {
            float x = pts[3 * i], y = pts[3 * i + 1];
            const float z = pts[3 * i + 2];
            if (off) { x += off[2 * i]; y += off[2 * i + 1]; }
            const int vx = (int)rintf(x / voxel), vy = (int)rintf(y / voxel), vz = (int)rintf(z / voxel);
            k = ((uint64_t)b << CL_BSHIFT) | (cl_field(vx, CL_VBIAS, CL_VBITS) << (2 * CL_VBITS)) |
                (cl_field(vy, CL_VBIAS, CL_VBITS) << CL_VBITS) | cl_field(vz, CL_VBIAS, CL_VBITS);
        }
        key[i] = k;
        idx[i] = (int)i;
    }
}

__global__ __launch_bounds__(CL_BLOCK) void cluster_heads(const uint64_t *__restrict__ key, int64_t n, int *head)
{
    for (int64_t j = (int64_t)blockIdx.x * CL_BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * CL_BLOCK) {
        const uint64_t k = key[j];
        head[j] = (k != CL_INVALID) && (j == 0 || key[j - 1] != k);
    }
}

// ---- 2. kept points + grid keys ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(CL_BLOCK) void cluster_reset(int64_t n, uint64_t *gk, int *gi, int *parent, int *size, int *root_min)
{
    for (int64_t j = (int64_t)blockIdx.x * CL_BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * CL_BLOCK) {
        gk[j] = CL_GINVALID;                   // slots >= M sort to the end
        gi[j] = (int)j;
        parent[j] = (int)j;
        size[j] = 0;
        root_min[j] = 0x7fffffff;
    }
}

// kept point of voxel u = the run head's point (a separate launch: slot vox[j] <= j belongs to another thread's reset)
__global__ __launch_bounds__(CL_BLOCK) void cluster_sub_points_fill(const float *__restrict__ pts, const float *__restrict__ off,
                                                                    const uint64_t *__restrict__ key, const int *__restrict__ idx,
                                                                    const int *__restrict__ head, const int *__restrict__ vox,
                                                                    int64_t n, float cell, float *sx, float *sy, uint64_t *gk)
{
#pragma clang fp contract(off)
    for (int64_t j = (int64_t)blockIdx.x * CL_BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * CL_BLOCK) {
        if (!head[j]) continue;
        const int u = vox[j];
        const int64_t i = idx[j];
        float x = pts[3 * i], y = pts[3 * i + 1];
        if (off) { x += off[2 * i]; y += off[2 * i + 1]; }
        sx[u] = x;
        sy[u] = y;
        const uint64_t b = key[j] >> CL_BSHIFT;
        const int gx = (int)floorf(x / cell), gy = (int)floorf(y / cell);
        gk[u] = (b << CL_GSHIFT) | (cl_field(gx, CL_GBIAS, CL_GBITS) << CL_GBITS) | cl_field(gy, CL_GBIAS, CL_GBITS);
    }
}

__device__ __forceinline__ int cl_lower_bound(const uint64_t *a, int n, uint64_t v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Grid-ordered coordinates, the extent of every point's own cell, and its candidate ranges: cells are 0.7 eps wide, so
// (a) two points of one cell are always within eps of each other and (b) anything within eps lies at most 2 cells away.
// Row r of `near` covers cells (gx + r-1, gy-1 .. gy+1); row r of `far` covers (gx + CL_ROW_DX[r], gy-2 .. gy+2), own row first.
__constant__ int CL_ROW_DX[5] = {0, -1, 1, -2, 2};

__global__ __launch_bounds__(CL_BLOCK) void cluster_rows(const uint64_t *__restrict__ gk, const int *__restrict__ g2u,
                                                         const float *__restrict__ sx, const float *__restrict__ sy,
                                                         const int *__restrict__ m_ptr, int64_t n, float *gpx, float *gpy,
                                                         int *cell_lo, int *cell_hi, int *near_lo, int *near_hi, int *far_lo, int *far_hi)
{
    const int m = *m_ptr;
    for (int k = blockIdx.x * CL_BLOCK + threadIdx.x; k < m; k += gridDim.x * CL_BLOCK) {
        const int u = g2u[k];
        gpx[k] = sx[u];
        gpy[k] = sy[u];
        const uint64_t key = gk[k];
        const uint64_t gy = key & ((1ull << CL_GBITS) - 1);
        const uint64_t hi_part = key >> CL_GBITS;                  // (sample, gx); the bias keeps gx, gy >= 2
        cell_lo[k] = cl_lower_bound(gk, m, key);
        cell_hi[k] = cl_lower_bound(gk, m, key + 1);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint64_t row = (hi_part + r - 1) << CL_GBITS;
            near_lo[(int64_t)r * n + k] = cl_lower_bound(gk, m, row | (gy - 1));
            near_hi[(int64_t)r * n + k] = cl_lower_bound(gk, m, (row | (gy + 1)) + 1);
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const uint64_t row = (hi_part + CL_ROW_DX[r]) << CL_GBITS;
            far_lo[(int64_t)r * n + k] = cl_lower_bound(gk, m, row | (gy - 2));
            far_hi[(int64_t)r * n + k] = cl_lower_bound(gk, m, (row | (gy + 2)) + 1);
        }
    }
}

// the KD-tree leaf test of sklearn.neighbors (rdist <= r*r on the float64 copy of the float32 coordinates)
__device__ __forceinline__ bool cl_in_range(float ax, float ay, float bx, float by, double r2)
{
#pragma clang fp contract(off)
    const double dx = (double)ax - (double)bx, dy = (double)ay - (double)by;
    double d = dx * dx;
    d = d + dy * dy;
    return d <= r2;
}

// ---- 3. core points (flags kept in GRID order: the candidate loops below then read contiguous memory) ---------------
__global__ __launch_bounds__(CL_BLOCK) void cluster_core(const float *__restrict__ gpx, const float *__restrict__ gpy,
                                                         const int *__restrict__ cell_lo, const int *__restrict__ cell_hi,
                                                         const int *__restrict__ far_lo, const int *__restrict__ far_hi,
                                                         const int *__restrict__ m_ptr, int64_t n, double r2, int min_samples,
                                                         int *core_g)
{
    const int m = *m_ptr;
    for (int k = blockIdx.x * CL_BLOCK + threadIdx.x; k < m; k += gridDim.x * CL_BLOCK) {
        int cnt = cell_hi[k] - cell_lo[k];                 // the own cell is in range as a whole
        if (cnt < min_samples) {
            const float x = gpx[k], y = gpy[k];
            cnt = 0;
            for (int r = 0; r < 5 && cnt < min_samples; ++r) {
                const int lo = far_lo[(int64_t)r * n + k], hi = far_hi[(int64_t)r * n + k];
                for (int c = lo; c < hi && cnt < min_samples; ++c) cnt += cl_in_range(x, y, gpx[c], gpy[c], r2) ? 1 : 0;
            }
        }
        core_g[k] = cnt >= min_samples;
    }
}

// ---- 4. union-find over core-core edges, on grid positions ----------------------------------------------------------------
__device__ __forceinline__ int uf_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int *parent, int x)
{
    int p = uf_load(&parent[x]);
    while (p != x) {
        const int gp = uf_load(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving
        x = p;
        p = gp;
    }
    return x;
}

// joins the sets of a and b; returns the surviving (smaller) root
__device__ __forceinline__ int uf_union(int *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return a;
        if (a < b) { const int t = a; a = b; b = t; }
        if (atomicCAS(&parent[a], a, b) == a) return b;                       // larger root goes under the smaller one
    }
}

// 4a. the core points of one cell are mutually in range: hang each under the cell's first core point (no atomics: a point
//     only writes its own, still untouched, slot)
__global__ __launch_bounds__(CL_BLOCK) void cluster_link_cell(const int *__restrict__ cell_lo, const int *__restrict__ core_g,
                                                              const int *__restrict__ m_ptr, int *parent)
{
    const int m = *m_ptr;
    for (int k = blockIdx.x * CL_BLOCK + threadIdx.x; k < m; k += gridDim.x * CL_BLOCK) {
        if (!core_g[k]) continue;
        int c = cell_lo[k];
        while (!core_g[c]) ++c;                                               // stops at k at the latest
        if (c < k) parent[k] = c;
    }
}

// 4b/4c. one edge between two cells joins all their core points, so a cell is left as soon as it is known to be in the
//     point's set (root check on its first core point) or one in-range core point has been joined.  Launched first over
//     the 8 adjacent cells, where an in-range partner turns up within a few candidates, then over the full 5x5 block,
//     where by then almost every cell of a dense object already passes the root check.
template <int ROWS>
__global__ __launch_bounds__(CL_BLOCK) void cluster_link(const float *__restrict__ gpx, const float *__restrict__ gpy,
                                                         const int *__restrict__ cell_hi, const int *__restrict__ row_lo,
                                                         const int *__restrict__ row_hi, const int *__restrict__ core_g,
                                                         const int *__restrict__ m_ptr, int64_t n, double r2, int *parent)
{
    const int m = *m_ptr;
    for (int k = blockIdx.x * CL_BLOCK + threadIdx.x; k < m; k += gridDim.x * CL_BLOCK) {
        if (!core_g[k]) continue;
        const float x = gpx[k], y = gpy[k];
        int rk = uf_find(parent, k);
        for (int r = 0; r < ROWS; ++r) {
            int c = row_lo[(int64_t)r * n + k];
            const int hi = row_hi[(int64_t)r * n + k];
            while (c < hi) {
                if (!core_g[c]) { ++c; continue; }
                const int end = cell_hi[c];
                if (uf_find(parent, c) != rk) {
                    for (int d = c; d < end; ++d) {
                        if (core_g[d] && cl_in_range(x, y, gpx[d], gpy[d], r2)) {
                            rk = uf_union(parent, rk, d);
                            break;
                        }
                    }
                }
                c = end;
            }
        }
    }
}

// after the joins: every core point directly under its root, so the passes below need one load instead of a walk
__global__ __launch_bounds__(CL_BLOCK) void cluster_flatten(const int *__restrict__ core_g, const int *__restrict__ m_ptr, int *parent)
{
    const int m = *m_ptr;
    for (int k = blockIdx.x * CL_BLOCK + threadIdx.x; k < m; k += gridDim.x * CL_BLOCK)
        if (core_g[k]) {
            const int r = uf_find(parent, k);
            if (r != k) __hip_atomic_store(&parent[k], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
}

// Neighbouring lanes mostly hold points of the same object: combine their contributions per distinct key inside the wave
// and issue one atomic per key (a dense object would otherwise send tens of thousands of atomics to one address).
// Both helpers must be reached by all 64 lanes.
__device__ __forceinline__ void wave_min_by_key(int *table, int key, int val, bool valid)
{
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int v = __shfl(key, leader, 64);
        const bool mine = valid && key == v;
        int best = mine ? val : 0x7fffffff;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(best, d, 64);
            best = o < best ? o : best;
        }
        if (lane_id() == leader && uf_load(&table[v]) > best) atomicMin(&table[v], best);
        todo &= ~__ballot(mine);
    }
}

__device__ __forceinline__ void wave_count_by_key(int *table, int key, bool valid)
{
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int v = __shfl(key, leader, 64);
        const unsigned long long same = __ballot(valid && key == v);
        if (lane_id() == leader) atomicAdd(&table[v], (int)__popcll(same));
        todo &= ~same;
    }
}

// lowest voxel rank among the core points of every component (scikit-learn numbers clusters by it)
__global__ __launch_bounds__(CL_BLOCK) void cluster_root_min(const int *__restrict__ g2u, const int *__restrict__ core_g,
                                                             const int *__restrict__ m_ptr, const int *__restrict__ parent,
                                                             int *root_min)
{
    const int m = *m_ptr;
    for (int base = blockIdx.x * CL_BLOCK; base < m; base += gridDim.x * CL_BLOCK) {
        const int k = base + threadIdx.x;
        const bool valid = k < m && core_g[k];
        wave_min_by_key(root_min, valid ? parent[k] : 0, valid ? g2u[k] : 0, valid);
    }
}

// ---- 5. component of every kept point (core: its own; border: the lowest among core neighbours; noise: -1) ----------
__global__ __launch_bounds__(CL_BLOCK) void cluster_assign(const float *__restrict__ gpx, const float *__restrict__ gpy,
                                                           const int *__restrict__ g2u, const int *__restrict__ cell_hi,
                                                           const int *__restrict__ far_lo, const int *__restrict__ far_hi,
                                                           const int *__restrict__ core_g, const int *__restrict__ root_min,
                                                           const int *__restrict__ m_ptr, int64_t n, double r2,
                                                           const int *__restrict__ parent, int *comp, int *size)
{
    const int m = *m_ptr;
    for (int base = blockIdx.x * CL_BLOCK; base < m; base += gridDim.x * CL_BLOCK) {
        const int k = base + threadIdx.x;
        int best = 0x7fffffff;
        if (k < m) {
            if (core_g[k]) {
                best = root_min[parent[k]];
            } else {
                const float x = gpx[k], y = gpy[k];
                for (int r = 0; r < 5; ++r) {
                    int c = far_lo[(int64_t)r * n + k];
                    const int hi = far_hi[(int64_t)r * n + k];
                    while (c < hi) {
                        if (!core_g[c]) { ++c; continue; }
                        const int end = cell_hi[c];
                        for (int d = c; d < end; ++d) {                       // a cell's core points share one component
                            if (core_g[d] && cl_in_range(x, y, gpx[d], gpy[d], r2)) {
                                const int cand = root_min[parent[d]];
                                best = cand < best ? cand : best;
                                break;
                            }
                        }
                        c = end;
                    }
                }
            }
            comp[g2u[k]] = best == 0x7fffffff ? -1 : best;
        }
        // size[] is only ever touched at a component's lowest core point
        wave_count_by_key(size, best, best != 0x7fffffff);
    }
}

// ---- 6. surviving clusters, per-sample rank, expansion through the inverse map -----------------------------------------
__global__ __launch_bounds__(CL_BLOCK) void cluster_survivors(const int *__restrict__ size, const int *__restrict__ m_ptr, int64_t n,
                                                              int min_p_cluster, int *flag)
{
    const int m = *m_ptr;
    for (int64_t u = (int64_t)blockIdx.x * CL_BLOCK + threadIdx.x; u < n; u += (int64_t)gridDim.x * CL_BLOCK)
        flag[u] = (u < m) && size[u] > 0 && size[u] >= min_p_cluster;
}

// one thread per sample: rank of its first kept voxel and the `sel.sum() > min_p_cluster` gate (cluster.py:66)
__global__ void cluster_sample_bases(const uint64_t *__restrict__ key, const int *__restrict__ vox, const int *__restrict__ rank,
                                     int64_t n, int n_batches, int min_p_cluster, int *base, int *gate)
{
    const int b = threadIdx.x;
    if (b >= n_batches) return;
    const int first = cl_lower_bound(key, (int)n, (uint64_t)b << CL_BSHIFT);
    const int last = cl_lower_bound(key, (int)n, (uint64_t)(b + 1) << CL_BSHIFT);   // invalid keys sort behind every sample
    gate[b] = (last - first) > min_p_cluster;
    base[b] = rank[vox[first]];          // vox[first] = voxel rank of the sample's first point (first == n -> vox[n] = M)
}

__global__ __launch_bounds__(CL_BLOCK) void cluster_expand(const uint64_t *__restrict__ key, const int *__restrict__ idx,
                                                           const int *__restrict__ head, const int *__restrict__ vox,
                                                           const int *__restrict__ comp, const int *__restrict__ flag,
                                                           const int *__restrict__ rank, const int *__restrict__ base,
                                                           const int *__restrict__ gate, int64_t n, int64_t *labels)
{
    for (int64_t j = (int64_t)blockIdx.x * CL_BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * CL_BLOCK) {
        const uint64_t k = key[j];
        int64_t lab = 0;
        if (k != CL_INVALID) {
            const int b = (int)(k >> CL_BSHIFT);
            const int u = vox[j] + head[j] - 1;
            const int c = comp[u];
            if (c >= 0 && flag[c] && gate[b]) lab = rank[c] - base[b] + 1;
        }
        labels[idx[j]] = lab;
    }
}

static void cluster_scan(const int *in, int64_t n, int *chunk, int *out, hipStream_t st)
{
    const int chunks = pcacc_chunks(n);
    hipLaunchKernelGGL(chunk_sums_i32, dim3(chunks), dim3(256), 0, st, in, n, chunk);
    hipLaunchKernelGGL(scan_chunk_sums, dim3(1), dim3(1024), 0, st, chunk, chunks, (int *)nullptr, -1);
    hipLaunchKernelGGL(chunk_scan_i32, dim3(chunks), dim3(256), 0, st, in, n, chunk, out, 1);
}

extern "C" int pcacc_cluster(const float *points, const float *offset, const uint8_t *sel, const int32_t *batch, int64_t n,
                             int32_t n_batches, float voxel_size, double eps, int32_t min_samples, int32_t min_p_cluster,
                             int64_t *labels, void *ws, size_t ws_bytes, void *stream)
{
    if (n < 0 || n >= (1ll << 31) - 1 || n_batches < 1 || n_batches > 64 || !(voxel_size > 0.f) || !(eps > 0.0)) return PCACC_E_ARG;
    if (n == 0) return 0;
    if (!points || !sel || !batch || !labels || !ws) return PCACC_E_ARG;
    ClusterWs w;
    if (ws_bytes < cluster_carve(&w, (char *)ws, n)) return PCACC_E_WORKSPACE;
    hipStream_t st = pcacc_stream(stream);
    const int grid = pcacc_grid(n, CL_BLOCK);
    const float cell = (float)eps * 0.7f;          // see cluster_rows
    const double r2 = eps * eps;

    hipLaunchKernelGGL(cluster_voxel_keys, dim3(grid), dim3(CL_BLOCK), 0, st, points, offset, sel, batch, n, n_batches, voxel_size,
                       w.key_a, w.idx_a);
    if (rocprim::radix_sort_pairs(w.sort_tmp, w.sort_tmp_bytes, w.key_a, w.key_b, w.idx_a, w.idx_b, (size_t)n, 0, 64, st) != hipSuccess)
        return PCACC_E_LAUNCH;
    hipLaunchKernelGGL(cluster_heads, dim3(grid), dim3(CL_BLOCK), 0, st, w.key_b, n, w.head);
    cluster_scan(w.head, n, w.chunk, w.vox, st);                                   // vox[n] = M
    hipLaunchKernelGGL(cluster_reset, dim3(grid), dim3(CL_BLOCK), 0, st, n, w.gk_a, w.gi_a, w.parent, w.size, w.root_min);
    hipLaunchKernelGGL(cluster_sub_points_fill, dim3(grid), dim3(CL_BLOCK), 0, st, points, offset, w.key_b, w.idx_b, w.head, w.vox, n,
                       cell, w.sx, w.sy, w.gk_a);
    if (rocprim::radix_sort_pairs(w.sort_tmp, w.sort_tmp_bytes, w.gk_a, w.gk_b, w.gi_a, w.gi_b, (size_t)n, 0, CL_GSHIFT + 6, st) != hipSuccess)
        return PCACC_E_LAUNCH;
    const int *m_ptr = w.vox + n;
    hipLaunchKernelGGL(cluster_rows, dim3(grid), dim3(CL_BLOCK), 0, st, w.gk_b, w.gi_b, w.sx, w.sy, m_ptr, n, w.gpx, w.gpy, w.cell_lo,
                       w.cell_hi, w.near_lo, w.near_hi, w.far_lo, w.far_hi);
    hipLaunchKernelGGL(cluster_core, dim3(grid), dim3(CL_BLOCK), 0, st, w.gpx, w.gpy, w.cell_lo, w.cell_hi, w.far_lo, w.far_hi, m_ptr, n,
                       r2, min_samples, w.core);
    hipLaunchKernelGGL(cluster_link_cell, dim3(grid), dim3(CL_BLOCK), 0, st, w.cell_lo, w.core, m_ptr, w.parent);
    hipLaunchKernelGGL(cluster_link<3>, dim3(grid), dim3(CL_BLOCK), 0, st, w.gpx, w.gpy, w.cell_hi, w.near_lo, w.near_hi, w.core, m_ptr,
                       n, r2, w.parent);
    hipLaunchKernelGGL(cluster_link<5>, dim3(grid), dim3(CL_BLOCK), 0, st, w.gpx, w.gpy, w.cell_hi, w.far_lo, w.far_hi, w.core, m_ptr, n,
                       r2, w.parent);
    hipLaunchKernelGGL(cluster_flatten, dim3(grid), dim3(CL_BLOCK), 0, st, w.core, m_ptr, w.parent);
    hipLaunchKernelGGL(cluster_root_min, dim3(grid), dim3(CL_BLOCK), 0, st, w.gi_b, w.core, m_ptr, w.parent, w.root_min);
    hipLaunchKernelGGL(cluster_assign, dim3(grid), dim3(CL_BLOCK), 0, st, w.gpx, w.gpy, w.gi_b, w.cell_hi, w.far_lo, w.far_hi, w.core,
                       w.root_min, m_ptr, n, r2, w.parent, w.comp, w.size);
    hipLaunchKernelGGL(cluster_survivors, dim3(grid), dim3(CL_BLOCK), 0, st, w.size, m_ptr, n, min_p_cluster, w.flag);
    cluster_scan(w.flag, n, w.chunk, w.rank, st);                                  // rank[n] = survivors in the whole batch
    hipLaunchKernelGGL(cluster_sample_bases, dim3(1), dim3(64), 0, st, w.key_b, w.vox, w.rank, n, n_batches, min_p_cluster, w.base,
                       w.gate);
    hipLaunchKernelGGL(cluster_expand, dim3(grid), dim3(CL_BLOCK), 0, st, w.key_b, w.idx_b, w.head, w.vox, w.comp, w.flag, w.rank, w.base,
                       w.gate, n, labels);
    PCACC_CHECK_LAUNCH();
    return 0;
}
