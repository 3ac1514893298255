// Point -> pillar CSR and the per-pillar reductions built on it.
//
// torch_scatter reduces with one atomic per (point, channel).  Here the points are grouped by pillar
// once per forward (counting sort: histogram, exclusive scan, cursor fill, then a tiny per-pillar
// index sort so that the order inside a pillar is ascending point index = the order a sequential CPU
// scatter visits them).  Every reduction afterwards is a contiguous, atomic-free, deterministic
// segmented loop: pillars have ~3 points on average (models/motionnet.py:142).
#include "scan.h"


#define CSR_SORT_MAX 64      // segments with more points: csr_list_long / csr_sort_long below ([r6]: ascending for ANY length)

// [r5] the counting pass keeps what its atomic returns -- the point's arrival number inside its segment -- and the fill pass places the point at
// seg_offsets[s] + that number: one atomic per point instead of two (the second pass re-counted every segment through a cursor: 124 + 164 us for the
// 3.2 M points / 1.17 M pillars of a step).  Arrival order is as arbitrary as cursor order was; csr_sort_segments makes it ascending either way.
__global__ __launch_bounds__(256) void csr_histogram(const int32_t *__restrict__ p2v, int64_t n, int *counts, int32_t *__restrict__ rank)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        rank[i] = atomicAdd(&counts[p2v[i]], 1);
}

__global__ __launch_bounds__(256) void csr_fill(const int32_t *__restrict__ p2v, int64_t n,
                                                const int32_t *__restrict__ seg_offsets, const int32_t *__restrict__ rank,
                                                int32_t *order)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        order[seg_offsets[p2v[i]] + rank[i]] = (int32_t)i;
}

// Few segments (TubeNet instances, K*T ~ 100 rows; m <= CSR_SMALL_M): one global atomic per point would serialise on a handful of addresses, and a
// segment holds tens of thousands of points -- far beyond any sorting network.  [r6] A stable counting sort instead, so that the order inside a segment is
// ascending point index BY CONSTRUCTION (rounds 1-5 placed the chunks' slices in arrival order: sums over such a segment changed in the last bit from run to
// run): (1) every 2048-point chunk counts its points per segment in LDS and writes its row of a [chunks][m] table; (2) a scan down the table's columns
// turns the counts into each chunk's start inside each segment; (3) the waves of a chunk walk their points in index order, 64 at a time: lanes holding the
// same segment find each other by ballot (one round per distinct segment in the wave), a lane's place is the segment's running cursor in LDS plus the
// number of lower lanes with the same segment, and the leader advances the cursor.  Integer atomics only (order-independent); no sort pass afterwards.
#define CSR_SMALL_M 2048

__global__ __launch_bounds__(256) void csr_histogram_small(const int32_t *__restrict__ p2v, int64_t n, int m, int *counts, int *__restrict__ table)
{
    __shared__ int hist[CSR_SMALL_M];
    for (int k = threadIdx.x; k < m; k += 256) hist[k] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * PCACC_CHUNK;
#pragma unroll
    for (int r = 0; r < PCACC_CHUNK_ROWS; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&hist[p2v[i]], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += 256) {
        const int c = hist[k];
        table[(int64_t)blockIdx.x * m + k] = c;
        if (c) atomicAdd(&counts[k], c);
    }
}

// table[chunk][k]: count of chunk `chunk` in segment k -> seg_offsets[k] + the counts of the chunks in front of it.  One wave per segment, 64 chunks per round.
__global__ __launch_bounds__(256) void csr_table_scan(int *__restrict__ table, int n_chunks, int m, const int32_t *__restrict__ seg_offsets)
{
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (k >= m) return;
    int carry = seg_offsets[k];
    for (int c0 = 0; c0 < n_chunks; c0 += 64) {
        const int c = c0 + lane;
        const int v = c < n_chunks ? table[(int64_t)c * m + k] : 0;
        const int incl = wave_inclusive_scan(v);
        if (c < n_chunks) table[(int64_t)c * m + k] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
}

// Four waves per 2048-point chunk, each owning 512 CONSECUTIVE points: a counting pass per wave (integer LDS atomics into the wave's own table), the four
// tables turned into the waves' starts inside every segment (wave 0 first), then every wave places its points in index order, 64 at a time.  Inside a wave the
// LDS instructions execute in program order, so the cursor read (all lanes) / cursor write (leader) pairs need no barrier; no wave touches another's table.
__global__ __launch_bounds__(256) void csr_fill_small(const int32_t *__restrict__ p2v, int64_t n, int m, const int *__restrict__ table, int32_t *__restrict__ order)
{
    extern __shared__ int cursors[];                                      // [4][m]
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int *cursor = cursors + wave * m;
    for (int k = threadIdx.x; k < 4 * m; k += 256) cursors[k] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * PCACC_CHUNK + wave * (PCACC_CHUNK / 4);
    int keys[PCACC_CHUNK / 256];
#pragma unroll
    for (int r = 0; r < PCACC_CHUNK / 256; ++r) {
        const int64_t i = base + r * 64 + lane;
        keys[r] = i < n ? p2v[i] : -1;
        if (keys[r] >= 0) atomicAdd(&cursor[keys[r]], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += 256) {
        int start = table[(int64_t)blockIdx.x * m + k];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = cursors[w * m + k];
            cursors[w * m + k] = start;
            start += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PCACC_CHUNK / 256; ++r) {
        const int key = keys[r];
        const bool live = key >= 0;
        unsigned long long todo = __ballot(live);
        int place = 0;
        while (todo) {                                                  // one round per distinct segment among the wave's 64 points
            const int leader = __ffsll((long long)todo) - 1;
            const int lk = __shfl(key, leader, 64);
            const unsigned long long same = __ballot(live && key == lk);
            const int start = cursor[lk];                                 // every lane reads the cursor before the leader moves it
            if (live && key == lk) place = start + __popcll(same & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (lane == leader) cursor[lk] = start + __popcll(same);
            __builtin_amdgcn_wave_barrier();
            todo &= ~same;
        }
        if (live) order[place] = (int32_t)(base + r * 64 + lane);
    }
}

// Ascending point index inside every segment.  One lane per segment: up to 8 entries are sorted in registers (Batcher's 19
// compare-exchanges); segments of 9..CSR_SORT_MAX entries are then taken one at a time by the whole wave (one entry per lane,
// bitonic network over xor-shuffles).  The former one-lane insertion sort in global memory spent ~110 us per call on the few
// crowded pillars of a sweep (dependent global loads, the rest of the wave idle).
#define CSR_CE(a, b) { const int32_t lo = min(v[a], v[b]), hi = max(v[a], v[b]); v[a] = lo; v[b] = hi; }
__global__ __launch_bounds__(256) void csr_sort_segments(const int32_t *__restrict__ seg_offsets, int64_t m,
                                                         int32_t *order)
{
    const int lane = lane_id();
    for (int64_t base = (int64_t)blockIdx.x * 256; base < m; base += (int64_t)gridDim.x * 256) {      // uniform trip count per wave
        const int64_t s = base + threadIdx.x;
        int b = 0, cnt = 0;
        if (s < m) { b = seg_offsets[s]; cnt = seg_offsets[s + 1] - b; }
        if (cnt >= 2 && cnt <= 8) {
            int32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = i < cnt ? order[b + i] : 0x7fffffff;
            CSR_CE(0, 1) CSR_CE(2, 3) CSR_CE(4, 5) CSR_CE(6, 7)
            CSR_CE(0, 2) CSR_CE(1, 3) CSR_CE(4, 6) CSR_CE(5, 7)
            CSR_CE(1, 2) CSR_CE(5, 6)
            CSR_CE(0, 4) CSR_CE(1, 5) CSR_CE(2, 6) CSR_CE(3, 7)
            CSR_CE(2, 4) CSR_CE(3, 5)
            CSR_CE(1, 2) CSR_CE(3, 4) CSR_CE(5, 6)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < cnt) order[b + i] = v[i];
        }
        unsigned long long crowded = __ballot(cnt > 8 && cnt <= CSR_SORT_MAX);
        while (crowded) {
            const int src = __ffsll((long long)crowded) - 1;
            crowded &= crowded - 1;
            const int sb = __shfl(b, src, 64), sc = __shfl(cnt, src, 64);
            int32_t v = lane < sc ? order[sb + lane] : 0x7fffffff;
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const int32_t other = __shfl_xor(v, j, 64);
                    const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
                    v = keep_min ? min(v, other) : max(v, other);
                }
            if (lane < sc) order[sb + lane] = v;
        }
    }
}
#undef CSR_CE

// [r6] Segments of more than CSR_SORT_MAX points on the many-segment path (crowded pillars of a LiDAR sweep, the cells of a foreground box in the bilinear
// backward's cell keys): rounds 1-5 left them in arrival order -- the one place where two runs of the same step summed the same numbers in a different
// order without an atomic add being involved.  The counting pass lists them (one append per such segment; the ORDER of the list does not matter, every
// entry is sorted on its own), and one workgroup per entry sorts the segment ascending: up to CSR_LONG_LDS entries in LDS, longer ones in place in global
// memory.  Both run the bitonic network in its all-ascending form (first step of every merge mirrors the partner index), which tolerates a length that
// is not a power of two: positions at and beyond the length are +infinity that never has to move.
#define CSR_LONG_LDS 4096

__global__ __launch_bounds__(256) void csr_list_long(const int32_t *__restrict__ seg_offsets, int64_t m, int *count, int32_t *__restrict__ list)
{
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < m; s += (int64_t)gridDim.x * 256)
        if (seg_offsets[s + 1] - seg_offsets[s] > CSR_SORT_MAX) list[atomicAdd(count, 1)] = (int32_t)s;
}

__global__ __launch_bounds__(256) void csr_sort_long(const int32_t *__restrict__ seg_offsets, const int *__restrict__ count, const int32_t *__restrict__ list,
                                                     int32_t *order)
{
    __shared__ int32_t v[CSR_LONG_LDS];
    const int n_long = *count;
    for (int q = blockIdx.x; q < n_long; q += gridDim.x) {
        const int s = list[q];
        const int b = seg_offsets[s], cnt = seg_offsets[s + 1] - b;
        int pow2 = 128;
        while (pow2 < cnt) pow2 <<= 1;
        if (cnt <= CSR_LONG_LDS) {
            for (int i = threadIdx.x; i < pow2; i += 256) v[i] = i < cnt ? order[b + i] : 0x7fffffff;
            __syncthreads();
            for (int k = 2; k <= pow2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = threadIdx.x; t < (pow2 >> 1); t += 256) {
                        const int lo = ((t / j) * 2 * j) + (t % j);                       // lower index of compare-exchange pair t at distance j
                        const int hi = (j == (k >> 1)) ? (lo ^ (k - 1)) : (lo + j);       // first step of a merge: mirrored partner (all-ascending network)
                        const int32_t a = v[lo], c = v[hi];
                        if (a > c) { v[lo] = c; v[hi] = a; }
                    }
                    __syncthreads();
                }
            for (int i = threadIdx.x; i < cnt; i += 256) order[b + i] = v[i];
            __syncthreads();
        } else {
            int32_t *g = order + b;
            for (int k = 2; k <= pow2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = threadIdx.x; t < (pow2 >> 1); t += 256) {
                        const int lo = ((t / j) * 2 * j) + (t % j);
                        const int hi = (j == (k >> 1)) ? (lo ^ (k - 1)) : (lo + j);
                        if (hi < cnt) {                                                     // beyond the length: +infinity, already in place
                            const int32_t a = g[lo], c = g[hi];
                            if (a > c) { g[lo] = c; g[hi] = a; }
                        }
                    }
                    __threadfence_block();
                    __syncthreads();
                }
        }
    }
}

static inline size_t csr_table_bytes(int64_t n, int64_t m) { return m <= CSR_SMALL_M ? pcacc_align((size_t)pcacc_chunks(n) * (size_t)m * 4) : 0; }

extern "C" int pcacc_csr_workspace_bytes(int64_t n, int64_t m, size_t *bytes)
{
    if (!bytes || n < 0 || m < 0) return PCACC_E_ARG;
    // counts (+ the long-segment counter), chunk sums, arrival numbers (many segments; afterwards the list of long segments) / the [chunks][m] table (few)
    *bytes = pcacc_align((size_t)(m + 2) * 4) + pcacc_align((size_t)(pcacc_chunks(m) + 1) * 4) + pcacc_align((size_t)n * 4) + csr_table_bytes(n, m);
    return PCACC_OK;
}

extern "C" int pcacc_csr_build(const int32_t *p2v, int64_t n, int64_t m, int32_t *seg_offsets, int32_t *order,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    size_t need;
    if (pcacc_csr_workspace_bytes(n, m, &need) != PCACC_OK || !seg_offsets) return PCACC_E_ARG;
    if (n > 0 && (!p2v || !order)) return PCACC_E_ARG;
    if (n >= 0x7fffffffLL || m >= 0x7fffffffLL) return PCACC_E_ARG;
    if (!workspace || workspace_bytes < need) return PCACC_E_WORKSPACE;
    hipStream_t s = pcacc_stream(stream);
    char *ws = static_cast<char *>(workspace);
    int *counts = reinterpret_cast<int *>(ws);
    int *sums = reinterpret_cast<int *>(ws + pcacc_align((size_t)(m + 2) * 4));
    int32_t *rank = reinterpret_cast<int32_t *>(ws + pcacc_align((size_t)(m + 2) * 4) + pcacc_align((size_t)(pcacc_chunks(m) + 1) * 4));
    int *table = reinterpret_cast<int *>(reinterpret_cast<char *>(rank) + pcacc_align((size_t)n * 4));
    if (m == 0 || n == 0) {                                   // no points: every segment is empty
        if (hipMemsetAsync(seg_offsets, 0, (size_t)(m + 1) * 4, s) != hipSuccess) return PCACC_E_LAUNCH;
        return PCACC_OK;
    }
    if (hipMemsetAsync(counts, 0, (size_t)(m + 2) * 4, s) != hipSuccess) return PCACC_E_LAUNCH;
    const int chunks = pcacc_chunks(m), n_chunks = pcacc_chunks(n);
    const bool small = m <= CSR_SMALL_M;
    if (small) csr_histogram_small<<<n_chunks, 256, 0, s>>>(p2v, n, (int)m, counts, table);
    else csr_histogram<<<pcacc_grid(n, 256), 256, 0, s>>>(p2v, n, counts, rank);
    chunk_sums_i32<<<chunks, 256, 0, s>>>(counts, m, sums);
    scan_chunk_sums<<<1, 1024, 0, s>>>(sums, chunks, nullptr, -1);
    chunk_scan_i32<<<chunks, 256, 0, s>>>(counts, m, sums, seg_offsets, 1);
    if (small) {
        csr_table_scan<<<(int)((m + 3) / 4), 256, 0, s>>>(table, n_chunks, (int)m, seg_offsets);
        csr_fill_small<<<n_chunks, 256, (size_t)4 * m * sizeof(int), s>>>(p2v, n, (int)m, table, order);      // ascending inside every segment by construction
    } else {
        csr_fill<<<pcacc_grid(n, 256), 256, 0, s>>>(p2v, n, seg_offsets, rank, order);
        csr_sort_segments<<<pcacc_grid(m, 256), 256, 0, s>>>(seg_offsets, m, order);
        int *n_long = counts + m + 1;                          // zero since the memset above; `rank` is free again: the list of long segments (<= n / 65 entries)
        csr_list_long<<<pcacc_grid(m, 256), 256, 0, s>>>(seg_offsets, m, n_long, rank);
        csr_sort_long<<<PCACC_CUS, 256, 0, s>>>(seg_offsets, n_long, rank, order);
    }
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---------------------------------------------------------------------------------------------------
// A3: per-pillar mean of xyz (+ max of an int64 label).  One lane per pillar; the sum runs in ascending
// point index (fp32, sequential) and is divided by the count -- torch_scatter's 'mean'.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_mean3_maxlabel(const float *__restrict__ pts, const int64_t *__restrict__ labels,
                                                          const int32_t *__restrict__ seg_offsets,
                                                          const int32_t *__restrict__ order, int64_t m,
                                                          float *__restrict__ mean, int64_t *__restrict__ max_label)
{
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < m; s += (int64_t)gridDim.x * 256) {
        const int b = seg_offsets[s], e = seg_offsets[s + 1];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        int64_t lab = 0;
        // [r5] four points at a time: their index loads go out together, then their coordinates and labels (clamped positions), then the adds in the
        // order they always had -- a pillar of ~3 points in two memory round trips instead of two per point; same sums bit for bit
        for (int k0 = b; k0 < e; k0 += 4) {
            int64_t idx[4];
            float px[4], py[4], pz[4];
            int64_t pl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[j] = order[min(k0 + j, e - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                px[j] = pts[idx[j] * 3 + 0];
                py[j] = pts[idx[j] * 3 + 1];
                pz[j] = pts[idx[j] * 3 + 2];
                pl[j] = labels ? labels[idx[j]] : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k0 + j >= e) break;
                sx = __fadd_rn(sx, px[j]);
                sy = __fadd_rn(sy, py[j]);
                sz = __fadd_rn(sz, pz[j]);
                if (labels) lab = (k0 + j == b || pl[j] > lab) ? pl[j] : lab;
            }
        }
        const float cnt = (float)(e - b);
        if (e > b) { sx = __fdiv_rn(sx, cnt); sy = __fdiv_rn(sy, cnt); sz = __fdiv_rn(sz, cnt); }
        mean[s * 3 + 0] = sx; mean[s * 3 + 1] = sy; mean[s * 3 + 2] = sz;
        if (labels) max_label[s] = lab;
    }
}

extern "C" int pcacc_segment_mean3_maxlabel(const float *points, const int64_t *labels, const int32_t *seg_offsets,
                                            const int32_t *order, int64_t m, float *mean, int64_t *max_label,
                                            void *stream)
{
    if (m < 0 || (m > 0 && (!points || !seg_offsets || !order || !mean))) return PCACC_E_ARG;
    if (labels && !max_label) return PCACC_E_ARG;
    if (m == 0) return PCACC_OK;
    seg_mean3_maxlabel<<<pcacc_grid(m, 256), 256, 0, pcacc_stream(stream)>>>(points, labels, seg_offsets, order, m,
                                                                           mean, max_label);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// ---------------------------------------------------------------------------------------------------
// A4 pooling: per-pillar channel-wise max with arg (lowest point index).  c/4 lanes per pillar, each lane
// owns 4 consecutive channels and streams the pillar's rows as float4 (a row of c floats is contiguous).
// ---------------------------------------------------------------------------------------------------
template <int LPP>   // lanes per pillar = c / 4
__global__ __launch_bounds__(256) void seg_max_kernel(const void *__restrict__ src, const int32_t *__restrict__ seg_offsets,
                                                      const int32_t *__restrict__ order, int64_t m,
                                                      void *__restrict__ out, int4 *__restrict__ arg, bool bf,
                                                      uint16_t *__restrict__ out16 = nullptr)      // out16 (f32 rows): a bf16 copy of `out` from the same store ('mixed' mode shadow)
{
    const int sub = threadIdx.x % LPP;
    const int64_t per_block = 256 / LPP;
    for (int64_t s = (int64_t)blockIdx.x * per_block + threadIdx.x / LPP; s < m; s += (int64_t)gridDim.x * per_block) {
        const int b = seg_offsets[s], e = seg_offsets[s + 1];
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 bi = make_int4(-1, -1, -1, -1);
        // four rows in flight: the index loads of a chunk go out together, then the four row loads (a pillar holds ~3 points,
        // so most segments are one chunk: two memory latencies instead of two per point)
        for (int k0 = b; k0 < e; k0 += 4) {
            int idx[4];
            float4 rows[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[j] = k0 + j < e ? order[k0 + j] : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rows[j] = idx[j] >= 0 ? pcacc_ld4(src, bf, (int64_t)idx[j] * LPP + sub) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = idx[j];
                if (i < 0) continue;
                const float4 v = rows[j];
                // strict '>' in ascending index order, but the lowest INDEX must win even when the cursor
                // order of a >64-point pillar is not sorted: tie-break on the index explicitly.
                if (bi.x < 0 || v.x > best.x || (v.x == best.x && i < bi.x)) { best.x = v.x; bi.x = i; }
                if (bi.y < 0 || v.y > best.y || (v.y == best.y && i < bi.y)) { best.y = v.y; bi.y = i; }
                if (bi.z < 0 || v.z > best.z || (v.z == best.z && i < bi.z)) { best.z = v.z; bi.z = i; }
                if (bi.w < 0 || v.w > best.w || (v.w == best.w && i < bi.w)) { best.w = v.w; bi.w = i; }
            }
        }
        pcacc_st4(out, bf, s * LPP + sub, best);
        if (out16) reinterpret_cast<uint2 *>(out16)[s * LPP + sub] = make_uint2(pcacc_pack_bf16x2(best.x, best.y), pcacc_pack_bf16x2(best.z, best.w));
        arg[s * LPP + sub] = bi;
    }
}


// ---------------------------------------------------------------------------------------------------
// Long segments (n/m large: per-instance poolings of the TubeNet, the offset loss): one lane group per segment
// would walk tens of thousands of rows serially.  Two levels instead: every segment is cut into pieces of
// SEG_PIECE rows of the sorted order; level 1 reduces pieces in parallel, level 2 reduces the (contiguous)
// pieces of each segment.  No atomics; pieces are found by binary search in the scanned piece counts.
// ---------------------------------------------------------------------------------------------------
#define SEG_PIECE 64

__global__ __launch_bounds__(256) void seg_piece_counts(const int32_t *__restrict__ seg_offsets, int64_t m, int *pieces)
{
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < m; s += (int64_t)gridDim.x * 256) {
        const int len = seg_offsets[s + 1] - seg_offsets[s];
        pieces[s] = len > 0 ? (len + SEG_PIECE - 1) / SEG_PIECE : 1;
    }
}

__device__ __forceinline__ int piece_segment(const int *__restrict__ piece_off, int m, int p)
{
    int lo = 0, hi = m;                        // largest s with piece_off[s] <= p
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (piece_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

template <int LPP, bool IS_MAX>
__global__ __launch_bounds__(256) void seg_level1(const void *__restrict__ src, bool src_bf, const int32_t *__restrict__ seg_offsets,
                                                  const int32_t *__restrict__ order, const int *__restrict__ piece_off, int m,
                                                  float4 *__restrict__ pval, int4 *__restrict__ parg)
{
    const int sub = threadIdx.x % LPP;
    const int per_block = 256 / LPP;
    const int n_pieces = piece_off[m];
    for (int p = blockIdx.x * per_block + threadIdx.x / LPP; p < n_pieces; p += gridDim.x * per_block) {
        const int s = piece_segment(piece_off, m, p);
        const int b = seg_offsets[s] + (p - piece_off[s]) * SEG_PIECE;
        const int e = min(b + SEG_PIECE, seg_offsets[s + 1]);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 bi = make_int4(-1, -1, -1, -1);
        for (int k0 = b; k0 < e; k0 += 4) {                               // four rows in flight (see seg_max_kernel); same order
            int idx[4];
            float4 rows[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[j] = k0 + j < e ? order[k0 + j] : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rows[j] = idx[j] >= 0 ? pcacc_ld4(src, src_bf, (int64_t)idx[j] * LPP + sub) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = idx[j];
                if (i < 0) continue;
                const float4 v = rows[j];
                if (IS_MAX) {
                    if (bi.x < 0 || v.x > acc.x || (v.x == acc.x && i < bi.x)) { acc.x = v.x; bi.x = i; }
                    if (bi.y < 0 || v.y > acc.y || (v.y == acc.y && i < bi.y)) { acc.y = v.y; bi.y = i; }
                    if (bi.z < 0 || v.z > acc.z || (v.z == acc.z && i < bi.z)) { acc.z = v.z; bi.z = i; }
                    if (bi.w < 0 || v.w > acc.w || (v.w == acc.w && i < bi.w)) { acc.w = v.w; bi.w = i; }
                } else {
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
        // [r6] sums take the second level too (rounds 1-5: one fp32 atomic per piece and channel into the zero-filled output -- the pieces of a segment then
        // met in arrival order): piece sums, added in piece order by seg_level2
        pval[(int64_t)p * LPP + sub] = acc;
        if (IS_MAX) parg[(int64_t)p * LPP + sub] = bi;
    }
}

template <int LPP, bool IS_MAX>
__global__ __launch_bounds__(256) void seg_level2(const float4 *__restrict__ pval, const int4 *__restrict__ parg,
                                                  const int *__restrict__ piece_off, int64_t m,
                                                  float4 *__restrict__ out, int4 *__restrict__ arg)
{
    const int sub = threadIdx.x % LPP;
    const int64_t per_block = 256 / LPP;
    for (int64_t s = (int64_t)blockIdx.x * per_block + threadIdx.x / LPP; s < m; s += (int64_t)gridDim.x * per_block) {
        const int b = piece_off[s], e = piece_off[s + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 bi = make_int4(-1, -1, -1, -1);
        for (int p = b; p < e; ++p) {
            const float4 v = pval[(int64_t)p * LPP + sub];
            if (IS_MAX) {
                const int4 a = parg[(int64_t)p * LPP + sub];
                if (a.x >= 0 && (bi.x < 0 || v.x > acc.x || (v.x == acc.x && a.x < bi.x))) { acc.x = v.x; bi.x = a.x; }
                if (a.y >= 0 && (bi.y < 0 || v.y > acc.y || (v.y == acc.y && a.y < bi.y))) { acc.y = v.y; bi.y = a.y; }
                if (a.z >= 0 && (bi.z < 0 || v.z > acc.z || (v.z == acc.z && a.z < bi.z))) { acc.z = v.z; bi.z = a.z; }
                if (a.w >= 0 && (bi.w < 0 || v.w > acc.w || (v.w == acc.w && a.w < bi.w))) { acc.w = v.w; bi.w = a.w; }
            } else {
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        out[s * LPP + sub] = acc;
        if (IS_MAX) arg[s * LPP + sub] = bi;
    }
}

static inline int64_t seg_max_pieces(int64_t n, int64_t m) { return m + n / SEG_PIECE + 1; }
static inline bool seg_use_two_level(int64_t n, int64_t m) { return m > 0 && n / m > 16; }

extern "C" int pcacc_segment_workspace_bytes(int64_t n, int64_t m, int c, size_t *bytes)
{
    if (!bytes || n < 0 || m < 0 || c <= 0) return PCACC_E_ARG;
    *bytes = 0;
    if (seg_use_two_level(n, m)) {
        const size_t P = (size_t)seg_max_pieces(n, m);
        *bytes = pcacc_align((size_t)(m + 1) * 4) * 2 + pcacc_align((size_t)(pcacc_chunks(m) + 1) * 4) +
                 pcacc_align(P * c * 4) * 2;
    }
    return PCACC_OK;
}

template <bool IS_MAX>
static int seg_two_level(const void *src, bool src_bf, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                         float *out, int32_t *arg, void *workspace, size_t workspace_bytes, hipStream_t s)
{
    size_t need;
    pcacc_segment_workspace_bytes(n, m, c, &need);
    if (!workspace || workspace_bytes < need) return PCACC_E_WORKSPACE;
    char *ws = static_cast<char *>(workspace);
    int *pieces = reinterpret_cast<int *>(ws); ws += pcacc_align((size_t)(m + 1) * 4);
    int *piece_off = reinterpret_cast<int *>(ws); ws += pcacc_align((size_t)(m + 1) * 4);
    int *sums = reinterpret_cast<int *>(ws); ws += pcacc_align((size_t)(pcacc_chunks(m) + 1) * 4);
    const size_t P = (size_t)seg_max_pieces(n, m);
    float4 *pval = reinterpret_cast<float4 *>(ws); ws += pcacc_align(P * c * 4);
    int4 *parg = reinterpret_cast<int4 *>(ws);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    int4 *arg4 = reinterpret_cast<int4 *>(arg);
    const int chunks = pcacc_chunks(m);
    seg_piece_counts<<<pcacc_grid(m, 256), 256, 0, s>>>(seg_offsets, m, pieces);
    chunk_sums_i32<<<chunks, 256, 0, s>>>(pieces, m, sums);
    scan_chunk_sums<<<1, 1024, 0, s>>>(sums, chunks, nullptr, -1);
    chunk_scan_i32<<<chunks, 256, 0, s>>>(pieces, m, sums, piece_off, 1);
#define LAUNCH2(L)                                                                                                   \
    seg_level1<L, IS_MAX><<<pcacc_grid((int64_t)P * L, 256), 256, 0, s>>>(src, src_bf, seg_offsets, order, piece_off, (int)m, pval, parg); \
    seg_level2<L, IS_MAX><<<pcacc_grid(m * L, 256), 256, 0, s>>>(pval, parg, piece_off, m, out4, arg4)
    switch (c / 4) {
        case 1: LAUNCH2(1); break;
        case 2: LAUNCH2(2); break;
        case 4: LAUNCH2(4); break;
        case 8: LAUNCH2(8); break;
        case 16: LAUNCH2(16); break;
        case 32: LAUNCH2(32); break;
        case 64: LAUNCH2(64); break;
        default: return PCACC_E_ARG;
    }
#undef LAUNCH2
    return PCACC_OK;
}

static int segment_max_any(const void *src, int dtype, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                           void *out, int32_t *arg, void *workspace, size_t workspace_bytes, void *stream, uint16_t *out16 = nullptr)
{
    if (m < 0 || n < 0 || c <= 0 || (c % 4) || c > 256 || (dtype != PCACC_F32 && dtype != PCACC_BF16)) return PCACC_E_ARG;
    const bool bf = dtype == PCACC_BF16;
    if (m > 0 && (!seg_offsets || !out || !arg || (n > 0 && (!src || !order)))) return PCACC_E_ARG;
    if (m == 0) return PCACC_OK;
    hipStream_t s = pcacc_stream(stream);
    if (out16 && (dtype != PCACC_F32 || seg_use_two_level(n, m))) return PCACC_E_ARG;      // the second output exists on the short-segment f32 path only
    if (seg_use_two_level(n, m)) {
        // long segments: rows in either type, result always f32 (m is small)
        const int rc = seg_two_level<true>(src, bf, c, seg_offsets, order, n, m, reinterpret_cast<float *>(out), arg, workspace,
                                           workspace_bytes, s);
        if (rc != PCACC_OK) return rc;
        PCACC_CHECK_LAUNCH();
        return PCACC_OK;
    }
    int4 *arg4 = reinterpret_cast<int4 *>(arg);
#define LAUNCH(L) seg_max_kernel<L><<<pcacc_grid(m * L, 256), 256, 0, s>>>(src, seg_offsets, order, m, out, arg4, bf, out16)
    switch (c / 4) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        default: return PCACC_E_ARG;
    }
#undef LAUNCH
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_segment_max(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                                 float *out, int32_t *arg, void *workspace, size_t workspace_bytes, void *stream)
{
    return segment_max_any(src, PCACC_F32, c, seg_offsets, order, n, m, out, arg, workspace, workspace_bytes, stream);
}

extern "C" int pcacc_segment_max_t(const void *src, int dtype, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n,
                                   int64_t m, void *out, int32_t *arg, void *workspace, size_t workspace_bytes, void *stream)
{
    return segment_max_any(src, dtype, c, seg_offsets, order, n, m, out, arg, workspace, workspace_bytes, stream);
}

extern "C" int pcacc_segment_max_dual(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m, float *out,
                                      uint16_t *out16, int32_t *arg, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!out16) return PCACC_E_ARG;
    return segment_max_any(src, PCACC_F32, c, seg_offsets, order, n, m, out, arg, workspace, workspace_bytes, stream, out16);
}

// ---------------------------------------------------------------------------------------------------
// [r6] A4 + A5 as ONE pass (models/pillar_encoder.py:119-122 -> :125-174): the encoder's last max-pooling writes the BEV canvas itself.  The kernel walks
// the CELLS of the canvas; an occupied cell reduces its pillar's point rows (ascending point index: the CSR) and stores the 32-channel maximum at the cell,
// an empty cell stores zeros -- the [M, C] pooled-row table of rounds 1-5 (written by the pooling, read back by two canvas fills in the 'mixed' mode: fp32
// twin + bf16 shadow) is never written.  Both canvases come from the same registers: fp32 with cached stores (the first convolution reads it next), the bf16
// shadow with streaming stores (read a whole forward later, by the backward).  arg[pillar] = the winners, for the backward.
// c / 4 lanes per cell; pillars are numbered in cell order (ops.PillarIndex), so the winners' table is written front to back.
// ---------------------------------------------------------------------------------------------------
typedef uint32_t seg_u32x2 __attribute__((ext_vector_type(2)));
typedef float seg_f32x4 __attribute__((ext_vector_type(4)));
typedef int seg_i32x4 __attribute__((ext_vector_type(4)));
// CPG cells per lane group and iteration: the chain cell table -> segment offsets -> order -> point rows is four dependent memory round trips per cell; with two
// cells in flight per group every round trip carries twice the requests (rows of ~3 points per pillar: the kernel is bound by what it keeps in flight).
// NT: the point rows and their order are read once, the winners are read a whole forward later: streaming policy for those; the fp32 canvas keeps cached stores.
template <int LPP, int CPG, bool NT>
__global__ __launch_bounds__(256) void seg_max_canvas_kernel(const float4 *__restrict__ src, const int32_t *__restrict__ seg_offsets,
                                                             const int32_t *__restrict__ order, const int32_t *__restrict__ cell2pillar, int64_t n_cells,
                                                             float4 *__restrict__ canvas32, uint16_t *__restrict__ canvas16, int4 *__restrict__ arg)
{
    const int sub = threadIdx.x % LPP;
    const int64_t per_block = 256 / LPP;
    const int64_t stride = (int64_t)gridDim.x * per_block;
    auto ld_row = [&](int64_t i4) -> float4 {
        if (NT) {
            const seg_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const seg_f32x4 *>(src) + i4);
            return make_float4(v.x, v.y, v.z, v.w);
        }
        return src[i4];
    };
    for (int64_t cell0 = (int64_t)blockIdx.x * per_block + threadIdx.x / LPP; cell0 < n_cells; cell0 += stride * CPG) {
        int s[CPG], b[CPG], e[CPG];
#pragma unroll
        for (int u = 0; u < CPG; ++u) {
            const int64_t cell = cell0 + u * stride;
            s[u] = cell < n_cells ? cell2pillar[cell] : -1;
        }
#pragma unroll
        for (int u = 0; u < CPG; ++u) {
            b[u] = s[u] >= 0 ? seg_offsets[s[u]] : 0;
            e[u] = s[u] >= 0 ? seg_offsets[s[u] + 1] : 0;
        }
        float4 best[CPG];
        int4 bi[CPG];
        int idx[CPG][4];
        float4 rows[CPG][4];
#pragma unroll
        for (int u = 0; u < CPG; ++u) {
            best[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            bi[u] = make_int4(-1, -1, -1, -1);
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[u][j] = b[u] + j < e[u] ? (NT ? __builtin_nontemporal_load(order + b[u] + j) : order[b[u] + j]) : -1;
        }
#pragma unroll
        for (int u = 0; u < CPG; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) rows[u][j] = idx[u][j] >= 0 ? ld_row((int64_t)idx[u][j] * LPP + sub) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < CPG; ++u) {
            int k0 = b[u];
            while (true) {                                                  // the first four rows of both cells are in flight together; longer pillars go on alone
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = idx[u][j];
                    if (i < 0) continue;
                    const float4 v = rows[u][j];
                    if (bi[u].x < 0 || v.x > best[u].x || (v.x == best[u].x && i < bi[u].x)) { best[u].x = v.x; bi[u].x = i; }
                    if (bi[u].y < 0 || v.y > best[u].y || (v.y == best[u].y && i < bi[u].y)) { best[u].y = v.y; bi[u].y = i; }
                    if (bi[u].z < 0 || v.z > best[u].z || (v.z == best[u].z && i < bi[u].z)) { best[u].z = v.z; bi[u].z = i; }
                    if (bi[u].w < 0 || v.w > best[u].w || (v.w == best[u].w && i < bi[u].w)) { best[u].w = v.w; bi[u].w = i; }
                }
                k0 += 4;
                if (k0 >= e[u]) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) idx[u][j] = k0 + j < e[u] ? order[k0 + j] : -1;
#pragma unroll
                for (int j = 0; j < 4; ++j) rows[u][j] = idx[u][j] >= 0 ? ld_row((int64_t)idx[u][j] * LPP + sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < CPG; ++u) {
            const int64_t cell = cell0 + u * stride;
            if (cell >= n_cells) continue;
            if (s[u] >= 0) {
                if (NT) {
                    const seg_i32x4 a = {bi[u].x, bi[u].y, bi[u].z, bi[u].w};
                    __builtin_nontemporal_store(a, reinterpret_cast<seg_i32x4 *>(arg) + (int64_t)s[u] * LPP + sub);
                } else arg[(int64_t)s[u] * LPP + sub] = bi[u];
            }
            canvas32[cell * LPP + sub] = best[u];
            const seg_u32x2 pk = {pcacc_pack_bf16x2(best[u].x, best[u].y), pcacc_pack_bf16x2(best[u].z, best[u].w)};
            __builtin_nontemporal_store(pk, reinterpret_cast<seg_u32x2 *>(canvas16) + cell * LPP + sub);
        }
    }
}

// [r6] Tried and removed (the commit before this one has the code; profiles/r06_fused_canvas_pipeline_ab.txt): the same pass as a four-stage software pipeline --
// iteration t of a lane group issues the table read of cell t, the offsets of cell t - 1, the point list of cell t - 2 and the rows of cell t - 3 back to back
// and reduces cell t - 4 meanwhile: one memory round trip per iteration instead of a chain of four.  Bit-identical, 93 VGPRs, and SLOWER: 241 vs 215 us alone,
// 250 vs 221 us in the step.  With 32 waves per CU the chain was already hidden; the kernel sits at 0.8 of what a plain copy of its 902 MB takes.
extern "C" int pcacc_segment_max_canvas(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                                        const int32_t *cell2pillar, int64_t n_cells, float *canvas32, uint16_t *canvas16, int32_t *arg,
                                        void *start_event, void *stop_event, void *stream)
{
    if (m < 0 || n < 0 || n_cells < 0 || c <= 0 || (c % 4) || c > 256 || (start_event == nullptr) != (stop_event == nullptr)) return PCACC_E_ARG;
    if (n_cells == 0) return PCACC_OK;
    if (!cell2pillar || !canvas32 || !canvas16 || (m > 0 && (!seg_offsets || !arg || (n > 0 && (!src || !order))))) return PCACC_E_ARG;
    if (seg_use_two_level(n, m)) return PCACC_E_ARG;              // long segments: the two-level pooling + the separate fills (the caller's fallback)
    const void *fn;
    int cpg = 1;
    // Measured (tools/bench_fused_canvas.py, profiles/r06_fused_canvas_variants.txt; alone, warm / behind 1 GiB of streamed lines): one cell per group with
    // cached loads 208 / 222 us, with streaming loads 198 / 200 us, two cells in flight 199 / 215 us, four 229 / 237 us -- bound by the random 128-byte row
    // reads themselves, not by how many of them a lane group keeps in flight.  INSIDE the step the rows were written by the layer in front and partly still
    // sit in the Infinity Cache: cached loads 191 us, streaming loads 199 - 209 us (three bench runs each) -- cached loads are the default.
    const char variant = pcacc_switches().scatter_variant;     // A/B (PCACC_SCATTER_VARIANT): 'c' streaming loads; 'b' two cells, cached loads; 'e' two cells, streaming; 'd' four
    switch (c / 4) {
        case 1: fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<1, 1, false>); break;
        case 2: fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<2, 1, false>); break;
        case 4: fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<4, 1, false>); break;
        case 8:
            fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<8, 1, false>);
            if (variant == 'c') fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<8, 1, true>);
            else if (variant == 'b') { fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<8, 2, false>); cpg = 2; }
            else if (variant == 'e') { fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<8, 2, true>); cpg = 2; }
            else if (variant == 'd') { fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<8, 4, true>); cpg = 4; }
            break;
        case 16: fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<16, 1, false>); break;
        case 32: fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<32, 1, false>); break;
        case 64: fn = reinterpret_cast<const void *>(seg_max_canvas_kernel<64, 1, false>); break;
        default: return PCACC_E_ARG;
    }
    void *args[] = {(void *)&src, (void *)&seg_offsets, (void *)&order, (void *)&cell2pillar, (void *)&n_cells, (void *)&canvas32, (void *)&canvas16, (void *)&arg};
    // events attached to the dispatch itself (see pillar_scatter_launch, canvas.hip): bench.py times this kernel live for its roofline object
    if (hipExtLaunchKernel(fn, dim3(pcacc_grid((n_cells + cpg - 1) / cpg * (c / 4), 256)), dim3(256), args, 0, pcacc_stream(stream), reinterpret_cast<hipEvent_t>(start_event),
                           reinterpret_cast<hipEvent_t>(stop_event), 0) != hipSuccess)
        return PCACC_E_LAUNCH;
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// Backward of the pass above: grad_src[i,k] = (arg[p2v[i],k] == i) ? grad_canvas[cell[p2v[i]],k] : 0 -- seg_max_bwd_kernel reading the canvas gradient in place
// through the pillars' cell numbers instead of a gathered [M, C] copy of it.
__global__ __launch_bounds__(256) void seg_max_canvas_bwd_kernel(const void *__restrict__ grad_canvas, const int4 *__restrict__ arg, const int32_t *__restrict__ p2v,
                                                                 const int32_t *__restrict__ cell, int64_t n, int lpp, void *__restrict__ grad_src, bool bf,
                                                                 bool out_bf)
{
    const int64_t total = n * lpp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / lpp;
        const int sub = (int)(e - i * lpp);
        const int64_t s = p2v[i];
        const int4 a = arg[s * lpp + sub];
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.x == (int)i || a.y == (int)i || a.z == (int)i || a.w == (int)i) {      // most points win nothing: their rows never touch the canvas gradient
            const float4 g = pcacc_ld4(grad_canvas, bf, (int64_t)cell[s] * lpp + sub);
            r.x = (a.x == (int)i) ? g.x : 0.f;
            r.y = (a.y == (int)i) ? g.y : 0.f;
            r.z = (a.z == (int)i) ? g.z : 0.f;
            r.w = (a.w == (int)i) ? g.w : 0.f;
        }
        pcacc_st4(grad_src, out_bf, e, r);
    }
}

extern "C" int pcacc_segment_max_canvas_backward(const void *grad_canvas, int dtype, const int32_t *arg, const int32_t *p2v, const int32_t *cell, int64_t n,
                                                 int c, void *grad_src, int out_dtype, void *stream)
{
    if (n < 0 || c <= 0 || (c % 4) || (dtype != PCACC_F32 && dtype != PCACC_BF16) || (out_dtype != PCACC_F32 && out_dtype != PCACC_BF16)) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (!grad_canvas || !arg || !p2v || !cell || !grad_src) return PCACC_E_ARG;
    seg_max_canvas_bwd_kernel<<<pcacc_grid(n * (c / 4), 256), 256, 0, pcacc_stream(stream)>>>(grad_canvas, reinterpret_cast<const int4 *>(arg), p2v, cell, n, c / 4,
                                                                                         grad_src, dtype == PCACC_BF16, out_dtype == PCACC_BF16);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

// grad_src[i,k] = (arg[p2v[i],k] == i) ? grad_out[p2v[i],k] : 0        (fully coalesced, no atomics)
// ACC: added to what grad_src holds (the rows' gradient from their other consumer: one pass instead of a dense result plus autograd's
// add of the two), with the largest magnitude of the sums into out_amax (256 zeroed slots; may be NULL)
template <bool ACC>
__global__ __launch_bounds__(256) void seg_max_bwd_kernel(const void *__restrict__ grad_out, const int4 *__restrict__ arg,
                                                          const int32_t *__restrict__ p2v, int64_t n, int lpp,
                                                          void *__restrict__ grad_src, bool bf, bool out_bf, float *__restrict__ out_amax)
{
    const int64_t total = n * lpp;
    float mx = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / lpp;
        const int sub = (int)(e - i * lpp);
        const int64_t s = p2v[i];
        const int4 a = arg[s * lpp + sub];
        const float4 g = pcacc_ld4(grad_out, bf, s * lpp + sub);
        float4 r;
        r.x = (a.x == (int)i) ? g.x : 0.f;
        r.y = (a.y == (int)i) ? g.y : 0.f;
        r.z = (a.z == (int)i) ? g.z : 0.f;
        r.w = (a.w == (int)i) ? g.w : 0.f;
        if (ACC) {
            const float4 o = pcacc_ld4(grad_src, out_bf, e);
            r = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
            if (!(r.x == r.x && r.y == r.y && r.z == r.z && r.w == r.w)) mx = __builtin_inff();
        }
        pcacc_st4(grad_src, out_bf, e, r);
    }
    if (ACC && out_amax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        if (lane_id() == 0) atomicMax(reinterpret_cast<unsigned *>(out_amax) + (blockIdx.x & 255), __float_as_uint(mx));
    }
}

static int segment_max_backward_any(const void *grad_out, int dtype, const int32_t *arg, const int32_t *p2v, int64_t n, int c,
                                    void *grad_src, int out_dtype, void *stream, bool accumulate = false, float *out_amax = nullptr)
{
    if (n < 0 || c <= 0 || (c % 4) || (dtype != PCACC_F32 && dtype != PCACC_BF16) || (out_dtype != PCACC_F32 && out_dtype != PCACC_BF16))
        return PCACC_E_ARG;
    if (n > 0 && (!grad_out || !arg || !p2v || !grad_src)) return PCACC_E_ARG;
    if (n == 0) return PCACC_OK;
    if (accumulate)
        seg_max_bwd_kernel<true><<<pcacc_grid(n * (c / 4), 256), 256, 0, pcacc_stream(stream)>>>(
            grad_out, reinterpret_cast<const int4 *>(arg), p2v, n, c / 4, grad_src, dtype == PCACC_BF16, out_dtype == PCACC_BF16, out_amax);
    else
        seg_max_bwd_kernel<false><<<pcacc_grid(n * (c / 4), 256), 256, 0, pcacc_stream(stream)>>>(
            grad_out, reinterpret_cast<const int4 *>(arg), p2v, n, c / 4, grad_src, dtype == PCACC_BF16, out_dtype == PCACC_BF16, nullptr);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_segment_max_backward(const float *grad_out, const int32_t *arg, const int32_t *p2v, int64_t n, int c,
                                          float *grad_src, void *stream)
{
    return segment_max_backward_any(grad_out, PCACC_F32, arg, p2v, n, c, grad_src, PCACC_F32, stream);
}

extern "C" int pcacc_segment_max_backward_t(const void *grad_out, int dtype, const int32_t *arg, const int32_t *p2v, int64_t n,
                                            int c, void *grad_src, int out_dtype, void *stream)
{
    return segment_max_backward_any(grad_out, dtype, arg, p2v, n, c, grad_src, out_dtype, stream);
}

extern "C" int pcacc_segment_max_backward_acc(const void *grad_out, int dtype, const int32_t *arg, const int32_t *p2v, int64_t n, int c,
                                              void *grad_src, int out_dtype, float *out_amax, void *stream)
{
    return segment_max_backward_any(grad_out, dtype, arg, p2v, n, c, grad_src, out_dtype, stream, true, out_amax);
}

// Backward of the [point_to_voxel_map] broadcast (models/pillar_encoder.py:116): per-pillar sum of point rows.
template <int LPP>
__global__ __launch_bounds__(256) void seg_sum_kernel(const void *__restrict__ src, const int32_t *__restrict__ seg_offsets,
                                                      const int32_t *__restrict__ order, int64_t m, void *__restrict__ out, bool bf)
{
    const int sub = threadIdx.x % LPP;
    const int64_t per_block = 256 / LPP;
    for (int64_t s = (int64_t)blockIdx.x * per_block + threadIdx.x / LPP; s < m; s += (int64_t)gridDim.x * per_block) {
        const int b = seg_offsets[s], e = seg_offsets[s + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = b; k0 < e; k0 += 4) {                               // four rows in flight (see seg_max_kernel); same sum order
            int idx[4];
            float4 rows[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[j] = k0 + j < e ? order[k0 + j] : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rows[j] = idx[j] >= 0 ? pcacc_ld4(src, bf, (int64_t)idx[j] * LPP + sub) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (idx[j] >= 0) { acc.x += rows[j].x; acc.y += rows[j].y; acc.z += rows[j].z; acc.w += rows[j].w; }
        }
        pcacc_st4(out, bf, s * LPP + sub, acc);
    }
}

static int segment_sum_any(const void *src, int dtype, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                           void *out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (m < 0 || n < 0 || c <= 0 || (c % 4) || c > 256 || (dtype != PCACC_F32 && dtype != PCACC_BF16)) return PCACC_E_ARG;
    const bool bf = dtype == PCACC_BF16;
    if (m > 0 && (!seg_offsets || !out || (n > 0 && (!src || !order)))) return PCACC_E_ARG;
    if (m == 0) return PCACC_OK;
    hipStream_t s = pcacc_stream(stream);
    if (seg_use_two_level(n, m)) {
        const int rc = seg_two_level<false>(src, bf, c, seg_offsets, order, n, m, reinterpret_cast<float *>(out), nullptr, workspace,
                                            workspace_bytes, s);
        if (rc != PCACC_OK) return rc;
        PCACC_CHECK_LAUNCH();
        return PCACC_OK;
    }
#define LAUNCH(L) seg_sum_kernel<L><<<pcacc_grid(m * L, 256), 256, 0, s>>>(src, seg_offsets, order, m, out, bf)
    switch (c / 4) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        default: return PCACC_E_ARG;
    }
#undef LAUNCH
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}

extern "C" int pcacc_segment_sum(const float *src, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n, int64_t m,
                                 float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return segment_sum_any(src, PCACC_F32, c, seg_offsets, order, n, m, out, workspace, workspace_bytes, stream);
}

extern "C" int pcacc_segment_sum_t(const void *src, int dtype, int c, const int32_t *seg_offsets, const int32_t *order, int64_t n,
                                   int64_t m, void *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return segment_sum_any(src, dtype, c, seg_offsets, order, n, m, out, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------
// Few output rows (instances x frames ~ 100, m*c <= 8192 floats): sums need no CSR at all.  Every workgroup accumulates its slice of the input into an
// LDS copy of the whole [m,c] output.  (models/tpointnet.py:227,251,283-284 `scatter(..., 'sum' | 'mean')`, libs/loss.py:216.)
// [r6] The LDS copy is 64-bit FIXED POINT: a workgroup first takes the largest magnitude of its slice (the slice is read twice; the second read hits the
// caches), scales by the power of two that puts it at 2^40, and adds integers (ds_add_u64) -- integer sums do not depend on the order of the additions,
// where the fp32 LDS / global atomics of rounds 2-5 made every few-row sum of the TubeNet differ in the last bits from run to run.  A slice is at most
// 2^22 elements, so a segment's sum stays below 2^62; an element loses what lies 2^-40 below the slice's maximum (an fp32 sum loses 2^-24 below its running
// value).  The workgroups' tables leave as fp32 partials and meet in a fixed order (pcacc_reduce_partials).  A non-finite input makes its workgroup's whole
// partial NaN -- the sum is non-finite either way, the step's finite check sees it.
// ---------------------------------------------------------------------------------------------------
#define SSS_SLICE_MAX (1 << 22)
#define SSS_THREADS 1024
#define SSS_MAX_GRID 128
// [r6, second form] 1 024 threads per workgroup (one table per CU needs all the CU's waves to hide its loads: the first form ran 4 waves per CU, one float per
// lane per trip, a 64-bit division per element: 75 us for 20 MB), whole rows per workgroup, 16-byte pieces where c % 4 == 0, at most 128 tables (the fixed-order
// reduce behind reads grid x m x c partials: 28 us at 256).  V = floats per piece.
template <int V>
__global__ __launch_bounds__(SSS_THREADS) void scatter_sum_small_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, int64_t n, int c,
                                                                        int mc, int64_t rows_per_block, float *__restrict__ partial)
{
    extern __shared__ unsigned long long acc64[];
    __shared__ float wmax[SSS_THREADS / 64];
    for (int j = threadIdx.x; j < mc; j += SSS_THREADS) acc64[j] = 0ull;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    const int cv = c / V;                                               // pieces per row
    const int n_pieces = r1 > r0 ? (int)((r1 - r0) * cv) : 0;          // < 2^22
    const float *mine_src = src + r0 * c;
    float mx = 0.f;
    bool bad = false;
#pragma unroll 4
    for (int q = threadIdx.x; q < n_pieces; q += SSS_THREADS) {
        float x[V];
        if (V == 4) { const float4 t = reinterpret_cast<const float4 *>(mine_src)[q]; x[0] = t.x; x[V > 1 ? 1 : 0] = t.y; x[V > 2 ? 2 : 0] = t.z; x[V > 3 ? 3 : 0] = t.w; }
        else x[0] = mine_src[q];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            mx = fmaxf(mx, fabsf(x[v]));
            bad |= !(fabsf(x[v]) < __builtin_inff());
        }
    }
    if (bad) mx = __builtin_inff();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if (lane_id() == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = wmax[0];
#pragma unroll
    for (int k = 1; k < SSS_THREADS / 64; ++k) mx = fmaxf(mx, wmax[k]);
    float *mine = partial + (int64_t)blockIdx.x * mc;
    if (!(mx < __builtin_inff())) {                                     // uniform
        for (int j = threadIdx.x; j < mc; j += SSS_THREADS) mine[j] = __builtin_nanf("");
        return;
    }
    int k = 0;
    if (mx > 0.f) frexpf(mx, &k);                                       // mx = f 2^k, f in [0.5, 1): scaled values below 2^40
    const float up = ldexpf(1.f, 40 - k > 126 ? 126 : 40 - k);          // (a slice of denormals only: scaled as far as a float factor reaches)
    const double down = 1.0 / (double)up;
#pragma unroll 4
    for (int q = threadIdx.x; q < n_pieces; q += SSS_THREADS) {
        const int r = q / cv, kk = q - r * cv;
        float x[V];
        if (V == 4) { const float4 t = reinterpret_cast<const float4 *>(mine_src)[q]; x[0] = t.x; x[V > 1 ? 1 : 0] = t.y; x[V > 2 ? 2 : 0] = t.z; x[V > 3 ? 3 : 0] = t.w; }
        else x[0] = mine_src[q];
        const int sl = idx[r0 + r];
        if (sl >= 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) atomicAdd(&acc64[sl * c + kk * V + v], (unsigned long long)(long long)llrintf(x[v] * up));
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < mc; j += SSS_THREADS) mine[j] = (float)((double)(long long)acc64[j] * down);
}

__global__ __launch_bounds__(1024) void scatter_sum_small_reduce_kernel(const float *__restrict__ partial, int n_parts, int elems, float *__restrict__ out)
{
    pcacc_reduce_partials<16>(partial, n_parts, elems, [&](int e, float v) { out[e] = v; });
}

static int sss_grid(int64_t n, int c, int64_t *rows_per_block)
{
    int64_t grid = (n * c + SSS_THREADS * 16 - 1) / (SSS_THREADS * 16);  // ~four 16-byte pieces per thread and pass before a second table pays
    if (grid > SSS_MAX_GRID) grid = SSS_MAX_GRID;
    if (grid < 1) grid = 1;
    int64_t rows = (n + grid - 1) / grid;
    const int64_t cap = SSS_SLICE_MAX / c;                              // a slice never exceeds 2^22 elements (the fixed-point head room)
    if (rows > cap) rows = cap;
    if (rows < 1) rows = 1;
    *rows_per_block = rows;
    return (int)((n + rows - 1) / rows);
}

extern "C" int pcacc_scatter_sum_small_workspace_bytes(int64_t n, int c, int m, size_t *bytes)
{
    if (!bytes || n < 0 || c <= 0 || m <= 0 || (int64_t)m * c > 8192) return PCACC_E_ARG;
    int64_t rows;
    *bytes = pcacc_align((size_t)(n > 0 ? sss_grid(n, c, &rows) : 1) * (size_t)m * c * sizeof(float));
    return PCACC_OK;
}

extern "C" int pcacc_scatter_sum_small(const float *src, const int32_t *idx, int64_t n, int c, int m, float *out, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    size_t need;
    if (n < 0 || c <= 0 || m <= 0 || (int64_t)m * c > 8192 || !out || pcacc_scatter_sum_small_workspace_bytes(n, c, m, &need) != PCACC_OK) return PCACC_E_ARG;
    hipStream_t s = pcacc_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(out, 0, (size_t)m * c * sizeof(float), s) != hipSuccess) return PCACC_E_LAUNCH;
        return PCACC_OK;
    }
    if (!src || !idx) return PCACC_E_ARG;
    if (!workspace || workspace_bytes < need) return PCACC_E_WORKSPACE;
    int64_t rows;
    const int grid = sss_grid(n, c, &rows), mc = m * c;
    const size_t lds = (size_t)mc * sizeof(unsigned long long);
    const bool vec = c % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    auto kern = vec ? scatter_sum_small_kernel<4> : scatter_sum_small_kernel<1>;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCACC_E_LAUNCH;
    float *partial = static_cast<float *>(workspace);
    kern<<<grid, SSS_THREADS, lds, s>>>(src, idx, n, c, mc, rows, partial);
    scatter_sum_small_reduce_kernel<<<(mc + 15) / 16, 1024, 0, s>>>(partial, grid, mc, out);
    PCACC_CHECK_LAUNCH();
    return PCACC_OK;
}
