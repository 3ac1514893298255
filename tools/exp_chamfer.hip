// Experiment harness for csrc/chamfer.hip (round 6): variants of the brute-force nearest-neighbour kernel timed against each other on
// 160 k x 160 k points, results compared bit for bit (distance AND index) with variant 0 = the round-5 kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/exp_chamfer tools/exp_chamfer.hip && build/exp_chamfer [n] [m]
// Variants: queries per lane (4 / 8), targets per skip test (8 / 16), scalar target loads one chunk ahead of their use, and the (query block x target split)
// task count rounded to a multiple of the chip's resident workgroups (no tail of a fifth workgroup on a quarter of the CUs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int QPT, int CHUNK, bool AHEAD>
__global__ __launch_bounds__(256) void nn_kernel(const float *__restrict__ q, int n, const float4 *__restrict__ tg, int m, int m_per_split,
                                                 unsigned long long *__restrict__ packed)
{
#pragma clang fp contract(off)
    const int k_begin = blockIdx.y * m_per_split;
    const int k_end = min(m, k_begin + m_per_split);
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * QPT;
    f32x2 qx[QPT / 2], qy[QPT / 2], qz[QPT / 2];
    float best[QPT];
    int besti[QPT];
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int j = min(j0 + r, n - 1);
        qx[r / 2][r & 1] = q[(int64_t)j * 3 + 0];
        qy[r / 2][r & 1] = q[(int64_t)j * 3 + 1];
        qz[r / 2][r & 1] = q[(int64_t)j * 3 + 2];
        best[r] = __builtin_inff();
        besti[r] = k_begin;
    }
    int k = k_begin;
    float4 cur[CHUNK], nxt[CHUNK];
    if (AHEAD && k + CHUNK <= k_end) {
#pragma unroll
        for (int c = 0; c < CHUNK; ++c) cur[c] = tg[k + c];
    }
    for (; k + CHUNK <= k_end; k += CHUNK) {
        if (AHEAD) {
            const int kn = (k + 2 * CHUNK <= k_end) ? k + CHUNK : k;      // the last chunk re-reads itself (uniform, cached)
#pragma unroll
            for (int c = 0; c < CHUNK; ++c) nxt[c] = tg[kn + c];
        } else {
#pragma unroll
            for (int c = 0; c < CHUNK; ++c) cur[c] = tg[k + c];
        }
        f32x2 d[CHUNK][QPT / 2];
#pragma unroll
        for (int c = 0; c < CHUNK; ++c) {
            const float tx = cur[c].x, ty = cur[c].y, tz = cur[c].z;
#pragma unroll
            for (int p = 0; p < QPT / 2; ++p) {
                const f32x2 x = (f32x2){tx, tx} - qx[p], y = (f32x2){ty, ty} - qy[p], z = (f32x2){tz, tz} - qz[p];
                d[c][p] = (x * x + y * y) + z * z;
            }
        }
        bool improve = false;
#pragma unroll
        for (int r = 0; r < QPT; ++r) {
            float mn = d[0][r / 2][r & 1];
#pragma unroll
            for (int c = 1; c < CHUNK; ++c) mn = fminf(mn, d[c][r / 2][r & 1]);
            improve |= mn < best[r];
        }
        if (__any(improve)) {
#pragma unroll
            for (int c = 0; c < CHUNK; ++c)
#pragma unroll
                for (int r = 0; r < QPT; ++r) {
                    const float dv = d[c][r / 2][r & 1];
                    const bool take = dv < best[r];
                    best[r] = take ? dv : best[r];
                    besti[r] = take ? k + c : besti[r];
                }
        }
        if (AHEAD) {
#pragma unroll
            for (int c = 0; c < CHUNK; ++c) cur[c] = nxt[c];
        }
    }
    for (; k < k_end; ++k) {
        const float4 t4 = tg[k];
#pragma unroll
        for (int r = 0; r < QPT; ++r) {
            const float x = t4.x - qx[r / 2][r & 1], y = t4.y - qy[r / 2][r & 1], z = t4.z - qz[r / 2][r & 1];
            const float dv = (x * x + y * y) + z * z;
            const bool take = dv < best[r];
            best[r] = take ? dv : best[r];
            besti[r] = take ? k : besti[r];
        }
    }
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int j = j0 + r;
        if (j >= n || k_begin >= k_end) continue;
        const unsigned long long key = ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)besti[r];
        atomicMin(&packed[j], key);
    }
}

// Targets staged through LDS in tiles of TILE (every lane reads the same 16 bytes: a broadcast ds_read_b128), all operands of the packed math in VGPRs.
template <int QPT, int CHUNK, int TILE>
__global__ __launch_bounds__(256) void nn_lds_kernel(const float *__restrict__ q, int n, const float4 *__restrict__ tg, int m, int m_per_split,
                                                     unsigned long long *__restrict__ packed)
{
#pragma clang fp contract(off)
    __shared__ float4 tile[TILE];
    const int k_begin = blockIdx.y * m_per_split;
    const int k_end = min(m, k_begin + m_per_split);
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * QPT;
    f32x2 qx[QPT / 2], qy[QPT / 2], qz[QPT / 2];
    float best[QPT];
    int besti[QPT];
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int j = min(j0 + r, n - 1);
        qx[r / 2][r & 1] = q[(int64_t)j * 3 + 0];
        qy[r / 2][r & 1] = q[(int64_t)j * 3 + 1];
        qz[r / 2][r & 1] = q[(int64_t)j * 3 + 2];
        best[r] = __builtin_inff();
        besti[r] = k_begin;
    }
    for (int t0 = k_begin; t0 < k_end; t0 += TILE) {
        __syncthreads();
        for (int i = threadIdx.x; i < TILE; i += 256) tile[i] = t0 + i < k_end ? tg[t0 + i] : make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);   // padding: never the nearest
        __syncthreads();
        const int cnt = min(TILE, k_end - t0);
        for (int kk = 0; kk < cnt; kk += CHUNK) {
            f32x2 d[CHUNK][QPT / 2];
#pragma unroll
            for (int c = 0; c < CHUNK; ++c) {
                const float4 t4 = tile[kk + c];
#pragma unroll
                for (int p = 0; p < QPT / 2; ++p) {
                    const f32x2 x = (f32x2){t4.x, t4.x} - qx[p], y = (f32x2){t4.y, t4.y} - qy[p], z = (f32x2){t4.z, t4.z} - qz[p];
                    d[c][p] = (x * x + y * y) + z * z;
                }
            }
            bool improve = false;
#pragma unroll
            for (int r = 0; r < QPT; ++r) {
                float mn = d[0][r / 2][r & 1];
#pragma unroll
                for (int c = 1; c < CHUNK; ++c) mn = fminf(mn, d[c][r / 2][r & 1]);
                improve |= mn < best[r];
            }
            if (__any(improve)) {
#pragma unroll
                for (int c = 0; c < CHUNK; ++c)
#pragma unroll
                    for (int r = 0; r < QPT; ++r) {
                        const float dv = d[c][r / 2][r & 1];
                        const bool take = dv < best[r] && t0 + kk + c < k_end;
                        best[r] = take ? dv : best[r];
                        besti[r] = take ? t0 + kk + c : besti[r];
                    }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int j = j0 + r;
        if (j >= n || k_begin >= k_end) continue;
        const unsigned long long key = ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)besti[r];
        atomicMin(&packed[j], key);
    }
}

// the same arithmetic without packed instructions (one query per VALU lane-op): is v_pk_* worth its encoding here?
template <int QPT, int CHUNK>
__global__ __launch_bounds__(256) void nn_scalar_kernel(const float *__restrict__ q, int n, const float4 *__restrict__ tg, int m, int m_per_split,
                                                        unsigned long long *__restrict__ packed)
{
#pragma clang fp contract(off)
    const int k_begin = blockIdx.y * m_per_split;
    const int k_end = min(m, k_begin + m_per_split);
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * QPT;
    float qx[QPT], qy[QPT], qz[QPT], best[QPT];
    int besti[QPT];
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int j = min(j0 + r, n - 1);
        qx[r] = q[(int64_t)j * 3 + 0], qy[r] = q[(int64_t)j * 3 + 1], qz[r] = q[(int64_t)j * 3 + 2];
        best[r] = __builtin_inff();
        besti[r] = k_begin;
    }
    int k = k_begin;
    for (; k + CHUNK <= k_end; k += CHUNK) {
        float d[CHUNK][QPT];
#pragma unroll
        for (int c = 0; c < CHUNK; ++c) {
            const float4 t4 = tg[k + c];
#pragma unroll
            for (int r = 0; r < QPT; ++r) {
                const float x = t4.x - qx[r], y = t4.y - qy[r], z = t4.z - qz[r];
                d[c][r] = (x * x + y * y) + z * z;
            }
        }
        bool improve = false;
#pragma unroll
        for (int r = 0; r < QPT; ++r) {
            float mn = d[0][r];
#pragma unroll
            for (int c = 1; c < CHUNK; ++c) mn = fminf(mn, d[c][r]);
            improve |= mn < best[r];
        }
        if (__any(improve)) {
#pragma unroll
            for (int c = 0; c < CHUNK; ++c)
#pragma unroll
                for (int r = 0; r < QPT; ++r) {
                    const bool take = d[c][r] < best[r];
                    best[r] = take ? d[c][r] : best[r];
                    besti[r] = take ? k + c : besti[r];
                }
        }
    }
    for (; k < k_end; ++k) {
        const float4 t4 = tg[k];
#pragma unroll
        for (int r = 0; r < QPT; ++r) {
            const float x = t4.x - qx[r], y = t4.y - qy[r], z = t4.z - qz[r];
            const float dv = (x * x + y * y) + z * z;
            const bool take = dv < best[r];
            best[r] = take ? dv : best[r];
            besti[r] = take ? k : besti[r];
        }
    }
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int j = j0 + r;
        if (j >= n || k_begin >= k_end) continue;
        atomicMin(&packed[j], ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)besti[r]);
    }
}

struct Variant { const char *name; int qpt; int balance; void (*launch)(dim3, const float *, int, const float4 *, int, int, unsigned long long *); };

template <int QPT, int CHUNK, bool AHEAD>
static void launch(dim3 grid, const float *q, int n, const float4 *tg, int m, int per, unsigned long long *packed)
{
    nn_kernel<QPT, CHUNK, AHEAD><<<grid, 256>>>(q, n, tg, m, per, packed);
}

template <int QPT, int CHUNK, int TILE>
static void launch_lds(dim3 grid, const float *q, int n, const float4 *tg, int m, int per, unsigned long long *packed)
{
    nn_lds_kernel<QPT, CHUNK, TILE><<<grid, 256>>>(q, n, tg, m, per, packed);
}
template <int QPT, int CHUNK>
static void launch_scalar(dim3 grid, const float *q, int n, const float4 *tg, int m, int per, unsigned long long *packed)
{
    nn_scalar_kernel<QPT, CHUNK><<<grid, 256>>>(q, n, tg, m, per, packed);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 160000, m = argc > 2 ? atoi(argv[2]) : 160000;
    std::vector<float> hq((size_t)n * 3);
    std::vector<float4> ht(m);
    srand(1);
    for (auto &v : hq) v = (rand() / (float)RAND_MAX - 0.5f) * 72.f;
    for (auto &t : ht) t = make_float4((rand() / (float)RAND_MAX - 0.5f) * 72.f, (rand() / (float)RAND_MAX - 0.5f) * 72.f, (rand() / (float)RAND_MAX) * 8.f - 2.f, 0.f);
    for (int i = 0; i < n; i += 7) hq[(size_t)i * 3 + 2] = ht[i % m].z, hq[(size_t)i * 3] = ht[i % m].x, hq[(size_t)i * 3 + 1] = ht[i % m].y;     // exact hits: distance 0, ties on duplicates
    for (int i = 0; i + 1 < m; i += 1001) ht[i + 1] = ht[i];                                                                                     // duplicate targets: the lower index must win
    float *q; float4 *tg; unsigned long long *packed, *ref;
    CK(hipMalloc(&q, hq.size() * 4)); CK(hipMalloc(&tg, (size_t)m * 16)); CK(hipMalloc(&packed, (size_t)n * 8)); CK(hipMalloc(&ref, (size_t)n * 8));
    CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(tg, ht.data(), (size_t)m * 16, hipMemcpyHostToDevice));
    Variant vs[] = {
        {"r5: 4 queries/lane, chunk 8", 4, 0, launch<4, 8, false>},
        {"4 q, chunk 8, balanced tasks", 4, 1, launch<4, 8, false>},
        {"4 q, chunk 8, loads ahead", 4, 0, launch<4, 8, true>},
        {"4 q, chunk 8, loads ahead, balanced", 4, 1, launch<4, 8, true>},
        {"4 q, chunk 16", 4, 1, launch<4, 16, false>},
        {"8 q, chunk 8", 8, 0, launch<8, 8, false>},
        {"8 q, chunk 8, balanced", 8, 1, launch<8, 8, false>},
        {"8 q, chunk 8, loads ahead, balanced", 8, 1, launch<8, 8, true>},
        {"8 q, chunk 4, balanced", 8, 1, launch<8, 4, false>},
        {"2 q, chunk 16, balanced", 2, 1, launch<2, 16, false>},
        {"4 q, chunk 8, LDS tiles of 1024, balanced", 4, 1, launch_lds<4, 8, 1024>},
        {"4 q, chunk 8, LDS tiles of 2048, balanced", 4, 1, launch_lds<4, 8, 2048>},
        {"8 q, chunk 8, LDS tiles of 1024, balanced", 8, 1, launch_lds<8, 8, 1024>},
        {"4 q, chunk 4, LDS tiles of 1024, balanced", 4, 1, launch_lds<4, 4, 1024>},
        {"4 q, chunk 8, no packed math, balanced", 4, 1, launch_scalar<4, 8>},
        {"2 q, chunk 8, no packed math, balanced", 2, 1, launch_scalar<2, 8>},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> href(n), hgot(n);
    for (size_t v = 0; v < sizeof(vs) / sizeof(vs[0]); ++v) {
        const int qblocks = (n + 256 * vs[v].qpt - 1) / (256 * vs[v].qpt);
        int splits;
        if (!vs[v].balance) {
            const int want = 256 * 4;
            splits = qblocks < want ? (want + qblocks - 1) / qblocks : 1;
        } else {
            // tasks = qblocks x splits as close as possible BELOW a multiple of 1024 (4 resident workgroups on each of 256 CUs), at least 2 rounds
            int best_s = 1; double best_fill = 0;
            for (int s = 1; s <= 64; ++s) {
                const int tasks = qblocks * s;
                if (tasks < 1024 || m / s < 2048) continue;
                const int rounds = (tasks + 1023) / 1024;
                const double fill = (double)tasks / (rounds * 1024.0);
                if (fill > best_fill + 1e-9) { best_fill = fill; best_s = s; }
            }
            splits = best_s;
        }
        int per = (m + splits - 1) / splits;
        per = (per + 15) / 16 * 16;
        splits = (m + per - 1) / per;
        float best_ms = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(packed, 0xFF, (size_t)n * 8));
            CK(hipEventRecord(e0));
            vs[v].launch(dim3(qblocks, splits), q, n, tg, m, per, packed);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_ms) best_ms = ms;
        }
        CK(hipMemcpy(hgot.data(), packed, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (v == 0) href = hgot;
        size_t bad = 0;
        for (int i = 0; i < n; ++i) bad += hgot[i] != href[i];
        printf("%-40s grid %4d x %2d  %7.3f ms  %6.1f TFLOP/s (8 flop per pair)  %s\n", vs[v].name, qblocks, splits, best_ms, 8.0 * n * (double)m / best_ms / 1e9,
               bad ? "RESULTS DIFFER" : "identical");
    }
    return 0;
}
