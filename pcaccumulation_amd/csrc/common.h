// Shared device helpers for libpcacc_hip.so (gfx950 only: wave64, 256 CUs, 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcacc.h"

#define PCACC_WAVE 64
#define PCACC_CUS 256

#define PCACC_CHECK_LAUNCH()                                   \
    do {                                                       \
        if (hipGetLastError() != hipSuccess) return PCACC_E_LAUNCH; \
    } while (0)

// A/B switches of the launchers (experiments and the equality tests; never set in production): environment variables read ONCE per process
// (a libc environment scan per convolution launch otherwise) -- pcacc_reload_switches() reads them again (tests change them at run time).
struct PcaccSwitches {
    bool conv_frame_major;   // PCACC_CONV_FRAME_MAJOR: bf16 resident convolution tiles frame by frame instead of frame-fastest
    bool conv_swz_off;       // PCACC_CONV_SWZ_OFF: 27-tap layers on the padded-row kernel of round 2
    bool rows_fm_off;        // PCACC_ROWS_FM_OFF: wide fp32x3 row layers on the weights-in-LDS kernel
    bool conv_plan;          // PCACC_CONV_PLAN: print the launch plans of the strip kernels
    char conv_res;           // PCACC_CONV_RES: '0' never the resident fp32x3 kernel, '2' whatever the size, 0 unset
    bool xcd_off;            // PCACC_XCD_REMAP=0: workgroups in launch order (round 4) instead of the XCD-contiguous walk of the multi-group convolution kernels
    char scatter_variant;    // PCACC_SCATTER_VARIANT: pillar_scatter_rows16 variants (pieces per lane / cache policy), 0 unset = the default kernel
    int scatter_blocks;      // PCACC_SCATTER_BLOCKS: workgroups per CU of the pillar-scatter launch (0 = default)
};
const PcaccSwitches &pcacc_switches();           // canvas.hip

static inline hipStream_t pcacc_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Grid for a grid-stride, HBM-bound kernel: enough workgroups to fill 256 CUs x 8 waves, never more
// than the work (cdna_hip_programming.md guideline 11).
static inline int pcacc_grid(int64_t work_items, int block, int max_blocks = PCACC_CUS * 8)
{
    int64_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

static inline size_t pcacc_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Hardware hands workgroup b of a launch to XCD b % 8 (each XCD has its own 4 MB L2).  pcacc_xcd_block maps b to a LOGICAL block id such that every XCD
// walks a CONTIGUOUS range of logical ids in increasing order: workgroups with neighbouring logical ids -- the output-channel groups of one pixel tile, the
// (co, ci) blocks of one strip -- run on the same XCD at about the same time and share their operand reads in its L2 instead of fetching them once per
// XCD (PMC, round 4: 1.9 - 2.7 x the algorithmic HBM bytes on these kernels).  A bijection on [0, n).
__device__ __forceinline__ int pcacc_xcd_block(int b, int n)
{
    const int per = (n + 7) >> 3, rem = n & 7, x = b & 7, i = b >> 3;
    return (rem == 0 || x < rem) ? x * per + i : rem * per + (x - rem) * (per - 1) + i;
}

// ---- bf16 <-> f32 (round to nearest even, the conversion torch uses) --------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// two fp32 -> two packed bf16 (round to nearest even) in one instruction: v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 pcacc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pcacc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pcacc_pack_bf16x2(float lo, float hi)
{
    const pcacc_f32x2 f = {lo, hi};
    const pcacc_bf16x2 r = __builtin_convertvector(f, pcacc_bf16x2);
    return *reinterpret_cast<const uint32_t *>(&r);
}
// the two halves of a packed pair as fp32
__device__ __forceinline__ float pcacc_bf16_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float pcacc_bf16_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
// 4 consecutive channels as float4, from f32 or packed-bf16 rows (index in units of 4 elements)
__device__ __forceinline__ float4 pcacc_ld4(const void *p, bool bf16, int64_t i4)
{
    if (!bf16) return reinterpret_cast<const float4 *>(p)[i4];
    const uint2 v = reinterpret_cast<const uint2 *>(p)[i4];
    return make_float4(pcacc_bf16_lo(v.x), pcacc_bf16_hi(v.x), pcacc_bf16_lo(v.y), pcacc_bf16_hi(v.y));
}
__device__ __forceinline__ void pcacc_st4(void *p, bool bf16, int64_t i4, float4 v)
{
    if (!bf16) reinterpret_cast<float4 *>(p)[i4] = v;
    else reinterpret_cast<uint2 *>(p)[i4] = make_uint2(pcacc_pack_bf16x2(v.x, v.y), pcacc_pack_bf16x2(v.z, v.w));
}

// ReLU backward on the fly: eight packed bf16 gradients, zeroed where the forward output y (same positions) is not > 0 -- what
// aten::threshold_backward(grad, y, 0) computes, fused into the staging of the kernels that consume the gradient.
__device__ __forceinline__ uint32_t pcacc_relu_mask2(uint32_t g, uint32_t y)
{
    const uint32_t lo = ((int32_t)(y << 16) > 0) ? 0x0000ffffu : 0u;
    const uint32_t hi = ((int32_t)(y & 0xffff0000u) > 0) ? 0xffff0000u : 0u;
    return g & (lo | hi);
}
__device__ __forceinline__ uint4 pcacc_relu_mask8(uint4 g, uint4 y)
{
    return make_uint4(pcacc_relu_mask2(g.x, y.x), pcacc_relu_mask2(g.y, y.y), pcacc_relu_mask2(g.z, y.z), pcacc_relu_mask2(g.w, y.w));
}

// Row stride (elements) of a channels-last bf16 LDS tile of `c` channels that is read with ds_read_b64_tr_b16: a 32-lane half of the
// instruction addresses 4 consecutive rows x 16 dwords (two 16-lane groups, 8 pieces of 8 bytes each), so the read is bank-conflict
// free iff the four row starts fall on the four 16-dword quarters of the 64 banks: stride = 16 or 48 dwords (mod 64).  The C + 4
// padding used before made rows 0 / 2 and 1 / 3 overlap: SQ_LDS_BANK_CONFLICT = 41-50 % of the LDS cycles of the weight-gradient
// kernels (profiles/r02_pmc_kernels_v1_summary.json).  32 channels need no padding at all.
__host__ __device__ constexpr int pcacc_tr_stride(int c)
{
    return c == 32 ? 32 : (c == 64 ? 96 : (c == 96 ? 96 : (c == 128 ? 160 : c + 4)));
}

// ---- [r6] precision-map experiment build (-DPCACC_X3_EXPERIMENT, tools/r06_precision_map.py; NEVER defined for the shipped library) ----------------------
// Which stages of the forward need all three terms of the fp32x3 product?  In this build the split kernels read a per-translation-unit device word that
// pcacc_x3_experiment() sets between stages: bit 0 -- activations without their lo half (two terms: (w_hi + w_lo) x_hi), bit 1 -- weights without theirs,
// bits 0 + 1 -- one fp16 term, bit 2 / 3 -- the hi half of activations / weights rounded to bf16's 8 significant bits (with the lo half dropped: what
// a one-term bf16 product of that stage computes, tensors still travelling as fp32).  The arithmetic is emulated at full cost: this build measures accuracy, not time.
#ifdef PCACC_X3_EXPERIMENT
__device__ __forceinline__ float pcacc_x_round_bf16(float v) { return bf16_to_f32(f32_to_bf16(v)); }
// (a, b) as the experiment sees them: rounded to bf16 precision when `f` has bit 2; -> true when the lo half is to be dropped (bit 0 or bit 2)
__device__ __forceinline__ bool pcacc_x_apply(int f, float &a, float &b)
{
    if (f & 4) { a = pcacc_x_round_bf16(a); b = pcacc_x_round_bf16(b); }
    return (f & 5) != 0;
}
#define PCACC_X_ACT(word) ((word) & 5)
#define PCACC_X_W(word) (((word) >> 1) & 5)
#endif

// ---- wave64 / block scans ------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// [r6] Fixed-order reduction of per-workgroup partial slots (weight / bias gradients): partial[p][e], p < n_parts, e < elems -> store(e, sum over p).
// The reduce kernels of rounds 2-5 cut the slots into 16-64 slices that met in `out` through fp32 atomicAdd: the order of those additions changed from run to
// run, and with it the last bit of every weight gradient -- a training step was not reproducible.  Here a workgroup of 1024 threads owns EL consecutive
// elements and SL = 1024 / EL interleaved slices of the slots (slice s takes slots s, s + SL, ...; the EL threads of a slice read consecutive elements of one
// slot); a thread keeps four running sums over its slots in a fixed order, and the SL slice sums of an element are added in slice order by one thread
// through LDS.  Same bits every run, no zero-filled output needed, no atomics.  Launch: (elems + EL - 1) / EL workgroups of 1024 threads.
template <int EL, class Store>
__device__ __forceinline__ void pcacc_reduce_partials(const float *__restrict__ partial, int n_parts, int elems, Store store)
{
    constexpr int SL = 1024 / EL;
    __shared__ float red[SL][EL + 1];
    const int el = threadIdx.x % EL, slice = threadIdx.x / EL;
    const int e = blockIdx.x * EL + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < elems) {
        int p = slice;
        for (; p + 3 * SL < n_parts; p += 4 * SL) {
            s0 += partial[(int64_t)p * elems + e];
            s1 += partial[(int64_t)(p + SL) * elems + e];
            s2 += partial[(int64_t)(p + 2 * SL) * elems + e];
            s3 += partial[(int64_t)(p + 3 * SL) * elems + e];
        }
        for (; p < n_parts; p += SL) s0 += partial[(int64_t)p * elems + e];
    }
    red[slice][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && e < elems) {
        float v = 0.f;
#pragma unroll 8
        for (int k = 0; k < SL; ++k) v += red[k][el];
        store(e, v);
    }
}
// elements per workgroup: long element ranges read whole 256-byte pieces of a slot per slice, short ones spread their slots over 64 slices
static inline int pcacc_reduce_el(int elems) { return elems >= 8192 ? 64 : 16; }

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}

// Exclusive prefix sum of `v` over a 256-thread block (4 waves); *total = block sum.
// `lds` must hold 4 ints.  Contains two __syncthreads().
__device__ __forceinline__ int block256_exclusive_scan(int v, int *lds, int *total)
{
    const int incl = wave_inclusive_scan(v);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 63) lds[w] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += (i < w) ? lds[i] : 0;
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + incl - v;
}

// The tiles of a persistent workgroup -- numbers first, first + stride, ... -- decoded 64 at a time on the vector unit: lane i holds the
// image, frame and tile coordinates of the workgroup's tile number base + i, and a pass picks its tile up with v_readlane.  Decoding
// tile -> (image, row, column) with scalar divisions in every pass came to ~400 scalar instructions per pass and wave next to 18 MFMAs:
// the waves of a workgroup did that arithmetic in step between two barriers with the matrix pipe idle (27-tap 32 -> 32 layer: MFMA
// 31 %, LDS 38 %, address unit 30 % busy, none of them the bound).
struct ConvTile { int img, fr, y0, x0; };
#define PCACC_WALK_OK(n_img, frames, tiles_y, tiles_x) ((n_img) < (1 << 24) && (frames) < 128 && (tiles_y) < 32768 && (tiles_x) < 65536)
struct ConvTileWalk {
    int first, stride, count;                                  // `count` tiles
    int tyx, tiles_x, frames, frame_fastest, th, tw;                // th x tw pixels per tile
    int v_if, v_yx;                                            // per lane: image | frame << 24 (PCACC_WALK_OK), tile row << 16 | tile column
    __device__ __forceinline__ void init(int lo, int hi, int slot, int slots, int tiles_y, int tiles_x_, int frames_, int ff, int th_, int tw_)
    {
        first = lo + slot; stride = slots; count = first < hi ? (hi - first + slots - 1) / slots : 0;
        tyx = tiles_y * tiles_x_; tiles_x = tiles_x_; frames = frames_; frame_fastest = ff; th = th_; tw = tw_;
        v_if = v_yx = 0;
    }
    __device__ __forceinline__ void refill(int k0)
    {
        const int t = first + (k0 + (int)(threadIdx.x & 63)) * stride;       // lanes past `count` decode numbers nobody reads
        int img, pos, fr;
        if (frame_fastest) {                                   // frame fastest inside a sample, then the position
            const int q = t / frames, smp = q / tyx;
            fr = t - q * frames; pos = q - smp * tyx; img = smp * frames + fr;
        } else {
            img = t / tyx; pos = t - img * tyx; fr = frames > 1 ? img % frames : 0;
        }
        const int ty = pos / tiles_x;
        v_if = img | fr << 24; v_yx = ty << 16 | (pos - ty * tiles_x);
    }
    __device__ __forceinline__ ConvTile get(int k)             // k ascending, each k once
    {
        if ((k & 63) == 0) refill(k);
        ConvTile t;
        const int a = __builtin_amdgcn_readlane(v_if, k & 63), b = __builtin_amdgcn_readlane(v_yx, k & 63);
        t.img = a & 0xffffff; t.fr = (unsigned)a >> 24;
        t.y0 = (b >> 16) * th; t.x0 = (b & 0xffff) * tw;
        return t;
    }
};
