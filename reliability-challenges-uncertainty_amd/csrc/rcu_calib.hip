// Calibration kernels: the ECE reliability histogram and the uncertainty-error counts.
//
// Reference semantics:
//   bin = np.digitize(p, linspace(0, 1+1e-8, n_bins+1)) - 1, then three bincounts
//                                                  common/evalutation/numpyfunctions.py:51-63
//   tp/tn/fp/fn and their "uncertain" subsets for a threshold on the uncertainty map
//                                                  common/evalutation/numpyfunctions.py:86-107
//   normalised entropy of [1-p, p]                 rechun/eval/analysis.py:196-203; numpyfunctions.py:166-168
//
// Both histograms are HBM scans over 6-10 bytes per voxel and must not be instruction-bound (a wave64 VALU
// instruction takes 4 cycles, so the budget at HBM speed is about 35 instructions per voxel).  Every lane owns a
// private column [bin][lane] of a per-wave LDS histogram and adds to it with ds_add (no return value, no
// conflicts, no cross-lane traffic in the streaming loop); once per workgroup the columns of its waves are added
// straight out of LDS (a few lanes per bin, a short wavefront butterfly to join them), the workgroup's sums go into
// per-volume accumulators with integer atomics, and a small second kernel writes the result.  Everything is integer
// arithmetic (the confidences as 2^-40 fixed point): counts and confidence sums are exact, whatever the order.
// Uncertainty thresholds that are not ascending take the general kernel (wavefront ballots per distinct key).
// Bin indices are bit-exact with np.digitize: p is compared against the float32 thresholds
// t_k = min{float32 t : t >= edge_k}.
#include "rcu_kernels.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <type_traits>

namespace rcu {

static constexpr int CB_THREADS = 256;
static constexpr int CB_WAVES = CB_THREADS / 64;
static constexpr int ELEMS_PER_BLOCK = CB_THREADS * 64;   // 16 rounds of 4 consecutive voxels per thread

struct BinThresholds {
    float t[MAX_BINS - 1];
    int n_bins;
};
static constexpr int UNC_CELLS = 64;

__host__ __device__ inline int unc_cell_of(float u, float lo, float scale)
{
    const float x = (u - lo) * scale;            // float32, the same two roundings on host and device
    int c = (x > 0.f) ? (int)x : 0;              // NaN and negatives -> cell 0
    return c < UNC_CELLS - 1 ? c : UNC_CELLS - 1;
}

struct UncThresholds {
    double t[MAX_THR];
    // float32 uncertainties: u > t[k] (in float64, as the reference compares) <=> u >= t32[k], t32[k] = the least float above t[k]
    float t32[MAX_THR];
    int n_thr;
    // float32 maps, ascending thresholds: cell table over f(u) = clamp(int((u - cell_lo) * cell_scale), 0, UNC_CELLS - 1).  f is monotone,
    // so with at most one threshold per cell  m = cell_base[f(u)] + (u >= cell_thr[f(u)])  counts exactly the thresholds u exceeds:
    // those in lower cells are below u, those in higher cells above it, the cell's own one is compared (n_cells = 0: no table).
    int n_cells;
    float cell_lo, cell_scale;
    float cell_thr[UNC_CELLS];
    unsigned char cell_base[UNC_CELLS];
};

// Uncertain-voxel sets of the reference as a function of the float32 probability (rcu_unc_counts_from_p): per threshold tau the set
// {p : ToEntropy([1 - p, p]) > tau} is, by an exhaustive run of the reference over every float32 in [0, 1]
// (tests/golden/generate_ue_boundaries.py -> rcu_ue_table.inc), an interval of bit patterns with, at each end, a ragged window of a few
// values described by a bit mask.
static constexpr int UE_P_CELLS = 256, UE_P_MASK_WORDS = 4;
struct UePRow {
    double thr;
    unsigned lo_first, lo_width, hi_first, hi_width;      // windows [first, first + width); members in between
    unsigned long long lo_mask[UE_P_MASK_WORDS], hi_mask[UE_P_MASK_WORDS];
};
static const UePRow UE_P_TABLE[] = {
#include "rcu_ue_table.inc"
};
static constexpr int UE_P_ROWS = (int)(sizeof(UE_P_TABLE) / sizeof(UE_P_TABLE[0]));
struct UePCell {          // one 1/256 slice of [0, 1]: m(p) = base + sign * past(p); past = p's bits >= end, or the mask bit inside [first, end)
    int first, end;
    short base, sign;
    unsigned mask_slot;
};
static_assert(sizeof(UePCell) == 16, "one ds_read_b128");

// One histogram word per voxel: the confidence as a 2^-40 fixed-point integer in bits 0..47, a count of one in bits 48..55 and the
// positive flag in bits 56..63.  A lane sees at most ECE_MAX_BLOCKS * ELEMS_PER_BLOCK / CB_THREADS = 128 voxels per workgroup, so neither
// count (<= 128) nor the sum (< 128 * 2 * 2^40 = 2^48) can carry into its neighbour, and ONE 64-bit LDS add records the voxel.  float32
// confidences >= 2^-16 convert exactly (24 mantissa bits above 2^-40), smaller ones are rounded to the nearest 2^-40, so a bin's confidence
// sum is the exact real sum to within 2^-41 per voxel -- and integer adds commute: the histogram does not depend on the launch geometry.
// Confidences are clamped to [0, 2): values a probability map cannot hold (the evaluation rejects them, rechun/eval/helper.py:8-12).
// (Round 3, later: 2^-40 and 8-bit fields instead of 2^-42 and 7-bit ones, so that a workgroup can take two blocks per prologue / reduction.)
static constexpr int ECE_FIX_BITS = 40, ECE_CNT_SHIFT = 48, ECE_POS_SHIFT = 56, ECE_MAX_BLOCKS = 2;

// hi_flags: the count / positive bytes of the word (bytes 2 and 3 of its high half)
__device__ __forceinline__ unsigned long long ece_word(float q, unsigned hi_flags)
{
    q = __builtin_amdgcn_fmed3f(q, 0.f, 1.99999988f);       // one instruction; NaN -> 0 (v_med3_f32 returns the minimum of the others)
    // q + 2^12 in float64 has its unit in the last place at 2^-40: the fraction field IS q * 2^40, rounded to nearest
    // (float32 values >= 2^-16 exactly); the bits of 4096.0 leave and the flags enter the high half in ONE three-operand add
    const unsigned long long d = (unsigned long long)__double_as_longlong((double)q + 4096.0);
    const unsigned hi = (unsigned)(d >> 32) + hi_flags + (0u - 0x40B00000u);
    return ((unsigned long long)hi << 32) | (unsigned)d;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__device__ __forceinline__ int bin_of(float p, const BinThresholds& th)
{
    int b = 0;
    for (int k = 0; k < th.n_bins - 1; ++k) b += (p >= th.t[k]) ? 1 : 0;
    return b;
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

static inline unsigned blocks_per_volume(size_t n) { return (unsigned)((n + ELEMS_PER_BLOCK - 1) / ELEMS_PER_BLOCK); }

// Blocks a workgroup of a streaming histogram kernel takes: its prologue (zeroing the columns) and its reduction cost a few per cent of
// a block's streaming time, so a large launch lets every workgroup take several consecutive blocks of its volume (measured, 160 BraTS
// volumes: 2 blocks +4 %, 8 blocks +5 %, 16 blocks -- too few workgroups left for the tail -- +2 %) while a small one keeps one block
// per workgroup and with it the chip full: at least four rounds of workgroups stay.  `forced`: rcu_calib_set_blocks_per_workgroup
// (test / tuning aid; 0 = this rule).
static int g_forced_blocks[2] = {0, 0};   // [ece, unc]
void calib_set_blocks_per_workgroup(int ece, int unc)
{
    g_forced_blocks[0] = ece;
    g_forced_blocks[1] = unc;
}
static inline unsigned blocks_per_workgroup(unsigned blocks, int n_volumes, unsigned cap, int forced)
{
    if (forced >= 1) return (unsigned)forced < cap ? (unsigned)forced : cap;
    const size_t resident = 256 * 6;   // workgroups the chip holds at a time (LDS-bound: 6-7 per CU)
    const size_t k = (size_t)blocks * n_volumes / (4 * resident);
    return (unsigned)(k < 1 ? 1 : k > cap ? cap : k);
}

// Workspace of the histogram: one accumulator row per volume and bin (count, positives, the fixed-point confidence sum as two
// 32-bit halves in 64-bit words: a volume's sum does not fit 64 bits, its halves do).  The workgroups add their sums with integer
// atomics -- exact and order-free -- and a one-workgroup kernel turns the rows into the result; the launcher zeroes the rows.
struct EceAccum {
    unsigned long long count, sum_pos, sum_lo, sum_hi;
};
size_t ece_workspace_bytes(size_t n_per_volume, int n_volumes)
{
    (void)n_per_volume;
    return (size_t)n_volumes * MAX_BINS * sizeof(EceAccum);
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// Dynamic LDS of the histogram kernel: [wave][bin][lane] histogram words (fixed-point confidence | count | positives, see above).
static inline size_t ece_lds_bytes(int n_bins) { return (size_t)CB_WAVES * n_bins * 64 * 8; }

// bin = #{k : p >= t_k}, computed as a candidate floor(p * n_bins) plus one table lookup.  The edges are
// k (1 + 1e-8) / n_bins, so p >= t_k implies p * n_bins > k and (k being representable, rounding monotonic)
// the float product is >= k: the candidate is the bin or the bin + 1; lut[c] = t_{c-1} (lut[0] = -inf) decides.
__device__ __forceinline__ int bin_lookup(float p, int n_bins, const float* lut)
{
    // clamped as a float (one v_med3_f32; NaN -> 0, +-inf saturate) and then truncated: the same integer as truncating first
    const int c = (int)__builtin_amdgcn_fmed3f(p * (float)n_bins, 0.f, (float)(n_bins - 1));
    return c - ((p < lut[c]) ? 1 : 0);
}

// threads per bin of the histogram kernel's final reduction: the largest power of two with parts * n_bins <= CB_THREADS (>= 8)
__device__ __forceinline__ int ece_parts(int nb)
{
    int parts = 64;
    while (parts * nb > CB_THREADS) parts >>= 1;
    return parts;
}

template <bool VEC>
__global__ __launch_bounds__(CB_THREADS) void ece_hist_kernel(const float* __restrict__ p, const uint8_t* __restrict__ target,
                                                               const uint8_t* __restrict__ mask, size_t n,
                                                               const BinThresholds th, EceAccum* __restrict__ accum, unsigned blocks_per_wg, unsigned nblocks)
{
    extern __shared__ unsigned long long ece_smem[];
    __shared__ float lut[MAX_BINS + 1];       // an object of its own: the compiler may then move the table reads across the histogram adds
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = th.n_bins;
    unsigned long long* const col = ece_smem + (size_t)wave * nb * 64 + lane;                // + bin * 64
    for (int b = 0; b < nb; ++b) col[b * 64] = 0ull;
    if (tid <= nb) lut[tid] = (tid == 0) ? -INFINITY : th.t[min(tid, MAX_BINS - 1) - 1];
    __syncthreads();
    const size_t vol = blockIdx.y;
    const float* pv = p + vol * n;
    const uint8_t* tv = target + vol * n;
    const uint8_t* mv = mask ? mask + vol * n : nullptr;
    // branch-free: a voxel outside the mask adds a zero word to whatever bin its confidence names (an exec-mask round trip per voxel
    // costs more than the LDS add it saves)
    auto add = [&](bool active, float q, bool pos) {
        const int b = bin_lookup(q, nb, lut);
        const unsigned hi = active ? ((1u << (ECE_CNT_SHIFT - 32)) | ((pos ? 1u : 0u) << (ECE_POS_SHIFT - 32))) : 0u;
        __hip_atomic_fetch_add(col + b * 64, ece_word(active ? q : 0.f, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    // four voxels of a mask / target word pair: A holds per byte 1 = inside the mask, P per byte 1 = positive and inside; a voxel's count /
    // positive fields -- bytes 6 and 7 of its word -- are ONE byte permute of the two (v_perm_b32: bytes 3, 2 <- P.k, A.k; bytes 1, 0 <- 0)
    auto nonzero_bytes = [](unsigned w) { return ((w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u; };
    auto add4 = [&](const float4& q, unsigned t4, unsigned m4) {
        const unsigned A = nonzero_bytes(m4);
        const unsigned P = A & nonzero_bytes(t4);
        const float qs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned hi = __builtin_amdgcn_perm(P, A, 0x00000c0cu | ((4u + k) << 24) | ((unsigned)k << 16));
            const int b = bin_lookup(qs[k], nb, lut);
            __hip_atomic_fetch_add(col + b * 64, ece_word(hi != 0u ? qs[k] : 0.f, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    };
    constexpr int ROUNDS = ELEMS_PER_BLOCK / (CB_THREADS * 4), BATCH = 4;
    // the workgroup's blocks_per_wg (<= ECE_MAX_BLOCKS) consecutive blocks of the volume: one prologue and one reduction for all of them
    const unsigned blk_end = min((blockIdx.x + 1) * blocks_per_wg, nblocks);
    for (unsigned blk = blockIdx.x * blocks_per_wg; blk < blk_end; ++blk) {
    const size_t base = (size_t)blk * ELEMS_PER_BLOCK;
    if (VEC && base + ELEMS_PER_BLOCK <= n) {
        // whole block inside the volume: the loads of BATCH rounds are issued before the first is consumed (with a bounds check per
        // round the compiler waits for every round), and the loads of the next BATCH rounds before this batch is worked on: a wave
        // keeps 6-12 KiB in flight through its arithmetic phases too
        float4 q[2][BATCH];
        unsigned t4[2][BATCH], m4[2][BATCH];
        auto load = [&](int r0, int s) {
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const size_t e = base + ((size_t)(r0 + i) * CB_THREADS + tid) * 4;
                q[s][i] = stream_load(reinterpret_cast<const float4*>(pv + e));      // read-once streams: non-temporal (rcu_kernels.h)
                t4[s][i] = stream_load(reinterpret_cast<const unsigned*>(tv + e));
                m4[s][i] = mv ? stream_load(reinterpret_cast<const unsigned*>(mv + e)) : 0x01010101u;
            }
        };
        load(0, 0);
#pragma unroll
        for (int g = 0; g < ROUNDS / BATCH; ++g) {
            if (g + 1 < ROUNDS / BATCH) load((g + 1) * BATCH, (g + 1) & 1);
#pragma unroll
            for (int i = 0; i < BATCH; ++i) add4(q[g & 1][i], t4[g & 1][i], m4[g & 1][i]);
        }
    } else if (VEC) {
        for (int r = 0; r < ROUNDS; ++r) {
            const size_t e = base + ((size_t)r * CB_THREADS + tid) * 4;
            if (e < n) {   // n % 4 == 0 on this path
                const float4 q = *reinterpret_cast<const float4*>(pv + e);
                const uchar4 t4 = *reinterpret_cast<const uchar4*>(tv + e);
                uchar4 m4 = make_uchar4(1, 1, 1, 1);
                if (mv) m4 = *reinterpret_cast<const uchar4*>(mv + e);
                add(m4.x != 0, q.x, t4.x != 0);
                add(m4.y != 0, q.y, t4.y != 0);
                add(m4.z != 0, q.z, t4.z != 0);
                add(m4.w != 0, q.w, t4.w != 0);
            }
        }
    } else {
        for (int r = 0; r < ELEMS_PER_BLOCK / CB_THREADS; ++r) {
            const size_t e = base + (size_t)r * CB_THREADS + tid;
            if (e < n) add(mv ? mv[e] != 0 : true, pv[e], tv[e] != 0);
        }
    }
    }
    // Reduction of the workgroup's CB_WAVES x 64 columns straight out of LDS (integers: exact whatever the order): bin b belongs to
    // `parts` consecutive threads (a power of two, 16 for ten bins), thread (b, k) adds the words j = k (mod parts) of the bin's 256
    // -- 16 reads per thread where a butterfly over the wave's lanes took 24 cross-lane moves per bin and lane --, then the parts meet
    // in a log2(parts)-step butterfly inside their wave.
    __syncthreads();
    {
        const int parts = ece_parts(nb);
        const int b = tid / parts, k = tid % parts;
        unsigned c = 0, cpos = 0;
        unsigned long long sm = 0;
        if (b < nb) {
            for (int i = 0; i < CB_WAVES * 64 / parts; ++i) {
                const int j = (k + (i + b) * parts) & (CB_WAVES * 64 - 1);   // start rotated by the bin: the bins of a wave read different banks
                const unsigned long long word = ece_smem[(size_t)(j >> 6) * nb * 64 + (size_t)b * 64 + (j & 63)];
                c += (unsigned)(word >> ECE_CNT_SHIFT) & 255u;
                cpos += (unsigned)(word >> ECE_POS_SHIFT);
                sm += word & ((1ull << ECE_CNT_SHIFT) - 1);
            }
        }
        for (int off = parts >> 1; off >= 1; off >>= 1) {
            c += __shfl_xor(c, off, 64);
            cpos += __shfl_xor(cpos, off, 64);
            sm += __shfl_xor(sm, off, 64);
        }
        if (k == 0 && b < nb && c != 0) {      // (an empty bin adds nothing)
            EceAccum* const row = accum + (size_t)vol * MAX_BINS + b;
            atomicAdd(&row->count, (unsigned long long)c);
            atomicAdd(&row->sum_pos, (unsigned long long)cpos);
            atomicAdd(&row->sum_lo, sm & 0xFFFFFFFFull);
            atomicAdd(&row->sum_hi, sm >> 32);
        }
    }
}

// Second stage: accumulator rows -> result.  The two halves of the fixed-point sum are joined and rounded ONCE to float64: the
// exact sum of the confidences to within 2^-41 per voxel, whatever order the workgroups added in.
static constexpr int RED_THREADS = 256;

__global__ __launch_bounds__(RED_THREADS) void ece_reduce_kernel(const EceAccum* __restrict__ accum, int n_volumes, EceResult* __restrict__ result)
{
    const int i = blockIdx.x * RED_THREADS + threadIdx.x;
    if (i >= n_volumes * MAX_BINS) return;
    const int vol = i / MAX_BINS, b = i % MAX_BINS;
    const EceAccum a = accum[i];
    unsigned long long hi = a.sum_hi + (a.sum_lo >> 32), lo = a.sum_lo & 0xFFFFFFFFull;
    result[vol].count[b] = a.count;
    result[vol].sum_pos[b] = a.sum_pos;
    result[vol].sum_conf[b] = ((double)hi * 4294967296.0 + (double)lo) * (1.0 / (double)(1ull << ECE_FIX_BITS));
}

hipError_t launch_ece_hist(const float* p, const uint8_t* target, const uint8_t* mask, size_t n, int n_volumes,
                           const float* thr_host, int n_bins, EceResult* result_dev, void* workspace, hipStream_t stream)
{
    if (n_bins < 1 || n_bins > MAX_BINS || n_volumes < 1) return hipErrorInvalidValue;
    BinThresholds th;
    th.n_bins = n_bins;
    for (int k = 0; k < MAX_BINS - 1; ++k) th.t[k] = (k < n_bins - 1) ? thr_host[k] : 0.f;
    const unsigned nb = blocks_per_volume(n);
    if (nb == 0) {   // empty input: all-zero histogram
        return hipMemsetAsync(result_dev, 0, sizeof(EceResult) * n_volumes, stream);
    }
    EceAccum* part = reinterpret_cast<EceAccum*>(workspace);
    {
        hipError_t e = hipMemsetAsync(part, 0, ece_workspace_bytes(n, n_volumes), stream);
        if (e != hipSuccess) return e;
    }
    const bool vec = (n % 4 == 0) && (reinterpret_cast<uintptr_t>(p) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(target) % 4 == 0) &&
                     (mask == nullptr || reinterpret_cast<uintptr_t>(mask) % 4 == 0);
    const size_t lds = ece_lds_bytes(n_bins);
    const void* fn = vec ? reinterpret_cast<const void*>(&ece_hist_kernel<true>) : reinterpret_cast<const void*>(&ece_hist_kernel<false>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const unsigned bpw = blocks_per_workgroup(nb, n_volumes, ECE_MAX_BLOCKS, g_forced_blocks[0]);
    const unsigned gx = (nb + bpw - 1) / bpw;
    if (vec)
        hipLaunchKernelGGL(ece_hist_kernel<true>, dim3(gx, n_volumes), dim3(CB_THREADS), lds, stream, p, target, mask, n, th,
                           part, bpw, nb);
    else
        hipLaunchKernelGGL(ece_hist_kernel<false>, dim3(gx, n_volumes), dim3(CB_THREADS), lds, stream, p, target, mask, n,
                           th, part, bpw, nb);
    hipLaunchKernelGGL(ece_reduce_kernel, dim3((unsigned)((n_volumes * MAX_BINS + RED_THREADS - 1) / RED_THREADS)), dim3(RED_THREADS), 0, stream, part,
                       n_volumes, result_dev);
    return hipGetLastError();
}

__global__ __launch_bounds__(CB_THREADS) void bin_ids_kernel(const float* __restrict__ p, size_t n, const BinThresholds th,
                                                              uint8_t* __restrict__ ids)
{
    const size_t i = (size_t)blockIdx.x * CB_THREADS + threadIdx.x;
    if (i < n) ids[i] = (uint8_t)bin_of(p[i], th);
}

hipError_t launch_bin_ids(const float* p, size_t n, const float* thr_host, int n_bins, uint8_t* ids, hipStream_t stream)
{
    if (n_bins < 1 || n_bins > MAX_BINS) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    BinThresholds th;
    th.n_bins = n_bins;
    for (int k = 0; k < MAX_BINS - 1; ++k) th.t[k] = (k < n_bins - 1) ? thr_host[k] : 0.f;
    hipLaunchKernelGGL(bin_ids_kernel, dim3((unsigned)((n + CB_THREADS - 1) / CB_THREADS)), dim3(CB_THREADS), 0, stream, p,
                       n, th, ids);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- uncertainty-error counts
// key = cell (tp=0, tn=1, fp=2, fn=3) | bitmask of exceeded thresholds << 2
static constexpr int UNC_SLOTS = (MAX_THR + 1) * 4;   // [t][cell], t == n_thr row holds the base counts
static constexpr int UNC_MAX_BLOCKS = 8;              // blocks per workgroup of the sorted kernel: 8 * 64 voxels per lane fit its 16-bit fields

// Workspace of the counts: one row of UNC_SLOTS accumulators per volume, added to with integer atomics by the workgroups (exact,
// order-free) and turned into the output layout by a small second kernel; the launcher zeroes it.
size_t unc_workspace_bytes(size_t n_per_volume, int n_volumes)
{
    (void)n_per_volume;
    return (size_t)n_volumes * UNC_SLOTS * sizeof(unsigned long long);
}

__device__ __forceinline__ void unc_wave_update(bool active, unsigned key, unsigned* w_slots, int n_thr, int lane)
{
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned k0 = __shfl(key, leader, 64);
        const unsigned long long grp = __ballot(active && key == k0);
        const unsigned c = (unsigned)__popcll(grp);
        const unsigned cell = k0 & 3u, bits = k0 >> 2;
        // lane t < n_thr owns threshold t; lane n_thr owns the base row
        if (lane < n_thr) {
            if ((bits >> lane) & 1u) w_slots[lane * 4 + cell] += c;
        } else if (lane == n_thr) {
            w_slots[n_thr * 4 + cell] += c;
        }
        todo &= ~grp;
    }
}

template <typename U>
__global__ __launch_bounds__(CB_THREADS) void unc_counts_kernel(const U* __restrict__ unc, const uint8_t* __restrict__ pred,
                                                                 const uint8_t* __restrict__ target,
                                                                 const uint8_t* __restrict__ mask, size_t n,
                                                                 const UncThresholds th,
                                                                 unsigned long long* __restrict__ partial)
{
    __shared__ unsigned s_slots[CB_WAVES][UNC_SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < CB_WAVES * UNC_SLOTS; i += CB_THREADS) (&s_slots[0][0])[i] = 0;
    __syncthreads();
    const size_t vol = blockIdx.y;
    const U* uv = unc + vol * n;
    const uint8_t* pv = pred + vol * n;
    const uint8_t* tv = target + vol * n;
    const uint8_t* mv = mask ? mask + vol * n : nullptr;
    const size_t base = (size_t)blockIdx.x * ELEMS_PER_BLOCK;
    for (int r = 0; r < ELEMS_PER_BLOCK / CB_THREADS; ++r) {
        const size_t e = base + (size_t)r * CB_THREADS + tid;
        const bool in = e < n;
        bool act = in;
        unsigned key = 0;
        if (in) {
            const double u = (double)uv[e];
            const bool pr = pv[e] != 0, tg = tv[e] != 0;
            if (mv) act = mv[e] != 0;
            const unsigned cell = tg ? (pr ? 0u : 3u) : (pr ? 2u : 1u);
            unsigned bits = 0;
            for (int t = 0; t < th.n_thr; ++t) bits |= (u > th.t[t]) ? (1u << t) : 0u;
            key = cell | (bits << 2);
        }
        unc_wave_update(act, key, s_slots[wave], th.n_thr, lane);
    }
    __syncthreads();
    if (tid < UNC_SLOTS) {
        unsigned long long c = 0;
        for (int w = 0; w < CB_WAVES; ++w) c += s_slots[w][tid];
        if (c != 0) atomicAdd(partial + (size_t)vol * UNC_SLOTS + tid, c);
    }
}

// Ascending thresholds: m = #{t : u > th_t} identifies the set of exceeded thresholds (the first m), so one
// ds_add into the lane's private column [m][cell] records the voxel; count_uncertain[t][cell] = sum_{m > t} col[m][cell]
// and the base counts are the column sums.  Writes the same partial layout as the general kernel.
__device__ __forceinline__ void load4(const float* src, float (&q)[4])
{
    const float4 v = stream_load(reinterpret_cast<const float4*>(src));
    q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
}
__device__ __forceinline__ void load4(const double* src, double (&q)[4])
{
    const double2 a = stream_load(reinterpret_cast<const double2*>(src)), b = stream_load(reinterpret_cast<const double2*>(src + 2));
    q[0] = a.x, q[1] = a.y, q[2] = b.x, q[3] = b.y;
}

// FROM_P: `unc` is the float32 foreground-probability map itself and "uncertain" is decided by the table of the reference's own
// float32 -> {uncertain, not} sets (UePCell below): no entropy map, no log.
template <typename U, bool FROM_P = false>
__global__ __launch_bounds__(CB_THREADS) void unc_counts_sorted_kernel(const U* __restrict__ unc, const uint8_t* __restrict__ pred,
                                                                        const uint8_t* __restrict__ target,
                                                                        const uint8_t* __restrict__ mask, size_t n,
                                                                        const UncThresholds th,
                                                                        unsigned long long* __restrict__ partial, unsigned blocks_per_wg, unsigned nblocks,
                                                                        const UePCell* __restrict__ p_cells = nullptr,
                                                                        const unsigned long long* __restrict__ p_masks = nullptr)
{
    extern __shared__ unsigned unc_smem[];           // [wave][(n_thr + 1) * 2][lane], two 16-bit cell counters per word
    __shared__ unsigned s_w[CB_WAVES][UNC_SLOTS];    // per wave: [m][cell] totals
    __shared__ float2 s_cell[FROM_P ? 1 : UNC_CELLS];   // (threshold of the cell, thresholds below the cell as an integer in a float's bits): ONE read
    __shared__ UePCell s_pcell[FROM_P ? UE_P_CELLS : 1];
    if constexpr (FROM_P) {
        static_assert(UE_P_CELLS == CB_THREADS, "one cell per thread");
        s_pcell[threadIdx.x] = p_cells[threadIdx.x];
    } else {
        if (threadIdx.x < UNC_CELLS) s_cell[threadIdx.x] = make_float2(th.cell_thr[threadIdx.x], __uint_as_float(th.cell_base[threadIdx.x]));
    }
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncol = (th.n_thr + 1) * 2;
    unsigned* const col = unc_smem + (size_t)wave * ncol * 64 + lane;
    for (int k = 0; k < ncol; ++k) col[k * 64] = 0u;
    const size_t vol = blockIdx.y;
    const U* uv = unc + vol * n;
    const uint8_t* pv = pred + vol * n;
    const uint8_t* tv = target + vol * n;
    const uint8_t* mv = mask ? mask + vol * n : nullptr;
    // m = number of thresholds u exceeds
    auto exceeded = [&](U u) {
        int m = 0;
        if constexpr (FROM_P) {
            // cell = the 1/256-wide slice of [0, 1] the probability sits in (NaN, negatives -> 0); a cell holds at most one edge of one
            // threshold's set: m = (sets that hold the cell's start) +- (is p past the edge?), the edge's ragged window by its bit mask
            const int b = (int)__float_as_uint(u);          // signed compares: negative floats sort below every edge
            const UePCell e = s_pcell[(int)__builtin_amdgcn_fmed3f(u * (float)UE_P_CELLS, 0.f, (float)(UE_P_CELLS - 1))];
            int past = (b >= e.end) ? 1 : 0;
            if (b >= e.first && b < e.end) {                // inside a ragged window (a handful of float32 values per table)
                const unsigned d = (unsigned)(b - e.first);
                past = (int)((p_masks[(size_t)e.mask_slot * UE_P_MASK_WORDS + (d >> 6)] >> (d & 63u)) & 1ull);
            }
            m = e.base + e.sign * past;
        } else if constexpr (std::is_same<U, float>::value) {   // full-rate float32 compares instead of float64 ones
            if (th.n_cells > 0) {
                // unc_cell_of with the clamp made on the float (one v_med3_f32; NaN -> 0): the same integer
                const int c = (int)__builtin_amdgcn_fmed3f((u - th.cell_lo) * th.cell_scale, 0.f, (float)(UNC_CELLS - 1));
                const float2 e = s_cell[c];
                m = (int)__float_as_uint(e.y) + ((u >= e.x) ? 1 : 0);
            } else {
                for (int t = 0; t < th.n_thr; ++t) m += (u >= th.t32[t]) ? 1 : 0;
            }
        } else {
            for (int t = 0; t < th.n_thr; ++t) m += ((double)u > th.t[t]) ? 1 : 0;
        }
        return m;
    };
    // cell = tp 0, tn 1, fp 2, fn 3 lives in column word (cell >> 1) = target XOR prediction, half (cell & 1) = NOT prediction
    auto add = [&](bool active, U u, bool pr, bool tg) {
        const int m = exceeded(u);
        const int cell = tg ? (pr ? 0 : 3) : (pr ? 2 : 1);
        // branch-free: a voxel outside the mask adds zero
        __hip_atomic_fetch_add(col + (m * 2 + (cell >> 1)) * 64, active ? (1u << ((cell & 1) * 16)) : 0u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    // four voxels of a (prediction, target, mask) word triple with the byte logic done once per word: X = target XOR prediction per
    // byte (the column word), W = per byte 0x01 (inside the mask, predicted positive: low half) or 0x10 (inside, predicted negative:
    // high half) or 0 -- (W_k * 0x1001) & 0x10001 is the voxel's increment
    auto nonzero_bytes = [](unsigned w) { return ((w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u; };
    auto add4 = [&](const U (&q)[4], unsigned p4, unsigned t4, unsigned m4) {
        const unsigned A = nonzero_bytes(m4), P = nonzero_bytes(p4), T = nonzero_bytes(t4);
        const unsigned X = T ^ P;
        const unsigned W = (A & P) | ((A & ~P) << 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = exceeded(q[k]);
            const unsigned x = (X >> (8 * k)) & 1u, wk = (W >> (8 * k)) & 0xffu;
            __hip_atomic_fetch_add(col + (m * 2 + (int)x) * 64, (wk * 0x1001u) & 0x10001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    };
    const bool vec = (n % 4 == 0) && (reinterpret_cast<uintptr_t>(uv) % (4 * sizeof(U)) == 0) &&
                     (reinterpret_cast<uintptr_t>(pv) % 4 == 0) && (reinterpret_cast<uintptr_t>(tv) % 4 == 0) &&
                     (mv == nullptr || reinterpret_cast<uintptr_t>(mv) % 4 == 0);
    constexpr int ROUNDS = ELEMS_PER_BLOCK / (CB_THREADS * 4), BATCH = 4;
    // the workgroup's blocks_per_wg (<= UNC_MAX_BLOCKS) consecutive blocks of the volume: one prologue and one reduction for all of them
    // (a lane adds at most 64 voxels per block to a 16-bit field)
    const unsigned blk_end = min((blockIdx.x + 1) * blocks_per_wg, nblocks);
    for (unsigned blk = blockIdx.x * blocks_per_wg; blk < blk_end; ++blk) {
    const size_t base = (size_t)blk * ELEMS_PER_BLOCK;
    if (vec && base + ELEMS_PER_BLOCK <= n) {
        // loads of BATCH rounds in flight together, and those of the next BATCH rounds issued before this batch is worked on
        U q[2][BATCH][4];
        unsigned p4[2][BATCH], t4[2][BATCH], m4[2][BATCH];
        auto load = [&](int r0, int s) {
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const size_t e = base + ((size_t)(r0 + i) * CB_THREADS + tid) * 4;
                load4(uv + e, q[s][i]);
                p4[s][i] = stream_load(reinterpret_cast<const unsigned*>(pv + e));
                t4[s][i] = stream_load(reinterpret_cast<const unsigned*>(tv + e));
                m4[s][i] = mv ? stream_load(reinterpret_cast<const unsigned*>(mv + e)) : 0x01010101u;
            }
        };
        load(0, 0);
#pragma unroll
        for (int g = 0; g < ROUNDS / BATCH; ++g) {
            if (g + 1 < ROUNDS / BATCH) load((g + 1) * BATCH, (g + 1) & 1);
#pragma unroll
            for (int i = 0; i < BATCH; ++i) add4(q[g & 1][i], p4[g & 1][i], t4[g & 1][i], m4[g & 1][i]);
        }
    } else if (vec) {
        for (int r = 0; r < ROUNDS; ++r) {
            const size_t e = base + ((size_t)r * CB_THREADS + tid) * 4;
            if (e < n) {
                U q[4];
                load4(uv + e, q);
                const uchar4 p4 = *reinterpret_cast<const uchar4*>(pv + e);
                const uchar4 t4 = *reinterpret_cast<const uchar4*>(tv + e);
                uchar4 m4 = make_uchar4(1, 1, 1, 1);
                if (mv) m4 = *reinterpret_cast<const uchar4*>(mv + e);
                add(m4.x != 0, q[0], p4.x != 0, t4.x != 0);
                add(m4.y != 0, q[1], p4.y != 0, t4.y != 0);
                add(m4.z != 0, q[2], p4.z != 0, t4.z != 0);
                add(m4.w != 0, q[3], p4.w != 0, t4.w != 0);
            }
        }
    } else {
        for (int r = 0; r < ELEMS_PER_BLOCK / CB_THREADS; ++r) {
            const size_t e = base + (size_t)r * CB_THREADS + tid;
            if (e < n) add(mv ? mv[e] != 0 : true, uv[e], pv[e] != 0, tv[e] != 0);
        }
    }
    }
    // Reduction of the wave's 64 columns straight out of LDS: lane (column k = lane >> 1, half h = lane & 1) adds the 32 words
    // k * 64 + h, + 2, ... of column k (two 16-bit fields each, at most 64 * UNC_MAX_BLOCKS per lane: no carry), then the two halves meet -- 32 reads
    // per lane for 32 columns at a time where a butterfly per column took 12 cross-lane moves per column and lane.
    __syncthreads();
    {
        const unsigned* const wcol = unc_smem + (size_t)wave * ncol * 64;
        for (int k0 = 0; k0 < ncol; k0 += 32) {
            const int k = k0 + (lane >> 1), h = lane & 1;
            unsigned lo = 0, hi = 0;
            if (k < ncol) {
                for (int j = h; j < 64; j += 2) {
                    const unsigned v = wcol[k * 64 + ((j + 2 * (lane >> 1)) & 63)];   // rotated start: the 64 lanes read 64 different banks
                    lo += v & 0xffffu;
                    hi += v >> 16;
                }
            }
            lo += __shfl_xor(lo, 1, 64);
            hi += __shfl_xor(hi, 1, 64);
            if (k < ncol && h == 0) {
                s_w[wave][2 * k] = lo;
                s_w[wave][2 * k + 1] = hi;
            }
        }
    }
    __syncthreads();
    if (tid < UNC_SLOTS) {
        const int t = tid / 4, cell = tid % 4;   // output slot [t][cell]; t == n_thr is the base row
        unsigned long long c = 0;
        if (t <= th.n_thr) {
            const int m_lo = (t < th.n_thr) ? t + 1 : 0;
            for (int w = 0; w < CB_WAVES; ++w)
                for (int m = m_lo; m <= th.n_thr; ++m) c += s_w[w][m * 4 + cell];
        }
        if (c != 0) atomicAdd(partial + (size_t)vol * UNC_SLOTS + tid, c);
    }
}

__global__ __launch_bounds__(RED_THREADS) void unc_reduce_kernel(const unsigned long long* __restrict__ accum, int n_volumes, int n_thr,
                                                                  unsigned long long* __restrict__ out)
{
    const int i = blockIdx.x * RED_THREADS + threadIdx.x;      // (volume, threshold, cell)
    if (i >= n_volumes * n_thr * 4) return;
    const int cell = i % 4, t = (i / 4) % n_thr, vol = i / (4 * n_thr);
    const unsigned long long* const row = accum + (size_t)vol * UNC_SLOTS;
    unsigned long long* o = out + ((size_t)vol * n_thr + t) * 8;
    o[cell] = row[n_thr * 4 + cell];      // tp, tn, fp, fn (same for every threshold)
    o[4 + cell] = row[t * 4 + cell];      // tpu, tnu, fpu, fnu
}

hipError_t launch_unc_counts(const void* unc, int unc_is_f64, const uint8_t* prediction, const uint8_t* target,
                             const uint8_t* mask, size_t n, int n_volumes, const double* thr_host, int n_thr,
                             unsigned long long* out_dev, void* workspace, hipStream_t stream)
{
    if (n_thr < 1 || n_thr > MAX_THR || n_volumes < 1) return hipErrorInvalidValue;
    UncThresholds th;
    th.n_thr = n_thr;
    for (int t = 0; t < MAX_THR; ++t) {
        th.t[t] = (t < n_thr) ? thr_host[t] : 0.0;
        float f = (float)th.t[t];                                   // nearest float
        if (!((double)f > th.t[t])) f = std::nextafter(f, INFINITY);  // the least float strictly above the threshold
        th.t32[t] = f;
    }
    th.n_cells = 0;
    th.cell_lo = 0.f;
    th.cell_scale = 0.f;
    for (int c = 0; c < UNC_CELLS; ++c) {
        th.cell_thr[c] = INFINITY;
        th.cell_base[c] = 0;
    }
    {
        bool asc = true;
        for (int t = 1; t < n_thr; ++t) asc = asc && (thr_host[t - 1] < thr_host[t]);
        const float lo = th.t32[0], hi = th.t32[n_thr - 1];
        if (asc && n_thr >= 2 && std::isfinite(lo) && std::isfinite(hi) && hi > lo) {
            const float scale = (float)(UNC_CELLS - 1) / (hi - lo);
            bool one_per_cell = std::isfinite(scale);
            int prev = -1;
            for (int t = 0; t < n_thr && one_per_cell; ++t) {
                const int c = unc_cell_of(th.t32[t], lo, scale);
                one_per_cell = c > prev;
                prev = c;
            }
            if (one_per_cell) {
                th.n_cells = UNC_CELLS;
                th.cell_lo = lo;
                th.cell_scale = scale;
                int below = 0;   // thresholds in lower cells
                for (int c = 0, t = 0; c < UNC_CELLS; ++c) {
                    th.cell_base[c] = (unsigned char)below;
                    if (t < n_thr && unc_cell_of(th.t32[t], lo, scale) == c) {
                        th.cell_thr[c] = th.t32[t];
                        ++t;
                        ++below;
                    }
                }
            }
        }
    }
    const unsigned nb = blocks_per_volume(n);
    if (nb == 0) return hipMemsetAsync(out_dev, 0, sizeof(unsigned long long) * 8 * n_thr * n_volumes, stream);
    unsigned long long* part = reinterpret_cast<unsigned long long*>(workspace);
    {
        hipError_t e = hipMemsetAsync(part, 0, unc_workspace_bytes(n, n_volumes), stream);
        if (e != hipSuccess) return e;
    }
    bool ascending = true;
    for (int t = 1; t < n_thr; ++t) ascending = ascending && (thr_host[t - 1] <= thr_host[t]);
    if (ascending) {
        const size_t lds = (size_t)CB_WAVES * (n_thr + 1) * 2 * 64 * sizeof(unsigned);
        const unsigned bpw = blocks_per_workgroup(nb, n_volumes, UNC_MAX_BLOCKS, g_forced_blocks[1]);
        const unsigned gx = (nb + bpw - 1) / bpw;
        if (unc_is_f64)
            hipLaunchKernelGGL(unc_counts_sorted_kernel<double>, dim3(gx, n_volumes), dim3(CB_THREADS), lds, stream,
                               reinterpret_cast<const double*>(unc), prediction, target, mask, n, th, part, bpw, nb);
        else
            hipLaunchKernelGGL(unc_counts_sorted_kernel<float>, dim3(gx, n_volumes), dim3(CB_THREADS), lds, stream,
                               reinterpret_cast<const float*>(unc), prediction, target, mask, n, th, part, bpw, nb);
    } else if (unc_is_f64)
        hipLaunchKernelGGL(unc_counts_kernel<double>, dim3(nb, n_volumes), dim3(CB_THREADS), 0, stream,
                           reinterpret_cast<const double*>(unc), prediction, target, mask, n, th, part);
    else
        hipLaunchKernelGGL(unc_counts_kernel<float>, dim3(nb, n_volumes), dim3(CB_THREADS), 0, stream,
                           reinterpret_cast<const float*>(unc), prediction, target, mask, n, th, part);
    hipLaunchKernelGGL(unc_reduce_kernel, dim3((unsigned)((n_volumes * n_thr * 4 + RED_THREADS - 1) / RED_THREADS)), dim3(RED_THREADS), 0, stream, part,
                       n_volumes, n_thr, out_dev);
    return hipGetLastError();
}

// ---- the same counts from the float32 probability map (rcu_unc_counts_from_p)
static inline int ue_p_cell_of(unsigned bits)
{
    float u;
    std::memcpy(&u, &bits, 4);
    const float x = u * (float)UE_P_CELLS;      // exact (a power of two): the device's product
    const int c = (x > 0.f) ? (int)x : 0;
    return c < UE_P_CELLS - 1 ? c : UE_P_CELLS - 1;
}

// Row of the table for a threshold (exact match of the double), or -1
static int ue_p_row_of(double thr)
{
    for (int r = 0; r < UE_P_ROWS; ++r)
        if (UE_P_TABLE[r].thr == thr) return r;
    return -1;
}

// Cells + masks for the thresholds thr[0..n_thr) (strictly ascending, all in the table).  false: a threshold is not in the table, or two
// edges share a cell / a window straddles a cell border (cannot happen with the shipped table: checked by tests/test_abi_cpu.py).
static bool ue_p_build(const double* thr, int n_thr, UePCell* cells, unsigned long long* masks)
{
    if (n_thr < 1 || n_thr > MAX_THR) return false;
    struct Edge { unsigned first, end; int sign; unsigned long long mask[UE_P_MASK_WORDS]; };
    Edge edges[2 * MAX_THR];
    int n_edges = 0;
    for (int t = 0; t < n_thr; ++t) {
        if (t > 0 && !(thr[t - 1] < thr[t])) return false;
        const int r = ue_p_row_of(thr[t]);
        if (r < 0) return false;
        const UePRow& R = UE_P_TABLE[r];
        Edge up{R.lo_first, R.lo_first + R.lo_width, +1, {}}, down{R.hi_first, R.hi_first + R.hi_width, -1, {}};
        for (int w = 0; w < UE_P_MASK_WORDS; ++w) {
            up.mask[w] = R.lo_mask[w];           // past the rising edge = member
            down.mask[w] = ~R.hi_mask[w];        // past the falling edge = NOT member
        }
        edges[n_edges++] = up;
        edges[n_edges++] = down;
    }
    // nested sets: rising edges ascend with the threshold, falling edges descend -- sort by position
    for (int i = 1; i < n_edges; ++i)
        for (int j = i; j > 0 && edges[j].first < edges[j - 1].first; --j) std::swap(edges[j], edges[j - 1]);
    for (int c = 0; c < UE_P_CELLS; ++c) cells[c] = UePCell{0x7fffffff, 0x7fffffff, 0, 0, 0u};
    for (int w = 0; w < UE_P_MASK_WORDS; ++w) masks[w] = 0ull;      // slot 0: no window
    int m = 0, prev_cell = -1;
    for (int e = 0; e < n_edges; ++e) {
        const Edge& E = edges[e];
        const int c0 = ue_p_cell_of(E.first), c1 = ue_p_cell_of(E.end);
        if (c0 != c1 || c0 <= prev_cell || E.end - E.first > 64u * UE_P_MASK_WORDS) return false;
        for (int c = prev_cell + 1; c < c0; ++c) cells[c].base = (short)m;
        cells[c0] = UePCell{(int)E.first, (int)E.end, (short)m, (short)E.sign, (unsigned)(e + 1)};
        for (int w = 0; w < UE_P_MASK_WORDS; ++w) masks[(size_t)(e + 1) * UE_P_MASK_WORDS + w] = E.mask[w];
        m += E.sign;
        if (m < 0 || m > n_thr) return false;
        prev_cell = c0;
    }
    for (int c = prev_cell + 1; c < UE_P_CELLS; ++c) cells[c].base = (short)m;
    return m == 0;
}

int unc_from_p_num_thresholds() { return UE_P_ROWS; }
double unc_from_p_threshold(int i) { return (i >= 0 && i < UE_P_ROWS) ? UE_P_TABLE[i].thr : -1.0; }
bool unc_from_p_supported(const double* thr, int n_thr)
{
    UePCell cells[UE_P_CELLS];
    unsigned long long masks[(2 * MAX_THR + 1) * UE_P_MASK_WORDS];
    return ue_p_build(thr, n_thr, cells, masks);
}

static constexpr size_t UE_P_TABLE_BYTES = sizeof(UePCell) * UE_P_CELLS + sizeof(unsigned long long) * (2 * MAX_THR + 1) * UE_P_MASK_WORDS;
size_t unc_from_p_workspace_bytes(size_t n_per_volume, int n_volumes)
{
    return unc_workspace_bytes(n_per_volume, n_volumes) + UE_P_TABLE_BYTES;
}

// Reference-side evaluation of the table for one probability (test aid + the launcher's tiny-input path)
int unc_from_p_exceeded_host(float p, const double* thr, int n_thr)
{
    unsigned b;
    std::memcpy(&b, &p, 4);
    int m = 0;
    for (int t = 0; t < n_thr; ++t) {
        const int r = ue_p_row_of(thr[t]);
        if (r < 0) return -1;
        const UePRow& R = UE_P_TABLE[r];
        bool in = b >= R.lo_first + R.lo_width && b < R.hi_first;
        if (b >= R.lo_first && b < R.lo_first + R.lo_width) in = (R.lo_mask[(b - R.lo_first) >> 6] >> ((b - R.lo_first) & 63u)) & 1ull;
        if (b >= R.hi_first && b < R.hi_first + R.hi_width) in = (R.hi_mask[(b - R.hi_first) >> 6] >> ((b - R.hi_first) & 63u)) & 1ull;
        m += in ? 1 : 0;
    }
    return m;
}

hipError_t launch_unc_counts_from_p(const float* p_fg, const uint8_t* prediction, const uint8_t* target, const uint8_t* mask, size_t n,
                                    int n_volumes, const double* thr_host, int n_thr, unsigned long long* out_dev, void* workspace,
                                    hipStream_t stream)
{
    if (n_volumes < 1) return hipErrorInvalidValue;
    // the table travels with every call (4.4 KB, stream-ordered; pageable source: the runtime stages it before the call returns)
    struct Host { UePCell cells[UE_P_CELLS]; unsigned long long masks[(2 * MAX_THR + 1) * UE_P_MASK_WORDS]; } host;
    static_assert(sizeof(Host) == UE_P_TABLE_BYTES, "workspace layout");
    if (!ue_p_build(thr_host, n_thr, host.cells, host.masks)) return hipErrorInvalidValue;
    const unsigned nb = blocks_per_volume(n);
    if (nb == 0) return hipMemsetAsync(out_dev, 0, sizeof(unsigned long long) * 8 * n_thr * n_volumes, stream);
    unsigned long long* part = reinterpret_cast<unsigned long long*>(workspace);
    char* table_dev = reinterpret_cast<char*>(workspace) + unc_workspace_bytes(n, n_volumes);
    hipError_t e = hipMemsetAsync(part, 0, unc_workspace_bytes(n, n_volumes), stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(table_dev, &host, sizeof host, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    UncThresholds th{};
    th.n_thr = n_thr;
    const size_t lds = (size_t)CB_WAVES * (n_thr + 1) * 2 * 64 * sizeof(unsigned);
    const unsigned bpw = blocks_per_workgroup(nb, n_volumes, UNC_MAX_BLOCKS, g_forced_blocks[1]);
    const unsigned gx = (nb + bpw - 1) / bpw;
    hipLaunchKernelGGL((unc_counts_sorted_kernel<float, true>), dim3(gx, n_volumes), dim3(CB_THREADS), lds, stream, p_fg, prediction, target,
                       mask, n, th, part, bpw, nb, reinterpret_cast<const UePCell*>(table_dev),
                       reinterpret_cast<const unsigned long long*>(table_dev + sizeof(UePCell) * UE_P_CELLS));
    hipLaunchKernelGGL(unc_reduce_kernel, dim3((unsigned)((n_volumes * n_thr * 4 + RED_THREADS - 1) / RED_THREADS)), dim3(RED_THREADS), 0, stream, part,
                       n_volumes, n_thr, out_dev);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- normalised entropy of [1-p, p]
// float32 products, float64 sum, divided by log 2 (numpyfunctions.py:166-168 via analysis.py:201).
__global__ __launch_bounds__(CB_THREADS) void norm_entropy_kernel(const float* __restrict__ p, size_t n,
                                                                   double* __restrict__ out64, float* __restrict__ out32)
{
    const size_t i = (size_t)blockIdx.x * CB_THREADS + threadIdx.x;
    if (i >= n) return;
    const float f = p[i];
    const float b = 1.0f - f;
    const double tf = (f > 0.f) ? (double)(f * logf(f)) : 0.0;
    const double tb = (b > 0.f) ? (double)(b * logf(b)) : 0.0;
    const double h = -(tb + tf) / 0.6931471805599453;
    if (out64) out64[i] = h;
    if (out32) out32[i] = (float)h;
}

hipError_t launch_norm_entropy(const float* p_fg, size_t n, double* out_f64, float* out_f32, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(norm_entropy_kernel, dim3((unsigned)((n + CB_THREADS - 1) / CB_THREADS)), dim3(CB_THREADS), 0,
                       stream, p_fg, n, out_f64, out_f32);
    return hipGetLastError();
}

}  // namespace rcu
