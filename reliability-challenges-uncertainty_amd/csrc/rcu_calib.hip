// Calibration kernels: the ECE reliability histogram and the uncertainty-error counts.
//
// Reference semantics:
//   bin = np.digitize(p, linspace(0, 1+1e-8, n_bins+1)) - 1, then three bincounts
//                                                  common/evalutation/numpyfunctions.py:51-63
//   tp/tn/fp/fn and their "uncertain" subsets for a threshold on the uncertainty map
//                                                  common/evalutation/numpyfunctions.py:86-107
//   normalised entropy of [1-p, p]                 rechun/eval/analysis.py:196-203; numpyfunctions.py:166-168
//
// Both histograms are HBM scans over 6-10 bytes per voxel.  Neighbouring voxels almost always fall
// into the same bin, so the histogram update is a wavefront reduction: each wave repeatedly picks
// the key of its first unprocessed lane, ballots the lanes sharing it, and adds one popcount (and
// one 64-lane sum of the confidences) to a per-wave LDS slot -- no atomics, a handful of
// iterations per 64 voxels.  Per-workgroup partials are combined by a second kernel in a fixed
// order, so counts are exact and the confidence sums are run-to-run deterministic.
// Bin indices are bit-exact with np.digitize: p is compared against the float32 thresholds
// t_k = min{float32 t : t >= edge_k}.
#include "rcu_kernels.h"

namespace rcu {

static constexpr int CB_THREADS = 256;
static constexpr int CB_WAVES = CB_THREADS / 64;
static constexpr int ELEMS_PER_BLOCK = CB_THREADS * 16;   // 4 rounds of 4 consecutive voxels per thread

struct BinThresholds {
    float t[MAX_BINS - 1];
    int n_bins;
};
struct UncThresholds {
    double t[MAX_THR];
    int n_thr;
};

struct EcePartial {
    unsigned long long count, sum_pos;
    double sum_conf;
};

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

size_t ece_workspace_bytes(size_t n_per_volume, int n_volumes)
{
    return (size_t)blocks_per_volume(n_per_volume) * n_volumes * MAX_BINS * sizeof(EcePartial);
}

// One voxel per lane: fold the wave's voxels into the per-wave LDS histogram.
__device__ __forceinline__ void ece_wave_update(bool active, int bin, bool pos, float p, unsigned* w_cnt, unsigned* w_pos,
                                                double* w_sum, int lane)
{
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __shfl(bin, leader, 64);
        const bool mine = active && (bin == b);
        const unsigned long long grp = __ballot(mine);
        const unsigned c = (unsigned)__popcll(grp);
        const unsigned cp = (unsigned)__popcll(__ballot(mine && pos));
        const double s = wave_sum(mine ? (double)p : 0.0);
        if (lane == 0) {
            w_cnt[b] += c;
            w_pos[b] += cp;
            w_sum[b] += s;
        }
        todo &= ~grp;
    }
}

template <bool VEC>
__global__ __launch_bounds__(CB_THREADS) void ece_hist_kernel(const float* __restrict__ p, const uint8_t* __restrict__ target,
                                                               const uint8_t* __restrict__ mask, size_t n,
                                                               const BinThresholds th, EcePartial* __restrict__ partial)
{
    __shared__ unsigned s_cnt[CB_WAVES][MAX_BINS];
    __shared__ unsigned s_pos[CB_WAVES][MAX_BINS];
    __shared__ double s_sum[CB_WAVES][MAX_BINS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < CB_WAVES * MAX_BINS; i += CB_THREADS) {
        (&s_cnt[0][0])[i] = 0;
        (&s_pos[0][0])[i] = 0;
        (&s_sum[0][0])[i] = 0.0;
    }
    __syncthreads();
    const size_t vol = blockIdx.y;
    const float* pv = p + vol * n;
    const uint8_t* tv = target + vol * n;
    const uint8_t* mv = mask ? mask + vol * n : nullptr;
    const size_t base = (size_t)blockIdx.x * ELEMS_PER_BLOCK;
    if (VEC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t e = base + ((size_t)r * CB_THREADS + tid) * 4;
            const bool in = e < n;   // n % 4 == 0 on this path
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            uchar4 t4 = make_uchar4(0, 0, 0, 0), m4 = make_uchar4(1, 1, 1, 1);
            if (in) {
                q = *reinterpret_cast<const float4*>(pv + e);
                t4 = *reinterpret_cast<const uchar4*>(tv + e);
                if (mv) m4 = *reinterpret_cast<const uchar4*>(mv + e);
            }
            ece_wave_update(in && m4.x, bin_of(q.x, th), t4.x != 0, q.x, s_cnt[wave], s_pos[wave], s_sum[wave], lane);
            ece_wave_update(in && m4.y, bin_of(q.y, th), t4.y != 0, q.y, s_cnt[wave], s_pos[wave], s_sum[wave], lane);
            ece_wave_update(in && m4.z, bin_of(q.z, th), t4.z != 0, q.z, s_cnt[wave], s_pos[wave], s_sum[wave], lane);
            ece_wave_update(in && m4.w, bin_of(q.w, th), t4.w != 0, q.w, s_cnt[wave], s_pos[wave], s_sum[wave], lane);
        }
    } else {
        for (int r = 0; r < 16; ++r) {
            const size_t e = base + (size_t)r * CB_THREADS + tid;
            const bool in = e < n;
            const float q = in ? pv[e] : 0.f;
            const bool act = in && (mv ? mv[e] != 0 : true);
            const bool pos = in && tv[e] != 0;
            ece_wave_update(act, bin_of(q, th), pos, q, s_cnt[wave], s_pos[wave], s_sum[wave], lane);
        }
    }
    __syncthreads();
    if (tid < MAX_BINS) {
        EcePartial out;
        out.count = 0;
        out.sum_pos = 0;
        out.sum_conf = 0.0;
        for (int w = 0; w < CB_WAVES; ++w) {   // fixed order
            out.count += s_cnt[w][tid];
            out.sum_pos += s_pos[w][tid];
            out.sum_conf += s_sum[w][tid];
        }
        partial[((size_t)vol * gridDim.x + blockIdx.x) * MAX_BINS + tid] = out;
    }
}

__global__ __launch_bounds__(64) void ece_reduce_kernel(const EcePartial* __restrict__ partial, unsigned nblocks,
                                                         EceResult* __restrict__ result)
{
    const int b = threadIdx.x;
    if (b >= MAX_BINS) return;
    const size_t vol = blockIdx.x;
    unsigned long long c = 0, sp = 0;
    double sc = 0.0;
    for (unsigned k = 0; k < nblocks; ++k) {   // fixed order -> deterministic
        const EcePartial q = partial[((size_t)vol * nblocks + k) * MAX_BINS + b];
        c += q.count;
        sp += q.sum_pos;
        sc += q.sum_conf;
    }
    result[vol].count[b] = c;
    result[vol].sum_pos[b] = sp;
    result[vol].sum_conf[b] = sc;
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
    EcePartial* part = reinterpret_cast<EcePartial*>(workspace);
    const bool vec = (n % 4 == 0) && (reinterpret_cast<uintptr_t>(p) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(target) % 4 == 0) &&
                     (mask == nullptr || reinterpret_cast<uintptr_t>(mask) % 4 == 0);
    if (vec)
        hipLaunchKernelGGL(ece_hist_kernel<true>, dim3(nb, n_volumes), dim3(CB_THREADS), 0, stream, p, target, mask, n, th,
                           part);
    else
        hipLaunchKernelGGL(ece_hist_kernel<false>, dim3(nb, n_volumes), dim3(CB_THREADS), 0, stream, p, target, mask, n,
                           th, part);
    hipLaunchKernelGGL(ece_reduce_kernel, dim3(n_volumes), dim3(64), 0, stream, part, nb, result_dev);
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

size_t unc_workspace_bytes(size_t n_per_volume, int n_volumes)
{
    return (size_t)blocks_per_volume(n_per_volume) * n_volumes * UNC_SLOTS * sizeof(unsigned long long);
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
    for (int r = 0; r < 16; ++r) {
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
        partial[((size_t)vol * gridDim.x + blockIdx.x) * UNC_SLOTS + tid] = c;
    }
}

__global__ __launch_bounds__(128) void unc_reduce_kernel(const unsigned long long* __restrict__ partial, unsigned nblocks,
                                                          int n_thr, unsigned long long* __restrict__ out)
{
    const int slot = threadIdx.x;   // [t][cell]
    const size_t vol = blockIdx.x;
    __shared__ unsigned long long s[UNC_SLOTS];
    if (slot < UNC_SLOTS) {
        unsigned long long c = 0;
        for (unsigned k = 0; k < nblocks; ++k) c += partial[((size_t)vol * nblocks + k) * UNC_SLOTS + slot];
        s[slot] = c;
    }
    __syncthreads();
    const int t = slot / 4, cell = slot % 4;
    if (slot < UNC_SLOTS && t < n_thr) {
        unsigned long long* o = out + ((size_t)vol * n_thr + t) * 8;
        o[cell] = s[n_thr * 4 + cell];   // tp, tn, fp, fn (same for every threshold)
        o[4 + cell] = s[slot];           // tpu, tnu, fpu, fnu
    }
}

hipError_t launch_unc_counts(const void* unc, int unc_is_f64, const uint8_t* prediction, const uint8_t* target,
                             const uint8_t* mask, size_t n, int n_volumes, const double* thr_host, int n_thr,
                             unsigned long long* out_dev, void* workspace, hipStream_t stream)
{
    if (n_thr < 1 || n_thr > MAX_THR || n_volumes < 1) return hipErrorInvalidValue;
    UncThresholds th;
    th.n_thr = n_thr;
    for (int t = 0; t < MAX_THR; ++t) th.t[t] = (t < n_thr) ? thr_host[t] : 0.0;
    const unsigned nb = blocks_per_volume(n);
    if (nb == 0) return hipMemsetAsync(out_dev, 0, sizeof(unsigned long long) * 8 * n_thr * n_volumes, stream);
    unsigned long long* part = reinterpret_cast<unsigned long long*>(workspace);
    if (unc_is_f64)
        hipLaunchKernelGGL(unc_counts_kernel<double>, dim3(nb, n_volumes), dim3(CB_THREADS), 0, stream,
                           reinterpret_cast<const double*>(unc), prediction, target, mask, n, th, part);
    else
        hipLaunchKernelGGL(unc_counts_kernel<float>, dim3(nb, n_volumes), dim3(CB_THREADS), 0, stream,
                           reinterpret_cast<const float*>(unc), prediction, target, mask, n, th, part);
    hipLaunchKernelGGL(unc_reduce_kernel, dim3(n_volumes), dim3(128), 0, stream, part, nb, n_thr, out_dev);
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
