// HBM-bound per-voxel kernels: input layout change, the fused 1x1 head + softmax + running MC
// statistics, the statistics finalisation (mean / predictive entropy / mutual information /
// variance) and the small aleatoric / argmax helpers.  One thread per voxel, voxel index fastest
// across lanes, so every plane access is a fully coalesced 256-B wave transaction.
//
// Reference semantics:
//   softmax over the class dim per pass                   rechun/dl/customsteps.py:24,33
//   mean over T, entropy of the mean (nat log, 0 log 0=0) customsteps.py:57-61; torchhelper.py:53-54
//   mutual info = H(mean) - mean_t H(p_t)                 customsteps.py:63-66
//   variance = mean_c unbiased var_t(p_c)                 customsteps.py:68-71
//   aleatoric sigma = |raw| or exp(raw)                   bin-dl/brats_test_aleatoric.py:63-73
// The T probability volumes are never materialised: each pass adds its softmax output into the
// statistics planes (sum p, optionally sum p^2 in double, optionally sum H).
#include "rcu_head_common.h"

namespace rcu {

static constexpr int PW_THREADS = 256;

static inline unsigned grid_for(size_t n) { return (unsigned)((n + PW_THREADS - 1) / PW_THREADS); }

// ------------------------------------------------------------------------------- NCHW -> padded NHWC
// `replicas` > 1: the N input images are written `replicas` times one after the other (pass groups).
__global__ __launch_bounds__(PW_THREADS) void pack_input_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                 int C, int CP, size_t HW, size_t V, size_t N)
{
    const size_t v = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;
    if (v >= V) return;
    const size_t n = (v / HW) % N, hw = v % HW;
    for (int c = 0; c < CP; c += 4) {
        float4 q;
        q.x = (c + 0 < C) ? x[(n * C + c + 0) * HW + hw] : 0.f;
        q.y = (c + 1 < C) ? x[(n * C + c + 1) * HW + hw] : 0.f;
        q.z = (c + 2 < C) ? x[(n * C + c + 2) * HW + hw] : 0.f;
        q.w = (c + 3 < C) ? x[(n * C + c + 3) * HW + hw] : 0.f;
        *reinterpret_cast<float4*>(out + v * CP + c) = q;
    }
}

hipError_t launch_pack_input(const float* x, float* out, int N, int C, int CP, int H, int W, int replicas, hipStream_t stream)
{
    const size_t HW = (size_t)H * W, V = HW * N * (size_t)replicas;
    hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(V)), dim3(PW_THREADS), 0, stream, x, out, C, CP, HW, V, (size_t)N);
    return hipGetLastError();
}

// The statistics entries of one voxel held in registers across the passes of a group: load, add pass after pass in
// the same order and with the same operations as accumulate_voxel, store -- bit-identical to one launch per pass.
template <int C>
struct VoxelStats {
    double d[2 * C + 1];
    float f[C + 1];
    __device__ __forceinline__ void load(const void* stats, size_t v, size_t V, int flags)
    {
        if (flags & MC_VAR) {
            const double* sd = reinterpret_cast<const double*>(stats);
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) d[k] = sd[(size_t)k * V + v];
            d[2 * C] = (flags & MC_MI) ? sd[(size_t)(2 * C) * V + v] : 0.0;
        } else {
            const float* sf = reinterpret_cast<const float*>(stats);
#pragma unroll
            for (int k = 0; k < C; ++k) f[k] = sf[(size_t)k * V + v];
            f[C] = (flags & MC_MI) ? sf[(size_t)C * V + v] : 0.f;
        }
    }
    __device__ __forceinline__ void add(int flags, const float (&p)[C])
    {
        if (flags & MC_VAR) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double pc = (double)p[c];
                d[c] += pc;
                d[C + c] += pc * pc;
            }
            if (flags & MC_MI) d[2 * C] += (double)entropy_of<C>(p);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) f[c] += p[c];
            if (flags & MC_MI) f[C] += entropy_of<C>(p);
        }
    }
    __device__ __forceinline__ void store(void* stats, size_t v, size_t V, int flags) const
    {
        if (flags & MC_VAR) {
            double* sd = reinterpret_cast<double*>(stats);
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) sd[(size_t)k * V + v] = d[k];
            if (flags & MC_MI) sd[(size_t)(2 * C) * V + v] = d[2 * C];
        } else {
            float* sf = reinterpret_cast<float*>(stats);
#pragma unroll
            for (int k = 0; k < C; ++k) sf[(size_t)k * V + v] = f[k];
            if (flags & MC_MI) sf[(size_t)C * V + v] = f[C];
        }
    }
};

// ------------------------------------------------------------------------------- fused head
// act[v][0..CPh) -> logits (1x1 conv, unet.py:161); optional twin on act[v][CPh..2CPh) -> sigma
// (unet.py:164); optional softmax + statistics update so that logits never reach HBM.
//
// A wave handles 64 consecutive voxels in 8 rounds of 8: in a round, 8 lanes share one voxel and each
// loads one float4 of its channel vector, so every load instruction covers 1 KiB of contiguous memory
// (a lane-per-voxel layout would touch 64 different 128-B lines per instruction and thrashes L1/L2:
// 4x over-fetch measured).  The 8 partial dot products are combined with three xor-shuffles, and after
// the 8 rounds one more shuffle per class hands voxel i to lane i, which makes the logits stores and
// the statistics read-modify-write fully coalesced.
template <int C>
__device__ __forceinline__ void head_dot8(const float* __restrict__ row, const float* __restrict__ w, int cph, int sub,
                                          float (&acc)[C])
{
    for (int seg = 0; seg < cph; seg += 32) {
        const float4 x = *reinterpret_cast<const float4*>(row + seg + sub * 4);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 wv = *reinterpret_cast<const float4*>(w + c * cph + seg + sub * 4);
            acc[c] = fmaf(wv.x, x.x, acc[c]);
            acc[c] = fmaf(wv.y, x.y, acc[c]);
            acc[c] = fmaf(wv.z, x.z, acc[c]);
            acc[c] = fmaf(wv.w, x.w, acc[c]);
        }
    }
}

// logits (and raw sigma) of voxel v0 + lane out of the activations starting at `act` (bias not yet added)
// SIG: the sigma twin is wanted (raw sigma out, or the running sigma sum of the aleatoric + MC extension).  A template parameter: as
// a run-time test inside the eight unrolled rounds it cost the plain path registers and a fifth of its speed (131 -> 162 us per launch
// between rounds 1 and 2).
template <int C, bool SIG>
__device__ __forceinline__ void head_logits(const HeadArgs& a, const float* __restrict__ act, size_t v0, int lane,
                                            float (&l)[C], float (&s)[C])
{
    const int sub = lane & 7, grp = lane >> 3;
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] = s[c] = 0.f;
#pragma unroll
    for (int round = 0; round < 8; ++round) {
        const size_t v = v0 + round * 8 + grp;
        float pl[C], ps[C];
#pragma unroll
        for (int c = 0; c < C; ++c) pl[c] = ps[c] = 0.f;
        if (v < a.V) {
            const float* row = act + v * a.CP;
            head_dot8<C>(row, a.w_cls, a.CPh, sub, pl);
            if constexpr (SIG) head_dot8<C>(row + a.CPh, a.w_sig, a.CPh, sub, ps);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                pl[c] += __shfl_xor(pl[c], off, 64);
                if constexpr (SIG) ps[c] += __shfl_xor(ps[c], off, 64);
            }
            // voxel (round*8 + j) lives in lanes 8j..8j+7; lane i wants voxel i = 8*(i>>3) + (i&7)
            const float tl = __shfl(pl[c], (lane & 7) * 8, 64);
            l[c] = (grp == round) ? tl : l[c];
            if constexpr (SIG) {
                const float ts = __shfl(ps[c], (lane & 7) * 8, 64);
                s[c] = (grp == round) ? ts : s[c];
            }
        }
    }
}

template <int C, bool SIG>
__global__ __launch_bounds__(PW_THREADS) void head_kernel(const HeadArgs a)
{
    const int lane = threadIdx.x & 63;
    const size_t wave_id = ((size_t)blockIdx.x * PW_THREADS + threadIdx.x) >> 6;
    const size_t v0 = wave_id * 64;
    if (v0 >= a.V) return;
    const size_t v = v0 + lane;
    float l[C], s[C];
    if (a.passes > 1) {   // pass group: statistics only; one read-modify-write for all passes of the group
        VoxelStats<C> st;
        const size_t n = v / a.HW, hw = v % a.HW;
        float ssum[C];   // sigma-head extension: the voxel's running sigma sum, added to in pass order like the statistics
        if (v < a.V) {
            st.load(a.stats, v, a.V, a.stats_flags);
            if (SIG && a.sigma_sum != nullptr) {
#pragma unroll
                for (int c = 0; c < C; ++c) ssum[c] = a.sigma_sum[(n * C + c) * a.HW + hw];
            }
        }
        for (int t = 0; t < a.passes; ++t) {
            head_logits<C, SIG>(a, a.act + (size_t)t * a.V * a.CP, v0, lane, l, s);
#pragma unroll
            for (int c = 0; c < C; ++c) l[c] += a.b_cls[c];
            softmax_inplace<C>(l);
            st.add(a.stats_flags, l);
            if (SIG && a.sigma_sum != nullptr) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float raw = s[c] + a.b_sig[c];
                    ssum[c] += a.sigma_log ? expf(raw) : fabsf(raw);
                }
            }
        }
        if (v < a.V) {
            st.store(a.stats, v, a.V, a.stats_flags);
            if (SIG && a.sigma_sum != nullptr) {
#pragma unroll
                for (int c = 0; c < C; ++c) a.sigma_sum[(n * C + c) * a.HW + hw] = ssum[c];
            }
        }
        return;
    }
    head_logits<C, SIG>(a, a.act, v0, lane, l, s);
    if (v >= a.V) return;
    const size_t n = v / a.HW, hw = v % a.HW;
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] += a.b_cls[c];
    if (a.logits != nullptr) {
#pragma unroll
        for (int c = 0; c < C; ++c) a.logits[(n * C + c) * a.HW + hw] = l[c];
    }
    if (SIG && a.sigma != nullptr) {
#pragma unroll
        for (int c = 0; c < C; ++c) a.sigma[(n * C + c) * a.HW + hw] = s[c] + a.b_sig[c];
    }
    if (SIG && a.sigma_sum != nullptr) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float raw = s[c] + a.b_sig[c];
            a.sigma_sum[(n * C + c) * a.HW + hw] += a.sigma_log ? expf(raw) : fabsf(raw);
        }
    }
    if (a.stats != nullptr) {
        softmax_inplace<C>(l);
        accumulate_voxel<C>(a.stats, v, a.V, a.stats_flags, l);
    }
}

#define RCU_DISPATCH_C(Cval, ...)                                    \
    switch (Cval) {                                                  \
        case 1: { constexpr int C_ = 1; __VA_ARGS__; break; }        \
        case 2: { constexpr int C_ = 2; __VA_ARGS__; break; }        \
        case 3: { constexpr int C_ = 3; __VA_ARGS__; break; }        \
        case 4: { constexpr int C_ = 4; __VA_ARGS__; break; }        \
        case 5: { constexpr int C_ = 5; __VA_ARGS__; break; }        \
        case 6: { constexpr int C_ = 6; __VA_ARGS__; break; }        \
        case 7: { constexpr int C_ = 7; __VA_ARGS__; break; }        \
        case 8: { constexpr int C_ = 8; __VA_ARGS__; break; }        \
        default: return hipErrorInvalidValue;                        \
    }

hipError_t launch_head(const HeadArgs& a, hipStream_t stream)
{
    if (a.sigma != nullptr || a.sigma_sum != nullptr) {
        RCU_DISPATCH_C(a.C, hipLaunchKernelGGL((head_kernel<C_, true>), dim3(grid_for(a.V)), dim3(PW_THREADS), 0, stream, a));
    } else {
        RCU_DISPATCH_C(a.C, hipLaunchKernelGGL((head_kernel<C_, false>), dim3(grid_for(a.V)), dim3(PW_THREADS), 0, stream, a));
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- standalone accumulate
template <int C>
__global__ __launch_bounds__(PW_THREADS) void mc_accumulate_kernel(const float* __restrict__ in, void* stats, size_t HW,
                                                                    size_t V, int flags)
{
    const size_t v = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;
    if (v >= V) return;
    const size_t n = v / HW, hw = v % HW;
    float l[C];
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] = in[(n * C + c) * HW + hw];
    if (!(flags & MC_INPUT_PROBS)) softmax_inplace<C>(l);
    accumulate_voxel<C>(stats, v, V, flags, l);
}

hipError_t launch_mc_accumulate(const float* in, void* stats, int C, size_t N, size_t HW, int flags, hipStream_t stream)
{
    const size_t V = N * HW;
    RCU_DISPATCH_C(C, hipLaunchKernelGGL(mc_accumulate_kernel<C_>, dim3(grid_for(V)), dim3(PW_THREADS), 0, stream, in,
                                         stats, HW, V, flags));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- finalize
template <int C>
__global__ __launch_bounds__(PW_THREADS) void mc_finalize_kernel(const void* stats, size_t HW, size_t V, int T, int flags,
                                                                  float* __restrict__ mean, float* __restrict__ entropy,
                                                                  float* __restrict__ mi, float* __restrict__ var)
{
    const size_t v = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;
    if (v >= V) return;
    const size_t n = v / HW, hw = v % HW;
    float p[C];
    float sum_h = 0.f;
    if (flags & MC_VAR) {
        const double* sd = reinterpret_cast<const double*>(stats);
        double vsum = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double s = sd[(size_t)c * V + v], q = sd[(size_t)(C + c) * V + v];
            p[c] = (float)(s / (double)T);
            vsum += (q - s * s / (double)T) / (double)(T - 1);   // unbiased, as torch.var (customsteps.py:70)
        }
        if (var != nullptr) var[v] = (float)(vsum / (double)C);
        if (flags & MC_MI) sum_h = (float)sd[(size_t)(2 * C) * V + v];
    } else {
        const float* sf = reinterpret_cast<const float*>(stats);
#pragma unroll
        for (int c = 0; c < C; ++c) p[c] = sf[(size_t)c * V + v] / (float)T;
        if (flags & MC_MI) sum_h = sf[(size_t)C * V + v];
    }
    if (mean != nullptr) {
#pragma unroll
        for (int c = 0; c < C; ++c) mean[(n * C + c) * HW + hw] = p[c];
    }
    const float h = entropy_of<C>(p);
    if (entropy != nullptr) entropy[v] = h;
    if (mi != nullptr && (flags & MC_MI)) mi[v] = h - sum_h / (float)T;
}

hipError_t launch_mc_finalize(const void* stats, int C, size_t N, size_t HW, int T, int flags, float* mean,
                              float* entropy, float* mi, float* var, hipStream_t stream)
{
    const size_t V = N * HW;
    RCU_DISPATCH_C(C, hipLaunchKernelGGL(mc_finalize_kernel<C_>, dim3(grid_for(V)), dim3(PW_THREADS), 0, stream, stats,
                                         HW, V, T, flags, mean, entropy, mi, var));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- softmax / aleatoric / argmax
template <int C>
__global__ __launch_bounds__(PW_THREADS) void softmax_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              size_t HW, size_t V)
{
    const size_t v = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;
    if (v >= V) return;
    const size_t n = v / HW, hw = v % HW;
    float l[C];
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] = in[(n * C + c) * HW + hw];
    softmax_inplace<C>(l);
#pragma unroll
    for (int c = 0; c < C; ++c) out[(n * C + c) * HW + hw] = l[c];
}

hipError_t launch_softmax_nchw(const float* logits, float* probs, int C, size_t N, size_t HW, hipStream_t stream)
{
    const size_t V = N * HW;
    RCU_DISPATCH_C(C, hipLaunchKernelGGL(softmax_kernel<C_>, dim3(grid_for(V)), dim3(PW_THREADS), 0, stream, logits,
                                         probs, HW, V));
    return hipGetLastError();
}

template <int C>
__global__ __launch_bounds__(PW_THREADS) void aleatoric_kernel(const float* __restrict__ logits,
                                                                const float* __restrict__ sigma_raw, size_t HW, size_t V,
                                                                int is_log_sigma, float* __restrict__ probs,
                                                                float* __restrict__ sigma_out,
                                                                uint8_t* __restrict__ prediction,
                                                                float* __restrict__ sigma_pred)
{
    const size_t v = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;
    if (v >= V) return;
    const size_t n = v / HW, hw = v % HW;
    float l[C], s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        l[c] = logits[(n * C + c) * HW + hw];
        const float r = sigma_raw[(n * C + c) * HW + hw];
        s[c] = is_log_sigma ? expf(r) : fabsf(r);
    }
    softmax_inplace<C>(l);
    int best = 0;
#pragma unroll
    for (int c = 1; c < C; ++c) best = (l[c] > l[best]) ? c : best;   // first maximum, as np.argmax
    float sp = s[0];
#pragma unroll
    for (int c = 1; c < C; ++c) sp = (best == c) ? s[c] : sp;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (probs != nullptr) probs[(n * C + c) * HW + hw] = l[c];
        if (sigma_out != nullptr) sigma_out[(n * C + c) * HW + hw] = s[c];
    }
    if (prediction != nullptr) prediction[v] = (uint8_t)best;
    if (sigma_pred != nullptr) sigma_pred[v] = sp;
}

hipError_t launch_aleatoric(const float* logits, const float* sigma_raw, int C, size_t N, size_t HW, int is_log_sigma,
                            float* probs, float* sigma_out, uint8_t* prediction, float* sigma_pred, hipStream_t stream)
{
    const size_t V = N * HW;
    RCU_DISPATCH_C(C, hipLaunchKernelGGL(aleatoric_kernel<C_>, dim3(grid_for(V)), dim3(PW_THREADS), 0, stream, logits,
                                         sigma_raw, HW, V, is_log_sigma, probs, sigma_out, prediction, sigma_pred));
    return hipGetLastError();
}

template <int C>
__global__ __launch_bounds__(PW_THREADS) void argmax_fg_kernel(const float* __restrict__ probs, size_t HW, size_t V,
                                                                uint8_t* __restrict__ prediction,
                                                                float* __restrict__ p_fg)
{
    const size_t v = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;
    if (v >= V) return;
    const size_t n = v / HW, hw = v % HW;
    float p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = probs[(n * C + c) * HW + hw];
    int best = 0;
#pragma unroll
    for (int c = 1; c < C; ++c) best = (p[c] > p[best]) ? c : best;
    if (prediction != nullptr) prediction[v] = (uint8_t)best;
    if (p_fg != nullptr) p_fg[v] = p[C > 1 ? 1 : 0];   // foreground class (brats_test_default.py:99)
}

hipError_t launch_argmax_fg(const float* probs, int C, size_t N, size_t HW, uint8_t* prediction, float* p_fg,
                            hipStream_t stream)
{
    const size_t V = N * HW;
    RCU_DISPATCH_C(C, hipLaunchKernelGGL(argmax_fg_kernel<C_>, dim3(grid_for(V)), dim3(PW_THREADS), 0, stream, probs, HW,
                                         V, prediction, p_fg));
    return hipGetLastError();
}

}  // namespace rcu
