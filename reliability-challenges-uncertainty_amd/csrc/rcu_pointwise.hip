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

#include <initializer_list>

namespace rcu {

static constexpr int PW_THREADS = 256;

static inline unsigned grid_for(size_t n) { return (unsigned)((n + PW_THREADS - 1) / PW_THREADS); }

// ------------------------------------------------------------------------------- NCHW -> padded NHWC
// `replicas` > 1: the N input images are written `replicas` times one after the other (pass groups).
// The output image is allocated PH x PW >= H x W (a padded level 0, rcu_api.hip): only the real pixels are written.
__global__ __launch_bounds__(PW_THREADS) void pack_input_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                 int C, int CP, size_t HW, size_t V, size_t N, int W, int PH, int PW)
{
    const size_t v = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;
    if (v >= V) return;
    const size_t s = v / HW, n = s % N, hw = v % HW;
    const size_t vo = (s * PH + hw / W) * PW + hw % W;
    for (int c = 0; c < CP; c += 4) {
        float4 q;
        q.x = (c + 0 < C) ? x[(n * C + c + 0) * HW + hw] : 0.f;
        q.y = (c + 1 < C) ? x[(n * C + c + 1) * HW + hw] : 0.f;
        q.z = (c + 2 < C) ? x[(n * C + c + 2) * HW + hw] : 0.f;
        q.w = (c + 3 < C) ? x[(n * C + c + 3) * HW + hw] : 0.f;
        *reinterpret_cast<float4*>(out + vo * CP + c) = q;
    }
}

hipError_t launch_pack_input(const float* x, float* out, int N, int C, int CP, int H, int W, int PH, int PW, int replicas, hipStream_t stream)
{
    const size_t HW = (size_t)H * W, V = HW * N * (size_t)replicas;
    hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(V)), dim3(PW_THREADS), 0, stream, x, out, C, CP, HW, V, (size_t)N, W, PH, PW);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- padded NHWC -> compact NHWC
__global__ __launch_bounds__(PW_THREADS) void crop_nhwc_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t quads, int W, int HW, int PH,
                                                                int PW, int CQ)
{
    const size_t i = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;      // float4 of the compact tensor
    if (i >= quads) return;
    const size_t v = i / CQ;
    const int q = (int)(i - v * CQ);
    const size_t n = v / HW;
    const int hw = (int)(v - n * HW), y = hw / W, x = hw - y * W;
    dst[i] = src[((n * PH + y) * PW + x) * CQ + q];
}

hipError_t launch_crop_nhwc(const float* src, float* dst, int N, int H, int W, int PH, int PW, int CP, hipStream_t stream)
{
    const size_t quads = (size_t)N * H * W * (CP / 4);
    hipLaunchKernelGGL(crop_nhwc_kernel, dim3(grid_for(quads)), dim3(PW_THREADS), 0, stream, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), quads, W, H * W, PH, PW, CP / 4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- fused head
// act[v][0..CPh) -> logits (1x1 conv, unet.py:161); optional twin on act[v][CPh..2CPh) -> sigma
// (unet.py:164); optional softmax + statistics update so that logits never reach HBM.
//
// A wave handles 64 consecutive voxels in 8 rounds of 8: in a round, 8 lanes share one voxel and each
// loads one float4 of its channel vector, so every load instruction covers 1 KiB of contiguous memory
// (a lane-per-voxel layout would touch 64 different 128-B lines per instruction and thrashes L1/L2:
// 4x over-fetch measured).  The 8 partial dot products are combined with three DPP adds (xor 1, xor 2 inside a quad,
// then the mirror image of the 8-lane group: the other quad's sum -- the same pairwise tree, and the same bits, as three
// xor-shuffles), every lane keeps the sum of the round that equals its position in its group, and after the 8 rounds ONE gather
// per class (the 8 x 8 transpose of the lane matrix) hands voxel i to lane i, which makes the logits stores and the statistics
// read-modify-write fully coalesced.  (Round 3 did all of it with ds_bpermute: 64 per wave and class pair instead of 2.)
template <int CTRL>
__device__ __forceinline__ float head_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float head_sum8(float x)
{
    x += head_dpp<0xB1>(x);    // quad_perm [1,0,3,2]: lane ^ 1
    x += head_dpp<0x4E>(x);    // quad_perm [2,3,0,1]: lane ^ 2
    x += head_dpp<0x141>(x);   // row_half_mirror: lane i <-> 7 - i of its group of eight = the other quad (every lane of a quad holds the quad's sum)
    return x;
}
template <int C>
__device__ __forceinline__ void head_dot8(const float* __restrict__ row, const float* __restrict__ w, int cph, int sub,
                                          float (&acc)[C])
{
    for (int seg = 0; seg < cph; seg += 32) {
        const float4 x = stream_load(reinterpret_cast<const float4*>(row + seg + sub * 4));
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
// CPH: the head unit's channel count when it is 32 (every shipped configuration: start_filters 32), else 0 = any multiple of 32 (run-time
// loop).  With it known, the lane's slice of the 1x1 weights is loaded ONCE and the activations of all eight rounds are requested
// before the first is used: eight 1-KB loads in flight per wave.  The round-3 form asked for a round's activations (and, again, its
// weights), waited, multiplied, and only then asked for the next round's -- eight memory round trips in a row per wave, and with
// ~24 waves per CU that is 24 KB in flight per CU: 4.7 TB/s at 1.3 us of loaded latency, which is what it measured.
// The activations of a wave's 64 voxels: eight rounds of one float4 per lane (and of the sigma twin's 32 channels).
template <bool SIG>
struct HeadTile {
    float4 xa[8], xs[8];
};
template <bool SIG>
__device__ __forceinline__ void head_load32(const HeadArgs& a, const float* __restrict__ act, size_t v0, int lane, HeadTile<SIG>& t)
{
    const int sub = lane & 7, grp = lane >> 3;
#pragma unroll
    for (int round = 0; round < 8; ++round) {
        const size_t v = v0 + round * 8 + grp;
        const float* row = act + (v < a.V ? v : a.V - 1) * a.CP;      // (a voxel behind the end reads the last one and is zeroed in the sums)
        t.xa[round] = stream_load(reinterpret_cast<const float4*>(row + sub * 4));      // read once: non-temporal (rcu_kernels.h)
        if constexpr (SIG) t.xs[round] = stream_load(reinterpret_cast<const float4*>(row + 32 + sub * 4));
    }
}
template <int C, bool SIG>
struct HeadWeights {
    float4 wl[C], wg[C];
    __device__ __forceinline__ void load(const HeadArgs& a, int lane)
    {
        const int sub = lane & 7;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            wl[c] = *reinterpret_cast<const float4*>(a.w_cls + c * 32 + sub * 4);
            if constexpr (SIG) wg[c] = *reinterpret_cast<const float4*>(a.w_sig + c * 32 + sub * 4);
        }
    }
};

template <int C, bool SIG, int CPH>
__device__ __forceinline__ void head_reduce(const HeadArgs& a, const float* __restrict__ act, size_t v0, int lane, const HeadWeights<C, SIG>& w,
                                            const HeadTile<SIG>& t, float (&l)[C], float (&s)[C])
{
    const int sub = lane & 7, grp = lane >> 3;
    float kl[C], ks[C];   // lane 8a + b keeps the sums of round b: voxel 8b + a
#pragma unroll
    for (int c = 0; c < C; ++c) kl[c] = ks[c] = 0.f;
#pragma unroll
    for (int round = 0; round < 8; ++round) {
        const size_t v = v0 + round * 8 + grp;
        float pl[C], ps[C];
#pragma unroll
        for (int c = 0; c < C; ++c) pl[c] = ps[c] = 0.f;
        if constexpr (CPH == 32) {      // the fmaf chain of head_dot8, operands from registers
#pragma unroll
            for (int c = 0; c < C; ++c) {
                pl[c] = fmaf(w.wl[c].w, t.xa[round].w, fmaf(w.wl[c].z, t.xa[round].z, fmaf(w.wl[c].y, t.xa[round].y, fmaf(w.wl[c].x, t.xa[round].x, 0.f))));
                if constexpr (SIG)
                    ps[c] = fmaf(w.wg[c].w, t.xs[round].w, fmaf(w.wg[c].z, t.xs[round].z, fmaf(w.wg[c].y, t.xs[round].y, fmaf(w.wg[c].x, t.xs[round].x, 0.f))));
                pl[c] = v < a.V ? pl[c] : 0.f;
                ps[c] = v < a.V ? ps[c] : 0.f;
            }
        } else if (v < a.V) {
            const float* row = act + v * a.CP;
            head_dot8<C>(row, a.w_cls, a.CPh, sub, pl);
            if constexpr (SIG) head_dot8<C>(row + a.CPh, a.w_sig, a.CPh, sub, ps);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float tl = head_sum8(pl[c]);
            kl[c] = (sub == round) ? tl : kl[c];
            if constexpr (SIG) {
                const float ts = head_sum8(ps[c]);
                ks[c] = (sub == round) ? ts : ks[c];
            }
        }
    }
    // voxel i = 8 r + j of the wave was summed in round r by group j and kept by lane 8 j + r: the transpose of the 8 x 8 lane matrix
    const int src = sub * 8 + grp;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        l[c] = __shfl(kl[c], src, 64);
        s[c] = SIG ? __shfl(ks[c], src, 64) : 0.f;
    }
}

template <int C, bool SIG, int CPH>
__device__ __forceinline__ void head_logits(const HeadArgs& a, const float* __restrict__ act, size_t v0, int lane,
                                            float (&l)[C], float (&s)[C])
{
    HeadWeights<C, SIG> w;
    HeadTile<SIG> t;
    if constexpr (CPH == 32) {
        w.load(a, lane);
        head_load32<SIG>(a, act, v0, lane, t);
    }
    head_reduce<C, SIG, CPH>(a, act, v0, lane, w, t, l, s);
}

// One 64-voxel group of the single-pass form: logits / sigma outputs, softmax, statistics update (the group's statistics entries in `st`).
template <int C, bool SIG>
__device__ __forceinline__ void head_finish(const HeadArgs& a, size_t v, float (&l)[C], const float (&s)[C], VoxelStats<C>& st)
{
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
        st.add(a.stats_flags, l);
        st.store(a.stats, v, a.V, a.stats_flags);
    }
}

// The single-pass form as a STREAM (32 head channels): a wave walks over its groups of 64 voxels (group g, g + waves, ...) and requests
// the next group's activations -- behind the current group's statistics entries, so that waiting for those does not wait for them:
// vector memory returns in order -- before it reduces the current group: 8-16 KB in flight per wave the whole time, not in bursts.
template <int C, bool SIG>
__global__ __launch_bounds__(PW_THREADS) void head_stream_kernel(const HeadArgs a)
{
    const int lane = threadIdx.x & 63;
    const size_t n_waves = (size_t)gridDim.x * (PW_THREADS / 64);
    const size_t groups = (a.V + 63) / 64;
    size_t g = ((size_t)blockIdx.x * PW_THREADS + threadIdx.x) >> 6;
    if (g >= groups) return;
    HeadWeights<C, SIG> w;
    w.load(a, lane);
    HeadTile<SIG> ta, tb;
    head_load32<SIG>(a, a.act, g * 64, lane, ta);
    float l[C], s[C];
    for (;;) {
        {
            VoxelStats<C> st;
            const size_t v = g * 64 + lane, gn = g + n_waves;
            if (a.stats != nullptr && v < a.V) st.load(a.stats, v, a.V, a.stats_flags);
            if (gn < groups) head_load32<SIG>(a, a.act, gn * 64, lane, tb);
            head_reduce<C, SIG, 32>(a, a.act, g * 64, lane, w, ta, l, s);
            head_finish<C, SIG>(a, v, l, s, st);
            if (gn >= groups) return;
            g = gn;
        }
        {
            VoxelStats<C> st;
            const size_t v = g * 64 + lane, gn = g + n_waves;
            if (a.stats != nullptr && v < a.V) st.load(a.stats, v, a.V, a.stats_flags);
            if (gn < groups) head_load32<SIG>(a, a.act, gn * 64, lane, ta);
            head_reduce<C, SIG, 32>(a, a.act, g * 64, lane, w, tb, l, s);
            head_finish<C, SIG>(a, v, l, s, st);
            if (gn >= groups) return;
            g = gn;
        }
    }
}

template <int C, bool SIG, int CPH>
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
            head_logits<C, SIG, CPH>(a, a.act + (size_t)t * a.V * a.CP, v0, lane, l, s);
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
    // the voxel's statistics entries are requested before the activations: their round trip runs beside the eight activation loads
    // instead of behind the softmax (load all planes, add, store all planes: the operations -- and bits -- of accumulate_voxel)
    VoxelStats<C> st;
    if (a.stats != nullptr && v < a.V) st.load(a.stats, v, a.V, a.stats_flags);
    head_logits<C, SIG, CPH>(a, a.act, v0, lane, l, s);
    head_finish<C, SIG>(a, v, l, s, st);
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
    const bool c32 = a.CPh == 32 && a.V > 0;
    const bool sig = a.sigma != nullptr || a.sigma_sum != nullptr;
#ifndef RCU_HEAD_WGS
#define RCU_HEAD_WGS 8      // resident workgroups per CU of the streaming form (experiment builds override it)
#endif
    if (c32 && a.passes <= 1) {   // a stream: persistent waves walk over the voxel groups
        const unsigned need = grid_for(a.V), cap = 256u * RCU_HEAD_WGS;
        const unsigned grid = need < cap ? need : cap;
        if (sig) {
            RCU_DISPATCH_C(a.C, hipLaunchKernelGGL((head_stream_kernel<C_, true>), dim3(grid), dim3(PW_THREADS), 0, stream, a));
        } else {
            RCU_DISPATCH_C(a.C, hipLaunchKernelGGL((head_stream_kernel<C_, false>), dim3(grid), dim3(PW_THREADS), 0, stream, a));
        }
        return hipGetLastError();
    }
    if (sig && c32) {
        RCU_DISPATCH_C(a.C, hipLaunchKernelGGL((head_kernel<C_, true, 32>), dim3(grid_for(a.V)), dim3(PW_THREADS), 0, stream, a));
    } else if (sig) {
        RCU_DISPATCH_C(a.C, hipLaunchKernelGGL((head_kernel<C_, true, 0>), dim3(grid_for(a.V)), dim3(PW_THREADS), 0, stream, a));
    } else if (c32) {
        RCU_DISPATCH_C(a.C, hipLaunchKernelGGL((head_kernel<C_, false, 32>), dim3(grid_for(a.V)), dim3(PW_THREADS), 0, stream, a));
    } else {
        RCU_DISPATCH_C(a.C, hipLaunchKernelGGL((head_kernel<C_, false, 0>), dim3(grid_for(a.V)), dim3(PW_THREADS), 0, stream, a));
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
    VoxelStats<C> st;      // every plane's entry requested at once (accumulate_voxel's operations, its loads no longer one behind the other)
    st.load(stats, v, V, flags);
    if (!(flags & MC_INPUT_PROBS)) softmax_inplace<C>(l);
    st.add(flags, l);
    st.store(stats, v, V, flags);
}

// Four consecutive voxels per thread (round 4): 16-byte loads and stores -- a wave instruction covers 1 KB of a plane instead of
// 256 bytes -- and 32-bit index arithmetic (the one-voxel form pays a 64-bit division per thread, a few dozen instructions on a
// kernel whose budget at HBM speed is a few dozen).  Per voxel the operations -- and so the bits -- are those of the scalar kernels.
// Needs HW % 4 == 0 (a group of four never straddles two samples), 16-byte aligned arrays and V / 4 < 2^31; anything else takes the
// one-voxel kernels.
static inline bool vec4_ok(size_t HW, size_t V, std::initializer_list<const void*> ptrs)
{
    if (HW % 4 != 0 || V / 4 >= ((size_t)1 << 31)) return false;
    for (const void* p : ptrs)
        if (p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15u) != 0) return false;
    return true;
}

template <int C>
__global__ __launch_bounds__(PW_THREADS) void mc_accumulate4_kernel(const float4* __restrict__ in, void* stats, uint32_t HW4,
                                                                     uint32_t V4, int flags)
{
    const uint32_t i = blockIdx.x * PW_THREADS + threadIdx.x;
    if (i >= V4) return;
    const uint32_t n = i / HW4, q = i - n * HW4;
    float4 x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = in[(size_t)(n * C + c) * HW4 + q];      // plain loads: see stream_load (rcu_kernels.h)
    float p[4][C];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < C; ++c) p[k][c] = reinterpret_cast<const float*>(&x[c])[k];
        if (!(flags & MC_INPUT_PROBS)) softmax_inplace<C>(p[k]);
    }
    {      // float32 planes only (launch_mc_accumulate sends float64 statistics to the one-voxel kernel)
        float4* sf = reinterpret_cast<float4*>(stats);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float4 a = sf[(size_t)c * V4 + i];
            a.x += p[0][c]; a.y += p[1][c]; a.z += p[2][c]; a.w += p[3][c];
            sf[(size_t)c * V4 + i] = a;
        }
        if (flags & MC_MI) {
            float4 e = sf[(size_t)C * V4 + i];
            e.x += entropy_of<C>(p[0]); e.y += entropy_of<C>(p[1]); e.z += entropy_of<C>(p[2]); e.w += entropy_of<C>(p[3]);
            sf[(size_t)C * V4 + i] = e;
        }
    }
}

hipError_t launch_mc_accumulate(const float* in, void* stats, int C, size_t N, size_t HW, int flags, hipStream_t stream)
{
    const size_t V = N * HW;
    // (float64 statistics -- 80 bytes of read-modify-write per voxel -- run as fast or faster one voxel per thread: measured 0.84 / 0.73 of
    // the HBM peak on one / four volumes against 0.83 / 0.67 with four voxels per thread; profiles/r04_aggregation.txt)
    if (!(flags & (MC_VAR | MC_EXACT)) && vec4_ok(HW, V, {in, stats})) {
        RCU_DISPATCH_C(C, hipLaunchKernelGGL(mc_accumulate4_kernel<C_>, dim3(grid_for(V / 4)), dim3(PW_THREADS), 0, stream,
                                             reinterpret_cast<const float4*>(in), stats, (uint32_t)(HW / 4), (uint32_t)(V / 4), flags));
        return hipGetLastError();
    }
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
    if (mc_is_f64(flags)) {
        const double* sd = reinterpret_cast<const double*>(stats);
        double vsum = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double s = sd[(size_t)c * V + v];
            p[c] = (float)(s / (double)T);
            if (flags & MC_VAR) {
                const double q = sd[(size_t)(C + c) * V + v];
                vsum += (q - s * s / (double)T) / (double)(T - 1);   // unbiased, as torch.var (customsteps.py:70)
            }
        }
        if (var != nullptr && (flags & MC_VAR)) var[v] = (float)fmax(vsum / (double)C, 0.0);      // (sum p^2 - (sum p)^2 / T can round a hair below 0 where all passes agree)
        if (flags & MC_MI) sum_h = (float)sd[(size_t)mc_h_plane(flags, C) * V + v];
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

template <int C>
__global__ __launch_bounds__(PW_THREADS) void mc_finalize4_kernel(const void* stats, uint32_t HW4, uint32_t V4, int T, int flags,
                                                                   float4* __restrict__ mean, float4* __restrict__ entropy,
                                                                   float4* __restrict__ mi, float4* __restrict__ var)
{
    const uint32_t i = blockIdx.x * PW_THREADS + threadIdx.x;
    if (i >= V4) return;
    const uint32_t n = i / HW4, q = i - n * HW4;
    float p[4][C], sum_h[4] = {0.f, 0.f, 0.f, 0.f}, vr[4] = {0.f, 0.f, 0.f, 0.f};
    if (mc_is_f64(flags)) {
        const double2* sd = reinterpret_cast<const double2*>(stats);
        const size_t P2 = 2 * (size_t)V4;
        double vsum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2 s2 = sd[(size_t)c * P2 + 2 * (size_t)i + h];
                p[2 * h][c] = (float)(s2.x / (double)T);
                p[2 * h + 1][c] = (float)(s2.y / (double)T);
                if (flags & MC_VAR) {
                    const double2 q2 = sd[(size_t)(C + c) * P2 + 2 * (size_t)i + h];
                    vsum[2 * h] += (q2.x - s2.x * s2.x / (double)T) / (double)(T - 1);   // unbiased, as torch.var (customsteps.py:70)
                    vsum[2 * h + 1] += (q2.y - s2.y * s2.y / (double)T) / (double)(T - 1);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) vr[k] = (float)fmax(vsum[k] / (double)C, 0.0);
        if (flags & MC_MI) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2 e = sd[(size_t)mc_h_plane(flags, C) * P2 + 2 * (size_t)i + h];
                sum_h[2 * h] = (float)e.x;
                sum_h[2 * h + 1] = (float)e.y;
            }
        }
    } else {
        const float4* sf = reinterpret_cast<const float4*>(stats);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 a = sf[(size_t)c * V4 + i];
            p[0][c] = a.x / (float)T; p[1][c] = a.y / (float)T; p[2][c] = a.z / (float)T; p[3][c] = a.w / (float)T;
        }
        if (flags & MC_MI) {
            const float4 e = sf[(size_t)C * V4 + i];
            sum_h[0] = e.x; sum_h[1] = e.y; sum_h[2] = e.z; sum_h[3] = e.w;
        }
    }
    if (mean != nullptr) {
#pragma unroll
        for (int c = 0; c < C; ++c) mean[(size_t)(n * C + c) * HW4 + q] = float4{p[0][c], p[1][c], p[2][c], p[3][c]};
    }
    float h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = entropy_of<C>(p[k]);
    if (entropy != nullptr) entropy[i] = float4{h[0], h[1], h[2], h[3]};
    if (mi != nullptr && (flags & MC_MI))
        mi[i] = float4{h[0] - sum_h[0] / (float)T, h[1] - sum_h[1] / (float)T, h[2] - sum_h[2] / (float)T, h[3] - sum_h[3] / (float)T};
    if (var != nullptr && (flags & MC_VAR)) var[i] = float4{vr[0], vr[1], vr[2], vr[3]};
}

hipError_t launch_mc_finalize(const void* stats, int C, size_t N, size_t HW, int T, int flags, float* mean,
                              float* entropy, float* mi, float* var, hipStream_t stream)
{
    const size_t V = N * HW;
    if (vec4_ok(HW, V, {stats, mean, entropy, mi, var})) {
        RCU_DISPATCH_C(C, hipLaunchKernelGGL(mc_finalize4_kernel<C_>, dim3(grid_for(V / 4)), dim3(PW_THREADS), 0, stream, stats,
                                             (uint32_t)(HW / 4), (uint32_t)(V / 4), T, flags, reinterpret_cast<float4*>(mean),
                                             reinterpret_cast<float4*>(entropy), reinterpret_cast<float4*>(mi),
                                             reinterpret_cast<float4*>(var)));
        return hipGetLastError();
    }
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

// ---------------------------------------------------------------------------------------------
// Dropout2d factors of the MC passes of a launch, all in ONE kernel (rcu_dropout_masks).  Drawn with torch -- one bernoulli_ per
// pass, a division, a gather into the group layout -- a 62k-element draw is a 5 us kernel that takes 160 us to get its turn beside the
// persistent conv kernels (profiles/r05_script_trace.txt): 20 of them per volume sit in front of the lanes' conv launches.
// The factor of (pass t, sample i, site s, channel c) is word e & 3 of Philox4x32-10 under key seeds[t], counter (e >> 2 as two words, 0, 0) with
// e = (first_sample + i) * per_sample + site_off[s] + c: a function of (seed, GLOBAL sample index, site, channel) alone -- the same value whatever
// the batch the sample arrives in, the group the pass is launched in, the lane and the rank.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t (&out)[4])
{
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__global__ __launch_bounds__(PW_THREADS) void dropout_masks_kernel(const MaskArgs a, float* __restrict__ out)
{
    const int r = blockIdx.x * PW_THREADS + threadIdx.x;      // element of the pass's own mask [site][n][C_site]
    const int t = blockIdx.y;
    if (r >= a.per_pass) return;
    int s = 0;
    while (r >= a.site_end[s]) ++s;                  // <= 40 sites, ascending: the site of element r
    const int begin = s ? a.site_end[s - 1] : 0, len = a.site_end[s] - begin;
    const int ch = a.site_ch[s], i = (r - begin) / ch, c = (r - begin) - i * ch;
    const unsigned long long e = (a.first_sample + (unsigned long long)i) * (unsigned long long)a.per_sample + (unsigned long long)(a.site_off[s] + c);
    uint32_t w[4];
    philox4x32_10((uint32_t)a.seed[t], (uint32_t)(a.seed[t] >> 32), (uint32_t)(e >> 2), (uint32_t)(e >> 34), w);
    const uint32_t word = (e & 3) == 0 ? w[0] : (e & 3) == 1 ? w[1] : (e & 3) == 2 ? w[2] : w[3];
    const float keep = a.site_keep[s];
    const float u = (float)(word >> 8) * (1.0f / 16777216.0f);        // uniform in [0, 1), 24 bits
    const float factor = keep < 0.f ? 1.f : (keep > 0.f && u < keep) ? 1.f / keep : 0.f;
    // group layout: [site][pass][n * C_site]
    out[(size_t)a.passes * begin + (size_t)(a.first + t) * len + (r - begin)] = factor;
}

hipError_t launch_dropout_masks(const MaskArgs& a, float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(dropout_masks_kernel, dim3((a.per_pass + PW_THREADS - 1) / PW_THREADS, a.count), dim3(PW_THREADS), 0, stream, a, out);
    return hipGetLastError();
}

}  // namespace rcu
