// Per-voxel arithmetic shared by the head kernel (rcu_pointwise.hip) and the head fused into the Winograd epilogue of
// conv_cls.0 (rcu_wino.hip): softmax, entropy, the update of the MC statistics planes.  Same functions, same bits.
#pragma once
#include "rcu_kernels.h"

namespace rcu {

// ------------------------------------------------------------------------------- shared device helpers
template <int C>
__device__ __forceinline__ void softmax_inplace(float (&l)[C])
{
    float mx = l[0];
#pragma unroll
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, l[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        l[c] = expf(l[c] - mx);
        s += l[c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] = l[c] / s;
}

template <int C>
__device__ __forceinline__ float entropy_of(const float (&p)[C])
{
    float h = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) h += (p[c] > 0.f) ? p[c] * logf(p[c]) : 0.f;
    return -h;
}

// ---- float64 statistics (MC_VAR and / or MC_EXACT): planes [sum p_c (C)] [sum p_c^2 (C) if MC_VAR] [sum H if MC_MI]
__device__ __forceinline__ bool mc_is_f64(int flags) { return (flags & (MC_VAR | MC_EXACT)) != 0; }
__device__ __forceinline__ int mc_h_plane(int flags, int C) { return (flags & MC_VAR) ? 2 * C : C; }   // float64 layouts only

// MC_EXACT: every addend is rounded (to nearest even) to a multiple of 2^-40 before it is added -- 1.5 * 2^12 has the float64 ulp 2^-40
// and x + 6144 stays inside [4096, 8192) for the addends here (p, p^2 <= 1, H <= log 8), so the sum of the two roundings is that multiple
// exactly.  Sums of multiples of 2^-40 below 2^13 are representable in float64's 53 bits: every addition is EXACT, hence associative and
// commutative -- the statistics of T <= 2048 passes carry the same bits whatever the order, the pass groups, the stream lanes, the ranks
// and the reduction tree of the collective that merges them.  (No fast-math: hipcc does not reassociate (x + c) - c away; the GPU test
// test_exact_statistics_* checks the quantum.)
__device__ __forceinline__ double mc_quantise(double x)
{
    const double shifted = __dadd_rn(x, 6144.0);
    return __dsub_rn(shifted, 6144.0);
}

template <int C>
__device__ __forceinline__ void accumulate_voxel(void* stats, size_t v, size_t V, int flags, const float (&p)[C])
{
    if (mc_is_f64(flags)) {
        double* sd = reinterpret_cast<double*>(stats);
        const bool exact = (flags & MC_EXACT) != 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double pc = (double)p[c];
            sd[(size_t)c * V + v] += exact ? mc_quantise(pc) : pc;
            if (flags & MC_VAR) sd[(size_t)(C + c) * V + v] += exact ? mc_quantise(pc * pc) : pc * pc;
        }
        if (flags & MC_MI) {
            const double h = (double)entropy_of<C>(p);
            sd[(size_t)mc_h_plane(flags, C) * V + v] += exact ? mc_quantise(h) : h;
        }
    } else {
        float* sf = reinterpret_cast<float*>(stats);
#pragma unroll
        for (int c = 0; c < C; ++c) sf[(size_t)c * V + v] += p[c];
        if (flags & MC_MI) sf[(size_t)C * V + v] += entropy_of<C>(p);
    }
}

// The statistics entries of one voxel held in registers across the passes of a group: load, add pass after pass in
// the same order and with the same operations as accumulate_voxel, store -- bit-identical to one launch per pass.
template <int C>
struct VoxelStats {
    double d[2 * C + 1];      // [sum p_c] [sum p_c^2] [sum H], whatever the planes of the blob
    float f[C + 1];
    __device__ __forceinline__ void load(const void* stats, size_t v, size_t V, int flags)
    {
        if (mc_is_f64(flags)) {
            const double* sd = reinterpret_cast<const double*>(stats);
#pragma unroll
            for (int k = 0; k < C; ++k) d[k] = sd[(size_t)k * V + v];
#pragma unroll
            for (int k = 0; k < C; ++k) d[C + k] = (flags & MC_VAR) ? sd[(size_t)(C + k) * V + v] : 0.0;
            d[2 * C] = (flags & MC_MI) ? sd[(size_t)mc_h_plane(flags, C) * V + v] : 0.0;
        } else {
            const float* sf = reinterpret_cast<const float*>(stats);
#pragma unroll
            for (int k = 0; k < C; ++k) f[k] = sf[(size_t)k * V + v];
            f[C] = (flags & MC_MI) ? sf[(size_t)C * V + v] : 0.f;
        }
    }
    __device__ __forceinline__ void add(int flags, const float (&p)[C])
    {
        if (mc_is_f64(flags)) {
            const bool exact = (flags & MC_EXACT) != 0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double pc = (double)p[c];
                d[c] += exact ? mc_quantise(pc) : pc;
                if (flags & MC_VAR) d[C + c] += exact ? mc_quantise(pc * pc) : pc * pc;
            }
            if (flags & MC_MI) {
                const double h = (double)entropy_of<C>(p);
                d[2 * C] += exact ? mc_quantise(h) : h;
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) f[c] += p[c];
            if (flags & MC_MI) f[C] += entropy_of<C>(p);
        }
    }
    __device__ __forceinline__ void store(void* stats, size_t v, size_t V, int flags) const
    {
        if (mc_is_f64(flags)) {
            double* sd = reinterpret_cast<double*>(stats);
#pragma unroll
            for (int k = 0; k < C; ++k) sd[(size_t)k * V + v] = d[k];
            if (flags & MC_VAR) {
#pragma unroll
                for (int k = 0; k < C; ++k) sd[(size_t)(C + k) * V + v] = d[C + k];
            }
            if (flags & MC_MI) sd[(size_t)mc_h_plane(flags, C) * V + v] = d[2 * C];
        } else {
            float* sf = reinterpret_cast<float*>(stats);
#pragma unroll
            for (int k = 0; k < C; ++k) sf[(size_t)k * V + v] = f[k];
            if (flags & MC_MI) sf[(size_t)C * V + v] = f[C];
        }
    }
};


}  // namespace rcu
