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

template <int C>
__device__ __forceinline__ void accumulate_voxel(void* stats, size_t v, size_t V, int flags, const float (&p)[C])
{
    if (flags & MC_VAR) {
        double* sd = reinterpret_cast<double*>(stats);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double pc = (double)p[c];
            sd[(size_t)c * V + v] += pc;
            sd[(size_t)(C + c) * V + v] += pc * pc;
        }
        if (flags & MC_MI) sd[(size_t)(2 * C) * V + v] += (double)entropy_of<C>(p);
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


}  // namespace rcu
