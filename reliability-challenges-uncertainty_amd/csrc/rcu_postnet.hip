// PostNet (common/model/postnet.py:6-18): nb_convs x [Conv2d 1x1 C->C + BatchNorm2d(eval) + ReLU] + Conv2d 1x1 C->nb_classes,
// the auxiliary confidence network of the auxiliary_feat runs (bin-dl/brats_test_auxiliary_feat.py:74-77), evaluated
// on the U-Net feature map (unet.py:178-179) as ONE kernel: a per-voxel MLP chained through the fp32 matrix cores.
//
// Orientation: D[cout][voxel] = W[cout][cin] * X[cin][voxel] (v_mfma_f32_32x32x2_f32, A = weights, B = activations).
// A lane (voxel n = lane % 32, half h = lane / 32) receives the 16 output channels perm(r, h) = 8 (r / 4) + 4 h + r % 4
// of ITS voxel in accumulator element r.  The K index of an MFMA step is only a label, so step s of the next layer uses
// channel perm(s, h) as its k-th input: the B operand of step s is accumulator element s of the same lane (after
// bias + ReLU) -- the activations never leave their registers between the layers -- and the A operand is
// W[m][perm(s, h)], i.e. float4 pieces of the row-major folded weight matrix, kept in LDS.  The bias enters as the
// initial accumulator.  The first layer reads the same pattern from the NHWC feature map: four float4 per lane.
// Exact fp32.  HBM: 128 B read + 4 nb_classes B written per voxel; MFMA: (nb_convs + 1) * 16 steps per 32 voxels.
#include "rcu_kernels.h"

namespace rcu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS image per layer: weights [j = 0..3][lane][4] (float4 = W[lane % 32][8 j + 4 (lane / 32) .. + 3]), then
// bias [j][h][4] (bias[8 j + 4 h .. + 3]) -- what postnet_pack_layer (rcu_api.hip) writes.
static constexpr int PN_W_FLOATS = 4 * 64 * 4;   // followed by 4 * 2 * 4 bias floats: PN_LAYER_FLOATS in rcu_kernels.h

__global__ __launch_bounds__(256) void postnet_kernel(const float* __restrict__ x, int channel_pitch, size_t nvox, int hw,
                                                      const float* __restrict__ packed, int n_layers, int nb_classes,
                                                      float* __restrict__ logits)
{
    extern __shared__ __attribute__((aligned(16))) float pn_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    for (int i = tid; i < n_layers * PN_LAYER_FLOATS / 4; i += 256)
        reinterpret_cast<f32x4*>(pn_smem)[i] = reinterpret_cast<const f32x4*>(packed)[i];
    __syncthreads();

    const size_t ngroups = (nvox + 31) / 32;
    const size_t stride = (size_t)gridDim.x * 4;
    size_t g = (size_t)blockIdx.x * 4 + wave;
    if (g >= ngroups) return;

    auto load = [&](size_t grp, f32x4 (&q)[4]) {
        size_t v = grp * 32 + n;
        v = v < nvox ? v : nvox - 1;            // tail lanes read a valid voxel, their results are not stored
        const float* src = x + v * channel_pitch + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const f32x4*>(src + 8 * j);
    };
    f32x4 cur[4], nxt[4];
    load(g, cur);
    for (; g < ngroups; g += stride) {
        const bool more = g + stride < ngroups;
        load(more ? g + stride : g, nxt);       // next group's features are in flight during this group's MFMAs
        f32x16 act;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            act[4 * j + 0] = cur[j].x, act[4 * j + 1] = cur[j].y, act[4 * j + 2] = cur[j].z, act[4 * j + 3] = cur[j].w;
        }
        for (int l = 0; l < n_layers; ++l) {
            const float* wl = pn_smem + (size_t)l * PN_LAYER_FLOATS;
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + PN_W_FLOATS + (j * 2 + h) * 4);
                acc[4 * j + 0] = b.x, acc[4 * j + 1] = b.y, acc[4 * j + 2] = b.z, acc[4 * j + 3] = b.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(wl + (j * 64 + lane) * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, act[4 * j + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, act[4 * j + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, act[4 * j + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, act[4 * j + 3], acc, 0, 0, 0);
            }
            if (l + 1 < n_layers) {
#pragma unroll
                for (int r = 0; r < 16; ++r) act[r] = fmaxf(acc[r], 0.f);
            } else {
                act = acc;
            }
        }
        const size_t v = g * 32 + n;
        if (v < nvox) {
            const size_t img = v / hw, pix = v % hw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 8 * (r >> 2) + 4 * h + (r & 3);
                if (c < nb_classes) logits[(img * nb_classes + c) * hw + pix] = act[r];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
}

hipError_t launch_postnet(const float* x, int channel_pitch, size_t nvox, int hw, const float* packed, int n_layers,
                          int nb_classes, float* logits, hipStream_t stream)
{
    if (nvox == 0) return hipSuccess;
    if (n_layers < 1 || n_layers > PN_MAX_LAYERS || nb_classes < 1 || nb_classes > 32 || channel_pitch < 32 || channel_pitch % 4)
        return hipErrorInvalidValue;
    const size_t ngroups = (nvox + 31) / 32;
    const size_t wgs = (ngroups + 3) / 4;
    const unsigned grid = (unsigned)(wgs < 2048 ? wgs : 2048);
    const size_t lds = (size_t)n_layers * PN_LAYER_FLOATS * sizeof(float);
    hipLaunchKernelGGL(postnet_kernel, dim3(grid), dim3(256), lds, stream, x, channel_pitch, nvox, hw, packed, n_layers,
                       nb_classes, logits);
    return hipGetLastError();
}

}  // namespace rcu
