// PostNet (common/model/postnet.py:6-18): nb_convs x [Conv2d 1x1 C->C + Dropout2d + BatchNorm2d(eval) + ReLU] + Conv2d 1x1
// C->nb_classes, the auxiliary confidence network of the auxiliary_feat runs (bin-dl/brats_test_auxiliary_feat.py:74-77),
// evaluated on the U-Net feature map (unet.py:178-179) as ONE kernel: a per-voxel MLP chained through the fp32 matrix cores.
//
// Orientation: D[cout][voxel] = W[cout][cin] * X[cin][voxel] (v_mfma_f32_32x32x2_f32, A = weights, B = activations), in
// CB x CB blocks of 32 channels (CB = 1 for the shipped 32-channel feature map; 2 and 3 for U-Nets with start_filters up to 96).
// A lane (voxel n = lane % 32, half h = lane / 32) receives the 16 output channels perm(r, h) = 8 (r / 4) + 4 h + r % 4 of
// ITS voxel and of every channel block in accumulator element r.  The K index of an MFMA step is only a label, so step s of the next
// layer uses channel perm(s, h) as its k-th input: the B operand of step s is accumulator element s of the same lane (after
// bias + ReLU) -- the activations never leave their registers between the layers -- and the A operand is
// W[m][perm(s, h)], i.e. float4 pieces of the row-major folded weight matrix, kept in LDS.  The bias enters as the
// initial accumulator.  The first layer reads the same pattern from the NHWC feature map: four float4 per lane and block.
// MC-dropout inside PostNet (a Dropout2d between conv and BatchNorm, Conv2dBnRelu, unet.py:14-15): with per-(sample, channel) factors m
// the unit is relu(m * (W' x + alpha b) + (beta - alpha mean)); the accumulator then starts from alpha b and the factor and the
// second constant are applied behind the MFMAs.  Exact fp32.
// HBM: 128 CB bytes read + 4 nb_classes bytes written per voxel; MFMA: (nb_convs CB^2 + CB) * 16 steps per 32 voxels.
#include "rcu_kernels.h"

namespace rcu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS image per layer (postnet_pack_layer, rcu_api.hip): CB x CB weight tiles [cob][cib][j = 0..3][lane][4]
// (float4 = W[32 cob + lane % 32][32 cib + 8 j + 4 (lane / 32) .. + 3]), then the start values of the accumulators
// [cob][j][h][4] (bias[32 cob + 8 j + 4 h .. + 3]) and the constants added behind the dropout factor, same layout.
template <int CB>
__global__ __launch_bounds__(256) void postnet_kernel(const float* __restrict__ x, int channel_pitch, size_t nvox, int hw,
                                                      const float* __restrict__ packed, int n_layers, int nb_classes, int channels,
                                                      const float* __restrict__ masks, int n_images, float* __restrict__ logits)
{
    extern __shared__ __attribute__((aligned(16))) float pn_smem[];
    constexpr int LAYER = pn_layer_floats(CB), W_FLOATS = CB * CB * 1024, B_FLOATS = CB * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    for (int i = tid; i < n_layers * LAYER / 4; i += 256)
        reinterpret_cast<f32x4*>(pn_smem)[i] = reinterpret_cast<const f32x4*>(packed)[i];
    __syncthreads();

    const size_t ngroups = (nvox + 31) / 32;
    const size_t stride = (size_t)gridDim.x * 4;
    size_t g = (size_t)blockIdx.x * 4 + wave;
    if (g >= ngroups) return;

    auto load = [&](size_t grp, f32x4 (&q)[CB][4]) {
        size_t v = grp * 32 + n;
        v = v < nvox ? v : nvox - 1;            // tail lanes read a valid voxel, their results are not stored
        const float* src = x + v * channel_pitch + 4 * h;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) q[cb][j] = *reinterpret_cast<const f32x4*>(src + 32 * cb + 8 * j);
    };
    f32x4 cur[CB][4], nxt[CB][4];
    load(g, cur);
    for (; g < ngroups; g += stride) {
        const bool more = g + stride < ngroups;
        load(more ? g + stride : g, nxt);       // next group's features are in flight during this group's MFMAs
        f32x16 act[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                act[cb][4 * j + 0] = cur[cb][j].x, act[cb][4 * j + 1] = cur[cb][j].y, act[cb][4 * j + 2] = cur[cb][j].z,
                               act[cb][4 * j + 3] = cur[cb][j].w;
            }
        const size_t v = g * 32 + n;
        const size_t img = (v < nvox ? v : nvox - 1) / hw;
        for (int l = 0; l < n_layers; ++l) {
            const float* wl = pn_smem + (size_t)l * LAYER;
            const bool last = l + 1 == n_layers;
            f32x16 acc[CB];
#pragma unroll
            for (int cob = 0; cob < CB; ++cob) {
                if (last && cob > 0) continue;      // conv_logits: at most 32 classes = one block of output channels
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(wl + W_FLOATS + ((cob * 4 + j) * 2 + h) * 4);
                    acc[cob][4 * j + 0] = b.x, acc[cob][4 * j + 1] = b.y, acc[cob][4 * j + 2] = b.z, acc[cob][4 * j + 3] = b.w;
                }
#pragma unroll
                for (int cib = 0; cib < CB; ++cib)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(wl + (((cob * CB + cib) * 4 + j) * 64 + lane) * 4);
                        acc[cob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, act[cib][4 * j + 0], acc[cob], 0, 0, 0);
                        acc[cob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, act[cib][4 * j + 1], acc[cob], 0, 0, 0);
                        acc[cob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, act[cib][4 * j + 2], acc[cob], 0, 0, 0);
                        acc[cob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, act[cib][4 * j + 3], acc[cob], 0, 0, 0);
                    }
            }
            if (!last) {
                if (masks != nullptr) {   // wave-uniform: Dropout2d factors [layer][image][channel] behind the conv, BatchNorm's shift after them
                    const float* const mrow = masks + ((size_t)l * n_images + img) * channels;
#pragma unroll
                    for (int cob = 0; cob < CB; ++cob)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int c = 32 * cob + 8 * (r >> 2) + 4 * h + (r & 3);
                            const float mk = c < channels ? mrow[c] : 1.f;
                            const float b2 = wl[W_FLOATS + B_FLOATS + ((cob * 4 + (r >> 2)) * 2 + h) * 4 + (r & 3)];
                            acc[cob][r] = fmaf(acc[cob][r], mk, b2);
                        }
                }
#pragma unroll
                for (int cob = 0; cob < CB; ++cob)
#pragma unroll
                    for (int r = 0; r < 16; ++r) act[cob][r] = fmaxf(acc[cob][r], 0.f);
            } else {
                act[0] = acc[0];
            }
        }
        if (v < nvox) {
            const size_t pix = v % hw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 8 * (r >> 2) + 4 * h + (r & 3);
                if (c < nb_classes) logits[(img * nb_classes + c) * hw + pix] = act[0][r];
            }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[cb][j] = nxt[cb][j];
    }
}

template <int CB>
static hipError_t launch_postnet_cb(const float* x, int channel_pitch, size_t nvox, int hw, const float* packed, int n_layers,
                                    int nb_classes, int channels, const float* masks, int n_images, float* logits, hipStream_t stream)
{
    const size_t ngroups = (nvox + 31) / 32;
    const size_t wgs = (ngroups + 3) / 4;
    const unsigned grid = (unsigned)(wgs < 2048 ? wgs : 2048);
    const size_t lds = (size_t)n_layers * pn_layer_floats(CB) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&postnet_kernel<CB>), (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(postnet_kernel<CB>, dim3(grid), dim3(256), lds, stream, x, channel_pitch, nvox, hw, packed, n_layers,
                       nb_classes, channels, masks, n_images, logits);
    return hipGetLastError();
}

hipError_t launch_postnet(const float* x, int channel_pitch, size_t nvox, int hw, const float* packed, int n_layers,
                          int nb_classes, int channels, const float* masks, float* logits, hipStream_t stream)
{
    if (nvox == 0) return hipSuccess;
    const int cb = (channels + 31) / 32;
    if (n_layers < 1 || n_layers > PN_MAX_LAYERS || nb_classes < 1 || nb_classes > 32 || cb < 1 || cb > PN_MAX_BLOCKS ||
        channel_pitch < 32 * cb || channel_pitch % 4)
        return hipErrorInvalidValue;
    const int n_images = (int)((nvox + hw - 1) / hw);
    switch (cb) {
        case 1: return launch_postnet_cb<1>(x, channel_pitch, nvox, hw, packed, n_layers, nb_classes, channels, masks, n_images, logits, stream);
        case 2: return launch_postnet_cb<2>(x, channel_pitch, nvox, hw, packed, n_layers, nb_classes, channels, masks, n_images, logits, stream);
        default: return launch_postnet_cb<3>(x, channel_pitch, nvox, hw, packed, n_layers, nb_classes, channels, masks, n_images, logits, stream);
    }
}

}  // namespace rcu
