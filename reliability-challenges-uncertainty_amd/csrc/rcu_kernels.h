// Internal declarations shared by the HIP translation units of librcu_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rcu {

// ---------------------------------------------------------------------------------------------
// conv3x3 implicit GEMM (rcu_conv.hip)
// ---------------------------------------------------------------------------------------------
// Activations are fp32 with the channel count padded to a multiple of 32 (the network input to a multiple of 8); padded
// channels hold zeros.  One launch covers every slice of the batch.  Two layouts of a [N][H][W][C] tensor:
//   NHWC     a pixel's C channels are contiguous                                   (every kernel)
//   blocked  [N][C/8][H][W][8]: a pixel's 8-channel group (the Cin chunk of the Winograd kernels) is 32 contiguous bytes and
//            the groups of neighbouring pixels follow each other, so an LDS-DMA instruction that stages 8 channels of 32..64
//            pixels touches 8..16 whole 128-byte lines instead of 32..64 quarter lines (Winograd kernels + rcu_first.hip only)
// Both are addressed as  sample * (H W C 4) + chunk * chunk_bytes + (y W + x) * pix_bytes + (c mod 8) * 4  with
//   NHWC: pix_bytes = 4 C, chunk_bytes = 32;   blocked: pix_bytes = 32, chunk_bytes = 32 H W.
struct ConvArgs {
    const float* src1;   // [N][H][W][C1]   first  K-range (channels [0, C1))
    const float* src2;   // [N][H][W][C2]   second K-range (cat-free decoder), may be null (C2 = 0)
    const float* wpack;  // [Cin/KC][NTW_total] tiles of [TAPS][BN][KC+4] floats, each padded to 1024-float multiples
    const float* alpha;  // [CoutP]  folded BN scale               (1 for bias-only convs)
    const float* betab;  // [CoutP]  alpha * conv bias
    const float* beta;   // [CoutP]  folded BN shift               (0 for bias-only convs)
    const float* mask;   // [N][Cmask] dropout factors {0, 1/(1-p)} of this site, or null (eval / no site)
    const float* mask2;  // second site for output channels >= Csplit (fused cls+sigma head unit), or null
    float* out;          // [N][H][W][CoutP]; sub-pixel up-conv kernels: [N][2H][2W][CoutP]
    float* pooled;       // [N][H/2][W/2][CoutP] or null
    int out_H, out_W;    // sub-pixel up-conv (direct kernels): extent of the output tensor when it is larger than 2H x 2W -- the
    int out_y0, out_x0;  //   reference's centre pad (common/model/unet.py:110-116) -- and where the 2H x 2W image sits in it; 0: 2H x 2W
    int N, H, W;         // input grid = pixel-tile grid
    // Padded levels (`part` = 1; Winograd kernels + rcu_first.hip; rcu_api.hip, choose_level_extents): the tensors of a level are allocated with
    // extents rounded up to whole tiles, H x W above are those ALLOCATED extents of the source level, and the pixels beyond the real image hold
    // zeros that no kernel ever writes -- so a tile that hangs over the real border reads exactly the zero padding the reference's conv applies
    // (common/model/unet.py:13: padding=1) and the load side of the kernels is the whole-tile one.  What changes is the store side:
    int part;            //   1: the fields below are set and the kernel's store side honours them
    int Hr, Wr;          //   real extent of the OUTPUT grid (up-convolutions: of the up-sampled grid): pixels at or beyond it are not stored
    int pool_H, pool_W;  //   allocated extent of `pooled` (out_H, out_W above: of `out`) -- a level's padding is its own, not half its parent's
    int C1, C2;          // padded channel counts of the two sources
    int cin_real;        // first-layer kernel (rcu_first.hip): input channels that are not padding
    const float* x_nchw; // first-layer kernel: when set, the caller's [n_images][cin_real][H][W] input is read in place of src1
    int n_images;        //   (sample n is image n % n_images: pass groups replicate the images)
    int CoutP;           // padded output channels (multiple of 32)
    int Cmask;           // real channel count of the dropout site (mask row length)
    int Csplit, Cmask2;  // mask2 row = [N][Cmask2], applies to channel co - Csplit
    int relu;            // 1: max(0, .) epilogue
    int accumulate;      // direct kernels: 1 = the unit's result is ADDED to what `out` holds (ConvResidualBlock: the 1x1 residual
                         // conv has been written there), before the 2x2 max-pool
    int tiles_y, tiles_x, slice_groups;
    int NT;              // output-channel tiles
    uint32_t magic_ntw, magic_tx, magic_ty;   // ceil(2^32 / d) for d = NTW_total, tiles_x, tiles_y (0: divide): the Winograd kernels
                                              // split a work item into tile coordinates with s_mul_hi instead of three divisions
    int NTW_total;       // weight tiles per Cin chunk = NT (3x3) or 4 * NT (sub-pixel: one set per parity class)
    uint32_t src1_bytes, src2_bytes, wpack_bytes;   // buffer-resource ranges (Winograd kernels)
    // layout of the sources (both alike), of `out` and of `pooled` (see above); 0 / 0 = NHWC for the kernels that know nothing else
    uint32_t in_pix_bytes, in_chunk_bytes;
    uint32_t out_pix_bytes, out_chunk_bytes;
    uint32_t pool_pix_bytes, pool_chunk_bytes;
    // fused 1x1 head of conv_cls.0 (rcu_wino.hip, rcu_wino4.hip; two classes): when head_w is set the conv unit's output stays on chip
    const float* head_w;   // [2][32] 1x1 weights, head_b[2] bias
    const float* head_b;
    float* head_logits;    // NCHW [N][2][H*W] or null
    void* head_stats;      // MC statistics blob or null
    int head_flags;        // MC_MI | MC_VAR
    size_t head_V;         // voxels of one pass = images * H * W
    int head_passes;       // MC passes in this launch: sample t * head_images + i is image i (pass groups); its statistics entry is i's
    int head_images;
};

enum ConvConfig {
    CONV_CFG_T8x16_N64 = 0,          // 8x16-pixel tile, 64 couts, Cin chunks of 8, LDS double-buffered (workhorse)
    CONV_CFG_T8x16_N32 = 1,          // 8x16-pixel tile, 32 couts, Cin chunks of 8, double-buffered (32-channel layers)
    CONV_CFG_T8x16_N32_FIRST = 2,    // first layer: Cin padded to 8 = a single chunk
    CONV_CFG_S2T12x8_N64 = 3,        // two whole 12x8 slices per workgroup (BraTS bottom level)
    CONV_CFG_UP_T8x16_N64 = 4,       // sub-pixel up-conv (2x2 taps on the low-res grid), 64 couts
    CONV_CFG_UP_T8x16_N32 = 5,       // sub-pixel up-conv, 32 couts
    CONV_CFG_UP_S2T12x8_N64 = 6,     // sub-pixel up-conv out of the 12x8 bottom level, chunks of 8
    CONV_CFG_T16x16_N32 = 7,         // 16x16-pixel tile, 32 couts: halves the weight staging per MFMA of the 32-channel layers
    CONV_CFG_UP_T16x16_N32 = 8,      // sub-pixel up-conv, 32 couts, 16x16 low-res tile
    CONV_CFG_T16x16_N64 = 9,         // 256 pixels x 64 couts: half the staging / barriers / fragment reads per MFMA
    CONV_CFG_S2T8x16_N64 = 10,       // the same tile as two 8x16 pieces of consecutive slices (heights not divisible by 16)
    CONV_CFG_COUNT,
    // Winograd F(2x2,3x3) kernels (rcu_wino.hip): 16 positions instead of 9 taps, weights host-transformed
    CONV_CFG_WINO_T16x16_N64 = CONV_CFG_COUNT,   // 16x16-pixel tile (64 Winograd tiles) x 64 couts
    CONV_CFG_WINO_T16x32_N32,                    // 16x32-pixel tile x 32 couts (32-channel layers)
    CONV_CFG_WINO_S2T8x16_N64,                   // two 8x16 pieces of consecutive slices x 64 couts
    CONV_CFG_WINO_S8T4x8_N64,                    // 4x8 strips of eight consecutive slices x 64 couts (8-pixel-wide level)
    CONV_CFG_WINO_T16x32_N32_HEAD,               // T16x32_N32 with the 1x1 head + softmax + statistics in the epilogue
    // Winograd F(2x2,2x2) sub-pixel up-convolutions (rcu_wino_up.hip): 18 positions = two parity classes per work item
    CONV_CFG_UPW_T16x16_N64,
    CONV_CFG_UPW_T16x32_N32,
    CONV_CFG_UPW_S2T8x16_N64,
    CONV_CFG_UPW_S8T4x8_N64,
    // first conv unit of the network, K = 9 taps x 4 (8) channels unpadded (rcu_first.hip)
    CONV_CFG_FIRST_T8x32,
    // Winograd F(4x4,3x3) kernels (rcu_wino4.hip): 36 positions, one wave per SIMD (288 accumulator registers), 32 couts
    CONV_CFG_WINO4_T32x32_N32,                   // 32x32-pixel tile (64 Winograd tiles of 4x4)
    CONV_CFG_WINO4_S2T16x32_N32,                 // 16x32 pixels of two consecutive slices
    CONV_CFG_WINO4_S8T8x16_N32,                  // 8x16 pixels of eight consecutive slices, images 16 pixels wide
    CONV_CFG_WINO4_S8T12x8_N32,                  // 12x8 pixels of eight consecutive slices (6 tiles per slice in the 8 tile slots of the S8 block), images 8 wide
    CONV_CFG_WINO4_T32x32_N32_HEAD,              // T32x32_N32 with the 1x1 head + softmax + statistics in the epilogue (run-time choice of forward_impl, never a plan entry)
    CONV_CFG_WINO4_S4T8x32_N32,                  // 8x32 pixels of four consecutive slices, images 32 pixels wide (round 6: ISIC's 24x32 level without padding)
    CONV_CFG_END
};

struct ConvConfigInfo {
    int TS, TH, TW, BN, KC, TAPS;
    const char* kernel_name;
    int KCP;   // floats per [tap][channel] row of a packed weight tile (KC, or KC + 4 in the padded layout)
    int SWZ;   // 1: the two 16-byte units of a row are swapped for output channels 16..31 (mod 32), see ConvTile
    int WINO;  // 1: Winograd F(2x2,3x3) kernel, TAPS = 16 positions; 2: F(2x2,2x2) up-conv, TAPS = 18 (two classes);
               // 3: F(4x4,3x3), TAPS = 36 positions; packed tile = [p][channel pair][cout][2] (rcu_wino.hip, rcu_wino_up.hip,
               // rcu_wino4.hip)
};
const ConvConfigInfo& conv_config_info(int cfg);
const ConvConfigInfo& wino_config_info(int cfg);
const ConvConfigInfo& wino_up_config_info(int cfg);
const ConvConfigInfo& wino4_config_info(int cfg);
const ConvConfigInfo& first_config_info();
hipError_t launch_conv3x3(int cfg, const ConvArgs& a, hipStream_t stream);
hipError_t launch_conv_wino(int cfg, const ConvArgs& a, hipStream_t stream);
hipError_t launch_upconv_wino(int cfg, const ConvArgs& a, hipStream_t stream);
hipError_t launch_conv_wino4(int cfg, const ConvArgs& a, hipStream_t stream);
hipError_t launch_conv_first(const ConvArgs& a, hipStream_t stream);

// Loads of data a kernel reads exactly once (activation / logits / probability / label streams of the per-voxel kernels): non-temporal, so that the
// stream does not push the lines that ARE reused -- statistics planes that are read-modify-written, weights -- out of L2 and the memory-side cache.
// Measured round 4 (profiles/r04_nt_loads.txt): head kernel 116 -> 98 us (0.61 -> 0.72 of the HBM peak), ece_hist / unc_counts 0.72 / 0.74 -> 0.79 /
// 0.82.  NOT for the conv kernels' input staging (their input was written by the previous kernel and neighbouring tiles re-read its halo: +4..5 %
// time with nt there) and NOT for mc_accumulate / mc_finalize (their logits and statistics were written by the kernel before them and one volume's
// worth sits in the 256 MB memory-side cache: finalize 13.3 -> 15.3 us with nt).  -DRCU_STREAM_NT=0: plain loads everywhere.
#ifndef RCU_STREAM_NT
#define RCU_STREAM_NT 1
#endif
#if defined(__HIPCC__)
template <class V>
__device__ __forceinline__ V stream_load_as(const void* p)
{
#if RCU_STREAM_NT
    return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#else
    return *reinterpret_cast<const V*>(p);
#endif
}
__device__ __forceinline__ float stream_load(const float* p) { return stream_load_as<float>(p); }
__device__ __forceinline__ unsigned stream_load(const unsigned* p) { return stream_load_as<unsigned>(p); }
__device__ __forceinline__ float4 stream_load(const float4* p)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 v = stream_load_as<v4>(p);
    return float4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ double2 stream_load(const double2* p)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2 v = stream_load_as<v2>(p);
    return double2{v.x, v.y};
}
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: set once per (kernel, device), so that one
// process driving several GPUs gets it on every one of them.
hipError_t set_max_dynamic_lds(const void* kernel, int bytes);

// ---------------------------------------------------------------------------------------------
// layout / head / aggregation kernels (rcu_pointwise.hip)
// ---------------------------------------------------------------------------------------------
// out_nhwc is allocated PH x PW >= H x W per image (a padded level 0): only the H x W real pixels are written
hipError_t launch_pack_input(const float* x_nchw, float* out_nhwc, int N, int C, int CP, int H, int W, int PH, int PW, int replicas,
                             hipStream_t stream);

// the H x W real pixels of an NHWC tensor allocated PH x PW per image -> a compact [N][H][W][CP] tensor (rcu_unet_features on a padded level 0)
hipError_t launch_crop_nhwc(const float* src, float* dst, int N, int H, int W, int PH, int PW, int CP, hipStream_t stream);

// MC statistics blob: planes over the voxel index v = n*HW + hw.
//   neither RCU_MC_VAR nor RCU_MC_EXACT : float planes  [sum_p[0..C-1]] [sum_H if MI]
//   RCU_MC_VAR and / or RCU_MC_EXACT    : double planes [sum_p[0..C-1]] [sum_p^2[0..C-1] if VAR] [sum_H if MI]
//   RCU_MC_EXACT: the addends are rounded to multiples of 2^-40 first, which makes every addition exact (rcu_head_common.h)
constexpr int MC_MI = 1;
constexpr int MC_VAR = 2;
constexpr int MC_INPUT_PROBS = 4;   // accumulate: input already holds probabilities (ensemble seam)
constexpr int MC_EXACT = 8;
constexpr int MAX_CLASSES = 8;

struct HeadArgs {
    const float* act;     // [V][CP] NHWC activations of the (fused) head conv unit
    const float* w_cls;   // [C][CPh]  1x1 weights (zero padded), bias b_cls[C]
    const float* b_cls;
    const float* w_sig;   // optional twin (reads channels [CPh, 2*CPh) of act)
    const float* b_sig;
    float* logits;        // NCHW [N][C][HW] or null
    float* sigma;         // NCHW or null
    float* sigma_sum;     // NCHW or null: += |sigma| (or exp(sigma), sigma_log) of this pass (aleatoric + MC extension)
    int sigma_log;
    void* stats;          // MC stats blob or null
    int C, CP, CPh, stats_flags;
    size_t V, HW;         // V = voxels of ONE pass (n * HW)
    int passes;           // > 1 (statistics only): act holds `passes` consecutive groups of V voxels, all accumulated
                          // into the same V statistics entries, in pass order
};
hipError_t launch_head(const HeadArgs& a, hipStream_t stream);

hipError_t launch_mc_accumulate(const float* in_nchw, void* stats, int C, size_t N, size_t HW, int flags,
                                hipStream_t stream);
hipError_t launch_mc_finalize(const void* stats, int C, size_t N, size_t HW, int T, int flags, float* mean,
                              float* entropy, float* mi, float* var, hipStream_t stream);
hipError_t launch_softmax_nchw(const float* logits, float* probs, int C, size_t N, size_t HW, hipStream_t stream);
hipError_t launch_aleatoric(const float* logits, const float* sigma_raw, int C, size_t N, size_t HW, int is_log_sigma,
                            float* probs, float* sigma_out, uint8_t* prediction, float* sigma_pred,
                            hipStream_t stream);
hipError_t launch_argmax_fg(const float* probs_nchw, int C, size_t N, size_t HW, uint8_t* prediction, float* p_fg,
                            hipStream_t stream);
// Dropout2d factors of the passes of a pass group, drawn in one launch (include/rcu.h: rcu_dropout_masks)
constexpr int MASK_MAX_PASSES = 32;      // seeds per launch (kernel arguments)
constexpr int MASK_MAX_SITES = 40;       // 2 * (2 * 8 + 1) + the head units: depth <= 8
struct MaskArgs {
    unsigned long long seed[MASK_MAX_PASSES];
    int site_end[MASK_MAX_SITES];        // exclusive prefix sums of n * C_site: site s covers [site_end[s - 1], site_end[s]) of a pass's own mask
    float site_keep[MASK_MAX_SITES];     // 1 - p of the site; < 0: the site is not active (factor 1); 0: p = 1 (factor 0)
    int passes;                          // passes of the whole group: rows per site of the output layout
    int first, count;                    // this launch draws passes first .. first + count - 1 (seed[0 .. count - 1])
    int sites, per_pass;
    int site_ch[MASK_MAX_SITES];         // channels of the site
    int site_off[MASK_MAX_SITES];        // offset of the site in ONE sample's factors (prefix sums of site_ch); per_sample = their total
    int per_sample;
    unsigned long long first_sample;     // global index of the batch's sample 0 (include/rcu.h, rcu_dropout_masks)
};
hipError_t launch_dropout_masks(const MaskArgs& a, float* out, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// calibration kernels (rcu_calib.hip)
// ---------------------------------------------------------------------------------------------
constexpr int MAX_BINS = 32;
constexpr int MAX_THR = 16;
struct EceResult {           // one per volume
    unsigned long long count[MAX_BINS];
    double sum_conf[MAX_BINS];
    unsigned long long sum_pos[MAX_BINS];
};
void calib_set_blocks_per_workgroup(int ece, int unc);   // 0 = the launchers' own rule
size_t ece_workspace_bytes(size_t n_per_volume, int n_volumes);
hipError_t launch_ece_hist(const float* p, const uint8_t* target, const uint8_t* mask, size_t n_per_volume,
                           int n_volumes, const float* thr_host, int n_bins, EceResult* result_dev, void* workspace,
                           hipStream_t stream);
hipError_t launch_bin_ids(const float* p, size_t n, const float* thr_host, int n_bins, uint8_t* ids,
                          hipStream_t stream);
size_t unc_workspace_bytes(size_t n_per_volume, int n_volumes);
// out_dev: [n_volumes][n_thr][8] u64 = tp, tn, fp, fn, tpu, tnu, fpu, fnu
hipError_t launch_unc_counts(const void* unc, int unc_is_f64, const uint8_t* prediction, const uint8_t* target,
                             const uint8_t* mask, size_t n_per_volume, int n_volumes, const double* thr_host,
                             int n_thr, unsigned long long* out_dev, void* workspace, hipStream_t stream);
hipError_t launch_norm_entropy(const float* p_fg, size_t n, double* out_f64, float* out_f32, hipStream_t stream);
// the same counts straight from the float32 foreground-probability map, "uncertain" decided by the table of the reference's own sets
int unc_from_p_num_thresholds();
double unc_from_p_threshold(int i);
bool unc_from_p_supported(const double* thr, int n_thr);
int unc_from_p_exceeded_host(float p, const double* thr, int n_thr);
size_t unc_from_p_workspace_bytes(size_t n_per_volume, int n_volumes);
hipError_t launch_unc_counts_from_p(const float* p_fg, const uint8_t* prediction, const uint8_t* target, const uint8_t* mask, size_t n_per_volume,
                                    int n_volumes, const double* thr_host, int n_thr, unsigned long long* out_dev, void* workspace,
                                    hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// PostNet: fused 1x1-conv stack on the U-Net feature map (rcu_postnet.hip)
// ---------------------------------------------------------------------------------------------
// packed layer of CB x CB blocks of 32 channels: weight tiles + accumulator start values + constants behind the dropout factor
constexpr int pn_layer_floats(int cb) { return cb * cb * 1024 + 2 * cb * 32; }
constexpr int PN_MAX_LAYERS = 12;
constexpr int PN_MAX_BLOCKS = 3;   // up to 96 feature channels (the packed layers live in LDS: n_layers * pn_layer_floats <= 160 KB)
// masks: null, or Dropout2d factors [n_layers - 1][images][channels] of an MC pass
hipError_t launch_postnet(const float* x_nhwc, int channel_pitch, size_t nvox, int hw, const float* packed, int n_layers,
                          int nb_classes, int channels, const float* masks, float* logits_nchw, hipStream_t stream);

}  // namespace rcu
