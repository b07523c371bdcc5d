/* rcu.h -- C ABI of librcu_hip.so: the MI355X (gfx950) implementation of the repeated-stochastic-
 * inference uncertainty path of alainjungo/reliability-challenges-uncertainty.
 *
 * The reference is pure Python and has no FFI of its own; the path sits behind three Python call
 * protocols.  Each group of entry points below names the reference
 * interface it stands behind (paths relative to the reference root):
 *
 *   model seam   context.model(images) -> logits | (logits, sigma)
 *                common/model/unet.py:123-186, called from rechun/dl/customsteps.py:23,32,
 *                common/trainloop/steps.py:84, bin-dl/brats_test_ensemble.py:85,89,
 *                bin-dl/brats_test_aleatoric.py:63
 *   step seam    BatchStep.__call__(batch_context, task_context, context)
 *                common/trainloop/steps.py:14-17; McPredictStep / MultiPredictionSummary
 *                rechun/dl/customsteps.py:10-71
 *   metric seam  EvaluationStrategy.__call__(to_evaluate, results)
 *                common/evalutation/eval.py:9-16; numpy kernels common/evalutation/numpyfunctions.py:6-107
 *
 * Conventions: every function returns 0 on success and a negative rcu_status otherwise and never
 * throws; rcu_last_error() gives the message of the calling thread's last failure.  Pointers
 * named *_dev are device pointers (e.g. torch tensor.data_ptr()), *_host are host pointers.
 * `stream` is a hipStream_t passed as void* (0 = the null stream); all device work is enqueued on
 * it and no call synchronises unless documented.  A handle is not thread-safe; distinct handles
 * are independent.  Tensors at the boundary use the reference's layout: float32, NCHW, contiguous.
 */
#ifndef RCU_H
#define RCU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rcu_status {
    RCU_OK = 0,
    RCU_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    RCU_ERR_HIP = -2,       /* a HIP runtime call failed */
    RCU_ERR_WEIGHTS = -3,   /* missing / mis-sized weight tensor */
    RCU_ERR_STATE = -4      /* call order violated (e.g. forward before finalize_weights) */
} rcu_status;

const char* rcu_last_error(void);
/* "librcu_hip <version> gfx950" */
const char* rcu_version(void);

/* ------------------------------------------------------------------------------------------
 * Model seam: UNet(nb_classes, in_channels, depth, start_filters, dropout, dropout_center,
 *                  residual=False, sigma_out, provide_features=False, bn)   (unet.py:128-130)
 * ------------------------------------------------------------------------------------------ */
typedef struct rcu_unet rcu_unet;

typedef struct rcu_unet_desc {
    int32_t nb_classes;      /* 1..8 */
    int32_t in_channels;
    int32_t depth;           /* number of down / up levels */
    int32_t start_filters;
    int32_t has_dropout;     /* 0 <=> dropout=None: the model has no Dropout2d modules at all */
    int32_t dropout_center;  /* -1 <=> None (dropout in every conv unit), else unet.py:74-82 */
    int32_t sigma_out;       /* 1: twin head, forward returns (logits, sigma) */
    int32_t bn;              /* 1: BatchNorm2d in every conv unit (folded, eval mode) */
    int32_t height, width;   /* per-slice size, each >= 2^depth; sizes not divisible by 2^depth take the reference's centre pad
                                (common/model/unet.py:110-116) and the direct kernels on the levels with odd sizes */
    int32_t max_batch;       /* largest N a forward call may pass; sizes the workspace */
    int32_t residual;        /* 1: ConvResidualBlock (common/model/unet.py:42-60) instead of ConvBlock: a block's second unit has no
                                ReLU and a 1x1 conv of the block input ("<block>.residual") is added to its output */
    int32_t provide_features; /* 1: rcu_unet_features will be called (unet.py:135-136, 178-179): the input of conv_cls.0 is kept
                                channels-last; otherwise the library is free to hold it in its channel-blocked layout */
} rcu_unet_desc;

int rcu_unet_create(const rcu_unet_desc* desc, rcu_unet** out);
int rcu_unet_destroy(rcu_unet* h);
/* bytes of device memory the handle OWNS: its packed weights, plus the activation workspace unless that is borrowed from a donor
 * (rcu_unet_create_with) */
int64_t rcu_unet_workspace_bytes(const rcu_unet* h);

/* Plan options: which kernel family / activation layout the planner may choose.  The defaults are the shipped path; the other
 * values exist for A/B measurements and for the parity tests that compare the kernel families on the same input (they replace the
 * RCU_CONV_WINO / RCU_CONV_WINO4 / RCU_CONV_FIRST / RCU_ACT_LAYOUT / RCU_FUSE_HEAD environment switches of earlier rounds: the library
 * reads no environment variable on the product path). */
typedef struct rcu_unet_options {
    int32_t conv_winograd;    /* 1 (default): Winograd kernels wherever their tiles fit; 0: the direct kernels everywhere */
    int32_t conv_winograd4;   /* 1 (default): F(4x4,3x3) wherever its tiles fit -- at the 12x8 level where, for the batch the plan is sized for, its few long
                                 work items fill the chip's rounds of workgroups (pick_config) --; 2: wherever its tiles fit, whatever the fill (parity
                                 tests on small batches); 0: F(2x2,3x3) only; 3: F(4x4,3x3) for the units with >= 64 output channels only (the round-2 selection) */
    int32_t conv_first;       /* 1 (default): the unpadded first-unit kernel; 0: the tiled kernel + channels-last input copy */
    int32_t act_layout;       /* 0 (default): channel-blocked activations between Winograd kernels; 1: channels-last everywhere */
    int32_t fuse_head;        /* 1 (default): 1x1 classifier + softmax + statistics in conv_cls.0's epilogue where the shapes allow;
                                 0: the standalone head kernel (also settable per handle at run time: rcu_unet_set_fuse_head) */
    int32_t head_winograd4;   /* 1 (default): conv_cls.0 -- with the classifier fused into its epilogue or not -- takes F(4x4,3x3) where the 32x32 tile
                                 fits (and conv_winograd4 is 1 or 2); 0: it stays on F(2x2,3x3), the plan of rounds 1-4 (A/B measurements) */
    int32_t pad_levels;       /* 1 (default): a level whose real extent (height >> l) x (width >> l) is not a whole number of Winograd tiles is ALLOCATED with
                                 a padded extent -- the padding holds zeros no kernel writes, which is the conv's own zero padding -- and runs on the Winograd
                                 kernels (the reference's BraTS slices are 240 x 240: levels 240, 120, 60, 30, 15; ISIC's 192 x 256 ends in a 12 x 16 level);
                                 0: real extents only, such levels take the direct kernels (the plans of rounds 1-5; A/B measurements) */
    int32_t reserved[1];      /* must be 0 */
} rcu_unet_options;
/* fills *opts with the defaults above */
void rcu_unet_default_options(rcu_unet_options* opts);
/* rcu_unet_create with explicit options (NULL = defaults) and an optional workspace DONOR: a handle of the same shape (desc and options
 * equal, max_batch <= the donor's) whose activation workspace the new handle shares instead of allocating its own -- the members of
 * an ensemble (bin-dl/brats_test_ensemble.py:44-57: K models resident at once) differ in 35 MB of packed weights, not in their
 * 6 GB of activations.  Handles that share a workspace must not run concurrently (launch them on ONE stream); the workspace is
 * freed when its last user is destroyed, in any order. */
int rcu_unet_create_with(const rcu_unet_desc* desc, const rcu_unet_options* opts, rcu_unet* workspace_donor, rcu_unet** out);
/* The plan alone -- which kernel runs which layer on which grid (rcu_unet_num_layers / rcu_unet_layer_info) -- without any device memory: a handle
 * that can be inspected and destroyed, nothing else (weights and forwards fail with RCU_ERR_STATE / RCU_ERR_HIP).  What a caller sizes its launches
 * by, and what the planner's tests read without a GPU. */
int rcu_unet_plan(const rcu_unet_desc* desc, const rcu_unet_options* opts, rcu_unet** out);
/* Run-time form of rcu_unet_options.fuse_head (benchmarks time the standalone head kernel on the plan of the timed run). */
int rcu_unet_set_fuse_head(rcu_unet* h, int on);

/* Dropout sites in execution order (= torch named_modules order of the Dropout2d modules). */
int rcu_unet_num_dropout_sites(const rcu_unet* h);
int rcu_unet_dropout_site_channels(const rcu_unet* h, int site);
/* state_dict-style name of the site, e.g. "down_convs.0.block.block.0.conv2d_batch_relu.dropout" */
const char* rcu_unet_dropout_site_name(const rcu_unet* h, int site);
/* sum of the sites' channel counts = floats of mask per sample and pass */
int rcu_unet_mask_floats_per_sample(const rcu_unet* h);

/* Weights: one call per state_dict tensor (torch.load(checkpoint)['state_dict'],
 * common/model/management.py:56-64), by its key (a leading "module." is ignored,
 * common/trainloop/context.py:167).  float32 host data in torch layout (conv: [Cout][Cin][kh][kw]).
 * Keys the path does not need (num_batches_tracked) are accepted and ignored. */
int rcu_unet_load_weight(rcu_unet* h, const char* name, const float* data_host, size_t count);
/* Folds BatchNorm (A = gamma / sqrt(var + 1e-5), B = beta - A * mean), repacks every conv into the
 * kernel layout and uploads.  Synchronous.  Fails with RCU_ERR_WEIGHTS naming the first missing key. */
int rcu_unet_finalize_weights(rcu_unet* h);

/* One forward pass of n <= max_batch slices.
 *   x_dev       [n][in_channels][H][W]
 *   masks_dev   NULL = eval mode (set_dropout_mode(model, False), common/utils/torchhelper.py:44-50);
 *               else the Dropout2d factors {0, 1/(1-p)} of this pass, sites concatenated:
 *               [site 0: n x C_0][site 1: n x C_1]...  (n * mask_floats_per_sample floats)
 *   logits_dev  [n][nb_classes][H][W] or NULL
 *   sigma_dev   [n][nb_classes][H][W] raw sigma head output, or NULL (must be NULL without sigma_out)
 */
int rcu_unet_forward(rcu_unet* h, const float* x_dev, int n, const float* masks_dev, float* logits_dev,
                     float* sigma_dev, void* stream);

/* Forward + softmax + accumulation into MC statistics, fused so that neither logits nor the
 * probability volume reach HBM (one pass of McPredictStep's loop body, customsteps.py:30-34, or one
 * ensemble member, brats_test_ensemble.py:85-92).  stats_dev / flags as for rcu_mc_accumulate. */
int rcu_unet_forward_accumulate(rcu_unet* h, const float* x_dev, int n, const float* masks_dev, void* stats_dev,
                                int flags, void* stream);
/* The same for `passes` stochastic passes at once: the n images run as one batch of n * passes samples (sample
 * t*n + i = image i under mask rows [site][t*n + i][C_site], masks_dev holds n * passes rows per site) and all passes
 * are added to the n statistics entries in pass order -- bit-identical to `passes` calls of
 * rcu_unet_forward_accumulate, but a small batch (the reference's batch_size 32, customsteps.py:30-34) fills the GPU.
 * n * passes <= max_batch. */
int rcu_unet_forward_accumulate_passes(rcu_unet* h, const float* x_dev, int n, int passes, const float* masks_dev,
                                       void* stats_dev, int flags, void* stream);

/* EXTENSION (BASELINE.json config "BraTS aleatoric + MC"; the reference has no such path: its McPredictStep cannot take the
 * (logits, sigma) tuple of a sigma_out model, customsteps.py:32-33, and bin-dl/brats_test_aleatoric.py:57-73 does ONE forward).
 * One stochastic pass of a sigma_out model: softmax(logits) goes into the MC statistics exactly as in
 * rcu_unet_forward_accumulate, and the pass's sigma = |raw| (exp(raw) with is_log_sigma, as brats_test_aleatoric.py:66-69)
 * is ADDED to sigma_sum_dev [n][nb_classes][H][W] (float32; zero it before the first pass, divide by T after the last). */
int rcu_unet_forward_accumulate_sigma(rcu_unet* h, const float* x_dev, int n, const float* masks_dev, void* stats_dev,
                                      int flags, float* sigma_sum_dev, int is_log_sigma, void* stream);

/* The same for a pass group (see rcu_unet_forward_accumulate_passes): `passes` stochastic passes of the n images as one batch of
 * n * passes samples; statistics and sigma sums are added to in pass order -- the bits of `passes` calls of
 * rcu_unet_forward_accumulate_sigma. */
int rcu_unet_forward_accumulate_sigma_passes(rcu_unet* h, const float* x_dev, int n, int passes, const float* masks_dev,
                                             void* stats_dev, int flags, float* sigma_sum_dev, int is_log_sigma, void* stream);

/* Per-layer introspection for benchmarks: canonical FLOPs (2*Cin*Cout*9*H*W per slice, real
 * channel counts) and the kernel configuration chosen. */
typedef struct rcu_layer_info {
    char name[96];
    char kernel[64];
    int32_t cin, cout, height, width;
    int32_t grid_height, grid_width; /* the grid the kernel's tiles walk: the ALLOCATED extent of the layer's level (an up-convolution's low-resolution
                                        level), larger than the real one on a padded level (rcu_unet_options.pad_levels) */
    int32_t upsample, pooled, dual_source;
    int32_t head_fusable;         /* 1: forwards that want logits or statistics (no sigma) run this unit with the classifier head in the kernel's epilogue
                                     while rcu_unet_options.fuse_head is on -- profilers see that kernel as `kernel` + "+head"; rcu_unet_run_layer runs the plain one */
    double flops_per_slice;       /* algorithmic: 2*cin*cout*9*H*W of the layer as the reference computes it */
    double mfma_flops_per_slice;  /* what the kernel issues to the MFMA pipe (padded channels; 4 of 9 taps for the
                                     sub-pixel up-convolution; whole tiles) */
} rcu_layer_info;
int rcu_unet_num_layers(const rcu_unet* h);
int rcu_unet_layer_info(const rcu_unet* h, int layer, rcu_layer_info* out);
/* Runs only conv layer `layer` on the handle's current workspace contents (benchmark aid; layer 0 then reads the
   workspace's channels-last input copy, which a forward pass only fills when its first-layer kernel does not read the
   caller's NCHW input directly -- timing only). */
int rcu_unet_run_layer(rcu_unet* h, int layer, int n, const float* masks_dev, void* stream);

/* Per-kernel timing of the next `max_forwards` forward calls with HIP events recorded on the caller's
 * stream between launches (slot 0 = input re-layout, slots 1..L = conv layers in rcu_unet_layer_info
 * order, slot L+1 = head kernel).  collect() waits for the recorded work, writes the summed
 * milliseconds per slot (L+2 doubles) and the number of forwards covered, and re-arms the pool. */
int rcu_unet_profile_begin(rcu_unet* h, int max_forwards);
int rcu_unet_profile_collect(rcu_unet* h, double* ms_sum, int* forwards);

/* Feature map of the last forward call = the input of conv_cls (UNet.features when provide_features is set,
 * common/model/unet.py:178-179): NHWC float32 in the handle's workspace, `channels` real channels at a pitch
 * of `channel_pitch` floats per voxel; valid until the next forward call on this handle. */
int rcu_unet_features(const rcu_unet* h, const float** features_dev, int* channels, int* channel_pitch);

/* ------------------------------------------------------------------------------------------
 * PostNet -- auxiliary confidence network on the U-Net features
 *   (common/model/postnet.py:6-18; bin-dl/brats_test_auxiliary_feat.py:74-77, isic_test_auxiliary_feat.py)
 *   nb_convs x [Conv2d 1x1 C->C, BatchNorm2d (eval), ReLU] + Conv2d 1x1 C->nb_classes, C <= 32, nb_classes <= 32.
 *   Weights by state_dict name: convs.<i>.conv2d_batch_relu.{conv.weight,conv.bias,bn.weight,bn.bias,
 *   bn.running_mean,bn.running_var}, conv_logits.{weight,bias}.
 * ------------------------------------------------------------------------------------------ */
typedef struct rcu_postnet rcu_postnet;
int rcu_postnet_create(int in_channels, int nb_classes, int nb_convs, int bn, rcu_postnet** out);
void rcu_postnet_destroy(rcu_postnet* h);
int rcu_postnet_load_weight(rcu_postnet* h, const char* name, const float* host_data, size_t count);
int rcu_postnet_finalize_weights(rcu_postnet* h);
/* in_channels <= 96.  features: NHWC float32 [n*hw][channel_pitch] (channel_pitch >= in_channels rounded up to 32, a multiple of 4;
 * the padding channels up to the next multiple of 32 must hold zeros or finite values -- their weights are zero);
 * masks_dev: NULL (eval / no Dropout2d) or the factors {0, 1/(1-p)} of one MC pass, float32 [nb_convs][n][in_channels]
 * (Dropout2d sits between conv and BatchNorm in every hidden unit, common/model/unet.py:14-15); logits_dev: float32 [n][nb_classes][hw]. */
int rcu_postnet_forward(rcu_postnet* h, const float* features_dev, int channel_pitch, int n, int hw, const float* masks_dev,
                        float* logits_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Step seam: per-voxel sufficient statistics over T passes / K members
 *   (McPredictStep + MultiPredictionSummary, customsteps.py:16-71; torchhelper.py:53-54)
 * ------------------------------------------------------------------------------------------ */
#define RCU_MC_MI 1           /* also track sum_t H(p_t)  -> mutual_info available */
#define RCU_MC_VAR 2          /* statistics in float64 incl. sum p^2 -> variance available */
#define RCU_MC_INPUT_PROBS 4  /* rcu_mc_accumulate input is already softmax-ed */
#define RCU_MC_EXACT 8        /* statistics in float64, every addend rounded to a multiple of 2^-40 first: all additions are exact */
#define RCU_MC_EXACT_MAX_PASSES 2048

/* Size of the statistics blob for n*hw voxels.  Layout: planes over voxel v = n_idx*hw + pix;
 * with neither RCU_MC_VAR nor RCU_MC_EXACT float32 planes [sum p_c (C)] [sum H (if MI)]; with either, float64 planes
 * [sum p_c (C)] [sum p_c^2 (C) (if VAR)] [sum H (if MI)].  The blob is plain additive: partial blobs of
 * disjoint pass subsets are merged by element-wise addition (one RCCL sum-reduce).
 * RCU_MC_EXACT (what the predict steps of the product use): an addend x (a pass's p_c, p_c^2 or entropy, all <= log 8) enters as
 * (x + 6144.0) - 6144.0 in float64, i.e. rounded to the nearest multiple of 2^-40; sums of such multiples below 2^13 are exactly
 * representable, so every addition -- in a kernel, between stream lanes, in the collective -- is exact and the merged statistics of
 * up to RCU_MC_EXACT_MAX_PASSES passes do not depend on the order, the pass groups, the lanes, the number of ranks or the
 * collective's reduction tree: a run on 8 GPUs writes the bytes a run on one GPU writes.  Cost against the reference's float32
 * mean: |delta p| <= 2^-41 per pass on probabilities below 2^-17. */
size_t rcu_mc_stats_bytes(size_t n, size_t hw, int nb_classes, int flags);
int rcu_mc_begin(void* stats_dev, size_t n, size_t hw, int nb_classes, int flags, void* stream);
/* in_dev: [n][C][hw] logits (softmax applied here) or probabilities (RCU_MC_INPUT_PROBS). */
int rcu_mc_accumulate(const float* in_dev, void* stats_dev, size_t n, size_t hw, int nb_classes, int flags,
                      void* stream);
/* T = number of accumulated passes.  Outputs (any may be NULL): mean [n][C][hw]; entropy, mutual_info,
 * variance [n][1][hw].  mutual_info needs RCU_MC_MI, variance needs RCU_MC_VAR. */
int rcu_mc_finalize(const void* stats_dev, size_t n, size_t hw, int nb_classes, int T, int flags, float* mean_dev,
                    float* entropy_dev, float* mutual_info_dev, float* variance_dev, void* stream);

/* F.softmax(logits, 1) (customsteps.py:24; steps.py:88) */
int rcu_softmax(const float* logits_dev, float* probs_dev, size_t n, size_t hw, int nb_classes, void* stream);
/* AleatoricPredictStep (bin-dl/brats_test_aleatoric.py:63-73) plus the writer's selection of the
 * predicted class' sigma (same file :95-97).  Outputs may be NULL. */
int rcu_aleatoric(const float* logits_dev, const float* sigma_raw_dev, size_t n, size_t hw, int nb_classes,
                  int is_log_sigma, float* probs_dev, float* sigma_dev, uint8_t* prediction_dev,
                  float* sigma_pred_dev, void* stream);
/* argmax over classes (first maximum) and the foreground-class probability map, as written to
 * *_prediction / *_probabilities.nii.gz (bin-dl/brats_test_default.py:96-99). */
int rcu_prediction_and_foreground(const float* probs_dev, size_t n, size_t hw, int nb_classes,
                                  uint8_t* prediction_dev, float* p_foreground_dev, void* stream);

/* Dropout2d factors of the MC passes of a launch, drawn on the device in one kernel: the `masks_dev` argument of
 * rcu_unet_forward_accumulate_passes (passes == 1: of rcu_unet_forward / rcu_unet_forward_accumulate) for seeded passes.
 * seeds_host[t]: the seed of pass t (the predict steps use job_seed(YAML seed, pass)); first_sample: the GLOBAL index of the batch's sample 0 --
 * its position in the run's stream of slices / images, whatever batches the loader cuts that stream into; site_channels_host / site_keep_host:
 * channels and 1 - p of the n_sites Dropout2d sites in execution order (site_keep < 0: the site is not active -- factor 1; 0: p = 1 -- factor 0).
 * out_dev: float32 [site][passes * n + i][C_site], sample t * n + i = image i in pass t.  The factor of (pass t, image i, site s, channel c) is
 * 1 / keep where the 24-bit uniform from word e & 3 of Philox4x32-10(key = seeds[t], counter = e >> 2) is below keep, else 0, with
 * e = (first_sample + i) * sum_s C_s + (C_0 + ... + C_{s-1}) + c: Bernoulli(1 - p) / (1 - p), the law of torch's Dropout2d
 * (common/model/unet.py:16), and a function of (seed, global sample index, site, channel) alone -- a slice's MC sample does not depend on the
 * batch_size it is loaded with (round 6; rounds 1-5 keyed the draw by the batch and the position in it), nor on the group, lane or rank the
 * pass is launched in. */
int rcu_dropout_masks(const uint64_t* seeds_host, int passes, int n, uint64_t first_sample, const int32_t* site_channels_host,
                      const float* site_keep_host, int n_sites, float* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Metric seam: calibration histograms (numpyfunctions.py:6-107)
 * ------------------------------------------------------------------------------------------ */
#define RCU_MAX_BINS 32
#define RCU_MAX_THRESHOLDS 16

typedef struct rcu_ece_result {           /* raw histogram of one volume, all RCU_MAX_BINS slots */
    uint64_t count[RCU_MAX_BINS];
    double sum_conf[RCU_MAX_BINS];
    uint64_t sum_pos[RCU_MAX_BINS];
} rcu_ece_result;

/* float32 thresholds t_k (k = 1..n_bins-1) with  sum_k [p >= t_k] == np.digitize(p, linspace(0,
 * 1+1e-8, n_bins+1)) - 1  for every float32 p in [0, 1]  (numpyfunctions.py:53-54). */
int rcu_ece_thresholds(int n_bins, float* thr_host);
size_t rcu_ece_workspace_bytes(size_t n_per_volume, int n_volumes);
/* Reliability histogram of n_volumes independent volumes of n_per_volume voxels each (volume v at
 * offset v * n_per_volume in every array).  mask_dev NULL = all voxels (ISIC), else voxels with
 * mask != 0 (BraTS brain mask, rechun/eval/analysis.py:118-125).  result_dev: n_volumes results. */
int rcu_ece_hist(const float* p_dev, const uint8_t* target_dev, const uint8_t* mask_dev, size_t n_per_volume,
                 int n_volumes, const float* thr_host, int n_bins, rcu_ece_result* result_dev, void* workspace_dev,
                 void* stream);
/* raw bin index per voxel (test aid: pins the binning bit-for-bit) */
int rcu_ece_bin_ids(const float* p_dev, size_t n, const float* thr_host, int n_bins, uint8_t* ids_dev, void* stream);

/* Test / tuning aid: consecutive 16,384-voxel blocks a workgroup of the histogram (ece) and of the count kernel (unc) takes;
 * 0 = the launcher's choice (default).  Integer sums: every value gives the same results.  Process-wide. */
int rcu_calib_set_blocks_per_workgroup(int ece_blocks, int unc_blocks);

size_t rcu_unc_workspace_bytes(size_t n_per_volume, int n_volumes);
/* counts_dev: [n_volumes][n_thr][8] uint64 = tp, tn, fp, fn, tpu, tnu, fpu, fnu with
 * "uncertain" := uncertainty > thr (compared in float64), numpyfunctions.py:86-107 for every
 * threshold of bin-eval/eval_uncertainty.py:239 in one pass.  unc_dev is float64 (unc_is_f64=1, what
 * ToEntropy yields) or float32. */
int rcu_unc_counts(const void* unc_dev, int unc_is_f64, const uint8_t* prediction_dev, const uint8_t* target_dev,
                   const uint8_t* mask_dev, size_t n_per_volume, int n_volumes, const double* thr_host, int n_thr,
                   uint64_t* counts_dev, void* workspace_dev, void* stream);
/* The same counts straight from the float32 foreground-probability map p, for the evaluation of the 'probabilities' confidence entry
 * (bin-eval/eval_uncertainty.py:176-202 on analysis.py:249-252: uncertainty = ToEntropy([1 - p, p])): that uncertainty is a function of the
 * float32 p alone, so {p : uncertainty(p) > thr} is a set of float32 values -- found for the script's 11 thresholds (eval_uncertainty.py:239)
 * by running the REFERENCE over every float32 in [0, 1] (tests/golden/generate_ue_boundaries.py, fixture g20, compiled in as
 * csrc/rcu_ue_table.inc): an interval of bit patterns per threshold with a ragged window of 0-3 values at either end.  The kernel looks p
 * up in that table: the counts equal the reference's integer for integer (no device log to disagree with numpy's in the last ulp) and the
 * 8-byte-per-voxel entropy map is never made (6 bytes per voxel instead of 7 + 12 for making the map).
 * thr_host: strictly ascending, every value one of rcu_unc_from_p_threshold(0 .. rcu_unc_from_p_num_thresholds() - 1), else RCU_ERR_INVALID
 * (rcu_unc_from_p_supported tells beforehand; other thresholds take rcu_normalised_entropy + rcu_unc_counts).  The table (4.4 KB) is
 * copied to the workspace with every call, stream-ordered.
 * "The reference's sets" are those of the build that made the table: numpy 2.2.6's float32 log on a CPU with AVX512 (the fixture records
 * numpy_version / cpu_features); a numpy that rounds log differently in the last ulp (another SIMD path) can disagree on the one to three
 * float32 values right at a boundary.  tests/test_oracle_golden.py::test_uncertain_voxel_table_holds_under_the_local_numpy re-evaluates
 * every probe value on the host it runs on, and bench.py re-checks the counts of its timed output on the GPU box's host (parity.ue_counts_equal);
 * where they differ, rcu_normalised_entropy + rcu_unc_counts (the map-based path) is the reference-of-that-host's arithmetic. */
int rcu_unc_from_p_num_thresholds(void);
double rcu_unc_from_p_threshold(int i);
int rcu_unc_from_p_supported(const double* thr_host, int n_thr);
/* number of the thresholds the probability p exceeds per the table (host side: tests pin the table against the fixture without a GPU); -1 = unsupported thresholds */
int rcu_unc_from_p_exceeded(float p, const double* thr_host, int n_thr);
size_t rcu_unc_from_p_workspace_bytes(size_t n_per_volume, int n_volumes);
int rcu_unc_counts_from_p(const float* p_foreground_dev, const uint8_t* prediction_dev, const uint8_t* target_dev, const uint8_t* mask_dev,
                          size_t n_per_volume, int n_volumes, const double* thr_host, int n_thr, uint64_t* counts_dev, void* workspace_dev,
                          void* stream);
/* ToEntropy (rechun/eval/analysis.py:196-203) on a foreground-probability map: float32 products,
 * float64 sum, / log 2.  Either output may be NULL. */
int rcu_normalised_entropy(const float* p_foreground_dev, size_t n, double* out_f64_dev, float* out_f32_dev,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RCU_H */
