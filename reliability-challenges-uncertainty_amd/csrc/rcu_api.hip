// C ABI of librcu_hip.so (see include/rcu.h): U-Net handle (layer plan, BN folding, weight
// packing, workspace) and thin wrappers around the kernel launchers.
#include "../../include/rcu.h"
#include "rcu_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace rcu;

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}
static int hip_fail(hipError_t e, const char* what)
{
    return fail(RCU_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define RCU_HIP(call)                                      \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return hip_fail(e_, #call);  \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
struct ConvLayer {
    std::string name;        // state_dict prefix of the conv ("....conv" for units, "...upconv.1" for up convs)
    std::string bn;          // prefix of the BatchNorm2d, empty when the conv has none
    std::string name2, bn2;  // fused twin (conv_sigma.0) stacked on the output channels, or empty
    int cin1 = 0, cin2 = 0;  // real channels per source
    int c1p = 0, c2p = 0;    // padded channels per source
    int cout = 0;            // real output channels (per twin)
    int coutp = 0;           // padded output channels (both twins)
    int csplit = 0;          // first channel of the twin
    int H = 0, W = 0;        // REAL extent of the unit's output grid (an up-convolution: of the up-sampled grid before the centre pad)
    int level = 0;           // level of the tile grid: the unit's own level, an up-convolution's LOW-resolution level
    int gh = 0, gw = 0;      // ALLOCATED extent of that level = the grid the tiles walk (== its real extent unless the level is padded)
    int upsample = 0, relu = 0;
    int is_1x1 = 0;          // ConvResidualBlock's residual conv: a 1x1 kernel, run as a 3x3 unit whose centre tap alone is non-zero
    int accumulate = 0;      // the unit's result is added to what its output tensor holds (the residual conv's output)
    int site = -1, site2 = -1;
    int cfg = 0, NT = 0;
    int t_src1 = -1, t_src2 = -1, t_out = -1, t_pool = -1;
    float *wpack = nullptr, *alpha = nullptr, *betab = nullptr, *beta = nullptr;
    size_t wpack_floats = 0;
};

struct Tensor {
    size_t floats_per_slice = 0;
    float* dev = nullptr;
    bool zero_fill = false;   // centre-padded up-conv output: the border is never written and must read as zero
    int H = 0, W = 0, cp = 0; // ALLOCATED extent (floats_per_slice = H W cp)
    int Hr = 0, Wr = 0;       // real extent; smaller on a padded level (choose_level_extents): the pixels beyond it hold zeros nobody writes
    bool padded() const { return H != Hr || W != Wr; }
    bool blocked = false;     // [N][C/8][H][W][8] instead of NHWC (rcu_kernels.h, ConvArgs): decided per tensor by assign_layouts
};

// The activation tensors of a plan: owned by the handle that allocated them, shared (shared_ptr) with the handles created with
// that handle as their workspace donor (rcu_unet_create_with); freed with the last user.
struct Workspace {
    std::vector<float*> dev;      // one allocation per plan tensor
    std::vector<size_t> bytes;
    int max_batch = 0;
    ~Workspace()
    {
        for (float* p : dev)
            if (p) (void)hipFree(p);
    }
};

struct rcu_unet {
    rcu_unet_desc d{};
    rcu_unet_options opt{};
    std::shared_ptr<Workspace> ws;
    bool ws_borrowed = false;
    std::vector<ConvLayer> layers;
    std::vector<Tensor> tensors;
    std::vector<std::pair<int, int>> level_ext;       // allocated (H, W) of the activation tensors of level 0 .. depth (choose_level_extents)
    std::vector<std::pair<std::string, int>> sites;   // (name, channels)
    std::vector<int> site_offset;                     // prefix sums of channels
    int mask_floats = 0;
    int in_cp = 0, head_cp = 0, head_cph = 0;
    int t_input = -1, t_head = -1;
    int t_features = -1;      // provide_features on a padded level 0: the compact [voxel][channel] copy rcu_unet_features hands out
    std::map<std::string, std::vector<float>> host_weights;
    float *w_cls = nullptr, *b_cls = nullptr, *w_sig = nullptr, *b_sig = nullptr;
    bool finalized = false;
    bool plan_only = false;   // rcu_unet_plan: no workspace -- the handle can be inspected, never run
    int64_t workspace_bytes = 0;
    std::vector<void*> allocs;
    // optional per-layer timing with HIP events on the caller's stream (rcu_unet_profile_*)
    std::vector<hipEvent_t> prof_events;   // [max_forwards][slots], slots = layers + 3
    int prof_capacity = 0, prof_used = 0;
};

static int prof_slots(const rcu_unet* h) { return (int)h->layers.size() + 3; }

// a tensor of level `level` (allocated with the level's extent), or -- level < 0 -- one that is exactly Hr x Wr
static int new_tensor(rcu_unet* h, int level, int Hr, int Wr, int cp)
{
    Tensor t;
    t.Hr = Hr; t.Wr = Wr; t.cp = cp;
    t.H = level >= 0 ? h->level_ext[level].first : Hr;
    t.W = level >= 0 ? h->level_ext[level].second : Wr;
    t.floats_per_slice = (size_t)t.H * t.W * cp;
    h->tensors.push_back(t);
    return (int)h->tensors.size() - 1;
}

static bool unit_has_dropout(const rcu_unet_desc& d, int level, bool is_down, int i)
{
    // common/model/unet.py:63-82
    if (!d.has_dropout) return false;
    if (d.dropout_center < 0) return true;                 // 'all'
    if (level == d.depth) return false;                    // 'no'
    if (level + d.dropout_center >= d.depth) return is_down ? (i == 1) : (i == 0);   // 'last' / 'first'
    return false;
}

// floats of one packed [TAPS][BN][KC+4] weight tile, rounded up to 256 threads x float4
static size_t conv_tile_floats(const ConvConfigInfo& ci)
{
    if (ci.WINO) return (size_t)ci.TAPS * ci.BN * ci.KCP;   // the LDS image, piece by piece (LDS-DMA)
    const size_t units = (size_t)ci.TAPS * ci.BN * ci.KCP / 4;
    return (units + 255) / 256 * 256 * 4;
}

static bool is_head_unit(const ConvLayer& L) { return L.name.rfind("conv_cls.0", 0) == 0; }

// Which kernel runs a layer.  (gh, gw) = L.gh x L.gw is the grid the tiles walk: the ALLOCATED extent of the layer's level (an up-convolution's
// low-resolution level), which is the real extent unless the level is padded (choose_level_extents); oh x ow the allocated extent of its output tensor.
static int pick_config(const ConvLayer& L, int n_slices, const rcu_unet_options& opt, int oh, int ow)
{
    // Winograd kernels (rcu_wino.hip, rcu_wino_up.hip): 16/36 (conv units) and 9/36 (up-convolutions) of the
    // multiplications.  They address activations through 32-bit byte offsets of a buffer resource (tensors < 2 GB)
    // and take whole tiles only.  Tiles that span two or eight slices are taken whatever the batch the plan is sized for (a last
    // group of slices that is not full reads zeros and stores nothing): the kernel of a layer -- and with it the bits of a slice's
    // result -- does not depend on the batch size, as long as every tensor stays below 2 GB.
    // rcu_unet_options.conv_winograd = 0 keeps every layer on the direct kernels of rcu_conv.hip (A/B tests)
    // a unit that adds to its output tensor (ConvResidualBlock's second unit) runs on the direct kernels, whose epilogue can
    const bool wino_on = opt.conv_winograd != 0 && !L.accumulate;
    const int gh = L.gh, gw = L.gw;
    // (an up-convolution reads the LOW-resolution grid: L.H x L.W is its output grid)
    const size_t in_px = (size_t)gh * gw;
    const size_t max_bytes = (size_t)n_slices * std::max(in_px * (size_t)std::max(L.c1p, L.c2p), (size_t)oh * ow * (size_t)L.coutp) * 4;
    const bool center_pad = L.upsample && (2 * (L.H / 2) != L.H || 2 * (L.W / 2) != L.W);   // unet.py:110-116: direct kernel only
    if (L.upsample && !center_pad && wino_on && L.c1p % 32 == 0 && L.c2p == 0 && max_bytes < ((size_t)1 << 31)) {
        const int lh = gh, lw = gw;
        if (L.coutp == 32 && lh % 16 == 0 && lw % 32 == 0) return CONV_CFG_UPW_T16x32_N32;
        if (L.coutp > 32 && lh % 16 == 0 && lw % 16 == 0) return CONV_CFG_UPW_T16x16_N64;
        if (L.coutp > 32 && lh % 8 == 0 && lw % 16 == 0) return CONV_CFG_UPW_S2T8x16_N64;
        if (L.coutp > 32 && lh % 4 == 0 && lw == 8) return CONV_CFG_UPW_S8T4x8_N64;
    }
    if (L.upsample) {   // sub-pixel form; tiles run over the low-res input grid
        if (L.coutp <= 32) return (gh % 16 == 0 && gw % 16 == 0) ? CONV_CFG_UP_T16x16_N32 : CONV_CFG_UP_T8x16_N32;
        if (gh == 12 && gw == 8) return CONV_CFG_UP_S2T12x8_N64;
        return CONV_CFG_UP_T8x16_N64;
    }
    if (L.c1p + L.c2p == 8) {
        // the network's first conv unit: unpadded K = 9 x 4 (8) on whole 8x32 tiles (rcu_first.hip); rcu_unet_options.conv_first = 0
        // keeps the tiled kernel (A/B tests)
        const bool first_on = opt.conv_first != 0;
        if (first_on && L.c2p == 0 && gh % 8 == 0 && gw % 32 == 0 && (L.coutp == 32 || L.coutp == 64) && L.t_pool < 0 &&
            L.name2.empty() && !L.accumulate)
            return CONV_CFG_FIRST_T8x32;
        return CONV_CFG_T8x16_N32_FIRST;
    }
    if (wino_on && (L.c1p + L.c2p) % 32 == 0 && (L.c2p == 0 || L.c2p == L.c1p) && max_bytes < ((size_t)1 << 31)) {
        // F(4x4,3x3) (rcu_wino4.hip) where its 32-pixel-wide tiles fit: 2.25 instead of 4 multiplications per output pixel.  Measured
        // per layer on the BraTS volume (tools/wino4_check.py, tools/layer_report.py): 1.1x at 32 -> 64 channels to 1.4x at 128 -> 128 and
        // 256 -> 128 over F(2x2,3x3).  The 32-channel full-resolution layers -- four Cin chunks per tile, so a tile switch (cold fetch +
        // epilogue) per four chunks -- lost to F(2x2,3x3) while the activations were channels-last (0.39-0.69 against 0.36-0.64 ms); with
        // the channel-blocked layout, whose chunks are cold one at a time, they win too (round 3: 0.31 / 0.47 / 0.28 against
        // 0.35 / 0.60 / 0.33 ms).  rcu_unet_options.conv_winograd4 = 0 keeps F(2x2,3x3) everywhere, 3 takes F(4x4,3x3) for the layers with
        // >= 64 output channels only (the round-2 choice; A/B tests).
        const int w4_mode = opt.conv_winograd4;
        // The head unit (conv_cls.0 alone: one 32-cout tile, no sigma twin) has its classifier fused into the epilogue of either family
        // (forward_impl); the F(4x4,3x3) form exists for the 32x32 tile only, elsewhere -- and under head_winograd4 = 0, the plan of
        // rounds 1-4 -- it stays on F(2x2,3x3).  The 64-cout cls + sigma twin unit has no fused form and takes F(4x4,3x3) like any layer.
        const bool lone_head = is_head_unit(L) && L.name2.empty();
        const bool w4_ok = w4_mode != 0 && (L.coutp >= 64 || w4_mode != 3) &&
                           !(lone_head && (opt.head_winograd4 == 0 || L.coutp != 32 || gh % 32 != 0 || gw % 32 != 0));
        if (w4_ok && gw % 32 == 0) {
            if (gh % 32 == 0) return CONV_CFG_WINO4_T32x32_N32;
            if (gh % 16 == 0) return CONV_CFG_WINO4_S2T16x32_N32;
        }
        // images exactly 32 wide whose height is a multiple of 8 but not of 16 (24x32: the fourth level of the reference's ISIC size 192x256): the
        // full-width tile of four slices takes them as they are (round 6; padded to 32x32 the level would compute a third more)
        if (w4_ok && gw == 32 && gh % 8 == 0) return CONV_CFG_WINO4_S4T8x32_N32;
        if (w4_ok && gw == 16 && gh % 8 == 0) return CONV_CFG_WINO4_S8T8x16_N32;
        // the 12x8 level (bottom_convs): F(4x4,3x3) with a slice's 6 tiles in the 8 tile slots of the S8 block -- 3 multiplications per output
        // pixel executed (2.25 x 4/3) against F(2x2,3x3)'s 4 (round 5; no pooled output in this geometry)
        if (w4_ok && gw == 8 && gh % 12 == 0 && L.t_pool < 0) {
            // ... where its few, long work items fill the chip's rounds: one item = 8 slices x 32 couts against F(2x2,3x3)'s 8 slices x a 4x8 strip x
            // 64 couts.  Measured per round of 256 workgroups (bottom_convs, tools/layer_report.py): 0.282 against 0.210 ms, so the folded form wins
            // when rounds_4 * 1.35 < rounds_2 -- 640 samples: 5 against 8 rounds (1.41 against 1.68 ms), 480: 4 against 6; but 160 samples
            // (an ensemble member's launch): 2 against 2 (0.57 against 0.44 ms), 320: 3 against 4 (a tie) -- there F(2x2,3x3) stays
            const long groups = (n_slices + 7) / 8;
            const long rounds4 = (groups * (gh / 12) * (L.coutp / 32) + 255) / 256;
            const long rounds2 = (groups * (gh / 4) * ((L.coutp + 63) / 64) + 255) / 256;
            if (w4_mode == 2 || L.coutp <= 32 || (double)rounds4 * 1.35 < (double)rounds2) return CONV_CFG_WINO4_S8T12x8_N32;   // (2: whatever the fill -- parity tests on small batches)
        }
        if (L.coutp == 32 && gh % 16 == 0 && gw % 32 == 0) return CONV_CFG_WINO_T16x32_N32;
        if (L.coutp > 32 && gh % 16 == 0 && gw % 16 == 0) return CONV_CFG_WINO_T16x16_N64;
        if (L.coutp > 32 && gh % 8 == 0 && gw % 16 == 0) return CONV_CFG_WINO_S2T8x16_N64;
        if (L.coutp > 32 && gh % 4 == 0 && gw == 8) return CONV_CFG_WINO_S8T4x8_N64;
    }
    if (L.coutp > 32) {
        if (gh == 12 && gw == 8) return CONV_CFG_S2T12x8_N64;
        // 128-pixel tiles run 3 workgroups per CU (768 slots), 256-pixel tiles 2 (512 slots) with half the staging,
        // barriers and fragment reads per MFMA.  Measured on the BraTS levels (tools/layer_report.py): the big tile
        // wins where every work item stages its own input tile (one channel tile) and where the small tile's last
        // round of work items is badly filled (24x16: 2.5 rounds); it loses 3 % at 48x32 (5 full rounds).
        const int nt = (L.coutp + 63) / 64;
        const long items = (long)((gh + 7) / 8) * ((gw + 15) / 16) * n_slices * nt;
        const double fill = (double)items / (double)((items + 767) / 768 * 768);
        if (nt == 1 || fill < 0.9) {
            if (gh % 16 == 0 && gw % 16 == 0) return CONV_CFG_T16x16_N64;
            if (gh % 8 == 0 && gw % 16 == 0 && n_slices % 2 == 0) return CONV_CFG_S2T8x16_N64;
        }
        return CONV_CFG_T8x16_N64;
    }
    return (gh % 16 == 0 && gw % 16 == 0) ? CONV_CFG_T16x16_N32 : CONV_CFG_T8x16_N32;
}

static void add_unit(rcu_unet* h, const std::string& prefix, int cin1, int c1p, int cin2, int c2p, int cout, int level, int H, int W,
                     bool dropout, int t_src1, int t_src2, int t_out, int t_pool, bool residual_tail = false)
{
    ConvLayer L;
    L.name = prefix + ".conv2d_batch_relu.conv";
    if (h->d.bn) L.bn = prefix + ".conv2d_batch_relu.bn";
    L.cin1 = cin1; L.c1p = c1p; L.cin2 = cin2; L.c2p = c2p;
    L.cout = cout; L.coutp = round_up(cout, 32); L.csplit = L.coutp;
    L.H = H; L.W = W; L.relu = residual_tail ? 0 : 1;
    L.level = level; L.gh = h->level_ext[level].first; L.gw = h->level_ext[level].second;
    L.accumulate = residual_tail ? 1 : 0;
    if (dropout) {
        L.site = (int)h->sites.size();
        h->sites.push_back({prefix + ".conv2d_batch_relu.dropout", cout});
    }
    L.t_src1 = t_src1; L.t_src2 = t_src2; L.t_out = t_out; L.t_pool = t_pool;
    h->layers.push_back(L);
}

// ConvResidualBlock (common/model/unet.py:42-60): conv1x1(block input) + bias into the block's output tensor; the block's second
// unit (no ReLU) then adds its result to it.  `prefix`: the block's module path ("down_convs.0.block", "bottom_convs", ...).
static void add_residual_conv(rcu_unet* h, const std::string& prefix, int cin1, int c1p, int cin2, int c2p, int cout, int level, int H, int W,
                              int t_src1, int t_src2, int t_out)
{
    ConvLayer L;
    L.name = prefix + ".residual";
    L.is_1x1 = 1;
    L.cin1 = cin1; L.c1p = c1p; L.cin2 = cin2; L.c2p = c2p;
    L.cout = cout; L.coutp = round_up(cout, 32); L.csplit = L.coutp;
    L.H = H; L.W = W; L.relu = 0;
    L.level = level; L.gh = h->level_ext[level].first; L.gw = h->level_ext[level].second;
    L.t_src1 = t_src1; L.t_src2 = t_src2; L.t_out = t_out;
    h->layers.push_back(L);
}

static void assign_layouts(rcu_unet* h);

// the kernels whose store side honours ConvArgs::part (a padded level): the Winograd families and rcu_first.hip
static bool cfg_handles_padding(int cfg)
{
    return cfg == CONV_CFG_FIRST_T8x32 || (cfg >= CONV_CFG_WINO_T16x16_N64 && cfg <= CONV_CFG_UPW_S8T4x8_N64) ||
           (cfg >= CONV_CFG_WINO4_T32x32_N32 && cfg < CONV_CFG_END);
}

// Builds layers and tensors for the level extents in h->level_ext (build_plan: the real extents; choose_level_extents tries padded ones).
// *valid = false when a layer that touches a padded tensor got a kernel that cannot: such a plan must not run.
static int build_plan_for(rcu_unet* h, bool* valid)
{
    const rcu_unet_desc& d = h->d;
    const int depth = d.depth;
    h->layers.clear();
    h->tensors.clear();
    h->sites.clear();
    h->in_cp = round_up(d.in_channels, 8);
    h->t_input = new_tensor(h, 0, d.height, d.width, h->in_cp);
    std::vector<int> skip(depth), skip_c(depth);
    int cur = h->t_input, cur_c = d.in_channels, cur_cp = h->in_cp;
    int c = d.start_filters;
    char buf[128];
    for (int l = 0; l < depth; ++l) {
        const int H = d.height >> l, W = d.width >> l, cp = round_up(c, 32);
        const int t_tmp = new_tensor(h, l, H, W, cp), t_skip = new_tensor(h, l, H, W, cp);
        const int t_pool = new_tensor(h, l + 1, H / 2, W / 2, cp);
        snprintf(buf, sizeof buf, "down_convs.%d.block.block.0", l);
        add_unit(h, buf, cur_c, cur_cp, 0, 0, c, l, H, W, unit_has_dropout(d, l, true, 0), cur, -1, t_tmp, -1);
        if (d.residual) {
            snprintf(buf, sizeof buf, "down_convs.%d.block", l);
            add_residual_conv(h, buf, cur_c, cur_cp, 0, 0, c, l, H, W, cur, -1, t_skip);
        }
        snprintf(buf, sizeof buf, "down_convs.%d.block.block.1", l);
        add_unit(h, buf, c, cp, 0, 0, c, l, H, W, unit_has_dropout(d, l, true, 1), t_tmp, -1, t_skip, t_pool, d.residual != 0);
        skip[l] = t_skip; skip_c[l] = c;
        cur = t_pool; cur_c = c; cur_cp = cp;
        c *= 2;
    }
    {
        const int H = d.height >> depth, W = d.width >> depth, cp = round_up(c, 32);
        const int t_tmp = new_tensor(h, depth, H, W, cp), t_out = new_tensor(h, depth, H, W, cp);
        add_unit(h, "bottom_convs.block.0", cur_c, cur_cp, 0, 0, c, depth, H, W, unit_has_dropout(d, depth, true, 0), cur, -1,
                 t_tmp, -1);
        if (d.residual) add_residual_conv(h, "bottom_convs", cur_c, cur_cp, 0, 0, c, depth, H, W, cur, -1, t_out);
        add_unit(h, "bottom_convs.block.1", c, cp, 0, 0, c, depth, H, W, unit_has_dropout(d, depth, true, 1), t_tmp, -1, t_out,
                 -1, d.residual != 0);
        cur = t_out; cur_c = c; cur_cp = cp;
    }
    for (int j = 0; j < depth; ++j) {
        const int l = depth - 1 - j;
        const int H = d.height >> l, W = d.width >> l;
        const int co = cur_c / 2, cop = round_up(co, 32);
        const int t_up = new_tensor(h, l, H, W, cop), t_tmp = new_tensor(h, l, H, W, cop), t_out = new_tensor(h, l, H, W, cop);
        ConvLayer U;   // nearest x2 + conv3x3 + bias, no BN / ReLU / dropout (unet.py:105)
        snprintf(buf, sizeof buf, "up_convs.%d.upconv.1", j);
        U.name = buf;
        U.cin1 = cur_c; U.c1p = cur_cp; U.cout = co; U.coutp = cop; U.csplit = cop;
        U.H = H; U.W = W; U.upsample = 1; U.relu = 0;
        U.level = l + 1; U.gh = h->level_ext[l + 1].first; U.gw = h->level_ext[l + 1].second;   // its tiles walk the low-resolution level
        U.t_src1 = cur; U.t_out = t_up;
        if (2 * (H / 2) != H || 2 * (W / 2) != W) h->tensors[t_up].zero_fill = true;   // centre pad (unet.py:110-116)
        h->layers.push_back(U);
        // cat((up, skip), 1) -> block: K split over the two tensors (unet.py:118-119)
        snprintf(buf, sizeof buf, "up_convs.%d.block.block.0", j);
        add_unit(h, buf, co, cop, skip_c[l], round_up(skip_c[l], 32), co, l, H, W, unit_has_dropout(d, l, false, 0), t_up,
                 skip[l], t_tmp, -1);
        if (d.residual) {
            snprintf(buf, sizeof buf, "up_convs.%d.block", j);
            add_residual_conv(h, buf, co, cop, skip_c[l], round_up(skip_c[l], 32), co, l, H, W, t_up, skip[l], t_out);
        }
        snprintf(buf, sizeof buf, "up_convs.%d.block.block.1", j);
        add_unit(h, buf, co, cop, 0, 0, co, l, H, W, unit_has_dropout(d, l, false, 1), t_tmp, -1, t_out, -1, d.residual != 0);
        cur = t_out; cur_c = co; cur_cp = cop;
    }
    {   // head unit(s): conv_cls.0 [+ conv_sigma.0 stacked on the output channels] (unet.py:161-164)
        const int cp = round_up(cur_c, 32);
        h->head_cph = cp;
        h->head_cp = d.sigma_out ? 2 * cp : cp;
        // the head unit's output is read by head_kernel as [voxel][channel]: exactly the real image, whatever level 0 is allocated with
        h->t_head = new_tensor(h, -1, d.height, d.width, h->head_cp);
        add_unit(h, "conv_cls.0", cur_c, cur_cp, 0, 0, cur_c, 0, d.height, d.width, d.has_dropout != 0, cur, -1, h->t_head,
                 -1);
        ConvLayer& L = h->layers.back();
        if (d.sigma_out) {
            L.name2 = "conv_sigma.0.conv2d_batch_relu.conv";
            if (d.bn) L.bn2 = "conv_sigma.0.conv2d_batch_relu.bn";
            L.csplit = cp;
            L.coutp = 2 * cp;
            if (d.has_dropout) {
                L.site2 = (int)h->sites.size();
                h->sites.push_back({"conv_sigma.0.conv2d_batch_relu.dropout", cur_c});
            }
        }
    }
    h->t_features = -1;
    if (d.provide_features && h->tensors[h->layers.back().t_src1].padded()) {
        // rcu_unet_features hands the input of conv_cls.0 out as [voxel][channel]: on a padded level 0 a compact copy is made behind the forward
        const Tensor& f = h->tensors[h->layers.back().t_src1];
        h->t_features = new_tensor(h, -1, f.Hr, f.Wr, f.cp);
    }
    h->site_offset.assign(h->sites.size() + 1, 0);
    for (size_t s = 0; s < h->sites.size(); ++s) h->site_offset[s + 1] = h->site_offset[s] + h->sites[s].second;
    h->mask_floats = h->site_offset.back();
    if (valid) *valid = true;
    for (ConvLayer& L : h->layers) {
        const Tensor& to = h->tensors[L.t_out];
        L.cfg = pick_config(L, h->d.max_batch, h->opt, to.H, to.W);
        const ConvConfigInfo& ci = conv_config_info(L.cfg);
        if ((L.c1p % ci.KC) != 0 || (L.c2p % ci.KC) != 0)
            return fail(RCU_ERR_INVALID, "internal: channel chunking does not divide for layer " + L.name);
        L.NT = (L.coutp + ci.BN - 1) / ci.BN;
        // a padded tensor may only be touched by a kernel that keeps its zeros: read as a tile grid and written by the Winograd families /
        // rcu_first.hip -- or written by a direct up-convolution, which places its image in a larger tensor anyway (the centre pad), or as
        // the pooled output of a direct conv unit (it writes the real pooled pixels at the tensor's own pitch)
        const bool src_padded = h->tensors[L.t_src1].padded() && L.t_src1 != h->t_input;
        const bool first_reads_caller = L.t_src1 == h->t_input && h->tensors[L.t_src1].padded();
        const bool out_padded = to.padded() && !(L.upsample && !cfg_handles_padding(L.cfg) && !h->tensors[L.t_src1].padded());
        if ((src_padded || first_reads_caller || out_padded) && !cfg_handles_padding(L.cfg) && valid) *valid = false;
    }
    assign_layouts(h);
    return RCU_OK;
}

// Rough time of a plan's conv stack per slice, arbitrary units: multiplications the layer's kernel executes on whole tiles of the grid it walks
// / the executed fraction of the matrix peak the kernel family has measured at (the direct kernels on the native BraTS
// volume: profiles/r06_layer_report_native_155_nopad.txt, 0.73-0.80).  It only has to rank "pad the level to whole Winograd tiles" against
// "keep the real extent on the direct kernels" and one padding against another.
static double plan_cost(const rcu_unet* h)
{
    double total = 0.0;
    for (const ConvLayer& L : h->layers) {
        const ConvConfigInfo& ci = conv_config_info(L.cfg);
        const double tiles = (double)((L.gh + ci.TH - 1) / ci.TH) * ((L.gw + ci.TW - 1) / ci.TW);
        const double px = tiles * ci.TH * ci.TW;           // pixels of the grid the tiles walk (an up-convolution: the low-resolution grid)
        const double kn = (double)(L.c1p + L.c2p) * (double)(L.NT * ci.BN);
        double mults, eff;
        if (L.cfg == CONV_CFG_FIRST_T8x32) { mults = 9.0; eff = 0.25; }                 // a write stream, not a matrix kernel
        else if (ci.WINO == 3) { mults = L.cfg == CONV_CFG_WINO4_S8T12x8_N32 ? 3.0 : 2.25; eff = (L.c1p + L.c2p) <= 32 ? 0.45 : 0.58; }
        else if (ci.WINO == 2) { mults = 9.0; eff = 0.72; }                              // four classes x 9 / 4 per low-resolution pixel
        else if (ci.WINO == 1) { mults = 4.0; eff = 0.62; }
        else if (L.upsample) { mults = 16.0; eff = 0.72; }                               // four classes x 2x2 taps per low-resolution pixel
        else { mults = 9.0; eff = 0.75; }
        total += px * kn * mults / eff;
    }
    return total;
}

// Allocated extent of every level.  A Winograd kernel takes whole tiles; where the real extent of a level (H >> l) x (W >> l) is not a
// whole number of them -- the reference's BraTS slices are 240 x 240 (levels 240, 120, 60, 30, 15: scripts/create_brats18_dataset.py:53-72 never
// crops), ISIC's 192 x 256 ends in a 12 x 16 level (scripts/prepare_isic_data.py:29-30) -- the level's tensors are ALLOCATED with a padded extent,
// the padding holds zeros that no kernel writes (ConvArgs::part), and the level runs on the Winograd kernels instead of the direct ones: 2.25 x
// (1 + padding) instead of 9 multiplications per pixel.  Every level chooses for itself: a pooled output and an up-convolution carry the extents
// of the tensors on both sides.  The choice minimises plan_cost over a small set of roundings per axis; a level keeps its real extent where
// nothing is gained (whole tiles fit: the benchmark's 192 x 128) or where a layer of the level has no kernel that keeps the zeros (residual
// blocks' adding units, feature taps).
static int choose_level_extents(rcu_unet* h)
{
    const rcu_unet_desc& d = h->d;
    h->level_ext.resize(d.depth + 1);
    for (int l = 0; l <= d.depth; ++l) h->level_ext[l] = {d.height >> l, d.width >> l};
    bool valid = true;
    int rc = build_plan_for(h, &valid);
    if (rc != RCU_OK || h->opt.pad_levels == 0 || h->opt.conv_winograd == 0) return rc;
    double best = plan_cost(h);
    static const int kRound[] = {4, 8, 12, 16, 32};
    for (int sweep = 0; sweep < 2; ++sweep)
        for (int l = 0; l <= d.depth; ++l) {
            const int Hl = d.height >> l, Wl = d.width >> l;
            std::pair<int, int> keep = h->level_ext[l];
            for (int rh : kRound)
                for (int rw : kRound) {
                    const std::pair<int, int> cand = {round_up(Hl, rh), round_up(Wl, rw)};
                    if (cand == keep || (rw == 12)) continue;
                    h->level_ext[l] = cand;
                    rc = build_plan_for(h, &valid);
                    if (rc != RCU_OK) return rc;
                    const double cost = plan_cost(h);
                    if (valid && cost < best * 0.999) {
                        best = cost;
                        keep = cand;
                    }
                }
            h->level_ext[l] = keep;
        }
#ifdef RCU_EXPERIMENTS
    // Experiment builds only (make EXTRA=-DRCU_EXPERIMENTS; tools/layer_report.py under RCU_HIP_LIBRARY): RCU_FORCE_LEVEL_EXTENTS="256x256,128x128,..."
    // overrides the chosen extents level by level (an entry that is smaller than the real extent, or a plan that is not valid, is ignored) --
    // how the planner's choice is measured against its alternatives (HISTORY, round-6 log, item 3).
    if (const char* force = getenv("RCU_FORCE_LEVEL_EXTENTS")) {
        std::vector<std::pair<int, int>> keep = h->level_ext;
        int l = 0;
        for (const char* p = force; *p && l <= d.depth; ++l) {
            int eh = 0, ew = 0, used = 0;
            if (sscanf(p, "%dx%d%n", &eh, &ew, &used) != 2) break;
            if (eh >= (d.height >> l) && ew >= (d.width >> l)) h->level_ext[l] = {eh, ew};
            p += used;
            if (*p == ',') ++p;
        }
        rc = build_plan_for(h, &valid);
        if (rc != RCU_OK || !valid) h->level_ext = keep;
    }
#endif
    return build_plan_for(h, &valid);
}

static int build_plan(rcu_unet* h) { return choose_level_extents(h); }

// Which activation tensors take the channel-blocked layout [N][C/8][H][W][8] (rcu_kernels.h, ConvArgs): every tensor all of whose
// producers and consumers are Winograd kernels (rcu_first.hip as a producer), except the network input, the head unit's output
// (head_kernel reads channels-last) and -- when the caller wants rcu_unet_features -- the feature tensor.  The two sources of a
// cat-free decoder unit share one layout.  rcu_unet_options.act_layout = 1 keeps everything channels-last (A/B tests).
static bool cfg_reads_blocked(int cfg)
{
    return (cfg >= CONV_CFG_WINO_T16x16_N64 && cfg <= CONV_CFG_UPW_S8T4x8_N64) || (cfg >= CONV_CFG_WINO4_T32x32_N32 && cfg < CONV_CFG_END);
}
static bool cfg_writes_blocked(int cfg) { return cfg_reads_blocked(cfg) || cfg == CONV_CFG_FIRST_T8x32; }

static void assign_layouts(rcu_unet* h)
{
    const bool on = h->opt.act_layout == 0;
    for (Tensor& t : h->tensors) t.blocked = on && t.cp % 8 == 0 && !t.zero_fill;
    h->tensors[h->t_input].blocked = false;
    h->tensors[h->t_head].blocked = false;
    if (h->d.provide_features) h->tensors[h->layers.back().t_src1].blocked = false;
    for (bool changed = true; changed;) {
        changed = false;
        auto clear = [&](int t) {
            if (t >= 0 && h->tensors[t].blocked) {
                h->tensors[t].blocked = false;
                changed = true;
            }
        };
        for (const ConvLayer& L : h->layers) {
            if (!cfg_reads_blocked(L.cfg)) {
                clear(L.t_src1);
                clear(L.t_src2);
            }
            if (!cfg_writes_blocked(L.cfg)) {
                clear(L.t_out);
                clear(L.t_pool);
            }
            if (L.t_src2 >= 0 && h->tensors[L.t_src1].blocked != h->tensors[L.t_src2].blocked) {
                clear(L.t_src1);
                clear(L.t_src2);
            }
        }
    }
}

extern "C" const char* rcu_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* rcu_version(void) { return "librcu_hip 0.1.0 gfx950"; }

extern "C" void rcu_unet_default_options(rcu_unet_options* opts)
{
    if (!opts) return;
    std::memset(opts, 0, sizeof *opts);
    opts->conv_winograd = 1;
    opts->conv_winograd4 = 1;
    opts->conv_first = 1;
    opts->act_layout = 0;
    opts->fuse_head = 1;
    opts->head_winograd4 = 1;
    opts->pad_levels = 1;
}

// desc / options checks + the plan (layers, tensors, level extents); no device memory
static int make_plan(const rcu_unet_desc* desc, const rcu_unet_options* opts, const rcu_unet* donor, rcu_unet** out)
{
    if (!desc || !out) return fail(RCU_ERR_INVALID, "rcu_unet_create: null argument");
    const rcu_unet_desc& d = *desc;
    if (d.nb_classes < 1 || d.nb_classes > MAX_CLASSES) return fail(RCU_ERR_INVALID, "nb_classes must be in 1..8");
    if (d.in_channels < 1 || d.depth < 1 || d.depth > 8 || d.start_filters < 1 || d.max_batch < 1)
        return fail(RCU_ERR_INVALID, "rcu_unet_create: bad in_channels / depth / start_filters / max_batch");
    if (d.in_channels > 8 && d.in_channels % 32 != 0)
        return fail(RCU_ERR_INVALID, "in_channels must be <= 8 or a multiple of 32");
    const int div = 1 << d.depth;
    if (d.height < div || d.width < div)
        return fail(RCU_ERR_INVALID, "height and width must be at least 2^depth (one pixel at the bottom level)");
    rcu_unet_options o;
    rcu_unet_default_options(&o);
    if (opts) {
        o = *opts;
        if ((o.conv_winograd | 1) != 1 || (o.conv_winograd4 < 0 || o.conv_winograd4 > 3) ||
            (o.conv_first | 1) != 1 || (o.act_layout | 1) != 1 || (o.fuse_head | 1) != 1 || (o.head_winograd4 | 1) != 1 || (o.pad_levels | 1) != 1 || o.reserved[0])
            return fail(RCU_ERR_INVALID, "rcu_unet_create_with: bad rcu_unet_options value");
    }
    rcu_unet* h = new rcu_unet();
    h->d = d;
    h->opt = o;
    // A borrower is planned for the DONOR's batch: the planner's choices (kernel family per layer, tensor layouts) depend on max_batch, and
    // two plans that share one workspace must agree on every tensor -- identical by construction instead of by luck
    if (donor && donor->ws && donor->ws->max_batch >= d.max_batch) h->d.max_batch = donor->ws->max_batch;
    int rc = build_plan(h);
    if (rc != RCU_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return RCU_OK;
}

extern "C" int rcu_unet_plan(const rcu_unet_desc* desc, const rcu_unet_options* opts, rcu_unet** out)
{
    int rc = make_plan(desc, opts, nullptr, out);
    if (rc == RCU_OK) (*out)->plan_only = true;
    return rc;
}

extern "C" int rcu_unet_create_with(const rcu_unet_desc* desc, const rcu_unet_options* opts, rcu_unet* donor, rcu_unet** out)
{
    rcu_unet* h = nullptr;
    int rc = make_plan(desc, opts, donor, &h);
    if (rc != RCU_OK) return rc;
    const rcu_unet_desc& d = *desc;
    // 32-bit element offsets inside the conv kernel
    for (const Tensor& t : h->tensors)
        if (t.floats_per_slice * (size_t)h->d.max_batch >= (size_t)1 << 31) {
            delete h;
            return fail(RCU_ERR_INVALID, "max_batch too large: an activation tensor would exceed 2^31 elements");
        }
    if (donor) {
        // the donor's plan must hold the same tensors (same shapes, same layouts, same never-written borders) for at least this batch
        bool same = donor->ws && donor->tensors.size() == h->tensors.size() && donor->ws->max_batch >= d.max_batch;
        for (size_t i = 0; same && i < h->tensors.size(); ++i) {
            const Tensor &a = h->tensors[i], &b = donor->tensors[i];
            same = a.floats_per_slice == b.floats_per_slice && a.H == b.H && a.W == b.W && a.Hr == b.Hr && a.Wr == b.Wr && a.cp == b.cp &&
                   a.blocked == b.blocked && a.zero_fill == b.zero_fill;
        }
        if (!same) {
            delete h;
            return fail(RCU_ERR_INVALID, "rcu_unet_create_with: the workspace donor's plan differs (shape, options) or is sized for a smaller batch");
        }
        h->ws = donor->ws;
        h->ws_borrowed = true;
        for (size_t i = 0; i < h->tensors.size(); ++i) h->tensors[i].dev = h->ws->dev[i];
        *out = h;
        return RCU_OK;
    }
    h->ws = std::make_shared<Workspace>();
    h->ws->max_batch = d.max_batch;
    bool zeroed = false;
    for (Tensor& t : h->tensors) {
        const size_t bytes = t.floats_per_slice * (size_t)d.max_batch * sizeof(float);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&t.dev), bytes);
        if (e != hipSuccess) {
            rcu_unet_destroy(h);
            return hip_fail(e, "hipMalloc(activation workspace)");
        }
        h->ws->dev.push_back(t.dev);
        h->ws->bytes.push_back(bytes);
        h->workspace_bytes += (int64_t)bytes;
        if (t.zero_fill || t.padded()) {   // pixels no kernel ever writes and every reader takes for the conv's zero padding
            e = hipMemset(t.dev, 0, bytes);
            if (e != hipSuccess) {
                rcu_unet_destroy(h);
                return hip_fail(e, "hipMemset(centre-pad border / level padding)");
            }
            zeroed = true;
        }
    }
    if (zeroed) {
        // a memset of device memory may return before it has run, and the forwards come on the caller's (non-blocking) streams, which the null
        // stream does not order: the zeros are in place before the handle is handed out
        hipError_t e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) {
            rcu_unet_destroy(h);
            return hip_fail(e, "hipStreamSynchronize(level padding)");
        }
    }
    *out = h;
    return RCU_OK;
}

extern "C" int rcu_unet_create(const rcu_unet_desc* desc, rcu_unet** out) { return rcu_unet_create_with(desc, nullptr, nullptr, out); }

extern "C" int rcu_unet_set_fuse_head(rcu_unet* h, int on)
{
    if (!h) return fail(RCU_ERR_INVALID, "rcu_unet_set_fuse_head: null handle");
    h->opt.fuse_head = on ? 1 : 0;
    return RCU_OK;
}

extern "C" int rcu_unet_destroy(rcu_unet* h)
{
    if (!h) return RCU_OK;
    for (void* p : h->allocs) (void)hipFree(p);     // packed weights, epilogue constants
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    delete h;                                       // drops its reference to the activation workspace
    return RCU_OK;
}

extern "C" int64_t rcu_unet_workspace_bytes(const rcu_unet* h) { return h ? h->workspace_bytes : 0; }
extern "C" int rcu_unet_num_dropout_sites(const rcu_unet* h) { return h ? (int)h->sites.size() : 0; }
extern "C" int rcu_unet_dropout_site_channels(const rcu_unet* h, int site)
{
    if (!h || site < 0 || site >= (int)h->sites.size()) return fail(RCU_ERR_INVALID, "bad dropout site index");
    return h->sites[site].second;
}
extern "C" const char* rcu_unet_dropout_site_name(const rcu_unet* h, int site)
{
    if (!h || site < 0 || site >= (int)h->sites.size()) return "";
    return h->sites[site].first.c_str();
}
extern "C" int rcu_unet_mask_floats_per_sample(const rcu_unet* h) { return h ? h->mask_floats : 0; }

extern "C" int rcu_unet_load_weight(rcu_unet* h, const char* name, const float* data, size_t count)
{
    if (!h || !name || (!data && count)) return fail(RCU_ERR_INVALID, "rcu_unet_load_weight: null argument");
    std::string key(name);
    if (key.rfind("module.", 0) == 0) key = key.substr(7);
    h->host_weights[key].assign(data, data + count);
    h->finalized = false;
    return RCU_OK;
}

static int get_weight(rcu_unet* h, const std::string& key, size_t count, const std::vector<float>** out)
{
    auto it = h->host_weights.find(key);
    if (it == h->host_weights.end()) return fail(RCU_ERR_WEIGHTS, "missing weight tensor '" + key + "'");
    if (it->second.size() != count)
        return fail(RCU_ERR_WEIGHTS, "weight tensor '" + key + "' has " + std::to_string(it->second.size()) +
                                         " elements, expected " + std::to_string(count));
    *out = &it->second;
    return RCU_OK;
}

template <typename T>
static int upload(rcu_unet* h, const std::vector<T>& host, T** dev)
{
    RCU_HIP(hipMalloc(reinterpret_cast<void**>(dev), host.size() * sizeof(T)));
    h->allocs.push_back(*dev);
    h->workspace_bytes += (int64_t)(host.size() * sizeof(T));
    RCU_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return RCU_OK;
}

// Fold one conv (+ optional BN) into (packed weights, alpha, betab, beta) at output-channel offset co0.
static int fold_conv(rcu_unet* h, const ConvLayer& L, const std::string& conv, const std::string& bn, int co0,
                     std::vector<float>& wpack, std::vector<float>& alpha, std::vector<float>& betab,
                     std::vector<float>& beta)
{
    const ConvConfigInfo& ci = conv_config_info(L.cfg);
    const int KC = ci.KC, KCP = ci.KCP, BN = ci.BN;
    const int cin = L.cin1 + L.cin2;
    const std::vector<float>*w, *b;
    std::vector<float> w_centre;
    int rc;
    if (L.is_1x1) {   // [cout][cin][1][1] -> [cout][cin][3][3] with the centre tap set
        if ((rc = get_weight(h, conv + ".weight", (size_t)L.cout * cin, &w))) return rc;
        w_centre.assign((size_t)L.cout * cin * 9, 0.f);
        for (size_t i = 0; i < (size_t)L.cout * cin; ++i) w_centre[i * 9 + 4] = (*w)[i];
        w = &w_centre;
    } else {
        rc = get_weight(h, conv + ".weight", (size_t)L.cout * cin * 9, &w);
    }
    if (rc) return rc;
    rc = get_weight(h, conv + ".bias", (size_t)L.cout, &b);
    if (rc) return rc;
    const std::vector<float>*g = nullptr, *bt = nullptr, *mu = nullptr, *var = nullptr;
    if (!bn.empty()) {
        if ((rc = get_weight(h, bn + ".weight", L.cout, &g))) return rc;
        if ((rc = get_weight(h, bn + ".bias", L.cout, &bt))) return rc;
        if ((rc = get_weight(h, bn + ".running_mean", L.cout, &mu))) return rc;
        if ((rc = get_weight(h, bn + ".running_var", L.cout, &var))) return rc;
    }
    for (int co = 0; co < L.cout; ++co) {
        float A = 1.f, B = 0.f;
        if (g) {
            A = (*g)[co] / std::sqrt((*var)[co] + 1e-5f);
            B = (*bt)[co] - A * (*mu)[co];
        }
        alpha[co0 + co] = A;
        betab[co0 + co] = A * (*b)[co];
        beta[co0 + co] = B;
    }
    const size_t tile_floats = conv_tile_floats(ci);   // padded to a whole number of float4 per thread
    // Sub-pixel up-conv: output parity a (rows) folds the 3 kernel rows onto 2 low-res rows,
    //   a = 0: low-res row y-1 <- {dy 0},   row y   <- {dy 1, 2}
    //   a = 1: low-res row y   <- {dy 0, 1}, row y+1 <- {dy 2}            (same for columns with b)
    auto fold_set = [](int parity, int t, int d) { return parity == 0 ? (t == 0 ? d == 0 : d >= 1) : (t == 0 ? d <= 1 : d == 2); };
    if (ci.WINO == 2) {
        // F(2x2,2x2) per parity class: U^{ab} = G w^{ab} G^T, G = [[1,0],[1,1],[0,1]], w^{ab} the 2x2 folded taps.  Packed
        // per Cin chunk as tiles (a, cout tile) of [p = 6 i + 3 b + j][channel pair][cout][2]; column j = 0 of class b = 1
        // is negated because the kernel feeds P.2 - P.1 where F(2,2) wants P.1 - P.2 (rcu_wino_up.hip).
        static const double G2[3][2] = {{1, 0}, {1, 1}, {0, 1}};
        for (int co = 0; co < L.cout; ++co) {
            const int cop = co0 + co;
            const int ntile = cop / BN, nn = cop % BN;
            for (int ci_ = 0; ci_ < cin; ++ci_) {
                const int chunk = ci_ / KC, kq = ci_ % KC;
                const float* w9 = w->data() + ((size_t)co * cin + ci_) * 9;
                for (int pa = 0; pa < 2; ++pa) {
                    const size_t tile0 = (((size_t)chunk * 2 + pa) * L.NT + ntile) * tile_floats;
                    for (int pb = 0; pb < 2; ++pb) {
                        double wc[2][2];
                        for (int ty = 0; ty < 2; ++ty)
                            for (int tx = 0; tx < 2; ++tx) {
                                double v = 0.0;
                                for (int dy = 0; dy < 3; ++dy)
                                    for (int dx = 0; dx < 3; ++dx)
                                        if (fold_set(pa, ty, dy) && fold_set(pb, tx, dx)) v += (double)w9[dy * 3 + dx];
                                wc[ty][tx] = v;
                            }
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j) {
                                double u = 0.0;
                                for (int ty = 0; ty < 2; ++ty)
                                    for (int tx = 0; tx < 2; ++tx) u += G2[i][ty] * wc[ty][tx] * G2[j][tx];
                                if (pb == 1 && j == 0) u = -u;
                                const int p = 6 * i + 3 * pb + j;
                                wpack[tile0 + (((size_t)p * 4 + (kq >> 1)) * BN + nn) * 2 + (kq & 1)] = (float)u;
                            }
                    }
                }
            }
        }
        return RCU_OK;
    }
    if (ci.WINO == 3) {
        // F(4x4,3x3): U = G g G^T (6x6) per (cout, cin) with the Lavin-Gray G for the points 0, +-1, +-2, inf; packed per Cin chunk and
        // cout tile as [position p = 6 i + j][channel pair q][cout][2] (rcu_wino4.hip).  Computed in double, rounded once.
        static const double G4[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                        {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
        for (int co = 0; co < L.cout; ++co) {
            const int cop = co0 + co;
            const int ntile = cop / BN, nn = cop % BN;
            for (int ci_ = 0; ci_ < cin; ++ci_) {
                const int kp = ci_ < L.cin1 ? ci_ : L.c1p + (ci_ - L.cin1);
                const int chunk = kp / KC, kq = kp % KC;
                const float* w9 = w->data() + ((size_t)co * cin + ci_) * 9;
                const size_t tile0 = ((size_t)chunk * L.NT + ntile) * tile_floats;
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) {
                        double u = 0.0;
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) u += G4[i][r] * (double)w9[r * 3 + c] * G4[j][c];
                        const int p = 6 * i + j;
                        wpack[tile0 + (((size_t)p * 4 + (kq >> 1)) * BN + nn) * 2 + (kq & 1)] = (float)u;
                    }
            }
        }
        return RCU_OK;
    }
    if (ci.WINO) {
        // U = G g G^T per (cout, cin), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; packed per Cin chunk and cout tile as
        // [position p][channel pair q][cout][2] (rcu_wino.hip).  Computed in double, rounded once.
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        for (int co = 0; co < L.cout; ++co) {
            const int cop = co0 + co;
            const int ntile = cop / BN, nn = cop % BN;
            for (int ci_ = 0; ci_ < cin; ++ci_) {
                const int kp = ci_ < L.cin1 ? ci_ : L.c1p + (ci_ - L.cin1);
                const int chunk = kp / KC, kq = kp % KC;
                const float* w9 = w->data() + ((size_t)co * cin + ci_) * 9;
                const size_t tile0 = ((size_t)chunk * L.NT + ntile) * tile_floats;
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        double u = 0.0;
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) u += G[i][r] * (double)w9[r * 3 + c] * G[j][c];
                        const int p = 4 * i + j;
                        wpack[tile0 + (((size_t)p * 4 + (kq >> 1)) * BN + nn) * 2 + (kq & 1)] = (float)u;
                    }
            }
        }
        return RCU_OK;
    }
    const int ncls = L.upsample ? 4 : 1;
    for (int co = 0; co < L.cout; ++co) {
        const int cop = co0 + co;
        const int ntile = cop / BN, nn = cop % BN;
        for (int ci_ = 0; ci_ < cin; ++ci_) {
            const int kp = ci_ < L.cin1 ? ci_ : L.c1p + (ci_ - L.cin1);   // position in the padded K range
            const int chunk = kp / KC, kq = kp % KC;
            const float* w9 = w->data() + ((size_t)co * cin + ci_) * 9;
            for (int cls = 0; cls < ncls; ++cls) {
                const size_t tile0 = ((size_t)chunk * (ncls * L.NT) + (size_t)cls * L.NT + ntile) * tile_floats;
                for (int tap = 0; tap < ci.TAPS; ++tap) {
                    float v;
                    if (!L.upsample) {
                        v = w9[tap];
                    } else {
                        const int a = cls >> 1, b = cls & 1, ty = tap >> 1, tx = tap & 1;
                        v = 0.f;
                        for (int dy = 0; dy < 3; ++dy)
                            for (int dx = 0; dx < 3; ++dx)
                                if (fold_set(a, ty, dy) && fold_set(b, tx, dx)) v += w9[dy * 3 + dx];
                    }
                    // swizzled layout: the two 16-byte units of a row swap places for channels 16..31 (mod 32)
                    const int kdst = ci.SWZ ? ((((kq >> 2) ^ ((nn >> 4) & 1)) << 2) | (kq & 3)) : kq;
                    wpack[tile0 + ((size_t)tap * BN + nn) * KCP + kdst] = v;
                }
            }
        }
    }
    return RCU_OK;
}

extern "C" int rcu_unet_finalize_weights(rcu_unet* h)
{
    if (!h) return fail(RCU_ERR_INVALID, "null handle");
    if (h->plan_only) return fail(RCU_ERR_STATE, "rcu_unet_finalize_weights: the handle is a plan without a workspace (rcu_unet_plan)");
    if (h->finalized) return RCU_OK;
    for (ConvLayer& L : h->layers) {
        const ConvConfigInfo& ci = conv_config_info(L.cfg);
        const int nchunks = (L.c1p + L.c2p) / ci.KC;
        L.wpack_floats = (size_t)nchunks * L.NT * (ci.WINO == 2 ? 2 : L.upsample ? 4 : 1) * conv_tile_floats(ci);
        std::vector<float> wpack(L.wpack_floats, 0.f);
        const int cpad = L.NT * ci.BN;
        std::vector<float> alpha(cpad, 0.f), betab(cpad, 0.f), beta(cpad, 0.f);
        int rc = fold_conv(h, L, L.name, L.bn, 0, wpack, alpha, betab, beta);
        if (rc) return rc;
        if (!L.name2.empty()) {
            rc = fold_conv(h, L, L.name2, L.bn2, L.csplit, wpack, alpha, betab, beta);
            if (rc) return rc;
        }
        if ((rc = upload(h, wpack, &L.wpack))) return rc;
        if ((rc = upload(h, alpha, &L.alpha))) return rc;
        if ((rc = upload(h, betab, &L.betab))) return rc;
        if ((rc = upload(h, beta, &L.beta))) return rc;
    }
    // 1x1 heads (unet.py:161, 164): [C][CPh] zero padded
    const int C = h->d.nb_classes, cph = h->head_cph, creal = h->layers.back().cout;
    auto head = [&](const std::string& key, float** w_dev, float** b_dev) -> int {
        const std::vector<float>*w, *b;
        int rc = get_weight(h, key + ".weight", (size_t)C * creal, &w);
        if (rc) return rc;
        if ((rc = get_weight(h, key + ".bias", (size_t)C, &b))) return rc;
        std::vector<float> wp((size_t)C * cph, 0.f);
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < creal; ++k) wp[(size_t)c * cph + k] = (*w)[(size_t)c * creal + k];
        if ((rc = upload(h, wp, w_dev))) return rc;
        return upload(h, *b, b_dev);
    };
    int rc = head("conv_cls.1", &h->w_cls, &h->b_cls);
    if (rc) return rc;
    if (h->d.sigma_out && (rc = head("conv_sigma.1", &h->w_sig, &h->b_sig))) return rc;
    h->host_weights.clear();
    h->finalized = true;
    return RCU_OK;
}

// `head`: when set (conv_cls.0 on the 32-cout Winograd tile, two classes), the 1x1 classifier + softmax + statistics run in
// the layer's epilogue and its output tensor is not written (rcu_wino.hip, wino_epilogue_head).
// the plan's last unit can take the classifier into its epilogue: two classes, no sigma twin, one 32-cout Winograd tile of either family
static bool head_fusable(const rcu_unet* h)
{
    const ConvLayer& last = h->layers.back();
    return (last.cfg == CONV_CFG_WINO_T16x32_N32 || last.cfg == CONV_CFG_WINO4_T32x32_N32) && h->d.nb_classes == 2 && last.name2.empty() &&
           h->head_cph == 32;
}

struct FusedHead {
    float* logits;
    void* stats;
    int flags;
    int passes;   // pass group: the batch holds `passes` x (n / passes) samples, all adding into the statistics of n / passes images
};

static int run_layer(rcu_unet* h, const ConvLayer& L, int n, const float* masks, hipStream_t stream, const FusedHead* head = nullptr,
                     const float* x_nchw = nullptr, int n_images = 0)
{
    const ConvConfigInfo& ci = conv_config_info(L.cfg);
    ConvArgs a{};
    a.src1 = h->tensors[L.t_src1].dev;
    a.src2 = L.t_src2 >= 0 ? h->tensors[L.t_src2].dev : nullptr;
    a.wpack = L.wpack; a.alpha = L.alpha; a.betab = L.betab; a.beta = L.beta;
    a.mask = (masks && L.site >= 0) ? masks + (size_t)n * h->site_offset[L.site] : nullptr;
    a.mask2 = (masks && L.site2 >= 0) ? masks + (size_t)n * h->site_offset[L.site2] : nullptr;
    a.out = h->tensors[L.t_out].dev;
    a.pooled = L.t_pool >= 0 ? h->tensors[L.t_pool].dev : nullptr;
    const int gh = L.gh, gw = L.gw;   // tile grid = the (allocated) input grid
    a.N = n; a.H = gh; a.W = gw;
    const Tensor& tsrc = h->tensors[L.t_src1];
    const Tensor& tout = h->tensors[L.t_out];
    const int os = L.upsample ? 2 : 1;
    if (cfg_handles_padding(L.cfg)) {
        // padded level on either side (ConvArgs::part): the tiles walk the allocated grid, the stores keep to the real image, the output and
        // pooled tensors have the extents of their own levels (the head unit's output: exactly the real image)
        const bool pool_differs = L.t_pool >= 0 && (h->tensors[L.t_pool].H != gh / 2 || h->tensors[L.t_pool].W != gw / 2);
        if (tsrc.padded() || tout.H != os * gh || tout.W != os * gw || pool_differs) {
            a.part = 1;
            a.Hr = L.upsample ? 2 * (L.H / 2) : L.H;
            a.Wr = L.upsample ? 2 * (L.W / 2) : L.W;
            a.out_H = tout.H; a.out_W = tout.W;
            if (L.t_pool >= 0) { a.pool_H = h->tensors[L.t_pool].H; a.pool_W = h->tensors[L.t_pool].W; }
        }
    } else if (L.t_pool >= 0 && h->tensors[L.t_pool].padded()) {
        a.pool_H = h->tensors[L.t_pool].H; a.pool_W = h->tensors[L.t_pool].W;   // a direct unit pooling into a padded level
    } else if (L.upsample && (2 * gh != tout.H || 2 * gw != tout.W)) {
        // the direct up-convolution places its 2 gh x 2 gw image in a larger tensor: the reference's centre pad
        // F.pad(up, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2)) (unet.py:110-116), and / or a padded level around it
        a.out_H = tout.H; a.out_W = tout.W;
        a.out_y0 = (L.H - 2 * gh) / 2; a.out_x0 = (L.W - 2 * gw) / 2;
    }
    a.C1 = L.c1p; a.C2 = L.c2p; a.CoutP = L.coutp;
    a.cin_real = L.cin1;
    a.x_nchw = L.cfg == CONV_CFG_FIRST_T8x32 ? x_nchw : nullptr;
    a.n_images = n_images;
    a.Cmask = L.cout; a.Csplit = L.csplit; a.Cmask2 = L.cout;
    a.relu = L.relu;
    a.accumulate = L.accumulate;
    a.tiles_y = (gh + ci.TH - 1) / ci.TH;
    a.tiles_x = (gw + ci.TW - 1) / ci.TW;
    a.slice_groups = (n + ci.TS - 1) / ci.TS;
    a.NT = L.NT;
    a.NTW_total = L.NT * (ci.WINO == 2 ? 2 : L.upsample ? 4 : 1);
    {
        const unsigned long long items = (unsigned long long)a.NTW_total * a.tiles_x * a.tiles_y * a.slice_groups;
        auto magic = [&](int d) -> uint32_t {
            return (d > 1 && items * (unsigned long long)d < (1ull << 32)) ? (uint32_t)(((1ull << 32) + d - 1) / d) : 0u;
        };
        a.magic_ntw = magic(a.NTW_total); a.magic_tx = magic(a.tiles_x); a.magic_ty = magic(a.tiles_y);
    }
    {
        auto strides = [](const Tensor& t, uint32_t& pix_bytes, uint32_t& chunk_bytes) {
            pix_bytes = t.blocked ? 32u : (uint32_t)t.cp * 4u;
            chunk_bytes = t.blocked ? (uint32_t)(t.H * t.W) * 32u : 32u;
        };
        strides(h->tensors[L.t_src1], a.in_pix_bytes, a.in_chunk_bytes);
        strides(h->tensors[L.t_out], a.out_pix_bytes, a.out_chunk_bytes);
        if (L.t_pool >= 0) strides(h->tensors[L.t_pool], a.pool_pix_bytes, a.pool_chunk_bytes);
    }
    a.src1_bytes = (uint32_t)std::min<size_t>(h->tensors[L.t_src1].floats_per_slice * (size_t)n * 4, 0xFFFFFFFFu);
    a.src2_bytes = L.t_src2 >= 0 ? (uint32_t)std::min<size_t>(h->tensors[L.t_src2].floats_per_slice * (size_t)n * 4, 0xFFFFFFFFu) : 0u;
    a.wpack_bytes = (uint32_t)std::min<size_t>(L.wpack_floats * 4, 0xFFFFFFFFu);
    int cfg = L.cfg;
    if (head) {
        cfg = L.cfg == CONV_CFG_WINO4_T32x32_N32 ? CONV_CFG_WINO4_T32x32_N32_HEAD : CONV_CFG_WINO_T16x32_N32_HEAD;
        a.head_w = h->w_cls; a.head_b = h->b_cls;
        a.head_logits = head->logits; a.head_stats = head->stats; a.head_flags = head->flags;
        a.head_passes = head->passes;
        a.head_images = n / head->passes;
        a.head_V = (size_t)a.head_images * L.H * L.W;
    }
    RCU_HIP(launch_conv3x3(cfg, a, stream));
    return RCU_OK;
}

// `passes` > 1 (statistics only): the n images run as ONE batch of n * passes samples -- sample t * n + i is image i
// under the mask rows [site][t * n + i] -- and the head adds all passes into the n statistics entries.
static int forward_impl(rcu_unet* h, const float* x, int n, const float* masks, float* logits, float* sigma, void* stats,
                        int flags, hipStream_t stream, int passes = 1, float* sigma_sum = nullptr, int sigma_log = 0)
{
    if (!h || !x) return fail(RCU_ERR_INVALID, "rcu_unet_forward: null argument");
    if (!h->finalized) return fail(RCU_ERR_STATE, "rcu_unet_forward before rcu_unet_finalize_weights");
    if (n < 1 || passes < 1 || (long)n * passes > h->d.max_batch)
        return fail(RCU_ERR_INVALID, "batch size (times passes) outside 1..max_batch");
    if (passes > 1 && (!stats || logits || sigma)) return fail(RCU_ERR_INVALID, "pass groups only feed the statistics");
    const int n_one = n;
    n *= passes;
    if ((sigma || sigma_sum) && !h->d.sigma_out) return fail(RCU_ERR_INVALID, "sigma output requested from a model without sigma_out");
    hipEvent_t* ev = nullptr;
    if (h->prof_capacity > 0 && h->prof_used < h->prof_capacity)
        ev = h->prof_events.data() + (size_t)(h->prof_used++) * prof_slots(h);
    if (ev) RCU_HIP(hipEventRecord(*ev++, stream));
    // the first-layer kernel reads the caller's NCHW input itself; every other first layer wants the channels-last copy
    const bool direct_input = h->layers.front().cfg == CONV_CFG_FIRST_T8x32;
    if (!direct_input)
        RCU_HIP(launch_pack_input(x, h->tensors[h->t_input].dev, n_one, h->d.in_channels, h->in_cp, h->d.height, h->d.width,
                                  h->tensors[h->t_input].H, h->tensors[h->t_input].W, passes, stream));
    if (ev) RCU_HIP(hipEventRecord(*ev++, stream));
    // conv_cls.0 and the classifier as one kernel where the shapes allow (the shipped configurations; rcu_unet_options.fuse_head = 0 /
    // rcu_unet_set_fuse_head keep them apart): two classes, no sigma twin, 32-cout Winograd tile; the passes of a pass group run back to back on the
    // workgroup that owns the tile, so their read-modify-writes of the statistics are ordered (pass 0 first, as head_kernel adds them)
    const ConvLayer& last = h->layers.back();
    const bool fuse = head_fusable(h) && sigma == nullptr && (logits != nullptr || stats != nullptr) && h->opt.fuse_head != 0;
    for (const ConvLayer& L : h->layers) {
        const FusedHead fh{logits, stats, flags, passes};
        int rc = run_layer(h, L, n, masks, stream, (fuse && &L == &last) ? &fh : nullptr, direct_input ? x : nullptr, n_one);
        if (rc) return rc;
        if (ev) RCU_HIP(hipEventRecord(*ev++, stream));
    }
    if (h->t_features >= 0) {   // (provide_features plans never fuse the head away from this point: the copy runs behind the last conv unit either way)
        const Tensor& f = h->tensors[h->layers.back().t_src1];
        RCU_HIP(launch_crop_nhwc(f.dev, h->tensors[h->t_features].dev, n, f.Hr, f.Wr, f.H, f.W, f.cp, stream));
    }
    if (fuse) {
        if (ev) RCU_HIP(hipEventRecord(*ev++, stream));
        return RCU_OK;
    }
    HeadArgs a{};
    a.act = h->tensors[h->t_head].dev;
    a.w_cls = h->w_cls; a.b_cls = h->b_cls; a.w_sig = h->w_sig; a.b_sig = h->b_sig;
    a.logits = logits; a.sigma = sigma; a.stats = stats;
    a.sigma_sum = sigma_sum; a.sigma_log = sigma_log;
    a.C = h->d.nb_classes; a.CP = h->head_cp; a.CPh = h->head_cph; a.stats_flags = flags;
    a.HW = (size_t)h->d.height * h->d.width;
    a.V = a.HW * n_one;
    a.passes = passes;
    RCU_HIP(launch_head(a, stream));
    if (ev) RCU_HIP(hipEventRecord(*ev++, stream));
    return RCU_OK;
}

extern "C" int rcu_unet_profile_begin(rcu_unet* h, int max_forwards)
{
    if (!h || max_forwards < 0) return fail(RCU_ERR_INVALID, "rcu_unet_profile_begin: bad argument");
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    h->prof_events.clear();
    h->prof_capacity = h->prof_used = 0;
    const size_t total = (size_t)max_forwards * prof_slots(h);
    h->prof_events.resize(total);
    for (size_t i = 0; i < total; ++i) {
        hipError_t e = hipEventCreate(&h->prof_events[i]);
        if (e != hipSuccess) {
            h->prof_events.resize(i);
            return hip_fail(e, "hipEventCreate");
        }
    }
    h->prof_capacity = max_forwards;
    return RCU_OK;
}

extern "C" int rcu_unet_profile_collect(rcu_unet* h, double* ms_sum, int* forwards)
{
    if (!h || !ms_sum || !forwards) return fail(RCU_ERR_INVALID, "rcu_unet_profile_collect: null argument");
    const int slots = prof_slots(h);
    for (int i = 0; i < slots - 1; ++i) ms_sum[i] = 0.0;
    for (int f = 0; f < h->prof_used; ++f) {
        hipEvent_t* ev = h->prof_events.data() + (size_t)f * slots;
        RCU_HIP(hipEventSynchronize(ev[slots - 1]));
        for (int i = 0; i < slots - 1; ++i) {
            float ms = 0.f;
            RCU_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            ms_sum[i] += ms;
        }
    }
    *forwards = h->prof_used;
    h->prof_used = 0;
    return RCU_OK;
}

extern "C" int rcu_unet_forward(rcu_unet* h, const float* x_dev, int n, const float* masks_dev, float* logits_dev,
                                float* sigma_dev, void* stream)
{
    return forward_impl(h, x_dev, n, masks_dev, logits_dev, sigma_dev, nullptr, 0, static_cast<hipStream_t>(stream));
}

extern "C" int rcu_unet_forward_accumulate(rcu_unet* h, const float* x_dev, int n, const float* masks_dev, void* stats_dev,
                                           int flags, void* stream)
{
    if (!stats_dev) return fail(RCU_ERR_INVALID, "rcu_unet_forward_accumulate: null stats");
    return forward_impl(h, x_dev, n, masks_dev, nullptr, nullptr, stats_dev, flags & (RCU_MC_MI | RCU_MC_VAR | RCU_MC_EXACT),
                        static_cast<hipStream_t>(stream));
}

extern "C" int rcu_unet_forward_accumulate_sigma(rcu_unet* h, const float* x_dev, int n, const float* masks_dev, void* stats_dev,
                                                 int flags, float* sigma_sum_dev, int is_log_sigma, void* stream)
{
    if (!stats_dev || !sigma_sum_dev) return fail(RCU_ERR_INVALID, "rcu_unet_forward_accumulate_sigma: null stats / sigma sum");
    return forward_impl(h, x_dev, n, masks_dev, nullptr, nullptr, stats_dev, flags & (RCU_MC_MI | RCU_MC_VAR | RCU_MC_EXACT),
                        static_cast<hipStream_t>(stream), 1, sigma_sum_dev, is_log_sigma ? 1 : 0);
}

extern "C" int rcu_unet_forward_accumulate_sigma_passes(rcu_unet* h, const float* x_dev, int n, int passes, const float* masks_dev,
                                                        void* stats_dev, int flags, float* sigma_sum_dev, int is_log_sigma, void* stream)
{
    if (!stats_dev || !sigma_sum_dev) return fail(RCU_ERR_INVALID, "rcu_unet_forward_accumulate_sigma_passes: null stats / sigma sum");
    return forward_impl(h, x_dev, n, masks_dev, nullptr, nullptr, stats_dev, flags & (RCU_MC_MI | RCU_MC_VAR | RCU_MC_EXACT),
                        static_cast<hipStream_t>(stream), passes, sigma_sum_dev, is_log_sigma ? 1 : 0);
}

extern "C" int rcu_unet_features(const rcu_unet* h, const float** features_dev, int* channels, int* channel_pitch)
{
    if (!h || !features_dev) return fail(RCU_ERR_INVALID, "rcu_unet_features: null argument");
    if (!h->finalized) return fail(RCU_ERR_STATE, "rcu_unet_features before rcu_unet_finalize_weights");
    for (const ConvLayer& L : h->layers)
        if (L.name == "conv_cls.0.conv2d_batch_relu.conv") {
            if (h->tensors[L.t_src1].blocked || (h->tensors[L.t_src1].padded() && h->t_features < 0))
                return fail(RCU_ERR_STATE, "rcu_unet_features: the handle was created without rcu_unet_desc.provide_features (the feature "
                                           "tensor is held channel-blocked / on a padded level)");
            *features_dev = h->t_features >= 0 ? h->tensors[h->t_features].dev : h->tensors[L.t_src1].dev;
            if (channels) *channels = L.cin1;
            if (channel_pitch) *channel_pitch = L.c1p;
            return RCU_OK;
        }
    return fail(RCU_ERR_STATE, "rcu_unet_features: no conv_cls layer in the plan");
}

// ------------------------------------------------------------------------------------------------
// PostNet
// ------------------------------------------------------------------------------------------------
struct rcu_postnet {
    int in_channels = 0, nb_classes = 0, nb_convs = 0, bn = 1;
    std::map<std::string, std::vector<float>> host_weights;
    float* packed = nullptr;      // eval image of the layers
    float* packed_mc = nullptr;   // MC-dropout image (see rcu_postnet_finalize_weights)
    bool finalized = false;
};

extern "C" int rcu_postnet_create(int in_channels, int nb_classes, int nb_convs, int bn, rcu_postnet** out)
{
    if (!out) return fail(RCU_ERR_INVALID, "rcu_postnet_create: null argument");
    if (in_channels < 1 || in_channels > 32 * PN_MAX_BLOCKS)
        return fail(RCU_ERR_INVALID, "PostNet kernel handles 1.." + std::to_string(32 * PN_MAX_BLOCKS) + " feature channels");
    if (nb_classes < 1 || nb_classes > 32) return fail(RCU_ERR_INVALID, "PostNet kernel handles 1..32 classes");
    if (nb_convs < 0 || nb_convs + 1 > PN_MAX_LAYERS)
        return fail(RCU_ERR_INVALID, "PostNet kernel handles at most " + std::to_string(PN_MAX_LAYERS - 1) + " hidden convs");
    const int cb = (in_channels + 31) / 32;
    if ((size_t)(nb_convs + 1) * pn_layer_floats(cb) * sizeof(float) > 160 * 1024)
        return fail(RCU_ERR_INVALID, "PostNet: the packed layers of " + std::to_string(nb_convs) + " hidden convs over " +
                                         std::to_string(in_channels) + " channels do not fit the 160 KB of LDS");
    rcu_postnet* h = new rcu_postnet;
    h->in_channels = in_channels; h->nb_classes = nb_classes; h->nb_convs = nb_convs; h->bn = bn ? 1 : 0;
    *out = h;
    return RCU_OK;
}

extern "C" void rcu_postnet_destroy(rcu_postnet* h)
{
    if (!h) return;
    if (h->packed) (void)hipFree(h->packed);
    if (h->packed_mc) (void)hipFree(h->packed_mc);
    delete h;
}

extern "C" int rcu_postnet_load_weight(rcu_postnet* h, const char* name, const float* data, size_t count)
{
    if (!h || !name || (!data && count)) return fail(RCU_ERR_INVALID, "rcu_postnet_load_weight: null argument");
    std::string key(name);
    if (key.rfind("module.", 0) == 0) key = key.substr(7);
    h->host_weights[key].assign(data, data + count);
    h->finalized = false;
    return RCU_OK;
}

static int postnet_weight(rcu_postnet* h, const std::string& key, size_t count, const std::vector<float>** out)
{
    auto it = h->host_weights.find(key);
    if (it == h->host_weights.end()) return fail(RCU_ERR_WEIGHTS, "missing weight tensor '" + key + "'");
    if (it->second.size() != count)
        return fail(RCU_ERR_WEIGHTS, "weight tensor '" + key + "' has " + std::to_string(it->second.size()) +
                                         " elements, expected " + std::to_string(count));
    *out = &it->second;
    return RCU_OK;
}

// folded [32 cb][32 cb] (zero padded) layer -> the LDS image of rcu_postnet.hip: weight tiles [cob][cib][j][lane][4], accumulator
// start values [cob][j][h][4], constants behind the dropout factor [cob][j][h][4]
static void postnet_pack_layer(int cb, const std::vector<float>& w, const std::vector<float>& start, const std::vector<float>& after,
                               float* dst)
{
    const int cp = 32 * cb;
    for (int cob = 0; cob < cb; ++cob)
        for (int cib = 0; cib < cb; ++cib)
            for (int j = 0; j < 4; ++j)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e)
                        dst[((((cob * cb + cib) * 4 + j) * 64) + lane) * 4 + e] =
                            w[(size_t)(32 * cob + (lane & 31)) * cp + 32 * cib + 8 * j + 4 * (lane >> 5) + e];
    float* bd = dst + cb * cb * 1024;
    for (int which = 0; which < 2; ++which)
        for (int cob = 0; cob < cb; ++cob)
            for (int j = 0; j < 4; ++j)
                for (int hf = 0; hf < 2; ++hf)
                    for (int e = 0; e < 4; ++e)
                        bd[which * cb * 32 + ((cob * 4 + j) * 2 + hf) * 4 + e] = (which ? after : start)[32 * cob + 8 * j + 4 * hf + e];
}

extern "C" int rcu_postnet_finalize_weights(rcu_postnet* h)
{
    if (!h) return fail(RCU_ERR_INVALID, "rcu_postnet_finalize_weights: null handle");
    const int C = h->in_channels, L = h->nb_convs + 1, cb = (C + 31) / 32, cp = 32 * cb;
    const size_t layer = pn_layer_floats(cb);
    // two images: eval (accumulators start from the whole folded bias) and MC (they start from alpha * conv bias; the dropout factor
    // and BatchNorm's shift follow the MFMAs)
    std::vector<float> packed((size_t)L * layer, 0.f), packed_mc((size_t)L * layer, 0.f);
    for (int l = 0; l < L; ++l) {
        const bool last = (l == h->nb_convs);
        const std::string pre = last ? "conv_logits" : "convs." + std::to_string(l) + ".conv2d_batch_relu.conv";
        const int cout = last ? h->nb_classes : C;
        const std::vector<float>*w, *b;
        int rc = postnet_weight(h, pre + ".weight", (size_t)cout * C, &w);
        if (rc) return rc;
        if ((rc = postnet_weight(h, pre + ".bias", cout, &b))) return rc;
        std::vector<float> wf((size_t)cp * cp, 0.f), b_conv(cp, 0.f), b_shift(cp, 0.f), b_all(cp, 0.f);
        for (int o = 0; o < cout; ++o) {
            float alpha = 1.f, beta = 0.f, mean = 0.f;
            if (!last && h->bn) {   // Conv -> [Dropout2d] -> BatchNorm(eval): y = gamma (m conv - mean) / sqrt(var + eps) + beta
                const std::string bp = "convs." + std::to_string(l) + ".conv2d_batch_relu.bn";
                const std::vector<float>*g, *be, *rm, *rv;
                if ((rc = postnet_weight(h, bp + ".weight", C, &g))) return rc;
                if ((rc = postnet_weight(h, bp + ".bias", C, &be))) return rc;
                if ((rc = postnet_weight(h, bp + ".running_mean", C, &rm))) return rc;
                if ((rc = postnet_weight(h, bp + ".running_var", C, &rv))) return rc;
                alpha = (float)((double)(*g)[o] / std::sqrt((double)(*rv)[o] + 1e-5));
                beta = (*be)[o];
                mean = (*rm)[o];
            }
            for (int i = 0; i < C; ++i) wf[(size_t)o * cp + i] = alpha * (*w)[(size_t)o * C + i];
            b_all[o] = alpha * ((*b)[o] - mean) + beta;
            b_conv[o] = alpha * (*b)[o];
            b_shift[o] = beta - alpha * mean;
        }
        postnet_pack_layer(cb, wf, b_all, std::vector<float>(cp, 0.f), packed.data() + (size_t)l * layer);
        postnet_pack_layer(cb, wf, last ? b_all : b_conv, b_shift, packed_mc.data() + (size_t)l * layer);
    }
    if (!h->packed) RCU_HIP(hipMalloc(reinterpret_cast<void**>(&h->packed), packed.size() * sizeof(float)));
    if (!h->packed_mc) RCU_HIP(hipMalloc(reinterpret_cast<void**>(&h->packed_mc), packed_mc.size() * sizeof(float)));
    RCU_HIP(hipMemcpy(h->packed, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    RCU_HIP(hipMemcpy(h->packed_mc, packed_mc.data(), packed_mc.size() * sizeof(float), hipMemcpyHostToDevice));
    h->finalized = true;
    return RCU_OK;
}

extern "C" int rcu_postnet_forward(rcu_postnet* h, const float* features_dev, int channel_pitch, int n, int hw,
                                   const float* masks_dev, float* logits_dev, void* stream)
{
    if (!h || !features_dev || !logits_dev) return fail(RCU_ERR_INVALID, "rcu_postnet_forward: null argument");
    if (!h->finalized) return fail(RCU_ERR_STATE, "rcu_postnet_forward before rcu_postnet_finalize_weights");
    if (n < 0 || hw < 1) return fail(RCU_ERR_INVALID, "rcu_postnet_forward: bad shape");
    const int cb = (h->in_channels + 31) / 32;
    if (channel_pitch < 32 * cb || channel_pitch % 4)
        return fail(RCU_ERR_INVALID, "rcu_postnet_forward: channel pitch must be >= in_channels rounded up to 32 and a multiple of 4");
    RCU_HIP(launch_postnet(features_dev, channel_pitch, (size_t)n * hw, hw, masks_dev ? h->packed_mc : h->packed, h->nb_convs + 1,
                           h->nb_classes, h->in_channels, masks_dev, logits_dev, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_unet_forward_accumulate_passes(rcu_unet* h, const float* x_dev, int n, int passes, const float* masks_dev,
                                                  void* stats_dev, int flags, void* stream)
{
    if (!stats_dev) return fail(RCU_ERR_INVALID, "rcu_unet_forward_accumulate_passes: null stats");
    return forward_impl(h, x_dev, n, masks_dev, nullptr, nullptr, stats_dev, flags & (RCU_MC_MI | RCU_MC_VAR | RCU_MC_EXACT),
                        static_cast<hipStream_t>(stream), passes);
}

extern "C" int rcu_unet_num_layers(const rcu_unet* h) { return h ? (int)h->layers.size() : 0; }

extern "C" int rcu_unet_layer_info(const rcu_unet* h, int layer, rcu_layer_info* out)
{
    if (!h || !out || layer < 0 || layer >= (int)h->layers.size()) return fail(RCU_ERR_INVALID, "bad layer index");
    const ConvLayer& L = h->layers[layer];
    std::memset(out, 0, sizeof *out);
    std::snprintf(out->name, sizeof out->name, "%s", L.name.c_str());
    std::snprintf(out->kernel, sizeof out->kernel, "%s", conv_config_info(L.cfg).kernel_name);
    out->cin = L.cin1 + L.cin2;
    out->cout = L.name2.empty() ? L.cout : 2 * L.cout;
    out->height = L.H; out->width = L.W;
    out->grid_height = L.gh; out->grid_width = L.gw;
    out->upsample = L.upsample; out->pooled = L.t_pool >= 0; out->dual_source = L.t_src2 >= 0;
    out->head_fusable = (layer + 1 == (int)h->layers.size() && head_fusable(h)) ? 1 : 0;
    // an up-convolution works on the up-sampled grid, which a centre pad leaves smaller than the skip tensor it is padded to
    out->flops_per_slice = 2.0 * out->cin * out->cout * (L.is_1x1 ? 1.0 : 9.0) * (L.upsample ? 4.0 * (L.H / 2) * (L.W / 2) : (double)L.H * L.W);
    {
        // executed on the matrix pipe: padded K and N, full tiles, 4 taps per output pixel for the sub-pixel form
        const ConvConfigInfo ci = conv_config_info(L.cfg);
        const int lh = L.gh, lw = L.gw;   // grid the tiles walk: the allocated extent of the level (rcu_layer_info.grid_height / grid_width)
        const double tiles = (double)((lh + ci.TH - 1) / ci.TH) * ((lw + ci.TW - 1) / ci.TW);
        const double px = tiles * ci.TH * ci.TW * (L.upsample ? 4.0 : 1.0);
        const double ncols = (double)L.NT * ci.BN;
        // Winograd: 16 multiplications per 2x2 output tile instead of 9 per pixel
        // and 9 per 2x2 low-resolution tile and parity class (= 9 per low-resolution pixel) for the up-convolutions
        // F(4x4,3x3): 36 per 4x4 output tile
        out->mfma_flops_per_slice = ci.WINO == 2 ? 2.0 * L.c1p * ncols * 9.0 * (px / 4.0)
                                    : ci.WINO == 3 ? 2.0 * (L.c1p + L.c2p) * ncols * 36.0 * (px / 16.0)
                                                   : 2.0 * (L.c1p + L.c2p) * ncols * (ci.WINO ? 4.0 : (double)ci.TAPS) * px;
        if (L.cfg == CONV_CFG_WINO4_S8T12x8_N32) out->mfma_flops_per_slice *= 4.0 / 3.0;   // 8 tile slots per slice execute, 6 hold tiles
        if (L.cfg == CONV_CFG_FIRST_T8x32)   // K = 4 channels per tap unless more than four are real
            out->mfma_flops_per_slice = 2.0 * (L.cin1 > 4 ? 8 : 4) * ncols * 9.0 * px;
    }
    return RCU_OK;
}

extern "C" int rcu_unet_run_layer(rcu_unet* h, int layer, int n, const float* masks_dev, void* stream)
{
    if (!h || layer < 0 || layer >= (int)h->layers.size()) return fail(RCU_ERR_INVALID, "bad layer index");
    if (!h->finalized) return fail(RCU_ERR_STATE, "rcu_unet_run_layer before rcu_unet_finalize_weights");
    if (n < 1 || n > h->d.max_batch) return fail(RCU_ERR_INVALID, "batch size outside 1..max_batch");
    return run_layer(h, h->layers[layer], n, masks_dev, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------
// step seam
// ------------------------------------------------------------------------------------------------
static int check_classes(int c)
{
    if (c < 1 || c > MAX_CLASSES) return fail(RCU_ERR_INVALID, "nb_classes must be in 1..8");
    return RCU_OK;
}

extern "C" size_t rcu_mc_stats_bytes(size_t n, size_t hw, int C, int flags)
{
    const size_t V = n * hw;
    const size_t mi = (flags & RCU_MC_MI) ? 1 : 0;
    if (flags & (RCU_MC_VAR | RCU_MC_EXACT)) return V * ((size_t)C + ((flags & RCU_MC_VAR) ? (size_t)C : 0) + mi) * sizeof(double);
    return V * ((size_t)C + mi) * sizeof(float);
}

extern "C" int rcu_mc_begin(void* stats, size_t n, size_t hw, int C, int flags, void* stream)
{
    if (!stats) return fail(RCU_ERR_INVALID, "rcu_mc_begin: null stats");
    if (check_classes(C)) return RCU_ERR_INVALID;
    RCU_HIP(hipMemsetAsync(stats, 0, rcu_mc_stats_bytes(n, hw, C, flags), static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_mc_accumulate(const float* in, void* stats, size_t n, size_t hw, int C, int flags, void* stream)
{
    if (!in || !stats) return fail(RCU_ERR_INVALID, "rcu_mc_accumulate: null argument");
    if (check_classes(C)) return RCU_ERR_INVALID;
    if (n * hw == 0) return RCU_OK;
    RCU_HIP(launch_mc_accumulate(in, stats, C, n, hw, flags, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_mc_finalize(const void* stats, size_t n, size_t hw, int C, int T, int flags, float* mean, float* entropy,
                               float* mi, float* var, void* stream)
{
    if (!stats) return fail(RCU_ERR_INVALID, "rcu_mc_finalize: null stats");
    if (check_classes(C)) return RCU_ERR_INVALID;
    if (T < 1) return fail(RCU_ERR_INVALID, "rcu_mc_finalize: T must be >= 1");
    if (mi && !(flags & RCU_MC_MI)) return fail(RCU_ERR_INVALID, "mutual_info requested without RCU_MC_MI statistics");
    if (var && !(flags & RCU_MC_VAR)) return fail(RCU_ERR_INVALID, "variance requested without RCU_MC_VAR statistics");
    if ((flags & RCU_MC_EXACT) && T > RCU_MC_EXACT_MAX_PASSES)
        return fail(RCU_ERR_INVALID, "RCU_MC_EXACT statistics hold at most 2048 passes (sums of multiples of 2^-40 below 2^13)");
    if (n * hw == 0) return RCU_OK;
    RCU_HIP(launch_mc_finalize(stats, C, n, hw, T, flags, mean, entropy, mi, var, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_softmax(const float* logits, float* probs, size_t n, size_t hw, int C, void* stream)
{
    if (!logits || !probs) return fail(RCU_ERR_INVALID, "rcu_softmax: null argument");
    if (check_classes(C)) return RCU_ERR_INVALID;
    if (n * hw == 0) return RCU_OK;
    RCU_HIP(launch_softmax_nchw(logits, probs, C, n, hw, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_aleatoric(const float* logits, const float* sigma_raw, size_t n, size_t hw, int C, int is_log_sigma,
                             float* probs, float* sigma, uint8_t* prediction, float* sigma_pred, void* stream)
{
    if (!logits || !sigma_raw) return fail(RCU_ERR_INVALID, "rcu_aleatoric: null argument");
    if (check_classes(C)) return RCU_ERR_INVALID;
    if (n * hw == 0) return RCU_OK;
    RCU_HIP(launch_aleatoric(logits, sigma_raw, C, n, hw, is_log_sigma, probs, sigma, prediction, sigma_pred,
                             static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_prediction_and_foreground(const float* probs, size_t n, size_t hw, int C, uint8_t* prediction,
                                             float* p_fg, void* stream)
{
    if (!probs) return fail(RCU_ERR_INVALID, "rcu_prediction_and_foreground: null argument");
    if (check_classes(C)) return RCU_ERR_INVALID;
    if (n * hw == 0) return RCU_OK;
    RCU_HIP(launch_argmax_fg(probs, C, n, hw, prediction, p_fg, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_dropout_masks(const uint64_t* seeds, int passes, int n, uint64_t first_sample, const int32_t* site_channels,
                                 const float* site_keep, int n_sites, float* out, void* stream)
{
    if (!seeds || !site_channels || !site_keep || !out) return fail(RCU_ERR_INVALID, "rcu_dropout_masks: null argument");
    if (passes < 1 || n < 1 || n_sites < 1 || n_sites > MASK_MAX_SITES)
        return fail(RCU_ERR_INVALID, "rcu_dropout_masks: passes, n >= 1 and 1 <= n_sites <= 40");
    MaskArgs a{};
    long end = 0;
    for (int s = 0; s < n_sites; ++s) {
        if (site_channels[s] < 1 || !(site_keep[s] <= 1.f)) return fail(RCU_ERR_INVALID, "rcu_dropout_masks: site_channels >= 1, site_keep <= 1");
        end += (long)n * site_channels[s];
        if (end * (long)passes >= (1l << 31)) return fail(RCU_ERR_INVALID, "rcu_dropout_masks: more than 2^31 factors");
        a.site_end[s] = (int)end;
        a.site_keep[s] = site_keep[s];
        a.site_ch[s] = site_channels[s];
        a.site_off[s] = a.per_sample;
        a.per_sample += site_channels[s];
    }
    a.first_sample = first_sample;
    a.sites = n_sites;
    a.per_pass = (int)end;
    a.passes = passes;
    // MASK_MAX_PASSES seeds travel as kernel arguments: a longer group takes several launches, each writing its passes' rows of `out`
    for (a.first = 0; a.first < passes; a.first += MASK_MAX_PASSES) {
        a.count = std::min(MASK_MAX_PASSES, passes - a.first);
        for (int t = 0; t < a.count; ++t) a.seed[t] = seeds[a.first + t];
        RCU_HIP(launch_dropout_masks(a, out, static_cast<hipStream_t>(stream)));
    }
    return RCU_OK;
}

// ------------------------------------------------------------------------------------------------
// metric seam
// ------------------------------------------------------------------------------------------------
static_assert(sizeof(rcu_ece_result) == sizeof(EceResult), "ABI struct mismatch");
static_assert(RCU_MAX_BINS == MAX_BINS && RCU_MAX_THRESHOLDS == MAX_THR, "ABI constant mismatch");

extern "C" int rcu_ece_thresholds(int n_bins, float* thr)
{
    if (n_bins < 1 || n_bins > MAX_BINS || !thr) return fail(RCU_ERR_INVALID, "n_bins must be in 1..32");
    // edges = np.linspace(0, 1 + 1e-8, n_bins + 1): start + k * step with step = (stop - start) / n_bins
    const double stop = 1.0 + 1e-8, step = stop / n_bins;
    for (int k = 1; k < n_bins; ++k) {
        const double edge = k * step;
        float t = (float)edge;
        if ((double)t < edge) t = std::nextafterf(t, INFINITY);
        thr[k - 1] = t;
    }
    return RCU_OK;
}

extern "C" size_t rcu_ece_workspace_bytes(size_t n, int nv) { return ece_workspace_bytes(n, nv < 1 ? 1 : nv); }

extern "C" int rcu_ece_hist(const float* p, const uint8_t* target, const uint8_t* mask, size_t n, int n_volumes,
                            const float* thr, int n_bins, rcu_ece_result* result, void* workspace, void* stream)
{
    if (!result || !thr || (n && (!p || !target || !workspace)))
        return fail(RCU_ERR_INVALID, "rcu_ece_hist: null argument");
    if (n_bins < 1 || n_bins > MAX_BINS || n_volumes < 1) return fail(RCU_ERR_INVALID, "rcu_ece_hist: bad n_bins / n_volumes");
    RCU_HIP(launch_ece_hist(p, target, mask, n, n_volumes, thr, n_bins, reinterpret_cast<EceResult*>(result), workspace,
                            static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_ece_bin_ids(const float* p, size_t n, const float* thr, int n_bins, uint8_t* ids, void* stream)
{
    if ((n && (!p || !ids)) || !thr) return fail(RCU_ERR_INVALID, "rcu_ece_bin_ids: null argument");
    if (n_bins < 1 || n_bins > MAX_BINS) return fail(RCU_ERR_INVALID, "n_bins must be in 1..32");
    RCU_HIP(launch_bin_ids(p, n, thr, n_bins, ids, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_calib_set_blocks_per_workgroup(int ece_blocks, int unc_blocks)
{
    if (ece_blocks < 0 || unc_blocks < 0) return fail(RCU_ERR_INVALID, "rcu_calib_set_blocks_per_workgroup: negative block count");
    calib_set_blocks_per_workgroup(ece_blocks, unc_blocks);
    return RCU_OK;
}

extern "C" size_t rcu_unc_workspace_bytes(size_t n, int nv) { return unc_workspace_bytes(n, nv < 1 ? 1 : nv); }

extern "C" int rcu_unc_counts(const void* unc, int unc_is_f64, const uint8_t* prediction, const uint8_t* target,
                              const uint8_t* mask, size_t n, int n_volumes, const double* thr, int n_thr, uint64_t* counts,
                              void* workspace, void* stream)
{
    if (!counts || !thr || (n && (!unc || !prediction || !target || !workspace)))
        return fail(RCU_ERR_INVALID, "rcu_unc_counts: null argument");
    if (n_thr < 1 || n_thr > MAX_THR || n_volumes < 1)
        return fail(RCU_ERR_INVALID, "rcu_unc_counts: n_thr must be in 1..16 and n_volumes >= 1");
    RCU_HIP(launch_unc_counts(unc, unc_is_f64, prediction, target, mask, n, n_volumes, thr, n_thr,
                              reinterpret_cast<unsigned long long*>(counts), workspace, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_unc_from_p_num_thresholds(void) { return unc_from_p_num_thresholds(); }
extern "C" double rcu_unc_from_p_threshold(int i) { return unc_from_p_threshold(i); }
extern "C" int rcu_unc_from_p_supported(const double* thr, int n_thr) { return (thr && unc_from_p_supported(thr, n_thr)) ? 1 : 0; }
extern "C" int rcu_unc_from_p_exceeded(float p, const double* thr, int n_thr)
{
    if (!thr || n_thr < 1 || n_thr > MAX_THR) return -1;
    return unc_from_p_exceeded_host(p, thr, n_thr);
}
extern "C" size_t rcu_unc_from_p_workspace_bytes(size_t n, int nv) { return unc_from_p_workspace_bytes(n, nv < 1 ? 1 : nv); }

extern "C" int rcu_unc_counts_from_p(const float* p, const uint8_t* prediction, const uint8_t* target, const uint8_t* mask, size_t n,
                                     int n_volumes, const double* thr, int n_thr, uint64_t* counts, void* workspace, void* stream)
{
    if (!thr || !counts || !workspace || (n && (!p || !prediction || !target)))
        return fail(RCU_ERR_INVALID, "rcu_unc_counts_from_p: null argument");
    if (n_thr < 1 || n_thr > MAX_THR || n_volumes < 1) return fail(RCU_ERR_INVALID, "rcu_unc_counts_from_p: bad n_thr / n_volumes");
    if (!unc_from_p_supported(thr, n_thr))
        return fail(RCU_ERR_INVALID, "rcu_unc_counts_from_p: thresholds must be strictly ascending values of the built-in table (rcu_unc_from_p_threshold)");
    RCU_HIP(launch_unc_counts_from_p(p, prediction, target, mask, n, n_volumes, thr, n_thr, reinterpret_cast<unsigned long long*>(counts),
                                     workspace, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

extern "C" int rcu_normalised_entropy(const float* p_fg, size_t n, double* out64, float* out32, void* stream)
{
    if (n && !p_fg) return fail(RCU_ERR_INVALID, "rcu_normalised_entropy: null argument");
    RCU_HIP(launch_norm_entropy(p_fg, n, out64, out32, static_cast<hipStream_t>(stream)));
    return RCU_OK;
}

#ifdef RCU_EXPERIMENTS
// Experiment builds only (tools/cu_contention_probe.py): `wgs` workgroups that each need a CU of their own (96 KB of LDS) and hold it for `usec`
// microseconds -- a stand-in for a collective's kernel running beside the persistent conv kernels on an 8-GPU node.
namespace rcu {
__global__ __launch_bounds__(256) void hog_kernel(unsigned long long ticks, float* sink)
{
    extern __shared__ float hog_lds[];
    hog_lds[threadIdx.x] = (float)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && hog_lds[threadIdx.x] < 0.f) sink[0] = 1.f;
}
}  // namespace rcu
extern "C" __attribute__((visibility("default"))) int rcu_debug_hog(int wgs, int usec, void* stream)
{
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rcu::hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(rcu::hog_kernel, dim3(wgs), dim3(256), 96 * 1024, (hipStream_t)stream, (unsigned long long)usec * 100ull, (float*)nullptr);
    return (int)hipGetLastError();
}
#endif
