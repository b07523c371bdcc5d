// conv3x3 (pad 1) as an implicit GEMM on the fp32 matrix cores of gfx950 (CDNA4), with the
// reference's whole conv unit fused into one kernel:
//
//     out = relu( alpha_c * m_{n,c} * (W * x) + (alpha_c * b_c * m_{n,c} + beta_c) )
//
// which is Conv2d(3x3, pad 1)+bias -> Dropout2d -> BatchNorm2d(eval) -> ReLU of the reference
// (common/model/unet.py:8-23) with BN folded to (alpha, beta) and the Dropout2d factor m_{n,c}
// in {0, 1/(1-p)} supplied per (slice, channel).  Optional extras, all fused into the same pass:
//   * second output = 2x2 max-pool of the result                  (DownConv, unet.py:85-95)
//   * K split over two source tensors instead of torch.cat        (UpConv,   unet.py:118)
//   * nearest x2 up-sampling + conv3x3 (UpConv, unet.py:105; helpers.py:15) in its sub-pixel form: the
//     output pixel (2y+a, 2x+b) only sees the 2x2 low-resolution neighbourhood (y+a-1.., x+b-1..), so
//     the layer is four 2x2-tap convolutions on the LOW-resolution grid (one per parity class (a,b))
//     with tap weights pre-summed on the host: 4 instead of 9 taps, 2.25x fewer FLOPs executed, and
//     the zero padding of the up-sampled grid coincides with zero padding of the low-resolution grid.
//
// GEMM view: M = pixels (N*H*W), N = output channels, K = taps x Cin.  v_mfma_f32_32x32x2_f32:
// A operand = 32 pixels (one 4-row x 8-column patch), B operand = 32 output channels.  With that
// pixel->row map every lane ends up owning a 4x4 pixel patch of one output channel, so the 2x2
// max-pool needs no cross-lane traffic and every store instruction writes 2 x 128 contiguous bytes.
//
// Per workgroup (256 threads = 4 waves, 2-3 workgroups per CU): an input tile with a 1-pixel halo and
// the weight slice of one Cin chunk (8 channels) are staged through LDS (register-staged: the global
// loads of chunk k+2 are in flight while chunk k is multiplied); the LDS images are unpadded and
// swizzled so that the ds_read_b128 fragment reads are bank-conflict free (ConvTile::SWZ); the
// fragment reads of step s+1 are issued before the MFMAs of step s.  Exact fp32 (MFMA f32 == fmaf chain).
#include "rcu_kernels.h"

#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

namespace rcu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector type: copies of HIP's struct float4 become memcpy in the IR and pin staging arrays to scratch
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int TS_, int TH_, int TW_, int BN_, int KC_, int WM_, int WN_, int TAPS_, int DB_, int SWZ_ = 0>
struct ConvTile {
    static constexpr int TS = TS_, TH = TH_, TW = TW_, BN = BN_, KC = KC_, WM = WM_, WN = WN_, TAPS = TAPS_;
    static constexpr bool DB = DB_ != 0;                     // LDS double buffering: one barrier per Cin chunk
    // SWZ: rows of KC = 8 floats WITHOUT padding.  ds_read_b128 is served over 64 banks (16 slots of 16 bytes) in
    // four 16-lane groups, {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} per wave half (MI355X_MICROARCH.md, LDS): of
    // the lane's 4 x 8 pixel patch a group holds 4 columns of each of the 4 rows -- columns 0-3 of two rows, 4-7 of the
    // other two -- all reading the same half h of their pixel row.  Slot = 2 * (R * PITCH + x) + h has one parity, so
    // a group would squeeze 16 reads into 8 slots.  Storing half h of an ODD halo row R in position h ^ 1 sends rows
    // R, R+2 to one parity and R+1, R+3 to the other, and a halo pitch that is a multiple of 4 pixels keeps the two
    // rows of equal parity 8 slots apart: conflict-free with 2/3 of the LDS bytes and staging traffic of the padded
    // layout.  Weights: a group reads channels {0-3, 12-15, 20-27} (or the complement) of a 32-channel block; swapping
    // the halves for channels 16..31 (mod 32) separates them the same way.  The packed weight tiles carry that
    // permutation (rcu_api.hip).
    static constexpr bool SWZ = SWZ_ != 0;
    static constexpr int TAPW = (TAPS == 9) ? 3 : 2;        // taps per window row
    static constexpr int THREADS = 256;
    static constexpr int KCP = SWZ ? KC : KC + 4;           // row length in LDS (floats)
    // Halo row pitch in pixels.  A 32-pixel MFMA row block is a 4x8 patch; with a pitch = 8 (mod 16) the four
    // 4-pixel runs a ds_read_b128 lane group touches fall on distinct LDS slots (conflict-free A fragment reads).
    static constexpr int PITCH = SWZ ? (TW + 2 + 3) / 4 * 4 : ((TW == 16) ? 24 : TW + 2);
    static constexpr int HW_ = (TH + 2) * PITCH;            // halo pixels per slice tile (incl. pitch padding)
    static constexpr int HPIX = TS * HW_;
    static constexpr int A_FLOATS = HPIX * KCP;
    static constexpr int W_FLOATS = TAPS * BN * KCP;
    static constexpr int BPS = (TH / 4) * (TW / 8);         // 32-pixel blocks per slice tile
    static constexpr int NBLK = TS * BPS;
    static constexpr int MT = NBLK / WM;                    // pixel blocks per wave
    static constexpr int NTW = BN / 32 / WN;                // channel blocks per wave
    static constexpr int HREAL = TS * (TH + 2) * (TW + 2);  // halo pixels that are really loaded
    static constexpr int A_UNITS = HREAL * (KC / 4);        // float4 units of the input tile
    static constexpr int NA = (A_UNITS + THREADS - 1) / THREADS;
    static constexpr int W_UNITS = W_FLOATS / 4;
    static constexpr int NW = (W_UNITS + THREADS - 1) / THREADS;
    // Both regions are rounded up to a whole number of float4 per thread so that staging is branch-free
    // (the packed weight tiles in global memory carry the same padding).
    static constexpr int W_UNITS_PAD = NW * THREADS;
    static constexpr int A_DUMP = (A_FLOATS + 3) / 4 * 4;    // 16 float4 slots where the tail units land
    static constexpr int A_REGION = A_DUMP + 16 * 4;         // floats
    // the LDS image holds exactly W_UNITS (the last staging round is predicated); only the packed tiles in global
    // memory are padded to W_UNITS_PAD
    static constexpr int BUF_FLOATS = A_REGION + W_UNITS * 4;
    static constexpr int LDS_BYTES = BUF_FLOATS * 4 * (DB ? 2 : 1);
    // resident workgroups per CU the streaming kernel is launched for (160 KB of LDS, <= 168 VGPRs for 3)
    static constexpr int WGS_PER_CU = (3 * LDS_BYTES <= 160 * 1024) ? 3 : 2;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(NBLK % WM == 0 && (BN / 32) % WN == 0, "wave tiling");
    static_assert(TH % 4 == 0 && TW % 8 == 0 && KC % 8 == 0, "block geometry");
    static_assert(TAPS == 9 || TAPS == 4, "3x3 window or the 2x2 window of the sub-pixel up-conv");
    static_assert(!SWZ || KC == 8, "the swizzle is defined for two 16-byte units per row");
    // float offset of (halo row R of the tile image, column x, unit sub) / of (output channel c of the tile, unit sub)
    static __device__ __forceinline__ int a_off(int R, int x, int sub) { return (R * PITCH + x) * KCP + (SWZ ? (sub ^ (R & 1)) : sub) * 4; }
    static __device__ __forceinline__ int b_off(int c, int sub) { return c * KCP + (SWZ ? (sub ^ ((c >> 4) & 1)) : sub) * 4; }
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

// Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  The `group` consecutive work
// items of one pixel tile (its channel tiles / sub-pixel classes) read the same input tile, so renumber the
// workgroups such that those run on ONE XCD -- the tile then comes from HBM once instead of once per XCD -- while
// successive groups still go round the XCDs (a partial last round stays spread over the whole chip).
__device__ __forceinline__ int xcd_virtual_block(int group)
{
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    if ((g & 7) != 0 || ((g >> 3) % group) != 0) return b;
    const int xcd = b & 7, slot = b >> 3;
    return ((slot / group) * 8 + xcd) * group + slot % group;
}

// Epilogue shared by both kernels: lane = output channel (lane & 31), 4x4 pixel patch per lane and pixel block.
//   out = relu(acc * alpha * mask + betab * mask + beta); optional 2x2 max-pool (lane-local).
// Output grid: (H, W), or (2H, 2W) with the lane's pixels at (2y+a, 2x+b) in sub-pixel mode.
template <class T>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x16 (&acc)[T::MT][T::NTW], int ntile, int pa, int pb,
                                              int n0, int y0, int x0, int wm, int wn, int m, int half)
{
    constexpr int MT = T::MT, NTW = T::NTW;
    constexpr bool SUBPIXEL = (T::TAPS == 4);
    constexpr int OS = SUBPIXEL ? 2 : 1;
    // the output tensor: the grid itself, or (sub-pixel mode with a centre pad) a larger zero-initialised tensor around it
    const int OH = (SUBPIXEL && a.out_H > 0) ? a.out_H : a.H * OS, OW = (SUBPIXEL && a.out_W > 0) ? a.out_W : a.W * OS;
    const int oy0 = SUBPIXEL ? a.out_y0 : 0, ox0 = SUBPIXEL ? a.out_x0 : 0;
    const bool full_tile = (y0 + T::TH <= a.H) && (x0 + T::TW <= a.W) && (n0 + T::TS <= a.N);
    const size_t row_stride = (size_t)OW * a.CoutP * OS, col_stride = (size_t)a.CoutP * OS;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    // the pooled tensor may belong to a padded level (ConvArgs::pool_H / pool_W, rcu_api.hip choose_level_extents): its own extents, our pixels
    const int PHs = a.pool_H > 0 ? a.pool_H : Hp, PWs = a.pool_W > 0 ? a.pool_W : Wp;
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
        const int co = ntile * T::BN + (wn * NTW + ni) * 32 + m;
        if (co >= a.CoutP) continue;
        const float al = a.alpha[co], bb = a.betab[co], be = a.beta[co];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int blk = wm * MT + mi;
            const int s = blk / T::BPS, rb = blk % T::BPS;
            const int by = rb / (T::TW / 8), bx = rb % (T::TW / 8);
            const int n = n0 + s;
            if (n >= a.N) continue;
            float mk = 1.f;
            if (a.mask != nullptr && co < a.Cmask) mk = a.mask[(size_t)n * a.Cmask + co];
            if (a.mask2 != nullptr && co >= a.Csplit && co - a.Csplit < a.Cmask2)
                mk = a.mask2[(size_t)n * a.Cmask2 + (co - a.Csplit)];
            const float scale = al * mk, shift = bb * mk + be;
            const int yb = y0 + 4 * by, xb = x0 + 8 * bx + 4 * half;
            float* const obase = a.out + ((size_t)(n * OH + oy0 + yb * OS + pa) * OW + ox0 + xb * OS + pb) * a.CoutP + co;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float t = acc[mi][ni][i] * scale + shift;
                if (a.accumulate && yb + (i >> 2) < a.H && xb + (i & 3) < a.W) t += obase[(i >> 2) * row_stride + (i & 3) * col_stride];
                v[i] = a.relu ? fmaxf(t, 0.f) : t;
            }
            if (full_tile) {
#pragma unroll
                for (int i = 0; i < 16; ++i) obase[(i >> 2) * row_stride + (i & 3) * col_stride] = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (yb + (i >> 2) < a.H && xb + (i & 3) < a.W)
                        obase[(i >> 2) * row_stride + (i & 3) * col_stride] = v[i];
            }
            if (!SUBPIXEL && a.pooled != nullptr) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const int i0 = (2 * pr) * 4 + 2 * pc;
                        const float mx = fmaxf(fmaxf(v[i0], v[i0 + 1]), fmaxf(v[i0 + 4], v[i0 + 5]));
                        const int py = (yb >> 1) + pr, px = (xb >> 1) + pc;
                        if (py < Hp && px < Wp) a.pooled[((size_t)(n * PHs + py) * PWs + px) * a.CoutP + co] = mx;
                    }
            }
        }
    }
}

template <class T>
__global__ __launch_bounds__(256, 2) void conv_igemm(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KC = T::KC, KCP = T::KCP, MT = T::MT, NTW = T::NTW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % T::WN;
    const int wm = wave / T::WN;

    // ---- workgroup -> (channel tile [, parity class], pixel tile), channel tile fastest; xcd_virtual_block
    // puts the workgroups that share one input tile on one XCD.
    const int bid = xcd_virtual_block(a.NTW_total < 4 ? 4 : a.NTW_total);   // >= 4: x-neighbours share halo columns
    const int wtile = bid % a.NTW_total;        // weight tile index = cls * NT + ntile
    const int ntile = wtile % a.NT;
    const int cls = wtile / a.NT;               // parity class (sub-pixel mode), else 0
    const int pa = cls >> 1, pb = cls & 1;
    int mtile = bid / a.NTW_total;
    const int tx = mtile % a.tiles_x;
    mtile /= a.tiles_x;
    const int ty = mtile % a.tiles_y;
    const int sg = mtile / a.tiles_y;
    const int n0 = sg * T::TS, y0 = ty * T::TH, x0 = tx * T::TW;

    // ---- per-thread staging plan for the input tile (fixed over the K loop).  Units outside the
    // image (zero padding) or outside the tile read element 0 and are multiplied by 0.
    uint32_t off1[T::NA], off2[T::NA];
    int adst[T::NA];
    float akeep[T::NA];
#pragma unroll
    for (int j = 0; j < T::NA; ++j) {
        const int u = tid + j * T::THREADS;
        const int q = u / (KC / 4);
        const int sub = u % (KC / 4);
        const int s = q / ((T::TH + 2) * (T::TW + 2));
        const int rem = q % ((T::TH + 2) * (T::TW + 2));
        const int yy = rem / (T::TW + 2), xx = rem % (T::TW + 2);
        const int n = n0 + s, gy = y0 + yy - 1, gx = x0 + xx - 1;
        const bool ok = u < T::A_UNITS && n < a.N && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const uint32_t pix = ok ? (uint32_t)((n * a.H + gy) * a.W + gx) : 0u;
        akeep[j] = ok ? 1.f : 0.f;
        adst[j] = u < T::A_UNITS ? T::a_off(s * (T::TH + 2) + yy, xx, sub) : T::A_DUMP + (tid & 15) * 4;   // tail units: dump slots
        off1[j] = ok ? pix * (uint32_t)a.C1 + sub * 4 : 0u;
        off2[j] = ok ? pix * (uint32_t)a.C2 + sub * 4 : 0u;
    }

    const int nchunks = (a.C1 + a.C2) / KC;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack) + (size_t)wtile * T::W_UNITS_PAD + tid;
    const size_t wchunk_stride = (size_t)a.NTW_total * T::W_UNITS_PAD;   // float4 units per Cin chunk

    f32x4 ra[T::NA], rw[T::NW];
#define RCU_PREFETCH(kc_)                                                                         \
    {                                                                                             \
        const int c0_ = (kc_) * KC;                                                               \
        const bool first_ = c0_ < a.C1;                                                           \
        const float* sp_ = first_ ? a.src1 + c0_ : a.src2 + (c0_ - a.C1);                         \
        _Pragma("unroll") for (int j = 0; j < T::NA; ++j)                                         \
            ra[j] = *reinterpret_cast<const f32x4*>(sp_ + (first_ ? off1[j] : off2[j]));          \
        const f32x4* wq_ = wp + (size_t)(kc_) * wchunk_stride;                                    \
        _Pragma("unroll") for (int j = 0; j < T::NW; ++j) rw[j] = wq_[j * T::THREADS];            \
    }
#define RCU_STAGE(buf_)                                                                           \
    {                                                                                             \
        float* const As_ = smem + (buf_) * T::BUF_FLOATS;                                         \
        _Pragma("unroll") for (int j = 0; j < T::NA; ++j)                                         \
            *reinterpret_cast<f32x4*>(As_ + adst[j]) = ra[j] * akeep[j];                          \
        _Pragma("unroll") for (int j = 0; j < T::NW; ++j)                                         \
            if ((j + 1) * T::THREADS <= T::W_UNITS || tid + j * T::THREADS < T::W_UNITS)          \
                reinterpret_cast<f32x4*>(As_ + T::A_REGION)[tid + j * T::THREADS] = rw[j];        \
    }

    // ---- fragment addresses (float offsets into As / Ws)
    const int m = lane & 31, half = lane >> 5;
    int a_addr[2][MT], b_addr[NTW];   // a_addr[p]: for window rows of parity p (the swizzle follows the halo row)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int blk = wm * MT + mi;
        const int s = blk / T::BPS, rb = blk % T::BPS;
        const int by = rb / (T::TW / 8), bx = rb % (T::TW / 8);
        // window origin of this lane's pixel inside the halo tile; the sub-pixel classes shift it by (a, b)
        const int R0 = s * (T::TH + 2) + 4 * by + (m >> 3) + pa, x0l = 8 * bx + (m & 7) + pb;
        a_addr[0][mi] = T::a_off(R0, x0l, half);
        a_addr[1][mi] = T::a_off(R0 + 1, x0l, half) - T::PITCH * KCP;
    }
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) b_addr[ni] = T::b_off((wn * NTW + ni) * 32 + m, half);

    f32x16 acc[MT][NTW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    // K loop over Cin chunks.
    //  single-buffered: barrier / regs -> LDS / barrier / issue the loads of chunk k+1 / multiply chunk k
    //  double-buffered: multiply chunk k out of buffer k&1 while this wave's share of chunk k+1 goes
    //                   regs -> LDS into the other buffer and the loads of chunk k+2 are issued; ONE barrier
    //                   per chunk, and no wave ever waits for its own LDS writes before issuing MFMAs.
    constexpr int STEPS = T::TAPS * (KC / 8);
    RCU_PREFETCH(0);
    if (T::DB) {
        RCU_STAGE(0);
        if (nchunks > 1) RCU_PREFETCH(1);
        __syncthreads();
    }
    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur_buf = T::DB ? (kc & 1) : 0;
        if (!T::DB) {
            __syncthreads();   // previous chunk fully consumed
            RCU_STAGE(0);
            __syncthreads();
            if (kc + 1 < nchunks) RCU_PREFETCH(kc + 1);   // lands while this chunk is multiplied
        }
        const float* const Ac = smem + cur_buf * T::BUF_FLOATS;
        const float* const Wc = Ac + T::A_REGION;
        // TAPS x KC/8 steps, software pipelined: the fragments of step s+1 are read from LDS before the
        // MFMAs of step s are issued, so one wave alone keeps its SIMD's matrix pipe busy.
        f32x4 av[2][MT], bv[2][NTW];
#define RCU_FRAGS(step_, buf_)                                                                              \
        {                                                                                                       \
            constexpr int tap_ = (step_) / (KC / 8), k8_ = (step_) % (KC / 8);                                  \
            constexpr int tapA_ = ((tap_ / T::TAPW) * T::PITCH + (tap_ % T::TAPW)) * KCP + k8_ * 8;             \
            constexpr int tapB_ = tap_ * T::BN * KCP + k8_ * 8;                                                 \
            _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                                                   \
                av[buf_][mi] = *reinterpret_cast<const f32x4*>(Ac + a_addr[(tap_ / T::TAPW) & 1][mi] + tapA_);  \
            _Pragma("unroll") for (int ni = 0; ni < NTW; ++ni)                                                  \
                bv[buf_][ni] = *reinterpret_cast<const f32x4*>(Wc + b_addr[ni] + tapB_);                        \
        }
        RCU_FRAGS(0, 0);
        static_for<0, STEPS>([&](auto step_c) {
            constexpr int step = decltype(step_c)::value;
            constexpr int cur = step & 1;
            if constexpr (step + 1 < STEPS) RCU_FRAGS(step + 1, cur ^ 1);
            if constexpr (T::DB && step == 0) {
                // behind the first fragment reads, ahead of the MFMAs: next chunk regs -> other LDS buffer,
                // then refill the registers with the chunk after that
                if (kc + 1 < nchunks) RCU_STAGE(cur_buf ^ 1);
                if (kc + 2 < nchunks) RCU_PREFETCH(kc + 2);
            }
            // pin the order: hipcc's scheduler otherwise sinks every ds_read down to its first use (lgkmcnt(0)
            // right before each MFMA group), which serialises LDS latency with the matrix pipe
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NTW; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi].x, bv[cur][ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi].y, bv[cur][ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi].z, bv[cur][ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi].w, bv[cur][ni].w, acc[mi][ni], 0, 0, 0);
                }
        });
#undef RCU_FRAGS
        if (T::DB) __syncthreads();   // everyone done with buffer k&1 and with writing buffer (k+1)&1
    }
#undef RCU_PREFETCH
#undef RCU_STAGE

    conv_epilogue<T>(a, acc, ntile, pa, pb, n0, y0, x0, wm, wn, m, half);
}


// ------------------------------------------------------------------------------------------------
// Streaming variant (double-buffered tiles only): a workgroup walks over several output tiles
// (items g, g + G, g + 2G, ... of the flat tile list) and runs ONE software pipeline across all of
// their Cin chunks: while chunk i is multiplied, chunk i+1 goes registers -> other LDS buffer and the
// loads of chunk i+2 are issued -- also when i+1 / i+2 belong to the NEXT tile.  The first-load latency,
// the index arithmetic and the workgroup launch of every tile but the first are hidden behind MFMAs.
// ------------------------------------------------------------------------------------------------
template <class T>
struct TilePlan {
    uint32_t off1[T::NA], off2[T::NA];
    float akeep[T::NA];
    int wtile, ntile, pa, pb, n0, y0, x0;
};

template <class T>
__device__ __forceinline__ void make_plan(TilePlan<T>& p, const ConvArgs& a, int item, int tid)
{
    p.wtile = item % a.NTW_total;
    p.ntile = p.wtile % a.NT;
    const int cls = p.wtile / a.NT;
    p.pa = cls >> 1;
    p.pb = cls & 1;
    int mtile = item / a.NTW_total;
    const int tx = mtile % a.tiles_x;
    mtile /= a.tiles_x;
    const int ty = mtile % a.tiles_y;
    const int sg = mtile / a.tiles_y;
    p.n0 = sg * T::TS;
    p.y0 = ty * T::TH;
    p.x0 = tx * T::TW;
#pragma unroll
    for (int j = 0; j < T::NA; ++j) {
        const int u = tid + j * T::THREADS;
        const int q = u / (T::KC / 4);
        const int sub = u % (T::KC / 4);
        const int s = q / ((T::TH + 2) * (T::TW + 2));
        const int rem = q % ((T::TH + 2) * (T::TW + 2));
        const int yy = rem / (T::TW + 2), xx = rem % (T::TW + 2);
        const int n = p.n0 + s, gy = p.y0 + yy - 1, gx = p.x0 + xx - 1;
        const bool ok = u < T::A_UNITS && n < a.N && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const uint32_t pix = ok ? (uint32_t)((n * a.H + gy) * a.W + gx) : 0u;
        p.akeep[j] = ok ? 1.f : 0.f;
        p.off1[j] = ok ? pix * (uint32_t)a.C1 + sub * 4 : 0u;
        p.off2[j] = ok ? pix * (uint32_t)a.C2 + sub * 4 : 0u;
    }
}

template <class T>
__global__ __launch_bounds__(256, T::WGS_PER_CU) void conv_igemm_stream(const ConvArgs a, const int total_items)
{
    static_assert(T::DB, "streaming kernel needs the double-buffered LDS layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KC = T::KC, KCP = T::KCP, MT = T::MT, NTW = T::NTW;
    constexpr int STEPS = T::TAPS * (KC / 8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % T::WN;
    const int wm = wave / T::WN;
    const int m = lane & 31, half = lane >> 5;
    const int nchunks = (a.C1 + a.C2) / KC;   // >= 2 (checked by the launcher)
    const size_t wchunk_stride = (size_t)a.NTW_total * T::W_UNITS_PAD;
    const f32x4* const wbase = reinterpret_cast<const f32x4*>(a.wpack) + tid;

    // LDS destination of every staging unit is tile independent
    int adst[T::NA];
#pragma unroll
    for (int j = 0; j < T::NA; ++j) {
        const int u = tid + j * T::THREADS;
        const int q = u / (KC / 4), sub = u % (KC / 4);
        const int s = q / ((T::TH + 2) * (T::TW + 2));
        const int rem = q % ((T::TH + 2) * (T::TW + 2));
        adst[j] = u < T::A_UNITS ? T::a_off(s * (T::TH + 2) + rem / (T::TW + 2), rem % (T::TW + 2), sub)
                                 : T::A_DUMP + (tid & 15) * 4;
    }
    int b_addr[NTW];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) b_addr[ni] = T::b_off((wn * NTW + ni) * 32 + m, half);

    TilePlan<T> cur, nxt;
    int item = xcd_virtual_block(a.NTW_total < 4 ? 4 : a.NTW_total);
    make_plan<T>(cur, a, item, tid);
    bool has_next = item + (int)gridDim.x < total_items;
    make_plan<T>(nxt, a, has_next ? item + (int)gridDim.x : item, tid);

    f32x4 ra[T::NA], rw[T::NW];
    float rkeep[T::NA];   // zero-padding factors of the data currently held in ra
#define RCU_PREFETCH_P(plan_, kc_)                                                                 \
    {                                                                                              \
        const int c0_ = (kc_) * KC;                                                                \
        const bool first_ = c0_ < a.C1;                                                            \
        const float* sp_ = first_ ? a.src1 + c0_ : a.src2 + (c0_ - a.C1);                          \
        _Pragma("unroll") for (int j = 0; j < T::NA; ++j) {                                        \
            ra[j] = *reinterpret_cast<const f32x4*>(sp_ + (first_ ? plan_.off1[j] : plan_.off2[j])); \
            rkeep[j] = plan_.akeep[j];                                                             \
        }                                                                                          \
        const f32x4* wq_ = wbase + (size_t)plan_.wtile * T::W_UNITS_PAD + (size_t)(kc_) * wchunk_stride; \
        _Pragma("unroll") for (int j = 0; j < T::NW; ++j) rw[j] = wq_[j * T::THREADS];             \
    }
#define RCU_STAGE_P(buf_)                                                                          \
    {                                                                                              \
        float* const As_ = smem + (buf_) * T::BUF_FLOATS;                                          \
        _Pragma("unroll") for (int j = 0; j < T::NA; ++j)                                          \
            *reinterpret_cast<f32x4*>(As_ + adst[j]) = ra[j] * rkeep[j];                           \
        _Pragma("unroll") for (int j = 0; j < T::NW; ++j)                                          \
            if ((j + 1) * T::THREADS <= T::W_UNITS || tid + j * T::THREADS < T::W_UNITS)           \
                reinterpret_cast<f32x4*>(As_ + T::A_REGION)[tid + j * T::THREADS] = rw[j];         \
    }

    RCU_PREFETCH_P(cur, 0);
    RCU_STAGE_P(0);
    RCU_PREFETCH_P(cur, 1);
    __syncthreads();

    int it = 0;   // flat chunk counter: LDS buffer = it & 1
    for (;;) {
        int a_addr[2][MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int blk = wm * MT + mi;
            const int s = blk / T::BPS, rb = blk % T::BPS;
            const int by = rb / (T::TW / 8), bx = rb % (T::TW / 8);
            const int R0 = s * (T::TH + 2) + 4 * by + (m >> 3) + cur.pa, x0l = 8 * bx + (m & 7) + cur.pb;
            a_addr[0][mi] = T::a_off(R0, x0l, half);
            a_addr[1][mi] = T::a_off(R0 + 1, x0l, half) - T::PITCH * KCP;
        }
        f32x16 acc[MT][NTW];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

        for (int kc = 0; kc < nchunks; ++kc, ++it) {
            const int cur_buf = it & 1;
            const float* const Ac = smem + cur_buf * T::BUF_FLOATS;
            const float* const Wc = Ac + T::A_REGION;
            f32x4 av[2][MT], bv[2][NTW];
#define RCU_FRAGS(step_, buf_)                                                                              \
            {                                                                                                   \
                constexpr int tap_ = (step_) / (KC / 8), k8_ = (step_) % (KC / 8);                              \
                constexpr int tapA_ = ((tap_ / T::TAPW) * T::PITCH + (tap_ % T::TAPW)) * KCP + k8_ * 8;         \
                constexpr int tapB_ = tap_ * T::BN * KCP + k8_ * 8;                                             \
                _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                                               \
                    av[buf_][mi] = *reinterpret_cast<const f32x4*>(Ac + a_addr[(tap_ / T::TAPW) & 1][mi] + tapA_); \
                _Pragma("unroll") for (int ni = 0; ni < NTW; ++ni)                                              \
                    bv[buf_][ni] = *reinterpret_cast<const f32x4*>(Wc + b_addr[ni] + tapB_);                    \
            }
            RCU_FRAGS(0, 0);
            static_for<0, STEPS>([&](auto step_c) {
                constexpr int step = decltype(step_c)::value;
                constexpr int cb = step & 1;
                if constexpr (step + 1 < STEPS) RCU_FRAGS(step + 1, cb ^ 1);
                if constexpr (step == 0) {
                    // chunk it+1 (held in registers) -> other buffer; then load chunk it+2.  Both may belong
                    // to the next tile of this workgroup.
                    if (kc + 1 < nchunks || has_next) RCU_STAGE_P(cur_buf ^ 1);
                    if (kc + 2 < nchunks) {
                        RCU_PREFETCH_P(cur, kc + 2);
                    } else if (has_next) {
                        RCU_PREFETCH_P(nxt, kc + 2 - nchunks);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][mi].x, bv[cb][ni].x, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][mi].y, bv[cb][ni].y, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][mi].z, bv[cb][ni].z, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][mi].w, bv[cb][ni].w, acc[mi][ni], 0, 0, 0);
                    }
            });
#undef RCU_FRAGS
            __syncthreads();   // everyone done with buffer it&1 and with writing buffer (it+1)&1
        }

        // ---- epilogue of the finished tile (registers only; the next tile's first chunk is already in LDS)
        conv_epilogue<T>(a, acc, cur.ntile, cur.pa, cur.pb, cur.n0, cur.y0, cur.x0, wm, wn, m, half);

        if (!has_next) break;
        // advance: next tile becomes current; plan the one after it
        item += (int)gridDim.x;
#pragma unroll
        for (int j = 0; j < T::NA; ++j) {
            cur.off1[j] = nxt.off1[j];
            cur.off2[j] = nxt.off2[j];
            cur.akeep[j] = nxt.akeep[j];
        }
        cur.wtile = nxt.wtile; cur.ntile = nxt.ntile; cur.pa = nxt.pa; cur.pb = nxt.pb;
        cur.n0 = nxt.n0; cur.y0 = nxt.y0; cur.x0 = nxt.x0;
        has_next = item + (int)gridDim.x < total_items;
        make_plan<T>(nxt, a, has_next ? item + (int)gridDim.x : item, tid);
    }
#undef RCU_PREFETCH_P
#undef RCU_STAGE_P
}

using Cfg0 = ConvTile<1, 8, 16, 64, 8, 2, 2, 9, 1, 1>;
using Cfg1 = ConvTile<1, 8, 16, 32, 8, 4, 1, 9, 1, 1>;
using Cfg2 = ConvTile<1, 8, 16, 32, 8, 4, 1, 9, 0, 1>;
using Cfg3 = ConvTile<2, 12, 8, 64, 8, 2, 2, 9, 1, 1>;
using Cfg4 = ConvTile<1, 8, 16, 64, 8, 2, 2, 4, 1, 1>;
using Cfg5 = ConvTile<1, 8, 16, 32, 8, 4, 1, 4, 1, 1>;
using Cfg6 = ConvTile<2, 12, 8, 64, 8, 2, 2, 4, 1, 1>;
using Cfg7 = ConvTile<1, 16, 16, 32, 8, 4, 1, 9, 1, 1>;
using Cfg8 = ConvTile<1, 16, 16, 32, 8, 4, 1, 4, 1, 1>;
using Cfg9 = ConvTile<1, 16, 16, 64, 8, 2, 2, 9, 1, 1>;
using Cfg10 = ConvTile<2, 8, 16, 64, 8, 2, 2, 9, 1, 1>;

static const ConvConfigInfo kInfo[CONV_CFG_COUNT] = {
    {Cfg0::TS, Cfg0::TH, Cfg0::TW, Cfg0::BN, Cfg0::KC, Cfg0::TAPS, "conv3x3_igemm<T8x16,N64,K8,db>", Cfg0::KCP, Cfg0::SWZ ? 1 : 0},
    {Cfg1::TS, Cfg1::TH, Cfg1::TW, Cfg1::BN, Cfg1::KC, Cfg1::TAPS, "conv3x3_igemm<T8x16,N32,K8,db>", Cfg1::KCP, Cfg1::SWZ ? 1 : 0},
    {Cfg2::TS, Cfg2::TH, Cfg2::TW, Cfg2::BN, Cfg2::KC, Cfg2::TAPS, "conv3x3_igemm<T8x16,N32,K8>", Cfg2::KCP, Cfg2::SWZ ? 1 : 0},
    {Cfg3::TS, Cfg3::TH, Cfg3::TW, Cfg3::BN, Cfg3::KC, Cfg3::TAPS, "conv3x3_igemm<S2T12x8,N64,K8,db>", Cfg3::KCP, Cfg3::SWZ ? 1 : 0},
    {Cfg4::TS, Cfg4::TH, Cfg4::TW, Cfg4::BN, Cfg4::KC, Cfg4::TAPS, "upconv_subpixel_igemm<T8x16,N64,K8,db>", Cfg4::KCP, Cfg4::SWZ ? 1 : 0},
    {Cfg5::TS, Cfg5::TH, Cfg5::TW, Cfg5::BN, Cfg5::KC, Cfg5::TAPS, "upconv_subpixel_igemm<T8x16,N32,K8,db>", Cfg5::KCP, Cfg5::SWZ ? 1 : 0},
    {Cfg6::TS, Cfg6::TH, Cfg6::TW, Cfg6::BN, Cfg6::KC, Cfg6::TAPS, "upconv_subpixel_igemm<S2T12x8,N64,K8,db>", Cfg6::KCP, Cfg6::SWZ ? 1 : 0},
    {Cfg7::TS, Cfg7::TH, Cfg7::TW, Cfg7::BN, Cfg7::KC, Cfg7::TAPS, "conv3x3_igemm<T16x16,N32,K8,db>", Cfg7::KCP, Cfg7::SWZ ? 1 : 0},
    {Cfg8::TS, Cfg8::TH, Cfg8::TW, Cfg8::BN, Cfg8::KC, Cfg8::TAPS, "upconv_subpixel_igemm<T16x16,N32,K8,db>", Cfg8::KCP, Cfg8::SWZ ? 1 : 0},
    {Cfg9::TS, Cfg9::TH, Cfg9::TW, Cfg9::BN, Cfg9::KC, Cfg9::TAPS, "conv3x3_igemm<T16x16,N64,K8,db>", Cfg9::KCP, Cfg9::SWZ ? 1 : 0},
    {Cfg10::TS, Cfg10::TH, Cfg10::TW, Cfg10::BN, Cfg10::KC, Cfg10::TAPS, "conv3x3_igemm<S2T8x16,N64,K8,db>", Cfg10::KCP, Cfg10::SWZ ? 1 : 0},
};

hipError_t set_max_dynamic_lds(const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({kernel, dev});
    return e;
}

const ConvConfigInfo& conv_config_info(int cfg)
{
    if (cfg >= CONV_CFG_WINO4_T32x32_N32) return wino4_config_info(cfg);
    if (cfg == CONV_CFG_FIRST_T8x32) return first_config_info();
    if (cfg >= CONV_CFG_UPW_T16x16_N64) return wino_up_config_info(cfg);
    return cfg >= CONV_CFG_COUNT ? wino_config_info(cfg) : kInfo[cfg];
}

template <class T>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t stream)
{
    const unsigned items = (unsigned)a.NTW_total * a.tiles_x * a.tiles_y * a.slice_groups;
    if constexpr (T::DB) {
        if ((a.C1 + a.C2) / T::KC >= 2) {
            hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_stream<T>), T::LDS_BYTES);
            if (e != hipSuccess) return e;
            // WGS_PER_CU resident workgroups per CU, each streaming through its share of the tiles
#ifdef RCU_EXPERIMENTS
            static const unsigned slots = [] {
                const char* e = getenv("RCU_CONV_WGS");
                return 256u * (unsigned)(e ? atoi(e) : T::WGS_PER_CU);
            }();
#else
            constexpr unsigned slots = 256u * (unsigned)T::WGS_PER_CU;
#endif
            const unsigned grid = items < slots ? items : slots;
            hipLaunchKernelGGL(conv_igemm_stream<T>, dim3(grid), dim3(T::THREADS), T::LDS_BYTES, stream, a, (int)items);
            return hipGetLastError();
        }
    }
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm<T>), T::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_igemm<T>, dim3(items), dim3(T::THREADS), T::LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t launch_conv3x3(int cfg, const ConvArgs& a, hipStream_t stream)
{
    if (cfg >= CONV_CFG_WINO4_T32x32_N32) return launch_conv_wino4(cfg, a, stream);
    if (cfg == CONV_CFG_FIRST_T8x32) return launch_conv_first(a, stream);
    if (cfg >= CONV_CFG_UPW_T16x16_N64) return launch_upconv_wino(cfg, a, stream);
    if (cfg >= CONV_CFG_COUNT) return launch_conv_wino(cfg, a, stream);
    switch (cfg) {
        case CONV_CFG_T8x16_N64: return launch_cfg<Cfg0>(a, stream);
        case CONV_CFG_T8x16_N32: return launch_cfg<Cfg1>(a, stream);
        case CONV_CFG_T8x16_N32_FIRST: return launch_cfg<Cfg2>(a, stream);
        case CONV_CFG_S2T12x8_N64: return launch_cfg<Cfg3>(a, stream);
        case CONV_CFG_UP_T8x16_N64: return launch_cfg<Cfg4>(a, stream);
        case CONV_CFG_UP_T8x16_N32: return launch_cfg<Cfg5>(a, stream);
        case CONV_CFG_UP_S2T12x8_N64: return launch_cfg<Cfg6>(a, stream);
        case CONV_CFG_T16x16_N32: return launch_cfg<Cfg7>(a, stream);
        case CONV_CFG_UP_T16x16_N32: return launch_cfg<Cfg8>(a, stream);
        case CONV_CFG_T16x16_N64: return launch_cfg<Cfg9>(a, stream);
        case CONV_CFG_S2T8x16_N64: return launch_cfg<Cfg10>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rcu
