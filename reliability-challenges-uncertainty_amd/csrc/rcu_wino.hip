// conv3x3 (pad 1) conv unit in Winograd F(2x2, 3x3) form on the fp32 matrix cores of gfx950.
//
//     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 2x2 output tile, 4x4 input patch d
//
// 16 multiplications per 4 output pixels and (cin, cout) pair instead of 36: the channel contraction -- the only dense
// part -- is 2.25x smaller than in rcu_conv.hip, and it is what runs on the MFMA pipe: one GEMM
// [tiles x Cin] * [Cin x Cout] per Winograd position p = 0..15.  Everything else of the reference's conv unit
// (common/model/unet.py:8-23: bias, Dropout2d factor, folded BatchNorm, ReLU, optional 2x2 max-pool of DownConv,
// unet.py:85-95, cat-free two-source K loop of UpConv, unet.py:118) is fused exactly as in rcu_conv.hip.
// fp32 throughout: measured against float64 the Winograd form is as accurate as the direct fp32 form here (the
// transforms only add/subtract; G's halves are exact), see DESIGN.md.
//
// Mapping (v_mfma_f32_16x16x4_f32):
//   * a wave owns 16 tiles (2 tile rows x 8 tile columns = 4 x 16 pixels) x 32 output channels x 16 positions
//     = 128 accumulator registers; a workgroup of 8 waves (2 per SIMD) owns WM tile blocks x WN channel blocks.
//   * A operand: lane (tile m = lane & 15, channel group kq = lane >> 4) reads its tile's 4x4 input patch for the
//     channel pair (2kq, 2kq+1) of the 8-channel chunk straight from the RAW input tile in LDS (16 ds_read_b64),
//     applies B^T d B in registers (64 adds) and feeds the 16 results per channel to the 16 positions' MFMAs: the
//     transformed input never exists in memory.
//   * B operand: host-transformed weights U = G g G^T, packed [chunk][cout tile][p][channel pair][cout][2] so that
//     one ds_read_b128 per position serves both 16-channel MFMA blocks of the wave (couts 2n, 2n+1 per lane).
//   * D: lane (n = lane & 15, g = lane >> 4) holds, for couts (2n, 2n+1), the four horizontally adjacent tiles
//     4(g & 1) .. +3 of tile row g >> 1: the output transform A^T M A, the epilogue and the 2x2 max-pool (= one tile)
//     are lane-local; neighbouring lanes then trade one pixel column (DPP), so that every store writes 16 bytes: 8 lanes
//     = one 128-B line per pixel.  With 8-pixel-wide images a 16-tile block is 2 tile rows x 4 tile columns of TWO
//     consecutive slices (WinoTile::SW) and the lane's four tiles are a whole row of one of them.
//   * staging by LDS-DMA only (`buffer_load_dwordx4 ... lds`, no staging registers, no ds_write): the LDS images are
//     lane-linear per wave instruction.  Input image: [channel half hh][slice][halo row R][position][4 channels] where
//     position = x ^ ((R >> 1) & 1) -- the swizzle is applied on the SOURCE address -- with a row pitch of 8k positions:
//     the 32 lanes a ds_read_b64 serves per cycle (2 channel pairs x 2 tile rows x 8 tile columns) then fall on 64
//     distinct banks.  Zero padding comes from the buffer resource: halo pixels outside the image carry an
//     out-of-range offset and the DMA writes zeros (tools/microbench/glds_oob_probe.hip).
//   * streaming: a workgroup walks over several output tiles and runs one double-buffered pipeline across all their
//     Cin chunks (chunk k+1 streams into the other LDS buffer while chunk k is multiplied, two DMA pieces per MFMA group
//     from the first group on; s_waitcnt vmcnt(0) + one barrier per chunk).  The per-lane DMA plan -- one packed geometry
//     register and one offset register per slot -- is recomputed for the next tile during the current tile's last chunk.
#include "rcu_head_common.h"
#include "rcu_wino_common.h"

#include <cstdlib>

namespace rcu {

// Output transform + conv-unit epilogue of one finished tile.  acc[b][p][r]: MFMA block b (cout 2n+b), position p,
// tile r of the lane's four.
// PART: the level is padded (ConvArgs::part) -- the tile may hang over the real image: pixels at or beyond (Hr, Wr) go out of range, where the
// buffer resource drops the write, and the output / pooled tensors have extents of their own.
template <class T, bool PART = false>
__device__ __forceinline__ void wino_epilogue(const ConvArgs& a, const f32x4 (&acc)[2][16], const WinoEpi& ep, int ntile, int n0, int y0,
                                              int x0, int wm, int wn, int lane)
{
    const int n16 = lane & 15, g = lane >> 4;
    const int co = ntile * T::BN + wn * 32 + 2 * n16;
    if (co >= a.CoutP) return;
    int bs, by, bx;
    T::block_origin(wm, bs, by, bx);
    // the lane's four tiles 4(g & 1) .. +3 of tile row g >> 1: columns 8(g & 1) .. of one slice, or (SW = 2) the whole
    // 8-pixel row of slice g & 1
    const int n = n0 + bs + (T::SW == 2 ? g & 1 : 0);
    if (n >= a.N) return;
    const int yb = y0 + by + 2 * (g >> 1);                       // top pixel row of the lane's tiles
    const int xb = x0 + bx + (T::SW == 2 ? 0 : 8 * (g & 1));     // left pixel column of the lane's first tile
    const float scale[2] = {ep.scale[0], ep.scale[1]}, shift[2] = {ep.shift[0], ep.shift[1]};
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();   // one v_max either way (a select per value otherwise)
    const int odd = n16 & 1;
    // Whole tiles only (checked by the launcher) and tensors below 2 GB: stores through a buffer resource, one per-lane byte offset
    // per tile, the steps between a lane's pixels in SGPRs -- no 64-bit address arithmetic, no per-store predicates.
    const int oH = PART ? a.out_H : a.H, oW = PART ? a.out_W : a.W;
    const uint32_t row_bytes = (uint32_t)oW * a.out_pix_bytes;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (uint32_t)(a.N * oH * oW) * (uint32_t)a.CoutP * 4u, 0x00020000);
    const uint32_t vo = wino_out_offset(n, oH * oW, a.CoutP, (uint32_t)(yb * oW + xb + odd), co - 2 * odd, a.out_pix_bytes, a.out_chunk_bytes);
    const uint32_t px_step = 2u * a.out_pix_bytes;   // two pixels to the right
    // PART: the lane stores pixels (yb + aa, xb + odd + 2 r): real iff aa < ylim && 2 r < xlim
    [[maybe_unused]] const int ylim = PART ? a.Hr - yb : 0, xlim = PART ? a.Wr - (xb + odd) : 0;
    f32x2 mx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f32x2 y[2][2];   // [row a][col bb], components = the two couts
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float nn[4][2];   // N[i][bb] = sum_j M[i][j] A[j][bb]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m0 = acc[b][4 * i + 0][r], m1 = acc[b][4 * i + 1][r], m2 = acc[b][4 * i + 2][r],
                            m3 = acc[b][4 * i + 3][r];
                nn[i][0] = m0 + m1 + m2;
                nn[i][1] = m1 - m2 - m3;
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const float v0 = nn[0][bb] + nn[1][bb] + nn[2][bb];
                const float v1 = nn[1][bb] - nn[2][bb] - nn[3][bb];
                const float t0 = v0 * scale[b] + shift[b], t1 = v1 * scale[b] + shift[b];
                y[0][bb][b] = fmaxf(t0, relu_floor);
                y[1][bb][b] = fmaxf(t1, relu_floor);
            }
        }
        // Neighbouring lanes (couts 2n, 2n+1 | 2n+2, 2n+3 of the same pixels) trade one pixel column each, so that the even
        // lane holds four couts of column x and the odd lane four couts of column x + 1: half as many store instructions
        // (one 16-byte store per lane and pixel row), which is what the epilogue's time goes into.
        f32x4 o[2];
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
            const f32x2 send = odd ? y[aa][0] : y[aa][1];
            f32x2 recv;
            recv.x = wino_swap_adjacent(send.x);
            recv.y = wino_swap_adjacent(send.y);
            o[aa] = odd ? f32x4{recv.x, recv.y, y[aa][1].x, y[aa][1].y} : f32x4{y[aa][0].x, y[aa][0].y, recv.x, recv.y};
        }
        wino_store16(o[0], ro, (PART && !(0 < ylim && 2 * r < xlim)) ? WINO_OOB : vo, r * px_step);
        wino_store16(o[1], ro, (PART && !(1 < ylim && 2 * r < xlim)) ? WINO_OOB : vo, r * px_step + row_bytes);
        if (a.pooled != nullptr) {
            mx[r].x = fmaxf(fmaxf(y[0][0].x, y[0][1].x), fmaxf(y[1][0].x, y[1][1].x));
            mx[r].y = fmaxf(fmaxf(y[0][0].y, y[0][1].y), fmaxf(y[1][0].y, y[1][1].y));
        }
    }
    if (a.pooled != nullptr) {
        // same trade for the pooled pixels (one per tile): the even lane stores four couts of tile r, the odd lane of tile r + 1
        const int Hp = PART ? a.pool_H : a.H >> 1, Wp = PART ? a.pool_W : a.W >> 1;
        // PART: pooled pixel ((yb >> 1), (xb >> 1) + odd + r) for r = 0, 2 of the pooled image of (Hr >> 1) x (Wr >> 1) real pixels
        [[maybe_unused]] const bool prow_in = PART ? (yb >> 1) < (a.Hr >> 1) : true;
        [[maybe_unused]] const int pxlim = PART ? (a.Wr >> 1) - ((xb >> 1) + odd) : 0;
        const __amdgpu_buffer_rsrc_t rp =
            __builtin_amdgcn_make_buffer_rsrc(a.pooled, 0, (uint32_t)(a.N * Hp * Wp * a.CoutP) * 4u, 0x00020000);
        const uint32_t vp = wino_out_offset(n, Hp * Wp, a.CoutP, (uint32_t)((yb >> 1) * Wp + (xb >> 1) + odd), co - 2 * odd, a.pool_pix_bytes,
                                            a.pool_chunk_bytes);
        const uint32_t pp_step = 2u * a.pool_pix_bytes;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2 send = odd ? mx[r] : mx[r + 1];
            f32x2 recv;
            recv.x = wino_swap_adjacent(send.x);
            recv.y = wino_swap_adjacent(send.y);
            const f32x4 o = odd ? f32x4{recv.x, recv.y, mx[r + 1].x, mx[r + 1].y} : f32x4{mx[r].x, mx[r].y, recv.x, recv.y};
            wino_store16(o, rp, (PART && !(prow_in && r < pxlim)) ? WINO_OOB : vp, (r >> 1) * pp_step);
        }
    }
}

// conv_cls.0 with the classifier behind it (common/model/unet.py:160-161, rechun/dl/customsteps.py:24,33): the finished
// 16x32-pixel x 32-channel tile goes to LDS instead of HBM ([pixel][34 floats]: conflict-free both ways), then every
// thread takes one pixel: 1x1 conv to two logits, and either the logits (NCHW) or softmax + the MC statistics update.
// The dot product is summed exactly as head_kernel sums it (eight 4-channel fmaf chains, pairwise tree), so a pass
// through this epilogue and a pass through head_kernel (rcu_unet_set_fuse_head(h, 0); sigma / feature outputs) give the same bits.
constexpr int WINO_HEAD_PITCH = 34;
// PART: the level is padded (ConvArgs::part) -- logits and statistics are the caller's arrays over the REAL Hr x Wr image.
template <class T, bool PART = false>
__device__ __forceinline__ void wino_epilogue_head(const ConvArgs& a, const f32x4 (&acc)[2][16], const WinoEpi& ep, int n0, int nstat, int y0,
                                                   int x0, int wm, int wn, int lane, int tid, float* hl)
{
    static_assert(T::TS == 1 && T::TH * T::TW == T::THREADS && T::BN == 32 && T::SW == 1, "one pixel per thread");
    // The statistics entries of the thread's voxel are requested first: their round trip runs beside the output transform, the LDS hand-over and
    // the dot products instead of behind the softmax -- one plane's load / add / store behind the other's, as accumulate_voxel does it, kept the
    // whole workgroup (both waves of every SIMD are in this phase together) waiting for two memory round trips per tile.  Same operations, same bits.
    const int gy = y0 + tid / T::TW, gx = x0 + tid % T::TW;
    const int Hi = PART ? a.Hr : a.H, Wi = PART ? a.Wr : a.W;   // the image the logits / statistics are arrays over
    const bool inside = n0 < a.N && gy < Hi && gx < Wi;
    const size_t hw = (size_t)gy * Wi + gx, HW = (size_t)Hi * Wi;
    VoxelStats<2> st;
    if (inside && a.head_stats != nullptr) st.load(a.head_stats, (size_t)nstat * HW + hw, a.head_V, a.head_flags);   // nstat: the image the sample is a pass of
    {
        const int n16 = lane & 15, g = lane >> 4;
        int bs, by, bx;
        T::block_origin(wm, bs, by, bx);
        const int yrel = by + 2 * (g >> 1), xrel = bx + 8 * (g & 1);
        const float relu_floor = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x2 y[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float nn[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float m0 = acc[b][4 * i + 0][r], m1 = acc[b][4 * i + 1][r], m2 = acc[b][4 * i + 2][r],
                                m3 = acc[b][4 * i + 3][r];
                    nn[i][0] = m0 + m1 + m2;
                    nn[i][1] = m1 - m2 - m3;
                }
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const float v0 = nn[0][bb] + nn[1][bb] + nn[2][bb];
                    const float v1 = nn[1][bb] - nn[2][bb] - nn[3][bb];
                    const float t0 = v0 * ep.scale[b] + ep.shift[b], t1 = v1 * ep.scale[b] + ep.shift[b];
                    y[0][bb][b] = fmaxf(t0, relu_floor);
                    y[1][bb][b] = fmaxf(t1, relu_floor);
                }
            }
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
                    *reinterpret_cast<f32x2*>(hl + ((yrel + aa) * T::TW + xrel + 2 * r + bb) * WINO_HEAD_PITCH + 2 * n16) = y[aa][bb];
        }
    }
    __syncthreads();
    if (inside) {
        const float* const row = hl + tid * WINO_HEAD_PITCH;
        float l[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float p[8];
#pragma unroll
            for (int sgm = 0; sgm < 8; ++sgm) {
                const f32x2 xa = *reinterpret_cast<const f32x2*>(row + 4 * sgm), xb = *reinterpret_cast<const f32x2*>(row + 4 * sgm + 2);
                const float* const w = a.head_w + c * 32 + 4 * sgm;
                float t = fmaf(w[0], xa.x, 0.f);
                t = fmaf(w[1], xa.y, t);
                t = fmaf(w[2], xb.x, t);
                p[sgm] = fmaf(w[3], xb.y, t);
            }
            l[c] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) + a.head_b[c];
        }
        if (a.head_logits != nullptr) {
            a.head_logits[((size_t)n0 * 2 + 0) * HW + hw] = l[0];
            a.head_logits[((size_t)n0 * 2 + 1) * HW + hw] = l[1];
        }
        if (a.head_stats != nullptr) {
            softmax_inplace<2>(l);
            st.add(a.head_flags, l);
            st.store(a.head_stats, (size_t)nstat * HW + hw, a.head_V, a.head_flags);
        }
    }
}

template <class T, bool HEAD = false, bool PART = false>
__global__ __launch_bounds__(512, 1) void conv_wino_stream(const ConvArgs a, const int total_items)
{
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource types and LDS-DMA builtins exist in the device pass only
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    constexpr int KC = T::KC;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % T::WN;
    const int wm = wave / T::WN;
    const int m16 = lane & 15, kq = lane >> 4;
    const int nchunks = (a.C1 + a.C2) / KC;   // even, >= 4 (checked by the launcher)
    const uint32_t wchunk_bytes = (uint32_t)a.NT * T::W_DW * 4u;
    const uint32_t in_chunk_bytes = a.in_chunk_bytes;

    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1), 0, a.src1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src2 ? a.src2 : a.src1), 0, a.src2 ? a.src2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);

    // fragment addresses (dword offsets inside a buffer).  Patch rows 0,1 share one swizzle bit, rows 2,3 the other.
    int aA[2], aB[2];
    {
        int bs, by, bx;
        T::block_origin(wm, bs, by, bx);
        const int tr = m16 >> 3, tcg = m16 & 7;
        const int sl = bs + (T::SW == 2 ? tcg >> 2 : 0);
        const int yy0 = by + 2 * tr, x0l = bx + 2 * (T::SW == 2 ? tcg & 3 : tcg);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int yy = yy0 + 2 * i2, swz = (yy >> 1) & 1;
#if RCU_WINO_IMAGE8
            const int rowbase = (sl * T::SLICE_POS + yy * T::PITCH) * 8 + kq * 2;   // [slice][row][position][8 channels]
            aA[i2] = rowbase + 8 * (x0l + swz);
            aB[i2] = rowbase + 8 * (x0l - swz);
#else
            const int rowbase = ((kq >> 1) * T::HALF_POS + sl * T::SLICE_POS + yy * T::PITCH) * 4 + (kq & 1) * 2;
            aA[i2] = rowbase + 4 * (x0l + swz);   // even patch columns j: pixel x0l + j sits at position x0l + j + swz
            aB[i2] = rowbase + 4 * (x0l - swz);   // odd  patch columns j: pixel x0l + j sits at position x0l + j - swz
#endif
        }
    }
    const int b_addr = T::A_DW + (kq * T::BN + wn * 32 + 2 * m16) * 2;
    const uint32_t w_voff = (uint32_t)(lane * 16);

    // dp*: the plan the LDS-DMA works from -- the current tile's until its last chunk is being multiplied, then the
    // next tile's; tile / ntile: coordinates of the tile being multiplied and of the workgroup's next one
    // HEAD: total_items counts the tiles of ONE pass; the workgroup that owns a tile runs the tile of every pass of the group back to
    // back (sample n0 + pass * head_images), so the passes' read-modify-writes of a voxel's statistics are ordered
    int item = wino_xcd_virtual_block(a.NT < 4 ? 4 : a.NT);
    [[maybe_unused]] int pass = 0;
    auto more_passes = [&]() {
        if constexpr (HEAD) return pass + 1 < wino_cold_args().head_passes;
        return false;
    };
    bool has_next = more_passes() || item + (int)gridDim.x < total_items;
    WinoTileId tile = wino_tile_id<T>(a, item), ntile = tile;
    uint32_t dp[T::NA], geo[T::NA];
    int dp_wtile = tile.wtile;
#pragma unroll
    for (int j = 0; j < T::NA; ++j) {
        geo[j] = wino_slot_plan<T>(a, wino_slot_geometry<T>(j, wave, lane));   // the slot's plan: tile-independent offset | border flags
        asm volatile("" : "+v"(geo[j]));   // one register per slot; keeps hipcc from carrying the unpacked fields instead
        dp[j] = wino_slot_offset(geo[j], wino_tile_offset<T>(a, tile));
    }

    // LDS-DMA of Cin chunk kc of a tile into LDS buffer `buf`: the wave's NW weight pieces and NA input pieces of 1 KB.
    // DmaJob holds the wave-uniform part; dma_piece(job, off1, off2, I) issues piece I (weights first).
    struct DmaJob {
        bool active, first;
        uint32_t cb, wso, lb;
    };
    auto dma_job = [&](int wtile, int kc, int buf, bool active) {
        DmaJob j;
        const int c0 = kc * KC;
        j.active = active;
        j.first = c0 < a.C1;
        j.cb = wino_chunk_offset(in_chunk_bytes, j.first ? c0 : c0 - a.C1);
        j.wso = (uint32_t)kc * wchunk_bytes + (uint32_t)wtile * (T::W_DW * 4u);
        j.lb = (uint32_t)buf * (T::BUF_DW * 4u);
        return j;
    };
    // LDS byte address of the dynamic shared memory as an integer: no generic-to-LDS pointer conversion (null check) per piece
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    auto dma_piece = [&](const DmaJob& job, auto i_c) {
        constexpr int I = decltype(i_c)::value;
        if constexpr (I < T::NW) {
            if (job.active && ((I + 1) * T::WAVES <= T::W_PIECES || I * T::WAVES + wave < T::W_PIECES))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(uintptr_t)(lds_base + job.lb + T::A_DW * 4 + (I * T::WAVES + wave) * 1024),
                                                         16, w_voff, job.wso + (uint32_t)(I * T::WAVES + wave) * 1024u, 0, 0);
        } else if constexpr (I < T::NW + T::NA) {
            constexpr int j = I - T::NW;
            if (job.active && ((j + 1) * T::WAVES <= T::A_PIECES || j * T::WAVES + wave < T::A_PIECES)) {
                if (job.first)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(uintptr_t)(lds_base + job.lb + (j * T::WAVES + wave) * 1024), 16,
                                                             dp[j], job.cb, 0, RCU_DMA_IN_AUX);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr_t)(uintptr_t)(lds_base + job.lb + (j * T::WAVES + wave) * 1024), 16,
                                                             dp[j], job.cb, 0, RCU_DMA_IN_AUX);
            }
        }
    };

    {
        const DmaJob job = dma_job(dp_wtile, 0, 0, true);
        wino_static_for<0, T::NW + T::NA>([&](auto i_c) { dma_piece(job, i_c); });
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();

    f32x4 acc[2][16];
    WinoEpiRaw epr;

    // One Cin chunk out of LDS buffer BUF; FIRST: the accumulators start from zero (first chunk of a tile).
    auto chunk = [&](auto buf_c, auto first_c, int kc) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        const float* const Ab = smem + BUF * T::BUF_DW;
        // the next chunk (of this tile or of the workgroup's next tile) streams into the other buffer meanwhile; its
        // LDS-DMA pieces are issued one per MFMA group below
        const bool more = kc + 1 < nchunks;
        if (!more && has_next) {   // last chunk of the tile: from here on the DMA works on the workgroup's next tile
            const WinoTileConsts ca = wino_tile_consts(wino_cold_args());   // tile counts and image extents are not kept in SGPRs either: one batch of scalar loads
            if (more_passes()) {
                ntile = tile;
                ntile.n0 = tile.n0 + wino_cold_args().head_images;
            } else {
                ntile = wino_tile_id<T>(ca, item + (int)gridDim.x);
            }
            dp_wtile = ntile.wtile;
            const WinoTileOffset nto = wino_tile_offset<T>(ca, ntile);
#pragma unroll
            for (int j = 0; j < T::NA; ++j) dp[j] = wino_slot_offset(geo[j], nto);
        }
        if (!more) epr = wino_epilogue_load<T>(wino_cold_args(), tile.wtile, tile.n0, wm, wn, lane);
        __builtin_amdgcn_sched_barrier(0);   // keep the address arithmetic above out of the register-heavy part below
        const DmaJob job = dma_job(dp_wtile, more ? kc + 1 : 0, BUF ^ 1, more || has_next);
        // raw 4x4 patch of the lane's tile, channel pair (2kq, 2kq+1): rows 0 and 2 first (position row 0 needs only them)
        f32x2 d[16];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = (ii & 1) * 2 + (ii >> 1);   // 0, 2, 1, 3
#pragma unroll
            for (int j = 0; j < 4; ++j)
                // volatile: hipcc otherwise fuses pairs of these reads into ds_read2_b64, whose 16-lane groups over 32 banks
                // conflict 2-way on this image (SQ_LDS_BANK_CONFLICT 33 % of the LDS cycles); ds_read_b64 serves 32
                // lanes over 64 banks
                d[4 * i + j] = *(const volatile __attribute__((address_space(3))) f32x2*)(Ab + ((j & 1) ? aB[i >> 1] : aA[i >> 1]) + (i & 1) * (T::PITCH * WINO_POS_DW) + WINO_POS_DW * j);
        }
        constexpr int AHEAD = 4;
        f32x4 bv[16];
#pragma unroll
        for (int p = 0; p < AHEAD; ++p) bv[p] = *reinterpret_cast<const f32x4*>(Ab + b_addr + p * (8 * T::BN));
        // B^T d B in place, spread over the MFMA groups: row transform t = B^T d, then per row i the column transform
        auto col_transform = [&](int i) {
            const f32x2 t0 = d[4 * i], t1 = d[4 * i + 1], t2 = d[4 * i + 2], t3 = d[4 * i + 3];
            d[4 * i] = t0 - t2;
            d[4 * i + 1] = t1 + t2;
            d[4 * i + 2] = t2 - t1;
            d[4 * i + 3] = t1 - t3;
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = d[j] - d[8 + j];
        col_transform(0);
        // 16 positions x (2 channel steps x 2 cout blocks); the weights of positions p+AHEAD.. are read while p multiplies
        wino_static_for<0, 8>([&](auto pp_c) {
            constexpr int G = decltype(pp_c)::value;
            constexpr int p0 = 2 * G, p1 = p0 + 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (p0 + AHEAD < 16) {
                bv[p0 + AHEAD] = *reinterpret_cast<const f32x4*>(Ab + b_addr + (p0 + AHEAD) * (8 * T::BN));
                bv[p1 + AHEAD] = *reinterpret_cast<const f32x4*>(Ab + b_addr + (p1 + AHEAD) * (8 * T::BN));
            }
            // two LDS-DMA pieces per group from the first group on, so that the last piece has 3/4 of the chunk to land (one per
            // group: -1.5 %; three per group: -1.4 %)
            dma_piece(job, std::integral_constant<int, 2 * G>{});
            dma_piece(job, std::integral_constant<int, 2 * G + 1>{});
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            acc[0][p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p0].x, bv[p0].x, FIRST ? z : acc[0][p0], 0, 0, 0);
            acc[1][p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p0].x, bv[p0].z, FIRST ? z : acc[1][p0], 0, 0, 0);
            acc[0][p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p1].x, bv[p1].x, FIRST ? z : acc[0][p1], 0, 0, 0);
            acc[1][p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p1].x, bv[p1].z, FIRST ? z : acc[1][p1], 0, 0, 0);
            acc[0][p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p0].y, bv[p0].y, acc[0][p0], 0, 0, 0);
            acc[1][p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p0].y, bv[p0].w, acc[1][p0], 0, 0, 0);
            acc[0][p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p1].y, bv[p1].y, acc[0][p1], 0, 0, 0);
            acc[1][p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p1].y, bv[p1].w, acc[1][p1], 0, 0, 0);
            // transform work for the coming position rows, in the shadow of this group's MFMAs
            if constexpr (G == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 x1 = d[4 + j], x2 = d[8 + j];
                    d[4 + j] = x1 + x2;
                    d[8 + j] = x2 - x1;
                    d[12 + j] = x1 - d[12 + j];
                }
            }
            if constexpr (G == 1) col_transform(1);
            if constexpr (G == 2) col_transform(2);
            if constexpr (G == 3) col_transform(3);
        });
        wino_static_for<16, T::NW + T::NA>([&](auto i_c) { dma_piece(job, i_c); });
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of the next chunk have landed
        __syncthreads();                      // everyone done with buffer BUF and with filling the other one
    };

    for (;;) {
        chunk(std::integral_constant<int, 0>{}, std::true_type{}, 0);
        chunk(std::integral_constant<int, 1>{}, std::false_type{}, 1);
        for (int kc = 2; kc < nchunks; kc += 2) {
            chunk(std::integral_constant<int, 0>{}, std::false_type{}, kc);
            chunk(std::integral_constant<int, 1>{}, std::false_type{}, kc + 1);
        }
        if constexpr (HEAD)
            wino_epilogue_head<T, PART>(wino_cold_args(), acc, wino_epilogue_fold(epr), tile.n0, tile.n0 - pass * wino_cold_args().head_images, tile.y0, tile.x0, wm,
                                        wn, lane, tid, smem + 2 * T::BUF_DW);
        else
            wino_epilogue<T, PART>(wino_cold_args(), acc, wino_epilogue_fold(epr), tile.wtile, tile.n0, tile.y0, tile.x0, wm, wn, lane);
        if (!has_next) break;
        if (more_passes()) {
            ++pass;
        } else {
            pass = 0;
            item += (int)gridDim.x;
        }
        tile = ntile;
        has_next = more_passes() || item + (int)gridDim.x < total_items;
    }
#endif
}

using WCfg0 = WinoTile<1, 16, 16, 64, 4, 2>;   // 256 pixels x 64 couts
using WCfg1 = WinoTile<1, 16, 32, 32, 8, 1>;   // 512 pixels x 32 couts (32-channel full-resolution layers)
using WCfg2 = WinoTile<2, 8, 16, 64, 4, 2>;    // two 8x16 pieces of consecutive slices (heights not divisible by 16)
using WCfg3 = WinoTile<8, 4, 8, 64, 4, 2, 2>;  // 4x8 strips of eight slices (8-pixel-wide bottom level)

static const ConvConfigInfo kWinoInfo[5] = {
    {WCfg0::TS, WCfg0::TH, WCfg0::TW, WCfg0::BN, 8, 16, "conv3x3_winograd<T16x16,N64,K8>", 8, 0, 1},
    {WCfg1::TS, WCfg1::TH, WCfg1::TW, WCfg1::BN, 8, 16, "conv3x3_winograd<T16x32,N32,K8>", 8, 0, 1},
    {WCfg2::TS, WCfg2::TH, WCfg2::TW, WCfg2::BN, 8, 16, "conv3x3_winograd<S2T8x16,N64,K8>", 8, 0, 1},
    {WCfg3::TS, WCfg3::TH, WCfg3::TW, WCfg3::BN, 8, 16, "conv3x3_winograd<S8T4x8,N64,K8>", 8, 0, 1},
    {WCfg1::TS, WCfg1::TH, WCfg1::TW, WCfg1::BN, 8, 16, "conv3x3_winograd<T16x32,N32,K8>+head", 8, 0, 1},
};

const ConvConfigInfo& wino_config_info(int cfg) { return kWinoInfo[cfg - CONV_CFG_WINO_T16x16_N64]; }

template <class T, bool HEAD = false>
static hipError_t launch_wino_cfg(const ConvArgs& a, hipStream_t stream)
{
    constexpr int lds_bytes = T::LDS_BYTES + (HEAD ? T::TH * T::TW * WINO_HEAD_PITCH * 4 : 0);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    if (HEAD && (a.head_w == nullptr || a.NT != 1 || a.pooled != nullptr || a.mask2 != nullptr || a.head_passes < 1 ||
                 a.head_images * a.head_passes != a.N || (a.head_passes > 1 && a.head_logits != nullptr)))
        return hipErrorInvalidValue;
    const int nchunks = (a.C1 + a.C2) / T::KC;
    if (nchunks < 4 || (nchunks & 1) != 0 || a.NTW_total != a.NT || a.src1_bytes == 0 || a.wpack_bytes == 0 ||
        (a.C2 != 0 && a.C2 != a.C1) || a.H % T::TH != 0 || a.W % T::TW != 0 ||
        (size_t)a.N * a.H * a.W * a.CoutP * 4 >= ((size_t)1 << 31))
        return hipErrorInvalidValue;
    // padded level: the real image lies inside the tile grid, the output / pooled tensors hold it
    if (a.part && (a.Hr < 1 || a.Wr < 1 || a.Hr > a.H || a.Wr > a.W || a.out_H < a.Hr || a.out_W < a.Wr ||
                   (size_t)a.N * a.out_H * a.out_W * a.CoutP * 4 >= ((size_t)1 << 31) ||
                   (a.pooled != nullptr && (a.pool_H < (a.Hr >> 1) || a.pool_W < (a.Wr >> 1)))))
        return hipErrorInvalidValue;
    // HEAD: the work items of one pass (TS == 1: a slice group is a sample); the kernel runs every pass of the group on each
    const unsigned items = (unsigned)a.NT * a.tiles_x * a.tiles_y * (HEAD ? a.head_images : a.slice_groups);
    const unsigned grid = wino_persistent_grid(items);
    auto launch = [&](auto part_c) {
        constexpr bool PART = decltype(part_c)::value;
        hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_stream<T, HEAD, PART>), lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((conv_wino_stream<T, HEAD, PART>), dim3(grid), dim3(T::THREADS), lds_bytes, stream, a, (int)items);
        return hipGetLastError();
    };
    return a.part ? launch(std::true_type{}) : launch(std::false_type{});
}

hipError_t launch_conv_wino(int cfg, const ConvArgs& a, hipStream_t stream)
{
    switch (cfg) {
        case CONV_CFG_WINO_T16x16_N64: return launch_wino_cfg<WCfg0>(a, stream);
        case CONV_CFG_WINO_T16x32_N32: return launch_wino_cfg<WCfg1>(a, stream);
        case CONV_CFG_WINO_S2T8x16_N64: return launch_wino_cfg<WCfg2>(a, stream);
        case CONV_CFG_WINO_S8T4x8_N64: return launch_wino_cfg<WCfg3>(a, stream);
        case CONV_CFG_WINO_T16x32_N32_HEAD: return launch_wino_cfg<WCfg1, true>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rcu
