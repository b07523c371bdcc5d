// conv3x3 (pad 1) conv unit in Winograd F(4x4, 3x3) form on the fp32 matrix cores of gfx950.
//
//     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 4x4 output tile, 6x6 input patch d
//
// 36 multiplications per 16 output pixels and (cin, cout) pair: 2.25 per pixel against 4 in F(2x2,3x3) (rcu_wino.hip) and 9 in
// the direct form -- the channel contraction, the only dense part and what the MFMA pipe executes, shrinks by another 1.78x.
// Interpolation points 0, +-1, +-2, inf (Lavin & Gray); in float32 the network stays as close to the float64 network as the
// direct float32 form does (tools/wino43_numerics.py: max |dlogit| 2.4e-7 against 1.3e-7 direct on the full-width U-Net), far
// inside the parity gate of the tests (2e-6 on logits).  Everything else of the reference's conv unit (common/model/unet.py:8-23:
// bias, Dropout2d factor, folded BatchNorm, ReLU; the 2x2 max-pool of DownConv, unet.py:85-95; the cat-free two-source K loop of
// UpConv, unet.py:118) is fused as in rcu_wino.hip.
//
// Mapping (v_mfma_f32_16x16x4_f32), where it differs from rcu_wino.hip:
//   * a wave owns 16 tiles x 32 output channels x 36 positions = 288 accumulator registers -- the accumulator half of the 512
//     registers of a lane -- so a workgroup is 4 waves, ONE per SIMD, one workgroup per CU: 64 tiles (1024 pixels) x 32 couts.
//     With the transform amortised over 32 couts the VALU work per MFMA is that of F(2x2,3x3) (288 operations per 144 MFMAs
//     and chunk against 64 per 64 there).
//   * A operand: lane (tile m = lane & 15, channel pair kq = lane >> 4) reads its tile's RAW 6x6 patch (36 ds_read_b64) and
//     applies B^T d B in registers (scalar operations, per channel), spread over the MFMA groups of the chunk: position rows are
//     multiplied in the order 0, 5, 1, 2, 3, 4 so that only T0 = 4 d0 - 5 d2 + d4 and its column transform precede the first MFMA.
//   * B operand: host-transformed weights U = G g G^T packed [chunk][cout tile][p][channel pair][cout][2] as in rcu_wino.hip: one
//     ds_read_b128 per position serves both 16-channel MFMA blocks of the wave (couts 2n, 2n+1 per lane).
//   * LDS input image: [slice][halo row R][position][8 channels] -- a pixel's 32 bytes of the chunk stay together, so adjacent lanes
//     of an LDS-DMA instruction fetch adjacent 16-byte halves (half the L2 requests of a [channel half][position] image, which
//     is what bounds the staging of these kernels) -- with position = x ^ swz, swz = (((x >> 3) ^ slice) & 1) | (((R >> 2) & 1) << 1),
//     applied on the SOURCE address of the LDS-DMA; with a row pitch of 4k and a slice stride of 16k positions the 32 lanes a
//     ds_read_b64 serves per cycle (2 channel pairs x 16 tiles, tiles 4 pixels = 32 dwords apart) fall two by two on the 32
//     two-dword slots of the 64 banks: 2-way, the minimum for 8-dword positions.
//   * D: lane (n = lane & 15, g = lane >> 4) holds the four horizontally adjacent tiles 4g .. 4g+3 of one tile row (4 x 16
//     pixels) for couts (2n, 2n+1): output transform A^T M A (100 operations per tile and cout), epilogue and 2x2 max-pool are
//     lane-local; neighbouring lanes trade pixel columns by DPP so that every store writes 16 bytes.
//   * staging by LDS-DMA only, double-buffered chunk pipeline running across the tiles of a workgroup, buffer-resource zero
//     padding: as rcu_wino.hip.
#include "rcu_wino_common.h"
#include "rcu_head_common.h"

#include <cstdlib>

namespace rcu {

// Workgroup tile: WS slice groups x WR block rows of MFMA row blocks; a block = SB slices x BR tile rows x BC tile columns = 16
// tiles of 4x4 pixels spanning the tile's full width.
// FULLW: the tile spans the image's width (checked by the launcher), so the halo columns are zero padding and are not staged at all:
// the LDS image holds TW positions per row and the lanes of the first / last tile column zero their outer patch column themselves.
// What it buys is LDS: eight slices of 8x16 pixels (the 24x16 level) fit the 160 KB only without the two halo columns.
// FOLD: 12x8-pixel images (the BraTS bottom level).  A slice is 3 x 2 = 6 tiles of 4x4 pixels; they go into the 8 tile slots (2 block rows x 4
// columns) a slice has in the S8 block geometry: slot k = 4 tr' + tc' holds tile (k / 2, k % 2) for k < 6, slots 6 and 7 are idle -- their MFMA rows
// read tile 5's patch again and are never stored: 48 of a workgroup's 64 tile slots work (a lane's four tiles, slots 4g .. 4g+3 of one slice, are a
// 2 x 2 block of tiles, the lower half of it idle for odd g), which still beats F(2x2,3x3)'s 4 multiplications per pixel with 2.25 x 4/3 = 3.
template <int SB_, int BR_, int BC_, int WS_, int WR_, bool FULLW_ = false, bool FOLD_ = false>
struct Wino4Tile {
    static constexpr int SB = SB_, BR = BR_, BC = BC_, WS = WS_, WR = WR_;
    static constexpr bool FULLW = FULLW_, FOLD = FOLD_;
    static constexpr int TS = SB * WS, TH = FOLD ? 12 : 4 * BR * WR, TW = FOLD ? 8 : 4 * BC;
    static constexpr int TCOLS = FOLD ? 2 : BC;                              // tile columns of the workgroup tile
    static constexpr int BN = 32, KC = 8, THREADS = 256, WAVES = 4, NPOS = 36;
    static constexpr int XCOLS = FULLW ? TW : TW + 2;                        // staged pixel columns per row: x = -1 .. TW, or 0 .. TW - 1
    static constexpr int PITCH = (XCOLS + 3) / 4 * 4;                        // positions per halo row, a multiple of 4
    static constexpr int SLICE_POS = ((TH + 2) * PITCH + 15) / 16 * 16;      // positions per slice image, a multiple of 16
    static constexpr int A_POS = 2 * TS * SLICE_POS;                         // 16-byte units: [slice][halo row][position][channel half]
    static constexpr int A_PIECES = (A_POS + 63) / 64;
    static constexpr int NA = (A_PIECES + WAVES - 1) / WAVES;
    static constexpr int A_DW = A_PIECES * 256;
    static constexpr int W_DW = NPOS * 4 * BN * 2;                           // [p][channel pair][cout][2]
    static constexpr int W_PIECES = W_DW / 256;
    static constexpr int NW = (W_PIECES + WAVES - 1) / WAVES;
    static constexpr int BUF_DW = A_DW + W_DW;
    static constexpr int LDS_BYTES = 2 * BUF_DW * 4;
    static_assert(SB * BR * BC == 16 && WS * WR == WAVES && BC % 4 == 0, "block geometry");
    static_assert(!FOLD || (SB == 2 && BR == 2 && BC == 4 && WS == 4 && WR == 1 && FULLW), "the folded 12x8 geometry rides on the S8 block");
    static_assert(W_DW % 256 == 0 && LDS_BYTES <= 160 * 1024, "LDS");
    // wave -> (first slice of its block inside the tile, top pixel row of its block inside the slice tile)
    static __device__ __forceinline__ void block_origin(int wave, int& s, int& y)
    {
        s = (wave / WR) * SB;
        y = (wave % WR) * (4 * BR);
    }
    // tile m of a block -> (slice inside the block, tile row, tile column)
    static __device__ __forceinline__ void tile_of(int m, int& sb, int& tr, int& tc)
    {
        sb = m / (BR * BC);
        if constexpr (FOLD) {
            const int k = min(m % 8, 5);      // idle slots 6, 7 read tile 5's patch
            tr = k >> 1;
            tc = k & 1;
        } else {
            tr = (m / BC) % BR;
            tc = m % BC;
        }
    }
    static __device__ __forceinline__ int swizzle(int x, int s, int R) { return (((x >> 3) ^ s) & 1) | (((R >> 2) & 1) << 1); }
};

template <class T>
__device__ __forceinline__ uint32_t wino4_slot_geometry(int j, int wave, int lane)
{
    const int f = (j * T::WAVES + wave) * 64 + lane;       // 16-byte unit of the LDS image: adjacent lanes fetch the two halves of a pixel's 32 bytes
    const int hh = f & 1, rem = f >> 1;
    const int s = rem / T::SLICE_POS, r2 = rem % T::SLICE_POS;
    const int yy = r2 / T::PITCH, pos = r2 % T::PITCH;
    const int x = pos ^ T::swizzle(pos, s, yy);             // the swizzle only moves a position inside its aligned group of four
    const bool real = f < T::A_POS && yy < T::TH + 2 && x < T::XCOLS;
    // the x field is the column in halo coordinates (image column + 1), whichever layout the image has
    return real ? (uint32_t)((x + (T::FULLW ? 1 : 0)) | (yy << 8) | (s << 16) | (hh << 24)) : 0xFFFFFFFFu;
}

template <class T>
__device__ __forceinline__ WinoEpiRaw wino4_epilogue_load(const ConvArgs& a, int ntile, int n0, int wave, int lane)
{
    WinoEpiRaw e;
    const int co = ntile * T::BN + 2 * (lane & 15);   // < NT * BN, the length of alpha / betab / beta
    int bs, by, sb, tr, tc;
    T::block_origin(wave, bs, by);
    T::tile_of(4 * (lane >> 4), sb, tr, tc);
    // the arguments in ONE batch of scalar loads (a wave alone on its SIMD sits out every scalar round trip: the mask branch used to fetch its
    // own arguments behind the branch)
    const float *alpha = a.alpha, *betab = a.betab, *beta = a.beta, *mask = a.mask, *mask2 = a.mask2;
    int N = a.N, Cmask = a.Cmask, Cmask2 = a.Cmask2, Csplit = a.Csplit;
    asm volatile("" : "+s"(alpha), "+s"(betab), "+s"(beta), "+s"(mask), "+s"(mask2), "+s"(N), "+s"(Cmask), "+s"(Cmask2), "+s"(Csplit));
    const int n = min(n0 + bs + sb, N - 1);
    e.al = *reinterpret_cast<const f32x2*>(alpha + co);
    e.bb = *reinterpret_cast<const f32x2*>(betab + co);
    e.be = *reinterpret_cast<const f32x2*>(beta + co);
    e.site = 0;
    e.mk[0] = e.mk[1] = 1.f;
    if (mask != nullptr) {   // wave-uniform
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = co + b;
            // the fused cls + sigma head unit: output channels >= Csplit belong to the twin's dropout site (as rcu_wino_common.h)
            const bool second = mask2 != nullptr && c >= Csplit;
            const int cm = second ? Cmask2 : Cmask, ci = second ? c - Csplit : c;
            const float* const row = second ? mask2 + (size_t)n * Cmask2 : mask + (size_t)n * Cmask;
            e.mk[b] = row[min(ci, cm - 1)];
            e.site |= (ci < cm ? 1 : 0) << b;
        }
    }
    return e;
}

// 288 accumulator registers per lane: 64 tuples of 4 in the accumulator file a[0:255], the other 8 in VGPRs.  hipcc gives a kernel
// ONE form of the MFMA (accumulators in AGPRs), treats all 512 registers as one pool and answers the overflow with
// v_accvgpr_read / _write pairs around MFMAs inside the loop, so the accumulator file is owned by the inline assembly below: every
// MFMA names its a[N:N+3] literally and every statement lists the whole file as clobbered (which also makes the kernel descriptor
// allocate it).  The compiler never holds a value there; the build is audited for that (tests/test_abi_cpu.py: no scratch, no
// v_accvgpr_* outside the asm statements).  The accumulators of cout block 1 at positions 18..25 are ordinary VGPR variables
// multiplied by the VGPR form of the instruction.
//   s_nop 11 (closing the VGPR-form MFMAs): to hipcc an asm output is ready when the statement ends; should it ever copy or spill one of
//   these accumulator tuples right behind the statement it must not read the MFMA's destination early (8 passes: 12 wait states).
//   The wave would wait for the matrix pipe in that time anyway: no measurable cost.
//   s_nop 1 (opening every MFMA; it costs no matrix time, tools/microbench/gen_mfma_valu_1wave.py): the wait states between a VALU
//   write of an operand and the MFMA reading it, which hipcc does not add inside asm -- and hipcc is free to sink a transform
//   operation down to the MFMA that consumes it (a build with the nop on the first MFMA of a group only gave wrong results).
#define WINO4_ALL_AGPRS \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
    "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
    "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
    "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
    "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
    "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
    "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
__host__ __device__ constexpr bool wino4_in_vgpr(int b, int p) { return b == 1 && p >= 18 && p < 26; }
// first AGPR of the accumulator tuple of (cout block b, position p)
__host__ __device__ constexpr int wino4_areg(int b, int p) { return 4 * (b == 0 ? p : 36 + (p < 18 ? p : p - 8)); }

template <int B, int P, bool ZERO, bool NOP, bool PROBE = false>
__device__ __forceinline__ void wino4_mfma(f32x4 (&accv)[8], float av, float wv)
{
    if constexpr (PROBE) asm volatile("s_nop 7\n\ts_nop 7");
    if constexpr (wino4_in_vgpr(B, P)) {
        if constexpr (ZERO)
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0\n\ts_nop 11" : "=v"(accv[P - 18]) : "v"(av), "v"(wv));
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 11" : "+v"(accv[P - 18]) : "v"(av), "v"(wv));
    } else {
        constexpr int R = wino4_areg(B, P);
        if constexpr (ZERO && NOP)
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, 0" ::"v"(av), "v"(wv), "i"(R), "i"(R + 3) : WINO4_ALL_AGPRS);
        else if constexpr (ZERO)
            asm volatile("v_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, 0" ::"v"(av), "v"(wv), "i"(R), "i"(R + 3) : WINO4_ALL_AGPRS);
        else if constexpr (NOP)
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(av), "v"(wv), "i"(R), "i"(R + 3) : WINO4_ALL_AGPRS);
        else
            asm volatile("v_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(av), "v"(wv), "i"(R), "i"(R + 3) : WINO4_ALL_AGPRS);
    }
}

// element r of the accumulator tuple of (b, p)
template <int B, int P, int R>
__device__ __forceinline__ float wino4_acc(const f32x4 (&accv)[8])
{
    if constexpr (wino4_in_vgpr(B, P)) {
        return accv[P - 18][R];
    } else {
        float v;
        asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(wino4_areg(B, P) + R));
        return v;
    }
}

// Per-lane parts of the epilogue's store offsets (output and pooled output): slice, pixel and cout pair of the lane's first tile relative to
// the workgroup tile's origin and cout tile -- made once per kernel; a tile adds scalars (wino4_epilogue).
struct Wino4StorePlan {
    uint32_t out, pool;
};
template <class T, bool PART>
__device__ __forceinline__ Wino4StorePlan wino4_store_plan(const ConvArgs& a, int wave, int lane)
{
    int bs, by, sb, tr, tc;
    T::block_origin(wave, bs, by);
    T::tile_of(4 * (lane >> 4), sb, tr, tc);
    const int n16 = lane & 15, odd = n16 & 1, c = 2 * n16 - 2 * odd;   // the even lane stores four couts of one pixel column, the odd lane of the next
    // PART (padded levels, ConvArgs::part): `out` and `pooled` have extents of their own
    const int oH = PART ? a.out_H : a.H, oW = PART ? a.out_W : a.W;
    const int Hp = PART ? a.pool_H : a.H >> 1, Wp = PART ? a.pool_W : a.W >> 1;
    Wino4StorePlan p;
    p.out = wino_out_offset(bs + sb, oH * oW, a.CoutP, (uint32_t)((by + 4 * tr) * oW + 4 * tc + odd), c, a.out_pix_bytes, a.out_chunk_bytes);
    p.pool = wino_out_offset(bs + sb, Hp * Wp, a.CoutP, (uint32_t)(((by + 4 * tr) >> 1) * Wp + 2 * tc + odd), c, a.pool_pix_bytes, a.pool_chunk_bytes);
    return p;
}

// Output transform A^T M A + conv-unit epilogue (scale, shift, ReLU) of tile R of the lane's four: y[row][col], components = the lane's two
// couts -- packed operations throughout (one instruction costs the same matrix time whether it is packed or not, see the chunk pipeline).
template <int R, int EV = 0>
__device__ __forceinline__ void wino4_output_tile(const f32x4 (&accv)[8], const WinoEpi& ep, float relu_floor, f32x2 (&y)[4][4])
{
    constexpr int r = R;
    const f32x2 scale = {ep.scale[0], ep.scale[1]}, shift = {ep.shift[0], ep.shift[1]}, floor2 = {relu_floor, relu_floor};
    f32x2 nn[6][4];   // N[i][q] = sum_j M[i][j] A[j][q]
    if constexpr ((EV & 4) != 0) {
        wino_static_for<0, 16>([&](auto k_c) {
            constexpr int k = decltype(k_c)::value;
            y[k >> 2][k & 3] = f32x2{wino4_acc<0, k, r>(accv), wino4_acc<1, k, r>(accv)};
        });
    } else {
        wino_static_for<0, 6>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            auto m = [&](auto j_c) {
                constexpr int j = decltype(j_c)::value;
                return f32x2{wino4_acc<0, 6 * i + j, r>(accv), wino4_acc<1, 6 * i + j, r>(accv)};
            };
            const f32x2 m0 = m(std::integral_constant<int, 0>{}), m1 = m(std::integral_constant<int, 1>{}), m2 = m(std::integral_constant<int, 2>{}),
                        m3 = m(std::integral_constant<int, 3>{}), m4 = m(std::integral_constant<int, 4>{}), m5 = m(std::integral_constant<int, 5>{});
            const f32x2 s1 = m1 + m2, s2 = m1 - m2, s3 = m3 + m4, s4 = m3 - m4;
            nn[i][0] = (m0 + s1) + s3;
            nn[i][1] = s2 + 2.f * s4;
            nn[i][2] = s1 + 4.f * s3;
            nn[i][3] = (s2 + 8.f * s4) + m5;
        });
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 s1 = nn[1][q] + nn[2][q], s2 = nn[1][q] - nn[2][q], s3 = nn[3][q] + nn[4][q], s4 = nn[3][q] - nn[4][q];
            const f32x2 v0 = (nn[0][q] + s1) + s3;
            const f32x2 v1 = s2 + 2.f * s4;
            const f32x2 v2 = s1 + 4.f * s3;
            const f32x2 v3 = (s2 + 8.f * s4) + nn[5][q];
            y[0][q] = __builtin_elementwise_max(v0 * scale + shift, floor2);
            y[1][q] = __builtin_elementwise_max(v1 * scale + shift, floor2);
            y[2][q] = __builtin_elementwise_max(v2 * scale + shift, floor2);
            y[3][q] = __builtin_elementwise_max(v3 * scale + shift, floor2);
        }
    }
}

// Output transform + conv-unit epilogue of one finished tile.  acc[b][p][r]: MFMA block b (cout 2n+b), position p = 6 i + j,
// tile r of the lane's four.
// PART: the level is padded (ConvArgs::part) -- the tile may hang over the real image: pixels at or beyond (Hr, Wr) go out of range like the
// slices beyond the batch, and the output / pooled tensors have extents of their own.
template <class T, int EV = 0, bool PART = false>   // EV (ablation build): 1 stores out of range, 2 no store instructions, 4 no output transform
__device__ __forceinline__ void wino4_epilogue(const ConvArgs& a, const f32x4 (&accv)[8], const WinoEpi& ep, int ntile, int n0, int y0, int x0,
                                               int wave, int lane, const Wino4StorePlan& plan)
{
    auto store16 = [&](const f32x4& o, const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t soff) {
        if constexpr ((EV & 2) != 0)
            asm volatile("" ::"v"(o));
        else
            wino_store16(o, rs, (EV & 1) != 0 ? WINO_OOB : voff, soff);
    };
    // every MFMA has long retired: the last chunk's barrier lies between them and this point; the nops cover the asm-to-asm case
    // (an MFMA's D read by v_accvgpr_read) the compiler cannot see
    asm volatile("s_nop 15\n\ts_nop 7");
    const int odd = lane & 1;
    // The kernel arguments the epilogue needs, read in ONE batch of scalar loads (a wave alone on its SIMD sits out every scalar-load round
    // trip: the branches of the earlier form -- outputs alive? pooled? -- each fetched their own arguments behind the branch).
    const int H = a.H, W = a.W, CoutP = a.CoutP, N = a.N;
    const uint32_t px_bytes = a.out_pix_bytes, chunk_bytes = a.out_chunk_bytes, ppx_bytes = a.pool_pix_bytes, pchunk_bytes = a.pool_chunk_bytes;
    float* const out = a.out;
    float* const pooled = a.pooled;
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
    const bool pool = pooled != nullptr;
    const int oH = PART ? a.out_H : H, oW = PART ? a.out_W : W;
    const int Hp = PART ? a.pool_H : H >> 1, Wp = PART ? a.pool_W : W >> 1;
    const uint32_t row_bytes = (uint32_t)oW * px_bytes, prow_bytes = (uint32_t)Wp * ppx_bytes;
    // store offsets = the lane's plan (made once per kernel: everything but the tile's origin and cout tile) + the tile's scalars.  Lanes of
    // slices beyond the batch need no flag: their offsets lie behind the tensor, where the buffer resource drops the write.
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (uint32_t)(N * oH * oW) * (uint32_t)CoutP * 4u, 0x00020000);
    const uint32_t vo = plan.out + wino_out_offset(n0, oH * oW, CoutP, (uint32_t)(y0 * oW + x0), ntile * T::BN, px_bytes, chunk_bytes);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(pool ? pooled : out, 0, pool ? (uint32_t)(N * Hp * Wp * CoutP) * 4u : 0u, 0x00020000);
    const uint32_t vp = plan.pool + wino_out_offset(n0, Hp * Wp, CoutP, (uint32_t)((y0 >> 1) * Wp + (x0 >> 1)), ntile * T::BN, ppx_bytes, pchunk_bytes);
    // PART: rows / columns of the real image left from the lane's first pixel (row by + 4 tr, column 4 tc + odd of the workgroup tile), and the same
    // for its first pooled pixel: a pixel `dy` rows down and `dx` columns right of the first is real iff dy < ylim && dx < xlim
    [[maybe_unused]] int ylim = 0, xlim = 0, pylim = 0, pxlim = 0;
    if constexpr (PART) {
        int bs, by, sb, tr, tc;
        T::block_origin(wave, bs, by);
        T::tile_of(4 * (lane >> 4), sb, tr, tc);
        const int Hr = a.Hr, Wr = a.Wr;
        ylim = Hr - (y0 + by + 4 * tr);
        xlim = Wr - (x0 + 4 * tc + odd);
        pylim = (Hr >> 1) - ((y0 + by + 4 * tr) >> 1);
        pxlim = (Wr >> 1) - (((x0 + 4 * tc) >> 1) + odd);
    }
    wino_static_for<0, 4>([&](auto r_c) {
        constexpr int r = decltype(r_c)::value;
        // where tile r of the lane's four sits relative to its first: the next tile column -- or, folded, a 2 x 2 block of tiles whose lower
        // half (r = 2, 3) does not exist for the lanes that hold slots 4..7 of a slice (odd g): their stores go out of range
        constexpr int tdx = T::FOLD ? 4 * (r & 1) : 4 * r, tdy = T::FOLD ? 4 * (r >> 1) : 0;
        const uint32_t vo_r = (T::FOLD && r >= 2 && ((lane >> 4) & 1) != 0) ? WINO_OOB : vo;
        f32x2 y[4][4];   // [row][col], components = the two couts
        wino4_output_tile<r, EV>(accv, ep, relu_floor, y);
        // Neighbouring lanes (couts 2n, 2n+1 | 2n+2, 2n+3 of the same pixels) trade pixel columns: the even lane ends up with four
        // couts of columns 0 and 2 of the tile, the odd lane with four couts of columns 1 and 3 -> 16-byte stores.
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x2 keep = odd ? y[aa][2 * h2 + 1] : y[aa][2 * h2];
                const f32x2 send = odd ? y[aa][2 * h2] : y[aa][2 * h2 + 1];
                f32x2 recv;
                recv.x = wino_swap_adjacent(send.x);
                recv.y = wino_swap_adjacent(send.y);
                const f32x4 o = odd ? f32x4{recv.x, recv.y, keep.x, keep.y} : f32x4{keep.x, keep.y, recv.x, recv.y};
                const uint32_t vo_s = (PART && !(tdy + aa < ylim && tdx + 2 * h2 < xlim)) ? WINO_OOB : vo_r;
                store16(o, ro, vo_s, (uint32_t)(tdx + 2 * h2) * px_bytes + (uint32_t)(tdy + aa) * row_bytes);
            }
        }
        if (!T::FOLD && pool) {   // wave-uniform; 2x2 pooled pixels per tile: the even lane stores four couts of pooled column 0, the odd lane of column 1 (the folded geometry -- the bottom level -- has no pooled output: launcher)
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
                f32x2 mx[2];
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
                    mx[b2] = __builtin_elementwise_max(__builtin_elementwise_max(y[2 * a2][2 * b2], y[2 * a2][2 * b2 + 1]),
                                                       __builtin_elementwise_max(y[2 * a2 + 1][2 * b2], y[2 * a2 + 1][2 * b2 + 1]));
                const f32x2 keep = odd ? mx[1] : mx[0];
                const f32x2 send = odd ? mx[0] : mx[1];
                f32x2 recv;
                recv.x = wino_swap_adjacent(send.x);
                recv.y = wino_swap_adjacent(send.y);
                const f32x4 o = odd ? f32x4{recv.x, recv.y, keep.x, keep.y} : f32x4{keep.x, keep.y, recv.x, recv.y};
                const uint32_t vp_s = (PART && !(a2 < pylim && 2 * r < pxlim)) ? WINO_OOB : vp;
                store16(o, rp, vp_s, (uint32_t)(2 * r) * ppx_bytes + (uint32_t)a2 * prow_bytes);
            }
        }
    });
}

// conv_cls.0 with the classifier behind it (common/model/unet.py:160-161, rechun/dl/customsteps.py:24,33), as rcu_wino.hip's wino_epilogue_head: the
// finished tile's 32 channels never reach HBM; every lane ends up with whole pixels, computes the 1x1 conv to two logits and either stores them
// (NCHW) or adds softmax / entropy into the MC statistics.  The dot product is summed exactly as head_kernel sums it (eight 4-channel fmaf
// chains, pairwise tree), so a pass through this epilogue and a pass through this kernel's plain epilogue + head_kernel
// (rcu_unet_set_fuse_head(h, 0)) give the same bits.
//
// Hand-over through LDS, WAVE-LOCAL (no workgroup barrier): for tile r of the lane's four, the 64 lanes of a wave hold 4 tiles (g) x 16 pixels
// x 32 channels; lane (n, g) writes its two channels of the 16 pixels and then takes pixel n of tile g with all 32 channels.  A wave's 8 KB live
// in ITS OWN 1 KB pieces of the weight region of LDS buffer 1 -- pieces wave, wave + 4, ... -- which nobody reads once the tile's last chunk is
// past its barrier (the chunk's remaining weights are in registers by then) and which only this wave's own LDS-DMA writes again (the next
// tile's second chunk, issued after this epilogue): the other waves may run ahead into the next tile meanwhile.
//   byte address of (pixel k of tile g, channel pair j) = area + 1024 wave + 4096 (k >> 1) + 512 (k & 1) + 128 g + 8 (j ^ k)
// -- the slab of a pixel index k is a compile-time offset of the writes, j ^ k spreads the 16 pixels a read instruction covers per tile over
// 16 bank pairs: ds_write_b64 and ds_read_b64 (32 lanes per cycle over 64 banks) are both free of conflicts, one v_xor per access.
// The statistics entries of the lane's voxel are requested before the output transform: their round trip runs beside it.  The classifier's
// 64 weights + 2 biases are read from LDS (`wlds`: copied there once per kernel, WINO4_HEAD_LDS_BYTES behind the two chunk buffers; every lane
// reads the same address: a broadcast) -- read through the kernel argument's pointer they became vector loads from global memory, four
// dependent round trips per tile quarter on a wave that has nothing else to run meanwhile.
constexpr int WINO4_HEAD_LDS_BYTES = 512;
// PART: the level is padded (ConvArgs::part) -- logits and statistics are the caller's arrays over the REAL Hr x Wr image; the lanes whose pixel lies
// beyond it take part in the hand-over and touch no memory.
template <class T, bool PART = false>
__device__ __forceinline__ void wino4_epilogue_head(const ConvArgs& a, const f32x4 (&accv)[8], const WinoEpi& ep, int n0, int nstat, int y0, int x0,
                                                    int wave, int lane, uint32_t area, uint32_t wlds)
{
    // the lane-constant parts of the addresses below are recomputed per tile: hoisted to the kernel's start they would be two more registers alive
    // through the chunk pipeline, which has none to spare (the build spilled them to scratch)
    asm volatile("" : "+v"(lane));
    static_assert(T::TS == 1 && !T::FOLD && T::BC % 4 == 0 && T::BN == 32, "whole tiles of one slice, one 32-cout tile");
    static_assert(7 * 4096 + 3 * 1024 + 1024 <= T::W_DW * 4, "the hand-over area lies inside the weight region");
    asm volatile("s_nop 15\n\ts_nop 7");   // as wino4_epilogue: the MFMAs' results are read by v_accvgpr_read behind the compiler's back
    const int n16 = lane & 15, g = lane >> 4;
    // one batch of scalar loads
    const int H = PART ? a.Hr : a.H, W = PART ? a.Wr : a.W, flags = a.head_flags;
    float* const logits = a.head_logits;
    void* const stats = a.head_stats;
    const size_t V = a.head_V;
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
    int bs, by, sb, tr, tc;
    T::block_origin(wave, bs, by);
    T::tile_of(4 * g, sb, tr, tc);
    // the lane's pixel of tile (g, r): row n16 >> 2, column n16 & 3 of the tile, r tiles (4 r pixels) to the right of the lane's first
    const size_t HW = (size_t)H * W;
    const size_t hw0 = (size_t)(y0 + by + 4 * tr + (n16 >> 2)) * W + (size_t)(x0 + 4 * tc + (n16 & 3));
    [[maybe_unused]] const bool row_in = y0 + by + 4 * tr + (n16 >> 2) < H;
    [[maybe_unused]] const int xlim = W - (x0 + 4 * tc + (n16 & 3));
    typedef volatile __attribute__((address_space(3))) f32x2 lds_f32x2;
    const uint32_t mine = area + (uint32_t)wave * 1024u + (uint32_t)g * 128u;
    const uint32_t wbase = mine + 8u * (uint32_t)n16;
    const uint32_t rbase = mine + (uint32_t)(n16 >> 1) * 4096u + (uint32_t)(n16 & 1) * 512u + 8u * (uint32_t)n16;
    wino_static_for<0, 4>([&](auto r_c) {
        constexpr int r = decltype(r_c)::value;
        const size_t hw = hw0 + 4 * r;
        // (requesting the entries of all four quarters up front measured the same at 640 samples per launch and 4 % slower at 160: the round
        // trip is not what the wave waits for)
        const bool inside = !PART || (row_in && 4 * r < xlim);
        VoxelStats<2> st;
        if (stats != nullptr && inside) st.load(stats, (size_t)nstat * HW + hw, V, flags);   // nstat: the image the sample is a pass of
        {
            f32x2 y[4][4];
            wino4_output_tile<r>(accv, ep, relu_floor, y);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                *(lds_f32x2*)(uintptr_t)((wbase ^ (8u * k)) + (uint32_t)((k >> 1) * 4096 + (k & 1) * 512)) = y[k >> 2][k & 3];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float row[32];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const f32x2 v = *(lds_f32x2*)(uintptr_t)(rbase ^ (8u * j));
            row[2 * j] = v.x;
            row[2 * j + 1] = v.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // tile r + 1 overwrites the area: in order behind these reads (one wave, one LDS queue)
        __builtin_amdgcn_wave_barrier();
        float l[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float p[8];
#pragma unroll
            for (int sgm = 0; sgm < 8; ++sgm) {
                const f32x4 w = *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)(wlds + (uint32_t)(c * 32 + 4 * sgm) * 4u);
                float t = fmaf(w.x, row[4 * sgm], 0.f);
                t = fmaf(w.y, row[4 * sgm + 1], t);
                t = fmaf(w.z, row[4 * sgm + 2], t);
                p[sgm] = fmaf(w.w, row[4 * sgm + 3], t);
            }
            const float bias = *(const __attribute__((address_space(3))) float*)(uintptr_t)(wlds + (uint32_t)(64 + c) * 4u);
            l[c] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) + bias;
        }
        if (logits != nullptr && inside) {
            logits[((size_t)n0 * 2 + 0) * HW + hw] = l[0];
            logits[((size_t)n0 * 2 + 1) * HW + hw] = l[1];
        }
        if (stats != nullptr && inside) {
            softmax_inplace<2>(l);
            st.add(flags, l);
            st.store(stats, (size_t)nstat * HW + hw, V, flags);
        }
    });
}

// position rows in the order they are multiplied (see the header), and the position multiplied in slot q of that order
__host__ __device__ constexpr int wino4_row_of(int k) { return k == 0 ? 0 : k == 1 ? 5 : k - 1; }
__host__ __device__ constexpr int wino4_pos_of(int q) { return 6 * wino4_row_of(q / 6) + q % 6; }

// VAR: timing ablations for tools/wino4_check.py (wrong results): bit 0 no LDS-DMA inside the chunks, bit 1 no epilogue, bit 2 no
// input transform, bit 3 no chunk barrier
#ifdef RCU_WINO4_ABLATIONS
// VAR bits 8..10: epilogue ablations (EV of wino4_epilogue); bit 7: s_memtime per tile and wave (tile start, chunks done, epilogue
// done, vmcnt(0), barrier, first fragments read) -- kept in a type that is empty for the other variants, so that they compile to what
// they would be without it (the kernel sits at 500 of 512 registers: a dead variable is enough to make hipcc spill)
__device__ uint64_t g_w4_trace[256 * 4 * 4 * 8];
extern "C" __attribute__((visibility("default"))) int rcu_debug_w4_trace(uint64_t* dst)   // copies the trace out and clears it
{
    void* p = nullptr;
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_w4_trace), sizeof(g_w4_trace));
    if (e == hipSuccess) e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_w4_trace));
    if (e == hipSuccess) e = hipMemset(p, 0, sizeof(g_w4_trace));
    return (int)e;
}
template <bool ON>
struct Wino4Trace {
    uint32_t t[8];   // 0 tile start, 1 chunks done, 2 epilogue done, 3 vmcnt(0), 4 barrier, 5 first fragments read, 6 / 7 end of the tile's chunk 0 / 1
    __device__ __forceinline__ void mark(int i) { t[i] = (uint32_t)__builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void flush(int item, int wave, int lane)
    {
        const int k = (item - (int)blockIdx.x) / (int)gridDim.x;   // the workgroup's k-th tile
        if (lane == 0 && k < 4) {
            uint64_t* o = g_w4_trace + (((size_t)blockIdx.x * 4 + wave) * 4 + k) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (1u << 30) + (t[i] - t[0]);
        }
        t[0] = t[5];
    }
};
template <>
struct Wino4Trace<false> {
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int, int, int) {}
};
#define WINO4_TRACE_DECL Wino4Trace<(VAR & 128) != 0> w4trace
#define WINO4_TRACE_MARK(i) w4trace.mark(i)
#define WINO4_TRACE_FLUSH() w4trace.flush(item, wave, lane)
#else
#define WINO4_TRACE_DECL
#define WINO4_TRACE_MARK(i)
#define WINO4_TRACE_FLUSH()
#endif

// HEAD: the classifier head in the epilogue (wino4_epilogue_head).  total_items counts the tiles of ONE pass; the workgroup that owns a tile runs
// the tile of every pass of the group back to back (sample n0 + pass * head_images), so the passes' read-modify-writes of a voxel's statistics
// are ordered (pass 0 first, as head_kernel adds them) -- as rcu_wino.hip.
template <class T, int VAR = 0, bool HEAD = false, bool PART = false>
__global__ __launch_bounds__(256, 1) void conv_wino4_stream(const ConvArgs a, const int total_items)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    constexpr int KC = T::KC;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kq = lane >> 4;
    const int nchunks = (a.C1 + a.C2) / KC;   // even, >= 4 (checked by the launcher)
    const uint32_t wchunk_bytes = (uint32_t)a.NT * T::W_DW * 4u;
    const uint32_t in_chunk_bytes = a.in_chunk_bytes;

    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1), 0, a.src1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src2 ? a.src2 : a.src1), 0, a.src2 ? a.src2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);

    // fragment addresses (dword offsets inside a buffer) of the lane's patch columns j = 0..5, for patch rows 0..3 ([0]) and 4..5
    // ([1]: the swizzle's row bit flips); the row itself is an immediate offset
    int aCol[2][6];
    bool zero_left = false, zero_right = false;   // FULLW: this lane's patch column 0 / 5 lies outside the image
    {
        int bs, by, sb, tr, tc;
        T::block_origin(wave, bs, by);
        T::tile_of(m16, sb, tr, tc);
        const int s = bs + sb, R0 = by + 4 * tr;
        const int base = (s * T::SLICE_POS + R0 * T::PITCH) * 8 + 2 * kq;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                int x = 4 * tc + j - (T::FULLW ? 1 : 0);   // staged column of patch column j
                if (T::FULLW) x = min(max(x, 0), T::TW - 1);   // the two columns outside read a neighbour and are zeroed below
                aCol[ip][j] = base + 8 * (x ^ T::swizzle(x, s, R0 + 4 * ip));
            }
        if (T::FULLW) {
            zero_left = tc == 0;
            zero_right = tc == T::TCOLS - 1;
        }
    }
    const int b_addr = T::A_DW + (kq * T::BN + 2 * m16) * 2;
    const uint32_t w_voff = (uint32_t)(lane * 16);

    int item = wino_xcd_virtual_block(a.NT < 4 ? 4 : a.NT);
    [[maybe_unused]] int pass = 0;
    auto more_passes = [&]() {
        if constexpr (HEAD) return pass + 1 < wino_cold_args().head_passes;
        return false;
    };
    bool has_next = more_passes() || item + (int)gridDim.x < total_items;
    WinoTileId tile = wino_tile_id<T>(a, item), ntile = tile;
    // dp / dp_wtile: staging offsets and cout tile of the tile whose chunk the FRONT part of a chunk's LDS-DMA fetches (the chunk after the
    // current one), dpn / dpn_wtile: of the tile the BACK part fetches (the chunk after that).  They differ in a tile's last-but-one chunk.
    uint32_t dp[T::NA], dpn[T::NA], geo[T::NA];
    int dp_wtile = tile.wtile, dpn_wtile = tile.wtile;
#pragma unroll
    for (int j = 0; j < T::NA; ++j) {
        geo[j] = wino_slot_plan<T>(a, wino4_slot_geometry<T>(j, wave, lane));   // the slot's plan: tile-independent offset | border flags
        asm volatile("" : "+v"(geo[j]));
        dp[j] = dpn[j] = wino_slot_offset(geo[j], wino_tile_offset<T>(a, tile));
    }

    Wino4StorePlan store_plan = wino4_store_plan<T, PART>(a, wave, lane);
    asm volatile("" : "+v"(store_plan.out), "+v"(store_plan.pool));   // two registers through the loop, not their ingredients

    // LDS-DMA of Cin chunk kc of a tile into LDS buffer `buf`: the wave's NW weight pieces and NA input pieces of 1 KB.  No branch
    // per piece: a job without work (behind the workgroup's last chunk) points out of range, where the buffer load writes zeros
    // into a buffer nobody reads; the source tensor's descriptor is picked once per chunk.
    struct DmaJob {
        __amdgpu_buffer_rsrc_t rs;
        uint32_t cb, wso, lb;
    };
    auto dma_job = [&](int wtile, int kc, int buf, bool active) {
        DmaJob j;
        const int c0 = kc * KC;
        const bool first = c0 < a.C1;
        j.rs = first ? rs1 : rs2;
        j.cb = active ? wino_chunk_offset(in_chunk_bytes, first ? c0 : c0 - a.C1) : WINO_OOB;
        j.wso = active ? (uint32_t)kc * wchunk_bytes + (uint32_t)wtile * (T::W_DW * 4u) : WINO_OOB;
        j.lb = (uint32_t)buf * (T::BUF_DW * 4u);
        return j;
    };
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    auto dma_piece = [&](const DmaJob& job, const uint32_t (&off)[T::NA], auto i_c) {
        constexpr int I = decltype(i_c)::value;
        // the input pieces first: they come from HBM (or another XCD's writes) and are waited for at the chunk's barrier, the weight pieces -- L2
        // hits, shared by every workgroup of the cout tile -- can afford to be the late ones
        if constexpr (I < T::NA) {
            constexpr int j = I;
            if ((j + 1) * T::WAVES <= T::A_PIECES || j * T::WAVES + wave < T::A_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(job.rs, (lds_ptr_t)(uintptr_t)(lds_base + job.lb + (j * T::WAVES + wave) * 1024), 16, off[j],
                                                         job.cb, 0, RCU_DMA_IN_AUX);
        } else if constexpr (I < T::NW + T::NA) {
            constexpr int k = I - T::NA;
            if ((k + 1) * T::WAVES <= T::W_PIECES || k * T::WAVES + wave < T::W_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(uintptr_t)(lds_base + job.lb + T::A_DW * 4 + (k * T::WAVES + wave) * 1024),
                                                         16, w_voff, job.wso + (uint32_t)(k * T::WAVES + wave) * 1024u, 0, 0);
        }
    };

    // The LDS-DMA runs TWO chunks ahead of the arithmetic: the first PB pieces of a chunk (input pieces: they come from HBM; PB = 4 measured best of 2 / 4 / 8) are issued right
    // behind the barrier of the chunk two before it -- into the buffer that barrier has just freed --, the rest in the front groups of the chunk
    // before it; its data is waited for at that chunk's barrier.  (One chunk ahead, the last input piece had 1.3k cycles to land.)
    constexpr int PB = (T::TS == 1 && T::TH == 32) ? 4 : 0;   // the 32x32-pixel tile only (the 96x64 and 192x128 levels: few chunks per tile, inputs from HBM); the others lose 1-2 % with it
    static_assert(PB <= T::NA, "the back part consists of input pieces");
    {
        const DmaJob job = dma_job(dp_wtile, 0, 0, true);
        wino_static_for<0, T::NW + T::NA>([&](auto i_c) { dma_piece(job, dp, i_c); });
    }
    if constexpr (HEAD) {   // the classifier: [2][32] weights, [2] biases (wino4_epilogue_head)
        if (tid < 66) smem[2 * T::BUF_DW + tid] = tid < 64 ? a.head_w[tid] : a.head_b[tid - 64];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    {
        const DmaJob job = dma_job(dp_wtile, 1, 1, true);   // the back part of the first tile's chunk 1
        wino_static_for<0, PB>([&](auto i_c) { dma_piece(job, dp, i_c); });
    }

    f32x4 accv[8];   // accumulators of (cout block 1, positions 18..25); all others in a[0:255] (wino4_areg)
    WinoEpiRaw epr;

    // Pipeline of one Cin chunk out of LDS buffer BUF (measured rules of the one-wave-per-SIMD regime, tools/microbench/
    // gen_mfma_valu_1wave.py: an LDS read between two MFMAs is free, a VALU instruction costs ~5 cycles of matrix time whether packed
    // or not and less in a clump than spread over the MFMA gaps):
    //   groups 0..13 of 8 MFMAs (two positions x 2 channel steps x 2 cout blocks), the weight reads of the positions two groups ahead
    //     and the chunk's LDS-DMA pieces between the MFMAs, the transform of the coming position rows as ONE packed clump per group;
    //   s_waitcnt vmcnt(0) + barrier: the next chunk (of this tile or of the workgroup's next tile) has landed in the other buffer;
    //   groups 14..17 -- they only need registers: their weights were read before the barrier -- with the RAW patch of the next
    //     chunk read between their MFMAs, so that a chunk starts with its transform clump instead of an LDS round trip.
    // dcur / dnext: the raw patch registers of this chunk and of the next one (the two arrays swap roles from chunk to chunk).
    constexpr int AHEAD = 4, NPRE = 14;
    auto load_patch_row = [&](f32x2 (&d)[36], const float* Ab, int i) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
            // volatile: keeps hipcc from fusing pairs of these reads into ds_read2_b64
            d[6 * i + j] = *(const volatile __attribute__((address_space(3))) f32x2*)(Ab + aCol[i >> 2][j] + i * (T::PITCH * 8));
        if constexpr (T::FULLW) {   // zero padding left and right of the image
            const f32x2 z = {0.f, 0.f};
            d[6 * i] = zero_left ? z : d[6 * i];
            d[6 * i + 5] = zero_right ? z : d[6 * i + 5];
        }
    };
    auto load_weights = [&](f32x4 (&bv)[36], const float* Ab, int p) { bv[p] = *reinterpret_cast<const f32x4*>(Ab + b_addr + p * (8 * T::BN)); };
    // B^T d B in place on channel pairs (packed operations).  One-dimensional transform of (x0..x5):
    //   t0 = 4 x0 - 5 x2 + x4,  t5 = 4 x1 - 5 x3 + x5,  with a = x4 - 4 x2, b = x3 - 4 x1, c = x4 - x2, e = x3 - x1:
    //   t1 = a + b, t2 = a - b, t3 = c + 2 e, t4 = c - 2 e.
    auto row_transform = [&](f32x2 (&d)[36], int i) {   // along the columns of position row i
        f32x2* const x = d + 6 * i;
        const f32x2 t0 = 4.f * x[0] + (x[4] - 5.f * x[2]);
        const f32x2 t5 = 4.f * x[1] + (x[5] - 5.f * x[3]);
        const f32x2 aa = x[4] - 4.f * x[2], bb = x[3] - 4.f * x[1], cc = x[4] - x[2], ee = x[3] - x[1];
        x[0] = t0;
        x[1] = aa + bb;
        x[2] = aa - bb;
        x[3] = cc + 2.f * ee;
        x[4] = cc - 2.f * ee;
        x[5] = t5;
    };
    auto transform_head = [&](f32x2 (&d)[36]) {   // before the first MFMA: T0 of every column, then the columns of position row 0
#pragma unroll
        for (int j = 0; j < 6; ++j) d[j] = 4.f * d[j] + (d[24 + j] - 5.f * d[12 + j]);
        row_transform(d, 0);
    };
    auto transform_clump = [&](f32x2 (&d)[36], int G) {   // 12 packed operations behind MFMA group G
        if (G == 0) {          // T5 of every column (rows 1, 3, 5 still raw)
#pragma unroll
            for (int j = 0; j < 6; ++j) d[30 + j] = 4.f * d[6 + j] + (d[30 + j] - 5.f * d[18 + j]);
        }
        if (G == 1) row_transform(d, 5);
        if (G == 2 || G == 3) {   // a -> row 2, c -> row 4, b -> row 1, e -> row 3 for three columns each
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int j = 3 * (G - 2) + jj;
                const f32x2 x1 = d[6 + j], x2 = d[12 + j], x3 = d[18 + j], x4 = d[24 + j];
                d[12 + j] = x4 - 4.f * x2;
                d[24 + j] = x4 - x2;
                d[6 + j] = x3 - 4.f * x1;
                d[18 + j] = x3 - x1;
            }
        }
        if (G == 4) {          // t1 = a + b, t2 = a - b
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const f32x2 aa = d[12 + j], bb = d[6 + j];
                d[6 + j] = aa + bb;
                d[12 + j] = aa - bb;
            }
        }
        if (G == 5) row_transform(d, 1);
        if (G == 6) row_transform(d, 2);
        if (G == 7) {          // t3 = c + 2 e, t4 = c - 2 e
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const f32x2 cc = d[24 + j], ee = d[18 + j];
                d[18 + j] = cc + 2.f * ee;
                d[24 + j] = cc - 2.f * ee;
            }
        }
        if (G == 8) row_transform(d, 3);
        if (G == 9) row_transform(d, 4);
    };

    auto chunk = [&](auto buf_c, auto first_c, int kc, f32x2 (&d)[36], f32x2 (&dn)[36], f32x4 (&bv)[36], f32x4 (&bvn)[36]) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        const float* const Ab = smem + BUF * T::BUF_DW;
        const float* const An = smem + (BUF ^ 1) * T::BUF_DW;
        const bool more = kc + 1 < nchunks, more2 = kc + 2 < nchunks;
        if (kc == nchunks - 2 && has_next) {   // last-but-one chunk of the tile: from its barrier on the DMA works on the workgroup's next tile
            const WinoTileConsts ca = wino_tile_consts(wino_cold_args());   // one batch of scalar loads, one wait
            if (more_passes()) {
                ntile = tile;
                ntile.n0 = tile.n0 + wino_cold_args().head_images;
            } else {
                ntile = wino_tile_id<T>(ca, item + (int)gridDim.x);
            }
            dpn_wtile = ntile.wtile;
            const WinoTileOffset nto = wino_tile_offset<T>(ca, ntile);
#pragma unroll
            for (int j = 0; j < T::NA; ++j) dpn[j] = wino_slot_offset(geo[j], nto);
        }
        if (!more) {   // last chunk: both parts fetch the next tile
            dp_wtile = dpn_wtile;
#pragma unroll
            for (int j = 0; j < T::NA; ++j) dp[j] = dpn[j];
        }
        // epilogue constants: loaded a chunk early, so that the barrier wait of the last-but-one chunk covers them and the epilogue
        // does not wait for memory (the LDS-DMA of the next tile is still in flight then)
        if (kc == nchunks - 2) epr = wino4_epilogue_load<T>(wino_cold_args(), tile.wtile, tile.n0, wave, lane);
        // front part: the rest of the next chunk (this tile's or, behind the last chunk, the next tile's first) into the other buffer;
        // back part (behind the barrier): the first pieces of the chunk after that into THIS chunk's buffer, which the barrier frees
        const DmaJob job = dma_job(dp_wtile, more ? kc + 1 : 0, BUF ^ 1, more || has_next);
        const DmaJob jobb = dma_job(dpn_wtile, more2 ? kc + 2 : kc + 2 - nchunks, BUF, more2 || has_next);
        if constexpr ((VAR & 4) == 0) transform_head(d);
        wino_static_for<0, 18>([&](auto g_c) {
            constexpr int G = decltype(g_c)::value;
            constexpr int p0 = wino4_pos_of(2 * G), p1 = wino4_pos_of(2 * G + 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (G == NPRE) {
                // behind a tile's last chunk the wait comes after the epilogue (main loop): the first chunk of the next tile is a
                // cold fetch, and the epilogue is the work to hide it behind
                if constexpr ((VAR & 8) == 0) {
                    // vmcnt(0): this wave's pieces of the next chunk (the next tile's first chunk behind a tile's last) have landed.
                    // INVARIANT the two-chunks-ahead LDS-DMA relies on: when a wave leaves __syncthreads(), every LDS read any wave has issued
                    // from buffer BUF is COMPLETE -- __syncthreads() lowers to `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier` (checked in the gfx950
                    // assembly of every in-loop barrier), and the weights of groups 14..17 were read in groups 10..13 -- so the back-part DMA issued
                    // right behind it may overwrite BUF.  A bare __builtin_amdgcn_s_barrier() here would be a write-after-read race on BUF; if
                    // the barrier is ever hand-rolled it needs an explicit s_waitcnt lgkmcnt(0) in front.
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    __syncthreads();   // everyone past the LDS reads of buffer BUF and done filling the other one
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // what goes between the 8 MFMAs of the group: unit u follows MFMA u
            auto filler = [&](auto u_c) {
                constexpr int U = decltype(u_c)::value;
                if constexpr (G < NPRE) {
                    // weights of the positions two groups ahead; in the last groups before the barrier also those of groups 14..17
                    if constexpr (U == 0 || U == 4) {
                        constexpr int q = 2 * G + AHEAD + (U == 4 ? 1 : 0);
                        if constexpr (q < 2 * NPRE) load_weights(bv, Ab, wino4_pos_of(q));
                    }
                    if constexpr (G >= NPRE - 4 && (U == 2 || U == 6)) {
                        constexpr int q = 2 * NPRE + 2 * (G - (NPRE - 4)) + (U == 6 ? 1 : 0);
                        load_weights(bv, Ab, wino4_pos_of(q));
                    }
                    if constexpr ((VAR & 1) == 0 && G < 10 && (U == 1 || U == 5)) dma_piece(job, dp, std::integral_constant<int, PB + 2 * G + (U == 5 ? 1 : 0)>{});
                } else {
                    if constexpr ((VAR & 1) == 0 && (U == 1 || U == 5) && 2 * (G - NPRE) + (U == 5 ? 1 : 0) < PB)
                        dma_piece(jobb, dpn, std::integral_constant<int, 2 * (G - NPRE) + (U == 5 ? 1 : 0)>{});
                    // behind the barrier: the raw patch of the next chunk (rows 0, 2, 4, 1, 3, 5) and its first weights.  Unconditional
                    // (a conditional load would keep the old contents of dn alive through the whole chunk); in a tile's last
                    // chunk the values are dropped and the next tile's first chunk is read again behind the epilogue (load_first):
                    // the epilogue needs the registers
                    {
                        constexpr int slot = 8 * (G - NPRE) + U;           // 0..31
                        constexpr int order[6] = {0, 2, 4, 1, 3, 5};
                        if constexpr (slot < 30) {
                            constexpr int r = order[slot / 5];
                            if constexpr (slot % 5 == 0) load_patch_row(dn, An, r);   // six reads per slot group of five gaps
                        }
                        if constexpr (slot == 30)
                            wino_static_for<0, AHEAD>([&](auto q_c) { load_weights(bvn, An, wino4_pos_of(decltype(q_c)::value)); });
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            wino4_mfma<0, p0, FIRST, true, (VAR & 16) != 0>(accv, d[p0].x, bv[p0].x);
            filler(std::integral_constant<int, 0>{});
            wino4_mfma<1, p0, FIRST, true, (VAR & 16) != 0>(accv, d[p0].x, bv[p0].z);
            filler(std::integral_constant<int, 1>{});
            wino4_mfma<0, p1, FIRST, true, (VAR & 16) != 0>(accv, d[p1].x, bv[p1].x);
            filler(std::integral_constant<int, 2>{});
            wino4_mfma<1, p1, FIRST, true, (VAR & 16) != 0>(accv, d[p1].x, bv[p1].z);
            filler(std::integral_constant<int, 3>{});
            wino4_mfma<0, p0, false, true, (VAR & 16) != 0>(accv, d[p0].y, bv[p0].y);
            filler(std::integral_constant<int, 4>{});
            wino4_mfma<1, p0, false, true, (VAR & 16) != 0>(accv, d[p0].y, bv[p0].w);
            filler(std::integral_constant<int, 5>{});
            wino4_mfma<0, p1, false, true, (VAR & 16) != 0>(accv, d[p1].y, bv[p1].y);
            filler(std::integral_constant<int, 6>{});
            wino4_mfma<1, p1, false, true, (VAR & 16) != 0>(accv, d[p1].y, bv[p1].w);
            filler(std::integral_constant<int, 7>{});
            // transform work for the coming position rows: one clump
            if constexpr ((VAR & 4) == 0) transform_clump(d, G);
        });
        if constexpr ((VAR & 64) != 0) {   // probe: read the next chunk's patch and first weights again behind the chunk
            __syncthreads();
#pragma unroll
            for (int ii = 0; ii < 6; ++ii) load_patch_row(dn, An, ii);
            wino_static_for<0, AHEAD>([&](auto q_c) { load_weights(bvn, An, wino4_pos_of(decltype(q_c)::value)); });
        }
    };

    f32x2 dA[36], dB[36];
    f32x4 bvA[36], bvB[36];
    // a tile's first chunk: nothing to hide the LDS round trip behind
    auto load_first = [&]() {
#pragma unroll
        for (int ii = 0; ii < 6; ++ii) load_patch_row(dA, smem, ii < 3 ? 2 * ii : 2 * (ii - 3) + 1);
        wino_static_for<0, AHEAD>([&](auto q_c) { load_weights(bvA, smem, wino4_pos_of(decltype(q_c)::value)); });
    };
    load_first();
    WINO4_TRACE_DECL;
    WINO4_TRACE_MARK(0);

    for (;;) {
        if constexpr ((VAR & 32) != 0) {   // probe (wrong results): no zero-accumulator form of the tile's first chunk -- two copies of the chunk code instead of four
            for (int kc = 0; kc < nchunks; kc += 2) {
                chunk(std::integral_constant<int, 0>{}, std::false_type{}, kc, dA, dB, bvA, bvB);
                chunk(std::integral_constant<int, 1>{}, std::false_type{}, kc + 1, dB, dA, bvB, bvA);
            }
        } else {
        chunk(std::integral_constant<int, 0>{}, std::true_type{}, 0, dA, dB, bvA, bvB);
        WINO4_TRACE_MARK(6);
        chunk(std::integral_constant<int, 1>{}, std::false_type{}, 1, dB, dA, bvB, bvA);
        WINO4_TRACE_MARK(7);
        for (int kc = 2; kc < nchunks; kc += 2) {
            chunk(std::integral_constant<int, 0>{}, std::false_type{}, kc, dA, dB, bvA, bvB);
            chunk(std::integral_constant<int, 1>{}, std::false_type{}, kc + 1, dB, dA, bvB, bvA);
        }
        }
        WINO4_TRACE_MARK(1);
        if constexpr (HEAD)
            wino4_epilogue_head<T, PART>(wino_cold_args(), accv, wino_epilogue_fold(epr), tile.n0, tile.n0 - pass * wino_cold_args().head_images, tile.y0, tile.x0,
                                         wave, lane, lds_base + (uint32_t)(T::BUF_DW + T::A_DW) * 4u, lds_base + (uint32_t)(2 * T::BUF_DW) * 4u);
        else if constexpr ((VAR & 2) == 0)
            wino4_epilogue<T, (VAR >> 8), PART>(wino_cold_args(), accv, wino_epilogue_fold(epr), tile.wtile, tile.n0, tile.y0, tile.x0, wave, lane, store_plan);
        WINO4_TRACE_MARK(2);
        if (!has_next) break;
        WINO4_TRACE_MARK(3);   // (no wait and no barrier here any more: the tile's last chunk waited for the next tile's first chunk at its barrier)
        WINO4_TRACE_MARK(4);
        load_first();
        WINO4_TRACE_MARK(5);
        WINO4_TRACE_FLUSH();
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

using W4Cfg0 = Wino4Tile<1, 2, 8, 1, 4>;   // 32x32 pixels of one slice
using W4Cfg1 = Wino4Tile<1, 2, 8, 2, 2>;   // 16x32 pixels of two consecutive slices (heights not divisible by 32)
using W4Cfg2 = Wino4Tile<2, 2, 4, 4, 1, true>;   // 8x16 pixels of eight consecutive slices, full image width (the 24x16 level)
using W4Cfg3 = Wino4Tile<2, 2, 4, 4, 1, true, true>;   // 12x8 pixels of eight consecutive slices, folded into the same block geometry (the 12x8 level)
using W4Cfg4 = Wino4Tile<1, 2, 8, 4, 1, true>;   // 8x32 pixels of four consecutive slices, full image width (round 6: 24x32 levels -- the reference's ISIC size 192x256 -- without padding to 32x32)

static const ConvConfigInfo kWino4Info[6] = {
    {W4Cfg0::TS, W4Cfg0::TH, W4Cfg0::TW, W4Cfg0::BN, 8, 36, "conv3x3_winograd4<T32x32,N32,K8>", 8, 0, 3},
    {W4Cfg1::TS, W4Cfg1::TH, W4Cfg1::TW, W4Cfg1::BN, 8, 36, "conv3x3_winograd4<S2T16x32,N32,K8>", 8, 0, 3},
    {W4Cfg2::TS, W4Cfg2::TH, W4Cfg2::TW, W4Cfg2::BN, 8, 36, "conv3x3_winograd4<S8T8x16,N32,K8>", 8, 0, 3},
    {W4Cfg3::TS, W4Cfg3::TH, W4Cfg3::TW, W4Cfg3::BN, 8, 36, "conv3x3_winograd4<S8T12x8,N32,K8>", 8, 0, 3},
    {W4Cfg0::TS, W4Cfg0::TH, W4Cfg0::TW, W4Cfg0::BN, 8, 36, "conv3x3_winograd4<T32x32,N32,K8>+head", 8, 0, 3},
    {W4Cfg4::TS, W4Cfg4::TH, W4Cfg4::TW, W4Cfg4::BN, 8, 36, "conv3x3_winograd4<S4T8x32,N32,K8>", 8, 0, 3},
};

const ConvConfigInfo& wino4_config_info(int cfg) { return kWino4Info[cfg - CONV_CFG_WINO4_T32x32_N32]; }

template <class T, int VAR, bool HEAD = false, bool PART = false>
static hipError_t launch_wino4_var(const ConvArgs& a, hipStream_t stream)
{
    constexpr int lds_bytes = T::LDS_BYTES + (HEAD ? WINO4_HEAD_LDS_BYTES : 0);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino4_stream<T, VAR, HEAD, PART>), lds_bytes);
    if (e != hipSuccess) return e;
    // HEAD: the work items of one pass (TS == 1: a slice group is a sample); the kernel runs every pass of the group on each
    const unsigned items = (unsigned)a.NT * a.tiles_x * a.tiles_y * (HEAD ? a.head_images : a.slice_groups);
    const unsigned grid = wino_persistent_grid(items);
    hipLaunchKernelGGL((conv_wino4_stream<T, VAR, HEAD, PART>), dim3(grid), dim3(T::THREADS), lds_bytes, stream, a, (int)items);
    return hipGetLastError();
}

template <class T, bool HEAD = false>
static hipError_t launch_wino4_cfg(const ConvArgs& a, hipStream_t stream)
{
    if (HEAD && (a.head_w == nullptr || a.head_b == nullptr || a.NT != 1 || a.pooled != nullptr || a.mask2 != nullptr || a.head_passes < 1 ||
                 a.head_images * a.head_passes != a.N || (a.head_passes > 1 && a.head_logits != nullptr) ||
                 (a.head_logits == nullptr && a.head_stats == nullptr)))
        return hipErrorInvalidValue;
    const int nchunks = (a.C1 + a.C2) / T::KC;
    if (nchunks < 4 || (nchunks & 1) != 0 || a.NTW_total != a.NT || a.src1_bytes == 0 || a.wpack_bytes == 0 ||
        (a.C2 != 0 && a.C2 != a.C1) || a.H % T::TH != 0 || a.W % T::TW != 0 || (T::FULLW && a.W != T::TW) || (T::FOLD && a.pooled != nullptr) ||
        (size_t)a.N * a.H * a.W * a.CoutP * 4 >= ((size_t)1 << 31))
        return hipErrorInvalidValue;
    // padded level: the real image lies inside the tile grid, the output / pooled tensors hold it
    if (a.part && (a.Hr < 1 || a.Wr < 1 || a.Hr > a.H || a.Wr > a.W || a.out_H < a.Hr || a.out_W < a.Wr ||
                   (size_t)a.N * a.out_H * a.out_W * a.CoutP * 4 >= ((size_t)1 << 31) ||
                   (a.pooled != nullptr && (a.pool_H < (a.Hr >> 1) || a.pool_W < (a.Wr >> 1)))))
        return hipErrorInvalidValue;
#ifdef RCU_WINO4_ABLATIONS   // timing experiments of tools/wino4_check.py (make EXTRA=-DRCU_WINO4_ABLATIONS); results are wrong
    const char* const v = HEAD ? nullptr : getenv("RCU_W4_VARIANT");
    switch (v ? atoi(v) : 0) {
        case 1: return launch_wino4_var<T, 1>(a, stream);
        case 2: return launch_wino4_var<T, 2>(a, stream);
        case 3: return launch_wino4_var<T, 3>(a, stream);
        case 4: return launch_wino4_var<T, 4>(a, stream);
        case 5: return launch_wino4_var<T, 5>(a, stream);
        case 6: return launch_wino4_var<T, 6>(a, stream);
        case 7: return launch_wino4_var<T, 7>(a, stream);
        case 8: return launch_wino4_var<T, 8>(a, stream);
        case 11: return launch_wino4_var<T, 11>(a, stream);
        case 15: return launch_wino4_var<T, 15>(a, stream);
        case 16: return launch_wino4_var<T, 16>(a, stream);
        case 64: return launch_wino4_var<T, 64>(a, stream);
        case 128: return launch_wino4_var<T, 128>(a, stream);
        case 128 + 32: return launch_wino4_var<T, 128 + 32>(a, stream);
        case 256: return launch_wino4_var<T, 256>(a, stream);
        case 512: return launch_wino4_var<T, 512>(a, stream);
        case 1024: return launch_wino4_var<T, 1024>(a, stream);
        case 128 + 256: return launch_wino4_var<T, 128 + 256>(a, stream);
        case 128 + 512: return launch_wino4_var<T, 128 + 512>(a, stream);
        case 128 + 1024: return launch_wino4_var<T, 128 + 1024>(a, stream);
        case 128 + 1024 + 512: return launch_wino4_var<T, 128 + 1024 + 512>(a, stream);
        default: break;
    }
#endif
    if (a.part) return launch_wino4_var<T, 0, HEAD, true>(a, stream);
    return launch_wino4_var<T, 0, HEAD>(a, stream);
}

hipError_t launch_conv_wino4(int cfg, const ConvArgs& a, hipStream_t stream)
{
    switch (cfg) {
        case CONV_CFG_WINO4_T32x32_N32: return launch_wino4_cfg<W4Cfg0>(a, stream);
        case CONV_CFG_WINO4_S2T16x32_N32: return launch_wino4_cfg<W4Cfg1>(a, stream);
        case CONV_CFG_WINO4_S8T8x16_N32: return launch_wino4_cfg<W4Cfg2>(a, stream);
        case CONV_CFG_WINO4_S8T12x8_N32: return launch_wino4_cfg<W4Cfg3>(a, stream);
        case CONV_CFG_WINO4_T32x32_N32_HEAD: return launch_wino4_cfg<W4Cfg0, true>(a, stream);
        case CONV_CFG_WINO4_S4T8x32_N32: return launch_wino4_cfg<W4Cfg4>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rcu
