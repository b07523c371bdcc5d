// Shared pieces of the Winograd kernels (rcu_wino.hip: F(2x2,3x3) conv unit; rcu_wino_up.hip: F(2x2,2x2) sub-pixel
// up-convolution): tile geometry, LDS-DMA staging plan, epilogue constants.  gfx950 only.
#pragma once
#include <cstdlib>
#include "rcu_kernels.h"

#include <type_traits>

namespace rcu {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Cache policy of the LDS-DMA loads of INPUT activations (the aux immediate of buffer_load ... lds: 1 sc0, 2 nt, 16 sc1).  0 = cached; experiment builds set
// it (profiles/r04_nt_loads.txt).
#ifndef RCU_DMA_IN_AUX
#define RCU_DMA_IN_AUX 0
#endif

// Exchange with the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2].
__device__ __forceinline__ float wino_swap_adjacent(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

// The kernel's arguments as they sit in the kernarg segment (ConvArgs is the first argument), re-read where the rarely used
// fields are needed -- epilogue constants, output pointers, next-tile arithmetic -- instead of living in SGPRs through the chunk
// loop (the kernels need ~150 SGPRs otherwise; the spills to VGPR lanes come back as v_readlane in the loop).  The empty
// assembly keeps hipcc from hoisting the loads back to the kernel entry.
__device__ __forceinline__ const ConvArgs& wino_cold_args()
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const ConvArgs*)p;
}

// 16-byte buffer store at rsrc[voff + soff].  hipcc lets a VALU write to the data registers follow such a store at once when
// the offset sits in an SGPR (the store-data hazard it knows is the immediate-offset one); on gfx950 that loses data now and
// then (measured: an up-conv epilogue whose selects overwrote v[8:11] two instructions after the store gave a few wrong
// pixels, different ones per run).  Two wait states, pinned right behind the store.
__device__ __forceinline__ void wino_store16(const f32x4& v, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff)
{
#ifndef RCU_STORE_AUX
#define RCU_STORE_AUX 0
#endif
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voff, soff, RCU_STORE_AUX);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}

template <int I, int N, class F>
__device__ __forceinline__ void wino_static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wino_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int wino_xcd_virtual_block(int group)
{
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    if ((g & 7) != 0 || ((g >> 3) % group) != 0) return b;
    const int xcd = b & 7, slot = b >> 3;
    return ((slot / group) * 8 + xcd) * group + slot % group;
}

// A 16-tile MFMA row block is 2 tile rows x 8 tile columns: 4 x 16 pixels of one slice (SW = 1) or, for images only
// 8 pixels wide, 4 x 8 pixels of each of SW = 2 consecutive slices.
template <int TS_, int TH_, int TW_, int BN_, int WM_, int WN_, int SW_ = 1, int NPOS_ = 16>
struct WinoTile {
    static constexpr int TS = TS_, TH = TH_, TW = TW_, BN = BN_, WM = WM_, WN = WN_, SW = SW_;
    static constexpr int NPOS = NPOS_;                        // Winograd positions a wave accumulates: 16 (F(2x2,3x3)) or 18 (two F(2x2,2x2) classes)
    static constexpr int KC = 8, THREADS = 512, WAVES = 8;
    static constexpr int TILES = TS * (TH / 2) * (TW / 2);
    static constexpr int BPS = (TH / 4) * (TW * SW / 16);     // 16-tile blocks per group of SW slices
    static constexpr int PITCH = (TW + 2 + 7) / 8 * 8;        // positions per halo row; a multiple of 8 (bank analysis)
    // positions per slice image; with SW = 2 an odd multiple of 8 so that the two slices of a block fall on different
    // halves of the 64 banks
    static constexpr int SLICE_POS = (TH + 2) * PITCH + (SW == 2 ? 8 : 0);
    static constexpr int HALF_POS = TS * SLICE_POS;           // positions of one channel-half image
    static constexpr int A_POS = 2 * HALF_POS;                // 16-byte positions of the input image
    static constexpr int A_PIECES = (A_POS + 63) / 64;        // 1-KB LDS-DMA pieces (one wave instruction each)
    static constexpr int NA = (A_PIECES + WAVES - 1) / WAVES; // pieces per wave
    static constexpr int A_DW = A_PIECES * 256;
    static constexpr int W_DW = NPOS * 4 * BN * 2;            // [p][q][n][2]
    static constexpr int W_PIECES = W_DW / 256;
    static constexpr int NW = (W_PIECES + WAVES - 1) / WAVES; // pieces per wave (piece j * WAVES + wave)
    static constexpr int BUF_DW = A_DW + W_DW;
    static constexpr int LDS_BYTES = 2 * BUF_DW * 4;
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(TILES == 16 * WM && BN == 32 * WN, "wave tiling");
    static_assert(TH % 4 == 0 && (TW * SW) % 16 == 0 && (SW == 1 || (SW == 2 && TW == 8 && TS % 2 == 0)), "block geometry");
    static_assert(SW == 1 || SLICE_POS % 16 == 8, "slice stride");
    static_assert(W_DW % 256 == 0, "weight staging");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    // block index -> (first slice of the block inside the tile, pixel origin of the block inside the slice tile)
    static __device__ __forceinline__ void block_origin(int blk, int& s, int& y, int& x)
    {
        const int grp = blk / BPS, rb = blk % BPS;
        s = grp * SW;
        y = 4 * (rb / (TW * SW / 16));
        x = SW == 1 ? 16 * (rb % (TW / 16)) : 0;
    }
};

// Per-lane constants of the conv-unit epilogue (the lane's two couts of the tile's slice): loaded while the tile's
// last-but-one Cin chunk is multiplied and folded at the start of the last one, so that the epilogue does not start with
// a global-memory round trip and the raw values are gone before the register-heavy last chunk.
struct WinoEpiRaw {
    f32x2 al, bb, be;   // folded BatchNorm / bias constants of the lane's two couts
    float mk[2];        // Dropout2d factors as loaded (garbage where no site applies: `site` says which do)
    int site;           // bit b: cout b of the lane belongs to a dropout site
};
struct WinoEpi {   // out = relu(acc * scale + shift)
    float scale[2], shift[2];
};
__device__ __forceinline__ WinoEpi wino_epilogue_fold(const WinoEpiRaw& r)
{
    WinoEpi e;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float mk = (r.site >> b) & 1 ? r.mk[b] : 1.f;
        e.scale[b] = r.al[b] * mk;
        e.shift[b] = r.bb[b] * mk + r.be[b];
    }
    return e;
}

// All loads are unconditional (clamped addresses) and nothing here depends on a loaded value, so the wave does not
// wait for memory where this is called (the start of a tile's last Cin chunk) but only in the epilogue.
template <class T>
__device__ __forceinline__ WinoEpiRaw wino_epilogue_load(const ConvArgs& a, int ntile, int n0, int wm, int wn, int lane)
{
    WinoEpiRaw e;
    const int co = ntile * T::BN + wn * 32 + 2 * (lane & 15);   // < NT * BN, the length of alpha / betab / beta
    int bs, by, bx;
    T::block_origin(wm, bs, by, bx);
    const int n = min(n0 + bs + (T::SW == 2 ? (lane >> 4) & 1 : 0), a.N - 1);
    e.al = *reinterpret_cast<const f32x2*>(a.alpha + co);
    e.bb = *reinterpret_cast<const f32x2*>(a.betab + co);
    e.be = *reinterpret_cast<const f32x2*>(a.beta + co);
    e.site = 0;
    e.mk[0] = e.mk[1] = 1.f;
    if (a.mask != nullptr) {   // wave-uniform
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = co + b;
            const bool second = a.mask2 != nullptr && c >= a.Csplit;
            const int cm = second ? a.Cmask2 : a.Cmask, ci = second ? c - a.Csplit : c;
            const float* const row = second ? a.mask2 + (size_t)n * a.Cmask2 : a.mask + (size_t)n * a.Cmask;
            e.mk[b] = row[min(ci, cm - 1)];
            e.site |= (ci < cm ? 1 : 0) << b;
        }
    }
    return e;
}

constexpr uint32_t WINO_OOB = 0x80000000u;

// Tile coordinates of a work item (wave-uniform).
struct WinoTileId {
    int wtile, n0, y0, x0;
};

// The kernel arguments the next-tile arithmetic needs (wino_tile_id + wino_tile_offset), fetched from the kernarg segment in ONE batch of scalar
// loads behind one wait.  Read field by field where they were used, the compiler put a load, a wait and a branch (magic == 0 ?) in front of each of
// the three divisions and another round trip in front of the offsets: four to five scalar-memory round trips in a row per tile, which a wave that
// is alone on its SIMD (rcu_wino4.hip) sits out in full.
struct WinoTileConsts {
    int NTW_total, tiles_x, tiles_y, H, W, C1;
    uint32_t magic_ntw, magic_tx, magic_ty, in_pix_bytes;
};
__device__ __forceinline__ WinoTileConsts wino_tile_consts(const ConvArgs& a)
{
    WinoTileConsts c{a.NTW_total, a.tiles_x, a.tiles_y, a.H, a.W, a.C1, a.magic_ntw, a.magic_tx, a.magic_ty, a.in_pix_bytes};
    asm volatile("" : "+s"(c.NTW_total), "+s"(c.tiles_x), "+s"(c.tiles_y), "+s"(c.H), "+s"(c.W), "+s"(c.C1), "+s"(c.magic_ntw), "+s"(c.magic_tx),
                      "+s"(c.magic_ty), "+s"(c.in_pix_bytes));
    return c;
}

template <class T, class A>
__device__ __forceinline__ WinoTileId wino_tile_id(const A& a, int item)
{
    WinoTileId t;
    // item / d by multiplication with ceil(2^32 / d): exact while item * d < 2^32 (the launcher falls back to magic = 0)
    auto div = [](int x, int d, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)x, magic) : x / d; };
    int mtile = div(item, a.NTW_total, a.magic_ntw);
    t.wtile = item - mtile * a.NTW_total;
    const int q = div(mtile, a.tiles_x, a.magic_tx);
    const int tx = mtile - q * a.tiles_x;
    const int sg = div(q, a.tiles_y, a.magic_ty);
    const int ty = q - sg * a.tiles_y;
    t.n0 = sg * T::TS;
    t.y0 = ty * T::TH;
    t.x0 = tx * T::TW;
    return t;
}

// Per-lane staging plan of one output tile: byte offset (into src1 and src2 alike) of the pixel chunk that slot j of this lane
// fetches by LDS-DMA; slots of zero padding / pitch padding / outside the batch point far out of range, where the
// buffer load returns zeros.  The tile-independent part -- which (channel half, slice, halo row, pixel column) the slot
// holds -- is packed into one register per slot at kernel start (wino_slot_geometry).
// The LDS input image of the F(2x2,3x3) / F(2x2,2x2) kernels is [channel half][slice][halo row][position][4 channels]: the patch reads
// (ds_read_b64: 32 lanes per cycle over 64 banks) are free of bank conflicts on it.  -DRCU_WINO_IMAGE8=1 (experiment builds) stages
// rcu_wino4.hip's image [slice][halo row][position][8 channels] instead -- adjacent lanes of an LDS-DMA instruction fetch the two 16-byte
// halves of one pixel's chunk, so an instruction touches 8 whole 128-byte lines of a channel-blocked tensor instead of 16 half lines --
// at the price of 2-way conflicts on the patch reads (16 tiles x 2 channel pairs cannot fall on 16 distinct 16-byte slots without a
// per-column half swizzle).  Measured round 4 (profiles/r04_layout_ab.txt): every one of the seven layers 0..+2.8 % SLOWER, 2.17 -> 2.19 ms
// per 160-slice forward, the headline unchanged: the halved L2 requests do not pay for the conflicts with two waves per SIMD.  Not the default.
#ifndef RCU_WINO_IMAGE8
#define RCU_WINO_IMAGE8 0
#endif
constexpr int WINO_POS_DW = RCU_WINO_IMAGE8 ? 8 : 4;   // dwords from one position of the input image to the next
template <class T>
__device__ __forceinline__ uint32_t wino_slot_geometry(int j, int wave, int lane)
{
    const int f = (j * T::WAVES + wave) * 64 + lane;       // 16-byte unit of the LDS image
#if RCU_WINO_IMAGE8
    const int hh = f & 1, rem = f >> 1;
#else
    const int hh = f / T::HALF_POS, rem = f % T::HALF_POS;
#endif
    const int s = rem / T::SLICE_POS, r2 = rem % T::SLICE_POS;
    const int yy = r2 / T::PITCH, pos = r2 % T::PITCH;
    const int x = pos ^ ((yy >> 1) & 1);                    // pixel column stored at this position
    const bool real = f < T::A_POS && yy < T::TH + 2 && x < T::TW + 2;
    return real ? (uint32_t)(x | (yy << 8) | (s << 16) | (hh << 24)) : 0xFFFFFFFFu;
}

// The tile-independent part of a slot's byte offset -- slice, halo row and pixel column relative to the tile's origin -- with the slot's
// border flags in its low four bits (offsets are multiples of 16): bit 0 the halo row above the tile, 1 the one below, 2 the halo column
// left of it, 3 the one right of it.  Computed once per kernel from the packed geometry; a slot that holds nothing (pitch padding) gets
// an offset beyond any tensor (tensors are < 2^31 bytes: checked by the launchers).
template <class T>
__device__ __forceinline__ uint32_t wino_slot_plan(const ConvArgs& a, uint32_t geo)
{
    const int x = geo & 0xFF, yy = (geo >> 8) & 0xFF, s = (geo >> 16) & 0xFF, hh = (geo >> 24) & 1;
    // sample * (H W C 4) + pixel * pix_bytes + half * 16 (ConvArgs: both layouts); the chunk's offset is added per chunk (wino_chunk_offset)
    const uint32_t off = (uint32_t)s * ((uint32_t)(a.H * a.W) * (uint32_t)a.C1 * 4u) + (uint32_t)((yy - 1) * a.W + (x - 1)) * a.in_pix_bytes + (uint32_t)hh * 16u;
    const uint32_t flags = (yy == 0 ? 1u : 0u) | (yy == T::TH + 1 ? 2u : 0u) | (x == 0 ? 4u : 0u) | (x == T::TW + 1 ? 8u : 0u);
    return geo != 0xFFFFFFFFu ? (off | flags) : 0x80000000u;   // C2 == C1 or 0 (checked by the launcher): one plan for both sources
}

// What a tile adds to the plans of its slots (wave-uniform: scalar arithmetic): the byte offset of its origin, and which of the four
// borders of the image it touches -- whole tiles only (launchers), so a halo row / column lies outside the image exactly when the tile
// touches that border.  Slices beyond the batch need no flag: their offsets lie behind the tensor, where the buffer load returns zeros.
struct WinoTileOffset {
    uint32_t off, border;
};
template <class T, class A>
__device__ __forceinline__ WinoTileOffset wino_tile_offset(const A& a, const WinoTileId& t)
{
    WinoTileOffset o;
    o.off = (uint32_t)t.n0 * ((uint32_t)(a.H * a.W) * (uint32_t)a.C1 * 4u) + (uint32_t)(t.y0 * a.W + t.x0) * a.in_pix_bytes;
    o.border = (t.y0 == 0 ? 1u : 0u) | (t.y0 + T::TH == a.H ? 2u : 0u) | (t.x0 == 0 ? 4u : 0u) | (t.x0 + T::TW == a.W ? 8u : 0u);
    return o;
}

// Byte offset (into src1 and src2 alike) slot `plan` fetches for the tile: five fast instructions per slot and tile (the bounds checks and
// multiplications of a from-scratch computation were ~20, two of them quarter-rate: 1.2k cycles per tile in the F(4x4,3x3) kernel).
__device__ __forceinline__ uint32_t wino_slot_offset(uint32_t plan, const WinoTileOffset& t)
{
    return (plan & t.border) != 0u ? WINO_OOB : ((plan + t.off) & ~15u);
}

// workgroups of a persistent Winograd kernel: one per CU (experiment builds, -DRCU_EXPERIMENTS: RCU_PERSISTENT_GRID overrides it --
// kernels of two streams side by side)
inline unsigned wino_persistent_grid(unsigned items)
{
#ifdef RCU_EXPERIMENTS
    static const unsigned cap = [] {
        const char* const v = getenv("RCU_PERSISTENT_GRID");
        const int n = v ? atoi(v) : 0;
        return n > 0 ? (unsigned)n : 256u;
    }();
#else
    constexpr unsigned cap = 256u;
#endif
    return items < cap ? items : cap;
}

// byte offset of the 8-channel chunk that starts at channel c0 of a source tensor
__device__ __forceinline__ uint32_t wino_chunk_offset(uint32_t in_chunk_bytes, int c0) { return (uint32_t)(c0 >> 3) * in_chunk_bytes; }

// byte offset of (pixel index pix inside a sample of `hw` pixels, channel c -- a multiple of 4) of output sample n
__device__ __forceinline__ uint32_t wino_out_offset(int n, int hw, int CoutP, uint32_t pix, int c, uint32_t pix_bytes, uint32_t chunk_bytes)
{
    return (uint32_t)n * ((uint32_t)hw * (uint32_t)CoutP * 4u) + (uint32_t)(c >> 3) * chunk_bytes + pix * pix_bytes + (uint32_t)(c & 7) * 4u;
}

}  // namespace rcu
