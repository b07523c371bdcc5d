// UpConv's "nearest x2 -> conv3x3 (+bias)" (common/model/unet.py:105, helpers.py:15) in its sub-pixel form -- output
// pixel (2y+a, 2x+b) is a 2x2-tap convolution of the LOW-resolution grid with tap weights pre-summed per parity class
// (a, b), see rcu_conv.hip -- with every class evaluated as Winograd F(2x2, 2x2):
//
//     Y^{ab} = A^T [ sum_c (G w^{ab}_c G^T) .* (B^T d^{ab}_c B) ] A,    d^{ab} = 3x3 patch, 9 multiplications per 4 outputs
//     B^T = [[1,-1,0],[0,1,0],[0,-1,1]]   G = [[1,0],[1,1],[0,1]]   A^T = [[1,1,0],[0,1,1]]
//
// i.e. 9 instead of 16 (sub-pixel) or 36 (up-sample, then 3x3) multiplications per (cin, cout) and 2x2 low-resolution
// tile: the up-convolutions execute 1/4 of their canonical FLOPs on the MFMA pipe.
//
// The four classes of a tile read sub-patches of ONE 4x4 low-resolution patch P (the F(2x2,3x3) patch of rcu_wino.hip):
// class (a, b) takes rows a..a+2 and columns b..b+2.  A work item fixes a; its waves compute both b at once:
//   rows     rho = (e0 - e1, e1, e2 - e1) of (e0, e1, e2) = P[a .. a+2]
//   columns  gamma = (P.0 - P.1, P.1, P.2 - P.1, P.2, P.3 - P.2): b = 0 multiplies (gamma0, gamma1, gamma2),
//            b = 1 multiplies (-gamma2, gamma3, gamma4) -- the sign is folded into the packed weights
// = 15 transformed values per (tile, channel) feeding 18 positions p = 6 i + 3 b + j: 144 accumulator registers per wave
// (16 tiles x 32 couts x 18).  Staging (LDS-DMA, swizzled lane-linear images, hardware zero padding), wave / lane
// mapping and the streaming pipeline are those of rcu_wino.hip; see there.
#include "rcu_wino_common.h"

#include <cstdlib>

namespace rcu {

// Output transform + epilogue.  acc[blk][p][r]; the lane's tile r of class (pa, b) yields the output pixels
// (2 (ly + u) + pa, 2 (lx + v) + b), u, v in {0, 1}, of the up-sampled grid.
// PART: a padded level on either side (ConvArgs::part) -- the tiles walk the ALLOCATED low-resolution grid, output pixels at or beyond the real
// up-sampled image (Hr, Wr) go out of range, where the buffer resource drops the write, and the output tensor has extents of its own.
template <class T, bool PART = false>
__device__ __forceinline__ void wino_up_epilogue(const ConvArgs& a, const f32x4 (&acc)[2][18], const WinoEpi& ep, int ntile, int pa,
                                                 int n0, int y0, int x0, int wm, int wn, int lane)
{
    const int n16 = lane & 15, g = lane >> 4;
    const int co = ntile * T::BN + wn * 32 + 2 * n16;
    if (co >= a.CoutP) return;
    int bs, by, bx;
    T::block_origin(wm, bs, by, bx);
    const int n = n0 + bs + (T::SW == 2 ? g & 1 : 0);
    if (n >= a.N) return;
    const int lyb = y0 + by + 2 * (g >> 1);                       // low-res row of the lane's tiles
    const int lxb = x0 + bx + (T::SW == 2 ? 0 : 8 * (g & 1));     // low-res column of the lane's first tile
    const int OH = PART ? a.out_H : 2 * a.H, OW = PART ? a.out_W : 2 * a.W;
    const int odd = n16 & 1;
    // PART: the lane stores output pixels (2 lyb + pa + 2 u, 2 (lxb + odd) + 4 r + b): real iff 2 u < ylim && 4 r + b < xlim
    [[maybe_unused]] const int ylim = PART ? a.Hr - (2 * lyb + pa) : 0, xlim = PART ? a.Wr - 2 * (lxb + odd) : 0;
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
    // whole tiles only (checked by the launcher), output below 2 GB: buffer stores, one per-lane byte offset per tile, scalar steps
    const uint32_t px_bytes = a.out_pix_bytes, row_bytes = (uint32_t)OW * px_bytes;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (uint32_t)(a.N * OH * OW) * (uint32_t)a.CoutP * 4u, 0x00020000);
    const uint32_t vo = wino_out_offset(n, OH * OW, a.CoutP, (uint32_t)((2 * lyb + pa) * OW + 2 * (lxb + odd)), co - 2 * odd, a.out_pix_bytes,
                                        a.out_chunk_bytes);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x2 y[2][2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                float s[2][3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float m0 = acc[blk][0 * 6 + b * 3 + j][r], m1 = acc[blk][1 * 6 + b * 3 + j][r], m2 = acc[blk][2 * 6 + b * 3 + j][r];
                    s[0][j] = m0 + m1;
                    s[1][j] = m1 + m2;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float t0 = (s[u][0] + s[u][1]) * ep.scale[blk] + ep.shift[blk];
                    const float t1 = (s[u][1] + s[u][2]) * ep.scale[blk] + ep.shift[blk];
                    y[u][0][blk] = fmaxf(t0, relu_floor);
                    y[u][1][blk] = fmaxf(t1, relu_floor);
                }
            }
            // neighbouring lanes trade one pixel column each: the even lane stores four couts of column v = 0, the odd lane
            // of v = 1 (one 16-byte store per lane instead of two 8-byte ones, see rcu_wino.hip)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x2 send = odd ? y[u][0] : y[u][1];
                f32x2 recv;
                recv.x = wino_swap_adjacent(send.x);
                recv.y = wino_swap_adjacent(send.y);
                const f32x4 o = odd ? f32x4{recv.x, recv.y, y[u][1].x, y[u][1].y} : f32x4{y[u][0].x, y[u][0].y, recv.x, recv.y};
                wino_store16(o, ro, (PART && !(2 * u < ylim && 4 * r + b < xlim)) ? WINO_OOB : vo, (4 * r + b) * px_bytes + u * 2 * row_bytes);
            }
        }
    }
}

template <class T, bool PART = false>
__global__ __launch_bounds__(512, 1) void upconv_wino_stream(const ConvArgs a, const int total_items)
{
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource types and LDS-DMA builtins exist in the device pass only
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    constexpr int KC = T::KC;
    static_assert(T::NPOS == 18, "two F(2x2,2x2) classes per wave");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % T::WN;
    const int wm = wave / T::WN;
    const int m16 = lane & 15, kq = lane >> 4;
    const int nchunks = a.C1 / KC;   // even, >= 4 (checked by the launcher); single source
    const uint32_t wchunk_bytes = (uint32_t)a.NTW_total * T::W_DW * 4u;
    const uint32_t in_chunk_bytes = a.in_chunk_bytes;

    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1), 0, a.src1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, a.wpack_bytes, 0x00020000);

    // fragment addresses (dword offsets inside a buffer) of patch rows 0 and 2; rows 1 and 3 follow one pitch later
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
            aA[i2] = rowbase + 4 * (x0l + swz);
            aB[i2] = rowbase + 4 * (x0l - swz);
#endif
        }
    }
    const int b_addr = T::A_DW + (kq * T::BN + wn * 32 + 2 * m16) * 2;
    const uint32_t w_voff = (uint32_t)(lane * 16);

    int item = wino_xcd_virtual_block(a.NTW_total < 4 ? 4 : a.NTW_total);
    bool has_next = item + (int)gridDim.x < total_items;
    WinoTileId tile = wino_tile_id<T>(a, item), ntile = tile;
    uint32_t dp[T::NA], geo[T::NA];
    int dp_wtile = tile.wtile;
#pragma unroll
    for (int j = 0; j < T::NA; ++j) {
        geo[j] = wino_slot_plan<T>(a, wino_slot_geometry<T>(j, wave, lane));   // the slot's plan: tile-independent offset | border flags
        asm volatile("" : "+v"(geo[j]));
        dp[j] = wino_slot_offset(geo[j], wino_tile_offset<T>(a, tile));
    }

    struct DmaJob {
        bool active;
        uint32_t cb, wso, lb;
    };
    auto dma_job = [&](int wtile, int kc, int buf, bool active) {
        DmaJob j;
        j.active = active;
        j.cb = wino_chunk_offset(in_chunk_bytes, kc * KC);
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
            if (job.active && ((j + 1) * T::WAVES <= T::A_PIECES || j * T::WAVES + wave < T::A_PIECES))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(uintptr_t)(lds_base + job.lb + (j * T::WAVES + wave) * 1024), 16, dp[j],
                                                         job.cb, 0, RCU_DMA_IN_AUX);
        }
    };

    {
        const DmaJob job = dma_job(dp_wtile, 0, 0, true);
        wino_static_for<0, T::NW + T::NA>([&](auto i_c) { dma_piece(job, i_c); });
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();

    f32x4 acc[2][18];

    auto chunk = [&](auto buf_c, auto first_c, int kc) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        const float* const Ab = smem + BUF * T::BUF_DW;
        const int pa = tile.wtile / a.NT;   // row parity class of this work item
        const bool more = kc + 1 < nchunks;
        if (!more && has_next) {   // last chunk of the tile: from here on the DMA works on the workgroup's next tile
            const WinoTileConsts ca = wino_tile_consts(wino_cold_args());   // tile counts and image extents are not kept in SGPRs either: one batch of scalar loads
            ntile = wino_tile_id<T>(ca, item + (int)gridDim.x);
            dp_wtile = ntile.wtile;
            const WinoTileOffset nto = wino_tile_offset<T>(ca, ntile);
#pragma unroll
            for (int j = 0; j < T::NA; ++j) dp[j] = wino_slot_offset(geo[j], nto);
        }
        __builtin_amdgcn_sched_barrier(0);
        const DmaJob job = dma_job(dp_wtile, more ? kc + 1 : 0, BUF ^ 1, more || has_next);
        // rows pa .. pa+2 of the lane's 4x4 patch, channel pair (2kq, 2kq+1): (e0, e1) now, e2 behind the first MFMA group
        f32x2 e0[4], e1[4], e2[4];
        auto read_row = [&](f32x2 (&e)[4], auto row_c) {
            constexpr int ROW = decltype(row_c)::value;   // patch row 0..3
#pragma unroll
            for (int j = 0; j < 4; ++j)
                e[j] = *(const volatile __attribute__((address_space(3))) f32x2*)(Ab + ((j & 1) ? aB[ROW >> 1] : aA[ROW >> 1]) +
                                                                                  (ROW & 1) * (T::PITCH * WINO_POS_DW) + WINO_POS_DW * j);
        };
        if (pa == 0) {
            read_row(e0, std::integral_constant<int, 0>{});
            read_row(e1, std::integral_constant<int, 1>{});
        } else {
            read_row(e0, std::integral_constant<int, 1>{});
            read_row(e1, std::integral_constant<int, 2>{});
        }
        constexpr int AHEAD = 2;   // the weights of positions p + 2, p + 3 are read while p, p + 1 multiply
        f32x4 bv[18];
#pragma unroll
        for (int p = 0; p < AHEAD; ++p) bv[p] = *reinterpret_cast<const f32x4*>(Ab + b_addr + p * (8 * T::BN));
        // V[5 i + c] = rho_i (x) gamma_c, one position row i at a time, just ahead of the groups that multiply it
        f32x2 V[15];
        auto col_transform = [&](int i, const f32x2 (&r)[4]) {
            V[5 * i + 0] = r[0] - r[1];
            V[5 * i + 1] = r[1];
            V[5 * i + 2] = r[2] - r[1];
            V[5 * i + 3] = r[2];
            V[5 * i + 4] = r[3] - r[2];
        };
        {
            f32x2 r0[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r0[j] = e0[j] - e1[j];
            col_transform(0, r0);
        }
        // 18 positions p = 6 i + 3 b + j x (2 channel steps x 2 cout blocks)
        wino_static_for<0, 9>([&](auto pp_c) {
            constexpr int G = decltype(pp_c)::value;
            constexpr int p0 = 2 * G, p1 = p0 + 1;
            constexpr int v0 = 5 * (p0 / 6) + 2 * ((p0 % 6) / 3) + p0 % 3, v1 = 5 * (p1 / 6) + 2 * ((p1 % 6) / 3) + p1 % 3;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (p0 + AHEAD < 18) {
                bv[p0 + AHEAD] = *reinterpret_cast<const f32x4*>(Ab + b_addr + (p0 + AHEAD) * (8 * T::BN));
                bv[p1 + AHEAD] = *reinterpret_cast<const f32x4*>(Ab + b_addr + (p1 + AHEAD) * (8 * T::BN));
            }
            if constexpr (G == 0) {
                if (pa == 0)
                    read_row(e2, std::integral_constant<int, 2>{});
                else
                    read_row(e2, std::integral_constant<int, 3>{});
            }
            // two LDS-DMA pieces per group from the first group on, so that the last piece has 3/4 of the chunk to land (one per
            // group: -1.5 %; three per group: -1.4 %)
            dma_piece(job, std::integral_constant<int, 2 * G>{});
            dma_piece(job, std::integral_constant<int, 2 * G + 1>{});
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            acc[0][p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[v0].x, bv[p0].x, FIRST ? z : acc[0][p0], 0, 0, 0);
            acc[1][p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[v0].x, bv[p0].z, FIRST ? z : acc[1][p0], 0, 0, 0);
            acc[0][p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[v1].x, bv[p1].x, FIRST ? z : acc[0][p1], 0, 0, 0);
            acc[1][p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[v1].x, bv[p1].z, FIRST ? z : acc[1][p1], 0, 0, 0);
            acc[0][p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[v0].y, bv[p0].y, acc[0][p0], 0, 0, 0);
            acc[1][p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[v0].y, bv[p0].w, acc[1][p0], 0, 0, 0);
            acc[0][p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[v1].y, bv[p1].y, acc[0][p1], 0, 0, 0);
            acc[1][p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[v1].y, bv[p1].w, acc[1][p1], 0, 0, 0);
            if constexpr (G == 1) col_transform(1, e1);            // multiplied from group 3 on
            if constexpr (G == 4) {                                // multiplied from group 6 on
                f32x2 r2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) r2[j] = e2[j] - e1[j];
                col_transform(2, r2);
            }
        });
        wino_static_for<18, T::NW + T::NA>([&](auto i_c) { dma_piece(job, i_c); });
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
        {
            const ConvArgs& ca = wino_cold_args();
            wino_up_epilogue<T, PART>(ca, acc, wino_epilogue_fold(wino_epilogue_load<T>(ca, tile.wtile % a.NT, tile.n0, wm, wn, lane)), tile.wtile % a.NT,
                                tile.wtile / a.NT, tile.n0, tile.y0, tile.x0, wm, wn, lane);
        }
        if (!has_next) break;
        item += (int)gridDim.x;
        tile = ntile;
        has_next = item + (int)gridDim.x < total_items;
    }
#endif
}

using UCfg0 = WinoTile<1, 16, 16, 64, 4, 2, 1, 18>;
using UCfg1 = WinoTile<1, 16, 32, 32, 8, 1, 1, 18>;
using UCfg2 = WinoTile<2, 8, 16, 64, 4, 2, 1, 18>;
using UCfg3 = WinoTile<8, 4, 8, 64, 4, 2, 2, 18>;

static const ConvConfigInfo kWinoUpInfo[4] = {
    {UCfg0::TS, UCfg0::TH, UCfg0::TW, UCfg0::BN, 8, 18, "upconv_winograd<T16x16,N64,K8>", 8, 0, 2},
    {UCfg1::TS, UCfg1::TH, UCfg1::TW, UCfg1::BN, 8, 18, "upconv_winograd<T16x32,N32,K8>", 8, 0, 2},
    {UCfg2::TS, UCfg2::TH, UCfg2::TW, UCfg2::BN, 8, 18, "upconv_winograd<S2T8x16,N64,K8>", 8, 0, 2},
    {UCfg3::TS, UCfg3::TH, UCfg3::TW, UCfg3::BN, 8, 18, "upconv_winograd<S8T4x8,N64,K8>", 8, 0, 2},
};

const ConvConfigInfo& wino_up_config_info(int cfg) { return kWinoUpInfo[cfg - CONV_CFG_UPW_T16x16_N64]; }

template <class T>
static hipError_t launch_wino_up_cfg(const ConvArgs& a, hipStream_t stream)
{
    const int nchunks = a.C1 / T::KC;
    if (nchunks < 4 || (nchunks & 1) != 0 || a.C2 != 0 || a.NTW_total != 2 * a.NT || a.src1_bytes == 0 || a.wpack_bytes == 0 ||
        a.H % T::TH != 0 || a.W % T::TW != 0 ||
        (!a.part && (size_t)a.N * a.H * a.W * a.CoutP * 16 >= ((size_t)1 << 31)))
        return hipErrorInvalidValue;
    // padded level on either side: the real up-sampled image lies inside twice the tile grid, the output tensor holds it
    if (a.part && (a.Hr < 1 || a.Wr < 1 || a.Hr > 2 * a.H || a.Wr > 2 * a.W || a.out_H < a.Hr || a.out_W < a.Wr ||
                   (size_t)a.N * a.out_H * a.out_W * a.CoutP * 4 >= ((size_t)1 << 31)))
        return hipErrorInvalidValue;
    const unsigned items = (unsigned)a.NTW_total * a.tiles_x * a.tiles_y * a.slice_groups;
    const unsigned grid = wino_persistent_grid(items);
    auto launch = [&](auto part_c) {
        constexpr bool PART = decltype(part_c)::value;
        hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&upconv_wino_stream<T, PART>), T::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((upconv_wino_stream<T, PART>), dim3(grid), dim3(T::THREADS), T::LDS_BYTES, stream, a, (int)items);
        return hipGetLastError();
    };
    return a.part ? launch(std::true_type{}) : launch(std::false_type{});
}

hipError_t launch_upconv_wino(int cfg, const ConvArgs& a, hipStream_t stream)
{
    switch (cfg) {
        case CONV_CFG_UPW_T16x16_N64: return launch_wino_up_cfg<UCfg0>(a, stream);
        case CONV_CFG_UPW_T16x32_N32: return launch_wino_up_cfg<UCfg1>(a, stream);
        case CONV_CFG_UPW_S2T8x16_N64: return launch_wino_up_cfg<UCfg2>(a, stream);
        case CONV_CFG_UPW_S8T4x8_N64: return launch_wino_up_cfg<UCfg3>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rcu
