// First conv unit of the network (common/model/unet.py:8-23 with in_channels <= 8: 4 BraTS modalities, 3 ISIC colours):
// conv3x3 (pad 1) + bias -> Dropout2d factor -> folded BatchNorm -> ReLU, 4 or 8 input channels -> 32 or 64 channels.
//
// K = 9 taps x 4 channels is exactly nine K steps of v_mfma_f32_16x16x4_f32, so nothing is padded (the tiled kernel of
// rcu_conv.hip pads the channels to its chunk of 8 and spends twice the matrix work on this layer), and the layer is what
// its 503 MB of output allow: an HBM write stream.
//   * M = output channels (A operand = the weights, held in registers for the whole launch: 9 taps x KS x NB values per
//     lane), N = 16 pixels of one image row (B operand = one float per lane from the input halo tile in LDS);
//     D: lane (pixel n = lane & 15, g = lane >> 4) ends up with four CONSECUTIVE output channels 4g .. 4g+3 of block b
//     for its pixel -> one 16-byte store per block, the four g of a pixel fill 64 contiguous bytes.
//   * workgroup = 4 waves = an 8 x 32 pixel tile (wave w: rows 2w, 2w+1), input halo tile 10 x 34 pixels x 4 (8) channels
//     = 5.4 (10.9) KB of LDS in [channel group][row][column][4] order: the 64 lanes of a B read (16 pixels x 4 channels) hit
//     64 consecutive banks.  Workgroups walk over the tiles (grid = a few per CU), weights and per-channel constants
//     are loaded once.  In a forward pass the tile is filled straight from the caller's NCHW input (ConvArgs::x_nchw): the
//     channels-last copy of the input (pack_input_kernel) is only made for the layer-by-layer API.
// fp32 throughout; the MFMA is an fmaf chain over (tap, channel) -- a different summation order than the tiled kernel's,
// the same as far as the parity tests (|dlogit| <= 1e-7 against the oracle) can tell.
#include "rcu_kernels.h"

namespace rcu {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FIRST_TH = 8, FIRST_TW = 32, FIRST_THREADS = 256;
constexpr int FIRST_HR = FIRST_TH + 2, FIRST_HC = FIRST_TW + 2;   // halo tile
constexpr int FIRST_TILE_FLOATS = 3072;                          // packed [tap][32 couts][8 channels], padded (rcu_api.hip)

template <int KS, int NB>
__global__ __launch_bounds__(FIRST_THREADS) void conv3x3_first_kernel(const ConvArgs a, const int tiles_total)
{
    __shared__ __attribute__((aligned(16))) float tile[KS][FIRST_HR][FIRST_HC][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, g = lane >> 4;

    // A operand: weights of output channel b * 16 + n16, input channel ks * 4 + g, per tap
    float wr[9][KS][NB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int co = b * 16 + n16;
                wr[tap][ks][b] = a.wpack[(size_t)(co >> 5) * FIRST_TILE_FLOATS + (tap * 32 + (co & 31)) * 8 + ks * 4 + g];
            }
    // epilogue constants of the lane's output channels b * 16 + 4 g + j
    f32x4 al[NB], bb[NB], be[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        al[b] = *reinterpret_cast<const f32x4*>(a.alpha + b * 16 + 4 * g);
        bb[b] = *reinterpret_cast<const f32x4*>(a.betab + b * 16 + 4 * g);
        be[b] = *reinterpret_cast<const f32x4*>(a.beta + b * 16 + 4 * g);
    }
    const int tiles_x = a.W / FIRST_TW, tiles_y = a.H / FIRST_TH;
    const float floor_v = a.relu ? 0.f : -__builtin_inff();
    // padded level (ConvArgs::part): H x W is the allocated extent of the output (and of src1), the caller's planes and the stored pixels are Hr x Wr
    const int Hr = a.part ? a.Hr : a.H, Wr = a.part ? a.Wr : a.W;

    // Pass groups (x_nchw set, N = passes * n_images: sample p * n_images + i is image i under the masks of pass p): the convolution
    // of an image tile is the same in every pass -- only the Dropout2d factors behind it differ --, so a tile is staged and multiplied
    // ONCE and written once per pass (tiles_total then counts the tiles of the n_images images).
    const int images = tiles_total / (tiles_x * tiles_y);     // (the launcher's rule: n_images for a pass group, N otherwise)
    const int passes = a.N / images;
    for (int t = blockIdx.x; t < tiles_total; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int y0 = ty * FIRST_TH, x0 = tx * FIRST_TW;
        __syncthreads();   // the previous tile's reads are done
        for (int i = tid; i < KS * FIRST_HR * FIRST_HC; i += FIRST_THREADS) {
            const int ks = i / (FIRST_HR * FIRST_HC), rem = i % (FIRST_HR * FIRST_HC);
            const int r = rem / FIRST_HC, c = rem % FIRST_HC;
            const int gy = y0 + r - 1, gx = x0 + c - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < Hr && gx >= 0 && gx < Wr) {
                if (a.x_nchw != nullptr) {   // the caller's NCHW planes: consecutive threads read consecutive columns of a plane
                    const size_t HW = (size_t)Hr * Wr;
                    const float* const px = a.x_nchw + (size_t)(n % a.n_images) * a.cin_real * HW + (size_t)gy * Wr + gx;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (ks * 4 + k < a.cin_real) v[k] = px[(size_t)(ks * 4 + k) * HW];
                } else {
                    v = *reinterpret_cast<const f32x4*>(a.src1 + ((size_t)(n * a.H + gy) * a.W + gx) * a.C1 + ks * 4);
                }
            }
            *reinterpret_cast<f32x4*>(&tile[ks][r][c][0]) = v;
        }
        __syncthreads();
        f32x4 acc[4][NB];
#pragma unroll
        for (int seg = 0; seg < 4; ++seg) {
            const int r = 2 * wave + (seg >> 1), c0 = 16 * (seg & 1);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[seg][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const float xv = tile[ks][r + tap / 3][c0 + n16 + tap % 3][g];
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[seg][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[tap][ks][b], xv, acc[seg][b], 0, 0, 0);
                }
        }
        for (int pass = 0; pass < passes; ++pass) {
            const int ns = pass * images + n;      // the sample this pass writes
            // Dropout2d factors of (sample ns, the lane's channels); channels beyond the site's width are padding
            f32x4 scale[NB], shift[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                f32x4 mk = {1.f, 1.f, 1.f, 1.f};
                if (a.mask != nullptr) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = b * 16 + 4 * g + j;
                        if (c < a.Cmask) mk[j] = a.mask[(size_t)ns * a.Cmask + c];
                    }
                }
                scale[b] = al[b] * mk;
                shift[b] = bb[b] * mk + be[b];
            }
#pragma unroll
            for (int seg = 0; seg < 4; ++seg) {
                const int r = 2 * wave + (seg >> 1), c0 = 16 * (seg & 1);
                if (y0 + r >= Hr || x0 + c0 + n16 >= Wr) continue;   // beyond the real image (padded level): the zeros there are never written
                // (sample, pixel, channel b * 16 + 4 g): NHWC or blocked [C/8][H][W][8], see ConvArgs
                char* const op = reinterpret_cast<char*>(a.out) + (size_t)ns * a.H * a.W * a.CoutP * 4 +
                                 (size_t)((y0 + r) * a.W + x0 + c0 + n16) * a.out_pix_bytes + (size_t)(g >> 1) * a.out_chunk_bytes + (g & 1) * 16;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    f32x4 v = acc[seg][b] * scale[b] + shift[b];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], floor_v);
                    *reinterpret_cast<f32x4*>(op + (size_t)(2 * b) * a.out_chunk_bytes) = v;
                }
            }
        }
    }
}

template <int KS, int NB>
hipError_t launch_first(const ConvArgs& a, hipStream_t stream)
{
    // pass groups of a forward call: one tile per IMAGE tile, written once per pass (see the kernel)
    const int images = (a.x_nchw != nullptr && a.n_images > 0 && a.N % a.n_images == 0) ? a.n_images : a.N;
    const int tiles = (a.H / FIRST_TH) * (a.W / FIRST_TW) * images;
    const int grid = tiles < 256 * 8 ? tiles : 256 * 8;
    hipLaunchKernelGGL((conv3x3_first_kernel<KS, NB>), dim3(grid), dim3(FIRST_THREADS), 0, stream, a, tiles);
    return hipGetLastError();
}

const ConvConfigInfo kFirstInfo = {1, FIRST_TH, FIRST_TW, 32, 8, 9, "conv3x3_first<T8x32,K36>", 8, 0, 0};

}  // namespace

const ConvConfigInfo& first_config_info() { return kFirstInfo; }

// a.cin_real: the input channels that are not padding (<= a.C1 = 8)
hipError_t launch_conv_first(const ConvArgs& a, hipStream_t stream)
{
    if (a.C1 != 8 || a.C2 != 0 || a.H % FIRST_TH != 0 || a.W % FIRST_TW != 0 || a.pooled != nullptr || a.mask2 != nullptr ||
        (a.CoutP != 32 && a.CoutP != 64))
        return hipErrorInvalidValue;
    const bool wide = a.cin_real > 4;
    if (a.CoutP == 32) return wide ? launch_first<2, 2>(a, stream) : launch_first<1, 2>(a, stream);
    return wide ? launch_first<2, 4>(a, stream) : launch_first<1, 4>(a, stream);
}

}  // namespace rcu
