// Micro-benchmark: would a two-waves-per-SIMD form of the F(4x4,3x3) conv unit beat the shipped one-wave form (csrc/rcu_wino4.hip,
// 0.50-0.56 of the fp32 matrix rate)?  A wave then owns 16 tiles x 16 couts x 36 positions = 144 accumulator registers (half the
// couts of the shipped kernel), so the input transform -- 144 packed operations per 8-channel chunk -- is done twice per 32 couts.
// The loop below is one chunk of such a kernel with everything that costs issue slots: 36 raw patch reads (ds_read_b64), the
// transform in F(4x4,3x3) order (position rows 0, 5, 1, 2, 3, 4), 36 weight reads (ds_read_b64), 72 v_mfma_f32_16x16x4_f32,
// 10 LDS-DMA pieces, vmcnt(0) + barrier.    hipcc --offload-arch=gfx950 -O3 wino4_2wave_bench.hip -o wino4_2wave_bench
// flags: 1 LDS reads, 2 transform, 4 LDS-DMA, 8 barrier
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void row0(f32x2 (&t)[6], const f32x2 (&d)[36], int j0)   // y0 = 4 x0 - 5 x2 + x4 over the 6 columns
{
#pragma unroll
    for (int j = 0; j < 6; ++j) t[j] = 4.f * d[j0 + j] - 5.f * d[12 + j] + d[24 + j];
}

template <int FLAGS>
__global__ __launch_bounds__(512) void bench(const float* src, float* out, int iters, unsigned nbytes)
{
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    f32x4 acc[36];
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 d[36], w[12];   // weights: a ring of two groups, read one group ahead
    for (int i = 0; i < 36; ++i) d[i] = f32x2{lane * 0.001f + i, 1.f};
    for (int i = 0; i < 12; ++i) w[i] = f32x2{1.f, lane * 0.002f};
    const float* la = lds + lane * 2;
    const float* lb = lds + 8192 + lane * 2;
    const unsigned voff = (threadIdx.x * 16u + blockIdx.x * 8192u) % (nbytes - 65536u);
    for (int it = 0; it < iters; ++it) {
        if (FLAGS & 1) {
#pragma unroll
            for (int i = 0; i < 36; ++i) d[i] = *(const volatile __attribute__((address_space(3))) f32x2*)(la + i * 128);
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = *(const volatile __attribute__((address_space(3))) f32x2*)(lb + i * 128);
        }
        // v[r][j]: row transform (over patch rows) for position row r, then the column transform in place
        f32x2 v[6][6];
        auto cols = [&](int r) {
            f32x2(&x)[6] = v[r];
            const f32x2 y0 = 4.f * x[0] - 5.f * x[2] + x[4];
            const f32x2 a = x[4] - 4.f * x[2], b = x[3] - 4.f * x[1];
            const f32x2 c = x[4] - x[2], e = x[3] - x[1];
            const f32x2 y5 = 4.f * x[1] - 5.f * x[3] + x[5];
            x[0] = y0, x[1] = a + b, x[2] = a - b, x[3] = c + 2.f * e, x[4] = c - 2.f * e, x[5] = y5;
        };
        if (FLAGS & 2) {
#pragma unroll
            for (int j = 0; j < 6; ++j) v[0][j] = 4.f * d[j] - 5.f * d[12 + j] + d[24 + j];
            cols(0);
        } else {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j) v[r][j] = d[6 * r + j];
        }
        // position rows in the order 0, 5, 1, 2, 3, 4; six groups of 12 MFMAs
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int r = g == 0 ? 0 : g == 1 ? 5 : g - 1;
            __builtin_amdgcn_sched_barrier(0);
            if ((FLAGS & 1) && g < 5) {
#pragma unroll
                for (int i = 0; i < 6; ++i) w[6 * ((g + 1) & 1) + i] = *(const volatile __attribute__((address_space(3))) f32x2*)(lb + (6 * (g + 1) + i) * 128);
            }
            if ((FLAGS & 4) && g < 5) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((char*)lds + 65536 + (wave * 10 + 2 * g) * 1024), 16, voff,
                                                         (unsigned)(((it & 15) * 16 + 2 * g) * 4096), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((char*)lds + 65536 + (wave * 10 + 2 * g + 1) * 1024), 16, voff,
                                                         (unsigned)(((it & 15) * 16 + 2 * g + 1) * 4096), 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[6 * r + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[r][j].x, w[6 * (g & 1) + j].x, acc[6 * r + j], 0, 0, 0);
                acc[6 * r + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[r][j].y, w[6 * (g & 1) + j].y, acc[6 * r + j], 0, 0, 0);
            }
            // transform work for the coming position rows, in the shadow of this group's MFMAs
            if (FLAGS & 2) {
                if (g == 0) {   // row 5, and the shared terms of rows 1..4
#pragma unroll
                    for (int j = 0; j < 6; ++j) v[5][j] = 4.f * d[6 + j] - 5.f * d[18 + j] + d[30 + j];
                    cols(5);
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const f32x2 a = d[24 + j] - 4.f * d[12 + j], b = d[18 + j] - 4.f * d[6 + j];
                        v[1][j] = a + b, v[2][j] = a - b;
                    }
                }
                if (g == 1) {
                    cols(1);
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const f32x2 c = d[24 + j] - d[12 + j], e = d[18 + j] - d[6 + j];
                        v[3][j] = c + 2.f * e, v[4][j] = c - 2.f * e;
                    }
                }
                if (g == 2) cols(2);
                if (g == 3) cols(3);
                if (g == 4) cols(4);
            }
        }
        if (FLAGS & 8) {
            if (FLAGS & 4) __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 36; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int FLAGS>
static void run(const char* name, const float* src, float* out, unsigned nbytes)
{
    const int iters = 4000, grid = 256;
    hipFuncSetAttribute((const void*)bench<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench<FLAGS>, dim3(grid), dim3(512), 160 * 1024, 0, src, out, iters, nbytes);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = (double)grid * 8 * iters * 72.0 * (16 * 16 * 4 * 2);
    printf("%-64s %7.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
}

int main()
{
    const unsigned nbytes = 64u << 20;
    float *src, *out;
    hipMalloc(&src, nbytes);
    hipMemset(src, 0, nbytes);
    hipMalloc(&out, 4096);
    run<0>("72 MFMA bare", src, out, nbytes);
    run<1>("+ 36 patch + 36 weight reads", src, out, nbytes);
    run<3>("+ reads + transform (144 packed ops)", src, out, nbytes);
    run<7>("+ reads + transform + 10 LDS-DMA", src, out, nbytes);
    run<11>("+ reads + transform + barrier", src, out, nbytes);
    run<15>("+ reads + transform + DMA + vmcnt(0) + barrier  (one chunk)", src, out, nbytes);
    return 0;
}
