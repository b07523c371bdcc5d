// Micro-benchmark: how close do two co-resident waves per SIMD get to the fp32 MFMA peak with v_mfma_f32_16x16x4_f32
// (32 cycles) streams, bare and with the Winograd kernel's per-chunk companions (LDS fragment reads, packed adds,
// LDS-DMA pieces, a workgroup barrier)?   hipcc --offload-arch=gfx950 -O3 mfma16_bench.hip -o mfma16_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// flags: 1 = 16 ds_read_b128 + 16 ds_read_b64 per 64 MFMA, 2 = 32 v_pk_add per 64 MFMA, 4 = 6 LDS-DMA pieces per 64 MFMA,
//        8 = barrier per 64 MFMA, 16 = use 32x32x2 (32 MFMA per iteration) instead of 16x16x4
template <int FLAGS>
__global__ __launch_bounds__(512) void bench(const float* src, float* out, int iters, unsigned nbytes, int waves_active)
{
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= waves_active) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    f32x4 acc[32];
    f32x16 big[8];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 16; ++k) big[i][k] = 0.f;
    f32x2 d[16];
    f32x4 b[16];
    for (int i = 0; i < 16; ++i) {
        d[i] = f32x2{lane * 0.001f, 1.f};
        b[i] = f32x4{1.f, lane * 0.002f, 0.5f, 0.25f};
    }
    const float* la = lds + lane * 2;
    const float* lb = lds + 8192 + lane * 4;
    const unsigned voff = (threadIdx.x * 16u + blockIdx.x * 8192u) % (nbytes - 65536u);
    for (int it = 0; it < iters; ++it) {
        if (FLAGS & 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = *(const volatile __attribute__((address_space(3))) f32x2*)(la + i * 128);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (FLAGS & 1) {
                b[2 * g] = *reinterpret_cast<const f32x4*>(lb + (2 * g) * 256);
                b[2 * g + 1] = *reinterpret_cast<const f32x4*>(lb + (2 * g + 1) * 256);
            }
            if ((FLAGS & 4) && g < 6)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((char*)lds + 65536 + (wave * 6 + g) * 1024), 16, voff,
                                                         (unsigned)(((it & 15) * 8 + g) * 4096), 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (FLAGS & 16) {
#pragma unroll
                for (int k = 0; k < 4; ++k) big[(g & 1) * 4 + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[2 * g].x, b[2 * g][k], big[(g & 1) * 4 + k], 0, 0, 0);
            } else {
                const int p0 = 2 * g, p1 = p0 + 1;
                acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p0].x, b[p0].x, acc[p0], 0, 0, 0);
                acc[16 + p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p0].x, b[p0].z, acc[16 + p0], 0, 0, 0);
                acc[p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p1].x, b[p1].x, acc[p1], 0, 0, 0);
                acc[16 + p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p1].x, b[p1].z, acc[16 + p1], 0, 0, 0);
                acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p0].y, b[p0].y, acc[p0], 0, 0, 0);
                acc[16 + p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p0].y, b[p0].w, acc[16 + p0], 0, 0, 0);
                acc[p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p1].y, b[p1].y, acc[p1], 0, 0, 0);
                acc[16 + p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[p1].y, b[p1].w, acc[16 + p1], 0, 0, 0);
            }
            if ((FLAGS & 2) && g < 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {   // 8 packed (or, FLAGS & 32, 16 plain) adds on values the NEXT iteration multiplies
                    if (FLAGS & 32) {
                        float* dd = reinterpret_cast<float*>(d);
                        const int i0 = 2 * (4 * g + k), i1 = 2 * ((4 * g + k + 8) & 15), i2 = 2 * ((4 * g + k + 4) & 15);
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(dd[i0]) : "v"(dd[i1]));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(dd[i0 + 1]) : "v"(dd[i1 + 1]));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(dd[i2]) : "v"(dd[i0]));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(dd[i2 + 1]) : "v"(dd[i0 + 1]));
                    } else {
                        d[4 * g + k] = d[4 * g + k] - d[(4 * g + k + 8) & 15];
                        d[(4 * g + k + 4) & 15] = d[(4 * g + k + 4) & 15] + d[4 * g + k];
                    }
                }
            }
        }
        if (FLAGS & 8) {
            if (FLAGS & 4) __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) s += big[i][0] + big[i][7];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int FLAGS>
static void run(const char* name, const float* src, float* out, unsigned nbytes, int waves)
{
    const int iters = 4000, grid = 256;
    hipFuncSetAttribute((const void*)bench<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench<FLAGS>, dim3(grid), dim3(512), 128 * 1024, 0, src, out, iters, nbytes, waves);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // 64 MFMA 16x16x4 (2048 flop each per ... 16*16*4*2) or 32 MFMA 32x32x2 per wave-iteration
    const double flop = (double)grid * waves * iters * 64.0 * (16 * 16 * 4 * 2);
    printf("%-58s %d waves/CU: %7.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", name, waves, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
}

int main()
{
    const unsigned nbytes = 64u << 20;
    float *src, *out;
    hipMalloc(&src, nbytes);
    hipMemset(src, 0, nbytes);
    hipMalloc(&out, 4096);
    for (int waves : {8}) {
        run<0>("16x16x4 bare", src, out, nbytes, waves);
        run<16>("32x32x2 bare", src, out, nbytes, waves);
        run<1>("16x16x4 + LDS fragment reads", src, out, nbytes, waves);
        run<3>("16x16x4 + reads + 32 pk_add", src, out, nbytes, waves);
        run<35>("16x16x4 + reads + 64 plain v_add/v_sub", src, out, nbytes, waves);
        run<47>("16x16x4 + reads + 64 plain adds + DMA + barrier", src, out, nbytes, waves);
        run<7>("16x16x4 + reads + pk_add + 6 LDS-DMA", src, out, nbytes, waves);
        run<15>("16x16x4 + reads + pk_add + DMA + vmcnt(0) + barrier", src, out, nbytes, waves);
        run<11>("16x16x4 + reads + pk_add + barrier", src, out, nbytes, waves);
        run<17>("32x32x2 + reads", src, out, nbytes, waves);
    }
    return 0;
}
