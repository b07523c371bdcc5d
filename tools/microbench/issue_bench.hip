// Micro-benchmark: what does a co-resident fp32-MFMA wave do to the VMEM / LDS instruction issue of another
// wave on the same SIMD (and vice versa)?   hipcc --offload-arch=gfx950 -O3 issue_bench.hip -o issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// role of a wave: 0 = idle, 1 = MFMA stream, 2 = buffer_load stream, 3 = ds_write stream, 4 = MFMA + 1 load per 8 MFMA
__global__ __launch_bounds__(512) void bench(const float* src, float* out, unsigned long long* cyc, int role_lo, int role_hi,
                                             int iters, unsigned nbytes)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wave < 4 ? role_lo : role_hi;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    f32x16 a0 = {}, a1 = {};
    f32x4 acc = {};
    float x = lane * 0.001f, y = 1.0f;
    const unsigned voff = (threadIdx.x * 16u + blockIdx.x * 8192u) % (nbytes - 65536u);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            }
        }
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)(k * 4096 + (i & 7) * 32768), 0));
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
    } else if (role == 3) {
        f32x4 v = {x, y, x, y};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) reinterpret_cast<f32x4*>(lds)[threadIdx.x + k * 512] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (role == 4) {
        for (int i = 0; i < iters; ++i) {
            f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)((i & 63) * 4096), 0));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            }
            acc += v;
        }
    } else if (role == 5) {   // MFMA + 1 ds_write per 8 MFMA
        f32x4 v = {x, y, x, y};
        for (int i = 0; i < iters; ++i) {
            reinterpret_cast<f32x4*>(lds)[threadIdx.x + (i & 7) * 512] = v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main()
{
    const unsigned nbytes = 64u << 20;
    float *src, *out;
    unsigned long long* cyc;
    hipMalloc(&src, nbytes);
    hipMemset(src, 0, nbytes);
    hipMalloc(&out, 4096);
    const int grid = 256;
    hipMalloc(&cyc, grid * 8 * sizeof(unsigned long long));
    hipFuncSetAttribute((const void*)bench, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    struct Case { const char* name; int lo, hi; } cases[] = {
        {"MFMA alone (waves 0-3), 4-7 idle", 1, 0},      {"MFMA both halves (2 waves/SIMD)", 1, 1},
        {"loads alone (waves 4-7)", 0, 2},               {"MFMA + co-resident load stream", 1, 2},
        {"ds_write alone (waves 4-7)", 0, 3},            {"MFMA + co-resident ds_write stream", 1, 3},
        {"MFMA with 1 load / 8 MFMA, alone", 4, 0},      {"MFMA with 1 load / 8 MFMA, both halves", 4, 4},
        {"MFMA with 1 ds_write / 8 MFMA, both halves", 5, 5}, {"loads both halves", 2, 2},
    };
    const int iters = 2000;
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(bench, dim3(grid), dim3(512), 100 * 1024, 0, src, out, cyc, c.lo, c.hi, iters, nbytes);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(grid * 8);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double lo = 0, hi = 0;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
        lo /= grid * 4;
        hi /= grid * 4;
        printf("%-46s  waves0-3: %8.0f ticks/iter (8 MFMA or 8 ops)   waves4-7: %8.0f ticks/iter\n", c.name, lo / iters, hi / iters);
    }
    return 0;
}
