// Micro-benchmark: what one LDS-DMA instruction (buffer_load_dwordx4 ... lds, 64 lanes x 16 B = 1 KB) costs a wave that is alone on
// its SIMD and otherwise issues v_mfma_f32_16x16x4_f32 back to back (the regime of csrc/rcu_wino4.hip), as a function of the address
// pattern of its 64 lanes -- all hits in L2 (every workgroup re-reads the same 256 KB):
//   0  contiguous: lane l reads bytes 16 l .. 16 l + 15                    (8 whole 128-byte lines per instruction)
//   1  lane pairs on 32 contiguous bytes, pairs 256 bytes apart             (rcu_wino4.hip at 64 channels: 32 lines per instruction)
//   2  every lane in its own 128-byte line                                  (rcu_wino.hip at 32 channels: 64 lines per instruction)
//   3  lane quads on 64 contiguous bytes, quads 256 bytes apart             (16 lines)
// hipcc --offload-arch=gfx950 -O3 dma_issue_bench.hip -o dma_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int PIECES>   // LDS-DMA instructions per 8 MFMAs
__global__ __launch_bounds__(256, 1) void bench(const float* src, float* out, int iters, unsigned nbytes, int pattern)
{
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = lane * 0.001f, b = 1.f + lane * 0.002f;
    unsigned voff;
    if (pattern == 0) voff = lane * 16u;
    else if (pattern == 1) voff = (lane >> 1) * 256u + (lane & 1) * 16u;
    else if (pattern == 2) voff = lane * 128u;
    else voff = (lane >> 2) * 256u + (lane & 3) * 16u;
    voff += wave * 16384u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((char*)lds + (wave * 8 + g) * 1024), 16, voff, (unsigned)((it & 7) * 8 + g) * 32u, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
        }
        if ((it & 7) == 7) __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int PIECES>
static double run(const float* src, float* out, unsigned nbytes, int pattern)
{
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench<PIECES>, dim3(grid), dim3(256), 64 * 1024, 0, src, out, iters, nbytes, pattern);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e-3 / iters * 2.4e9;   // nominal cycles per iteration (8 MFMAs)
}

int main()
{
    const unsigned nbytes = 4u << 20;
    float *src, *out;
    (void)hipMalloc(&src, nbytes);
    (void)hipMemset(src, 0, nbytes);
    (void)hipMalloc(&out, 4096);
    const double bare = run<0>(src, out, nbytes, 0);
    printf("bare 8 MFMA: %.1f nominal cycles per iteration (%.1f per MFMA)\n", bare, bare / 8);
    const char* names[4] = {"contiguous (8 lines)", "lane pairs, 32 B (32 lines)", "one line per lane (64 lines)", "lane quads, 64 B (16 lines)"};
    for (int p = 0; p < 4; ++p) {
        const double c1 = run<1>(src, out, nbytes, p), c2 = run<2>(src, out, nbytes, p), c4 = run<4>(src, out, nbytes, p);
        printf("%-32s +%.1f cycles per LDS-DMA instruction at 1 per 8 MFMAs, +%.1f at 2, +%.1f at 4\n", names[p], c1 - bare, (c2 - bare) / 2, (c4 - bare) / 4);
    }
    return 0;
}
