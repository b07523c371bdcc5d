// Micro-benchmark: the HBM WRITE rate a streaming kernel reaches on this box, by store width, run length per wave instruction and
// cache policy -- the ceiling the output side of the 192x128-level conv units (1 GB per launch) is to be judged against.
//   hipcc --offload-arch=gfx950 -O3 write_bw_bench.hip -o write_bw_bench && ./write_bw_bench
#include <hip/hip_runtime.h>
#include <cstdio>

// RUN: consecutive lanes that write consecutive 16-byte units (64: 1 KB per wave instruction; 4: the 64-byte runs of the channel-blocked
// layout's stores, the runs of a wave 32 KB apart)
template <int RUN, int AUX>
__global__ __launch_bounds__(256) void write_kernel(float4* __restrict__ dst, size_t n4, float v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const float4 val = {v, v + 1.f, v + 2.f, v + 3.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        size_t j = i;
        if (RUN < 64) {   // permute inside a 64 KB block: lane group g of the wave goes to run g of a different 1 KB row
            const size_t blk = i / 4096, r = i % 4096, lane = r % 64, row = r / 64;
            const size_t grp = lane / RUN, in = lane % RUN;
            j = blk * 4096 + ((row + grp * (64 / (64 / RUN))) % 64) * 64 + grp * RUN + in;
        }
        if (AUX == 0) dst[j] = val;
        else {
            typedef float v4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(v4{val.x, val.y, val.z, val.w}, reinterpret_cast<v4*>(&dst[j]));
        }
    }
}

// U stores per lane and loop trip: SEQ = the lane's own consecutive 16-byte units (64 B per lane and trip for U = 4), else U wave-contiguous 1 KB rows
template <int U, bool SEQ>
__global__ __launch_bounds__(256) void write_unrolled(float4* __restrict__ dst, size_t n4, float v)
{
    const float4 val = {v, v + 1.f, v + 2.f, v + 3.f};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U; base < n4; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = SEQ ? base + (size_t)threadIdx.x * U + u : base + (size_t)u * 256 + threadIdx.x;
            if (j < n4) dst[j] = val;
        }
    }
}

// The Winograd epilogue's pattern on the channel-blocked layout: a wave instruction writes 64 / RUN runs of RUN 16-byte units, the runs FAR
// units apart (other channel planes / tiles), and the wave's next instruction writes the units right behind each run (RUN = 4: the second
// 64-byte half of every 128-byte line).  Covers a [FAR-unit columns] x [64 / RUN rows] panel per wave and FAR / RUN trips.
template <int RUN>
__global__ __launch_bounds__(256) void write_panels(float4* __restrict__ dst, size_t n4, float v)
{
    constexpr int FAR = 2048, RUNS = 64 / RUN;                   // 32 KB between the runs of an instruction
    const float4 val = {v, v + 1.f, v + 2.f, v + 3.f};
    const size_t panel = (size_t)FAR * RUNS;                      // units per wave panel
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64, waves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x % 64, run = lane / RUN, in = lane % RUN;
    for (size_t p = wave; (p + 1) * panel <= n4; p += waves) {
        float4* const base = dst + p * panel + (size_t)run * FAR + in;
#pragma unroll 4
        for (int t = 0; t < FAR / RUN; ++t) base[(size_t)t * RUN] = val;
    }
}

int main()
{
    const size_t n4 = (size_t)64 << 20;   // 64 Mi float4 = 1 GiB per launch
    float4* p;
    hipMalloc(&p, n4 * sizeof(float4));
    hipMemset(p, 0, n4 * sizeof(float4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timed = [&](auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        return ms / 5;
    };
    const double gb = (double)n4 * 16 / 1e9;
    for (int blocks : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        const float a = timed([&] { hipLaunchKernelGGL((write_kernel<64, 0>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float b = timed([&] { hipLaunchKernelGGL((write_kernel<4, 0>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float c = timed([&] { hipLaunchKernelGGL((write_kernel<64, 1>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float d = timed([&] { hipLaunchKernelGGL((write_kernel<8, 0>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        printf("blocks %5d: 1 KB runs %.0f GB/s | 64 B runs %.0f GB/s | 128 B runs %.0f GB/s | 1 KB runs nontemporal %.0f GB/s\n", blocks, gb / a * 1e3, gb / b * 1e3,
               gb / d * 1e3, gb / c * 1e3);
    }
    for (int blocks : {256, 256 * 2, 256 * 4, 256 * 8, 256 * 32}) {
        const float a = timed([&] { hipLaunchKernelGGL((write_unrolled<4, false>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float b = timed([&] { hipLaunchKernelGGL((write_unrolled<4, true>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float c = timed([&] { hipLaunchKernelGGL((write_unrolled<8, false>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float d = timed([&] { hipLaunchKernelGGL((write_unrolled<16, false>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        printf("blocks %5d: 4 rows per trip %.0f GB/s | 64 B per lane %.0f GB/s | 8 rows %.0f GB/s | 16 rows %.0f GB/s\n", blocks, gb / a * 1e3, gb / b * 1e3, gb / c * 1e3,
               gb / d * 1e3);
    }
    for (int blocks : {256, 256 * 2, 256 * 8}) {
        const float a = timed([&] { hipLaunchKernelGGL((write_panels<4>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float b = timed([&] { hipLaunchKernelGGL((write_panels<8>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float c = timed([&] { hipLaunchKernelGGL((write_panels<16>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        const float d = timed([&] { hipLaunchKernelGGL((write_panels<64>), dim3(blocks), dim3(256), 0, 0, p, n4, 1.f); });
        printf("blocks %5d, panels (runs of an instruction 32 KB apart, consecutive instructions adjacent): 64 B runs %.0f GB/s | 128 B %.0f | 256 B %.0f | 1 KB %.0f\n", blocks,
               gb / a * 1e3, gb / b * 1e3, gb / c * 1e3, gb / d * 1e3);
    }
    const float m = timed([&] { hipMemsetAsync(p, 0, n4 * sizeof(float4), 0); });
    printf("hipMemsetAsync: %.0f GB/s\n", gb / m * 1e3);
    return 0;
}
