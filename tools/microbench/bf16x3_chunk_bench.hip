// Micro-benchmark for VERDICT r05 next #2 (gate 1): would the channel contraction of the F(4x4,3x3) conv unit run >= 1.8x faster on the bf16 matrix
// pipe with an error-free 3-piece split -- every float32 operand = hi + mid + lo bf16 pieces (24 significand bits), six products
// (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid) accumulated in float32 by v_mfma_f32_16x16x32_bf16 -- than on v_mfma_f32_16x16x4_f32?
//
// The unit of comparison is what one wave of csrc/rcu_wino4.hip owns (ONE wave per SIMD: its 288 accumulator registers are why): 16 tiles x 32
// output channels x 36 positions, over 32 input channels:
//   float32 today : 4 chunks of 8 channels x 144 MFMAs (32 cycles each) = 18.4 k cycles of matrix time, 26.4 k cycles in the shipped kernel
//                   (6.6 k per chunk: profiles/r03_wino4_trace.txt)
//   bf16 x 6      : 36 positions x 2 cout blocks x 6 products = 432 MFMAs of K = 32 (16 cycles each) = 6.9 k cycles of matrix time
// and what has to happen beside those 432 MFMAs in a real kernel:
//   flag 1  the B operands: 3 pieces x 2 cout blocks of weights per position = 216 ds_read_b128 per wave (each operand is used by ONE MFMA: the
//           wave has one 16-tile block) -- 1 KB per MFMA per SIMD = 256 B/clk per CU asked of an LDS that delivers 128
//   flag 2  the split of the A operands: 8 transformed float32 values per lane and position -> 3 x 8 bf16 (11 VALU operations per value pair)
//   flag 4  the input transform's VALU work (1152 packed operations per 32 channels, as the shipped kernel's 288 per 8)
//   flag 8  the raw patch reads (72 ds_read_b128 per wave)
//   hipcc --offload-arch=gfx950 -O3 bf16x3_chunk_bench.hip -o bf16x3_chunk_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = hi + mid + lo exactly (three bf16 pieces of a float32: 8 + 8 + 8 significand bits); two values per call, packed pieces out
__device__ __forceinline__ void split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo)
{
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    hi = __builtin_bit_cast(unsigned, h);
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xFFFF0000u);
    const bf16x2 m = {(__bf16)ra, (__bf16)rb};
    mid = __builtin_bit_cast(unsigned, m);
    const float sa = ra - __builtin_bit_cast(float, mid << 16), sb = rb - __builtin_bit_cast(float, mid & 0xFFFF0000u);
    const bf16x2 l = {(__bf16)sa, (__bf16)sb};
    lo = __builtin_bit_cast(unsigned, l);
}

template <int FLAGS>
__global__ __launch_bounds__(256, 1) void bench_bf16(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 36 * 1024; i += 256) lds[i] = 0.001f * (i & 1023);
    __syncthreads();
    f32x4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ah = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, am = ah, al = ah;
    u32x4 bw[6];
    for (int i = 0; i < 6; ++i) bw[i] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i};
    f32x2 raw[4] = {{1.f + lane, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f + wave}};
    f32x2 tr[16];
    for (int i = 0; i < 16; ++i) tr[i] = f32x2{0.5f * i, 1.f + lane * 0.01f};
    const float* wl = lds + lane * 4;                       // weights: [position][piece x block][64 lanes][4 dwords]: lane-linear, conflict-free
    const float* pl = lds + 24 * 1024 + lane * 4;
    for (int it = 0; it < iters; ++it) {
        const float itf = (float)it;
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            __builtin_amdgcn_sched_barrier(0);
            if (FLAGS & 1) {
#pragma unroll
                for (int k = 0; k < 6; ++k) bw[k] = *(const volatile __attribute__((address_space(3))) u32x4*)(wl + ((p % 4) * 6 + k) * 256);
            }
            if (FLAGS & 8) {   // the lane's 8 channels of a raw patch pixel: 2 x 16 bytes
                const f32x4 r0 = *(const volatile __attribute__((address_space(3))) f32x4*)(pl + (p % 8) * 512), r1 = *(const volatile __attribute__((address_space(3))) f32x4*)(pl + (p % 8) * 512 + 256);
                raw[0] = f32x2{r0.x, r0.y}; raw[1] = f32x2{r0.z, r0.w}; raw[2] = f32x2{r1.x, r1.y}; raw[3] = f32x2{r1.z, r1.w};
            }
            if (FLAGS & 4) {   // 32 packed operations per position: the transform's share (1152 per 32 channels)
#pragma unroll
                for (int k = 0; k < 16; ++k) tr[k] = tr[k] * 4.f + raw[k & 3];
#pragma unroll
                for (int k = 0; k < 16; ++k) tr[k] = tr[k] - tr[(k + 5) & 15];
            }
            if (FLAGS & 2) {   // the position's 8 values (4 channel pairs) -> three packed pieces each
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x2 v = (FLAGS & 4) ? tr[4 * (p & 3) + k] : raw[k] * (1.f + p) + itf;      // (itf: nothing to hoist out of the loop)
                    unsigned h_, m_, l_;
                    split3(v.x, v.y, h_, m_, l_);
                    ah[k] = h_; am[k] = m_; al[k] = l_;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
            {   // six products per cout block, the two blocks' accumulators in turn (no MFMA reads the result of the one before it)
                f32x4& c0 = acc[2 * (p % 12)];
                f32x4& c1 = acc[2 * (p % 12) + 1];
                const bf16x8 B0h = __builtin_bit_cast(bf16x8, bw[0]), B0m = __builtin_bit_cast(bf16x8, bw[1]), B0l = __builtin_bit_cast(bf16x8, bw[2]);
                const bf16x8 B1h = __builtin_bit_cast(bf16x8, bw[3]), B1m = __builtin_bit_cast(bf16x8, bw[4]), B1l = __builtin_bit_cast(bf16x8, bw[5]);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B0h, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B1h, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B0m, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B1m, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, B0h, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, B1h, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B0l, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B1l, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, B0h, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, B1h, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, B0m, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, B1m, c1, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 24; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 16; ++i) s += tr[i].x;
    if (s == 123.456f) out[threadIdx.x] = s;
}

// The same unit software-pipelined the way a real kernel would have to be: position p's 12 MFMAs are issued with the NEXT position's work between them --
// its weight pieces and raw patch requested first (landing behind the MFMAs), its transform share and its 3-piece split as VALU slices behind
// each MFMA (one wave per SIMD issues in order: VALU work that is not between MFMAs is not beside them).
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void bench_bf16_pipelined(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 36 * 1024; i += 256) lds[i] = 0.001f * (i & 1023);
    __syncthreads();
    f32x4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a3[2][3];
    u32x4 bw[2][6];
    for (int h = 0; h < 2; ++h) {
        for (int i = 0; i < 3; ++i) a3[h][i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i};
        for (int i = 0; i < 6; ++i) bw[h][i] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i};
    }
    f32x2 raw[4] = {{1.f + lane, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f + wave}};
    f32x2 tr[16];
    for (int i = 0; i < 16; ++i) tr[i] = f32x2{0.5f * i, 1.f + lane * 0.01f};
    const float* wl = lds + lane * 4;
    const float* pl = lds + 24 * 1024 + lane * 4;
    for (int it = 0; it < iters; ++it) {
        const float itf = (float)it;
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            constexpr int dummy = 0;
            (void)dummy;
            const int cur = p & 1, nxt = cur ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
            if (FLAGS & 1) {
#pragma unroll
                for (int k = 0; k < 6; ++k) bw[nxt][k] = *(const volatile __attribute__((address_space(3))) u32x4*)(wl + (((p + 1) % 4) * 6 + k) * 256);
            }
            if (FLAGS & 8) {
                r0 = *(const volatile __attribute__((address_space(3))) f32x4*)(pl + (p % 8) * 512);
                r1 = *(const volatile __attribute__((address_space(3))) f32x4*)(pl + (p % 8) * 512 + 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, a3[cur][0]), Am = __builtin_bit_cast(bf16x8, a3[cur][1]), Al = __builtin_bit_cast(bf16x8, a3[cur][2]);
            const bf16x8 B0h = __builtin_bit_cast(bf16x8, bw[cur][0]), B0m = __builtin_bit_cast(bf16x8, bw[cur][1]), B0l = __builtin_bit_cast(bf16x8, bw[cur][2]);
            const bf16x8 B1h = __builtin_bit_cast(bf16x8, bw[cur][3]), B1m = __builtin_bit_cast(bf16x8, bw[cur][4]), B1l = __builtin_bit_cast(bf16x8, bw[cur][5]);
            f32x4& c0 = acc[2 * (p % 12)];
            f32x4& c1 = acc[2 * (p % 12) + 1];
            // VALU slice `u` (0..11) of the next position's work
            auto slice = [&](int u) {
                if ((FLAGS & 4) && u < 4) {          // the transform's share: 32 packed operations in four slices
#pragma unroll
                    for (int k = 4 * u; k < 4 * u + 4; ++k) tr[k] = tr[k] * 4.f + raw[k & 3];
#pragma unroll
                    for (int k = 4 * u; k < 4 * u + 4; ++k) tr[k] = tr[k] - tr[(k + 5) & 15];
                }
                if ((FLAGS & 8) && u == 4) {
                    raw[0] = f32x2{r0.x, r0.y}; raw[1] = f32x2{r0.z, r0.w}; raw[2] = f32x2{r1.x, r1.y}; raw[3] = f32x2{r1.z, r1.w};
                }
                if ((FLAGS & 2) && u >= 4 && (u & 1) == 0) {      // the split of channel pair (u - 4) / 2 in slices u, u + 1 (issued here as one clump of 11)
                    const int k = (u - 4) >> 1;
                    const f32x2 v = (FLAGS & 4) ? tr[4 * (p & 3) + k] : raw[k] * (1.f + p) + itf;
                    unsigned h_, m_, l_;
                    split3(v.x, v.y, h_, m_, l_);
                    a3[nxt][0][k] = h_; a3[nxt][1][k] = m_; a3[nxt][2][k] = l_;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B0h, c0, 0, 0, 0); slice(0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B1h, c1, 0, 0, 0); slice(1);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B0m, c0, 0, 0, 0); slice(2);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B1m, c1, 0, 0, 0); slice(3);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, B0h, c0, 0, 0, 0); slice(4);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, B1h, c1, 0, 0, 0); slice(5);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B0l, c0, 0, 0, 0); slice(6);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B1l, c1, 0, 0, 0); slice(7);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, B0h, c0, 0, 0, 0); slice(8);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, B1h, c1, 0, 0, 0); slice(9);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, B0m, c0, 0, 0, 0); slice(10);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, B1m, c1, 0, 0, 0); slice(11);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 24; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 16; ++i) s += tr[i].x;
    if (s == 123.456f) out[threadIdx.x] = s;
}

// the same unit on the float32 pipe: 4 chunks of 8 channels x 144 MFMAs; flag 1: the chunk's 36 weight reads (ds_read_b128), flag 4: its 288 packed
// transform operations, flag 8: its 36 patch reads (ds_read_b64)
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void bench_f32(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 36 * 1024; i += 256) lds[i] = 0.001f * (i & 1023);
    __syncthreads();
    f32x4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 w = {1.f, 2.f + lane, 3.f, 4.f};
    f32x2 d = {1.f + lane, 2.f};
    f32x2 tr[8];
    for (int i = 0; i < 8; ++i) tr[i] = f32x2{0.5f * i, 1.f + lane * 0.01f};
    const float* wl = lds + lane * 4;
    const float* pl = lds + 24 * 1024 + lane * 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int chunk = 0; chunk < 4; ++chunk) {
#pragma unroll
            for (int p = 0; p < 36; ++p) {
                __builtin_amdgcn_sched_barrier(0);
                if (FLAGS & 1) w = *(const volatile __attribute__((address_space(3))) f32x4*)(wl + (p % 24) * 256);
                if (FLAGS & 8) d = *(const volatile __attribute__((address_space(3))) f32x2*)(pl + (p % 24) * 128);
                if (FLAGS & 4) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) tr[k] = tr[k] * 4.f + (k & 1 ? d : tr[(k + 3) & 7]);
                    d = tr[p & 7];
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4& c0 = acc[2 * (p % 12)];
                f32x4& c1 = acc[2 * (p % 12) + 1];
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d.x, w.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d.x, w.z, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d.y, w.y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d.y, w.w, c1, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 24; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <class K>
static double time_kernel(K kernel, float* out, int iters)
{
    hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(256), dim3(256), 160 * 1024, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main()
{
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 2000;
    // one "unit" = 16 tiles x 32 couts x 36 positions x 32 input channels per wave; canonical work of a unit in the Winograd domain: 2 * 16 * 32 * 36 * 32 flop
    const double unit_flop = 2.0 * 16 * 32 * 36 * 32, units = 256.0 * 4 * iters;
    auto report = [&](const char* name, double ms, double ref_ms) {
        const double us_per_unit = ms * 1e3 / iters;                 // all waves run their units side by side: time per unit of one wave
        printf("%-78s %8.3f ms  %7.2f us per unit  %6.1f TFLOP/s in the Winograd domain  %5.2fx\n", name, ms, us_per_unit, units * unit_flop / ms / 1e9,
               ref_ms / ms);
    };
    const double f0 = time_kernel(bench_f32<0>, out, iters);
    const double f13n = time_kernel(bench_f32<13>, out, iters);
    // THE REFERENCE (1.00x): the shipped kernel's unit -- 6.6 k cycles per 8-channel chunk (profiles/r03_wino4_trace.txt: hand-scheduled fillers,
    // LDS-DMA, barrier) = 26.4 k cycles, on this box's clock as the bare float32 stream shows it (576 MFMAs x 32 cycles = 18,432 cycles)
    const double f13 = f0 * 26400.0 / 18432.0;
    report("float32: 576 v_mfma_f32_16x16x4_f32, operands in registers", f0, f13);
    report("float32: + reads + transform VALU as THIS file schedules them (not the reference)", f13n, f13);
    report("float32: the shipped kernel's unit, 26.4 k cycles on this clock  (= the reference, 1.00x)", f13, f13);
    report("bf16 x 6: 432 v_mfma_f32_16x16x32_bf16, operands in registers", time_kernel(bench_bf16<0>, out, iters), f13);
    report("bf16 x 6: + 216 ds_read_b128 of weight pieces", time_kernel(bench_bf16<1>, out, iters), f13);
    report("bf16 x 6: + the 3-piece split of the A operands (VALU)", time_kernel(bench_bf16<2>, out, iters), f13);
    report("bf16 x 6: + weight reads + split", time_kernel(bench_bf16<3>, out, iters), f13);
    report("bf16 x 6: + weight reads + split + transform VALU", time_kernel(bench_bf16<7>, out, iters), f13);
    report("bf16 x 6: + weight reads + split + transform VALU + raw patch reads", time_kernel(bench_bf16<15>, out, iters), f13);
    report("bf16 x 6 PIPELINED (next position's reads + VALU between the MFMAs): weight reads", time_kernel(bench_bf16_pipelined<1>, out, iters), f13);
    report("bf16 x 6 PIPELINED: split", time_kernel(bench_bf16_pipelined<2>, out, iters), f13);
    report("bf16 x 6 PIPELINED: weight reads + split", time_kernel(bench_bf16_pipelined<3>, out, iters), f13);
    report("bf16 x 6 PIPELINED: weight reads + split + transform VALU", time_kernel(bench_bf16_pipelined<7>, out, iters), f13);
    report("bf16 x 6 PIPELINED: weight reads + split + transform VALU + raw patch reads", time_kernel(bench_bf16_pipelined<15>, out, iters), f13);
    printf("(the shipped float32 kernel spends 6.6 k cycles per 8-channel chunk = 26.4 k per unit incl. LDS-DMA, barriers and the cold first chunk;\n"
           " the reference row above is the same unit without those)\n");
    return 0;
}
