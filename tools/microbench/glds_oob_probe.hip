// Probe: does `buffer_load_dwordx4 ... lds` (LDS-DMA) write zeros for lanes whose offset is out of range of the
// buffer resource?  The Winograd conv kernel relies on it for the zero padding of halo pixels.
//   hipcc --offload-arch=gfx950 -O3 glds_oob_probe.hip -o glds_oob_probe && ./glds_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* src, float* out, unsigned nbytes)
{
    __shared__ __attribute__((aligned(16))) float smem[256];
    for (int i = threadIdx.x; i < 256; i += 64) smem[i] = -7.f;   // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000u;      // odd lanes: far out of range
    if (threadIdx.x == 62) voff = nbytes - 8;     // straddles the end: partial
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}

int main()
{
    const int n = 256;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *d, *o;
    hipMalloc(&d, n * 4);
    hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, (unsigned)(n * 4));
    std::vector<float> r(n);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        for (int c = 0; c < 4; ++c) {
            const float got = r[l * 4 + c];
            float want = (l & 1) ? 0.f : h[l * 4 + c];
            if (l == 62) want = c < 2 ? h[n - 2 + c] : 0.f;
            if (got != want) {
                if (bad < 10) printf("lane %d comp %d: got %g want %g\n", l, c, got, want);
                ++bad;
            }
        }
    }
    printf(bad ? "glds_oob_probe: %d mismatches\n" : "glds_oob_probe: OK (out-of-range lanes write zeros)\n", bad);
    return bad != 0;
}
