// mfma_clock.hip -- the shader clock an MI355X sustains while every SIMD streams v_mfma_f32_16x16x4_f32 for ~2 ms
// (the length of one SARL look-ahead), and the TFLOP/s that gives.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_clock.hip -o tools/microbench/mfma_clock
// s_memtime counts shader clocks, s_memrealtime a constant 100 MHz: their ratio over the kernel is the clock.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamp, int iters)
{
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamp[2 * w] = c1 - c0; stamp[2 * w + 1] = r1 - r0;
    }
}

int main()
{
    const int blocks = 512, waves = 4;            // two 4-wavefront workgroups per CU: 2 wavefronts per SIMD
    float *out; unsigned long long *st;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&st, 16 * blocks * waves);
    for (int iters : {1000, 40000, 40000, 40000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks * waves);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> ghz;
        for (int w = 0; w < blocks * waves; ++w) ghz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] * 10.0));   // cycles per ns
        std::sort(ghz.begin(), ghz.end());
        const double flop = (double)blocks * waves * iters * 64.0 * 2048.0;
        printf("%6d iterations x 64 MFMAs per wavefront, 2 wavefronts per SIMD: %.3f ms, %.1f TFLOP/s, shader clock median %.3f GHz "
               "(min %.3f, max %.3f); 157.3 TFLOP/s assumes 2.4 GHz\n", iters, ms, flop / ms / 1e9, ghz[ghz.size() / 2],
               ghz.front(), ghz.back());
    }
    return 0;
}
