// mfma_bf16_shapes.hip -- cycles per instruction of the bf16 MFMA shapes on one gfx950 SIMD (one and two wavefronts):
// v_mfma_f32_16x16x32_bf16 (gfx950's K = 32 form) against the legacy v_mfma_f32_16x16x16_bf16 (K = 16, "_1k").
// Question: could the ragged last 16 features of a 100-wide layer run as ONE K = 16 instruction at half the cycles of a
// K = 32 one (the bf16x3 layers pad 112 -> 128 today)?   Build: hipcc -O3 --offload-arch=gfx950 <this> -o mfma_bf16_shapes
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void k(float *out, unsigned long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a8, b8;
    bf16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(1.0f + lane * 1e-3f + i); b8[i] = (__bf16)(0.5f + i * 1e-2f); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (SHAPE == 32) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[r & 3], 0, 0, 0);
            else             acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[r & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

int main()
{
    const int blocks = 256, iters = 2000;
    for (int waves : {4, 8}) {
        float *out; unsigned long long *cyc;
        hipMalloc(&out, (size_t)blocks * waves * 64 * 4); hipMalloc(&cyc, (size_t)blocks * waves * 8);
        for (int shape : {32, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, iters);
                else             hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, iters);
            }
            hipDeviceSynchronize();
            std::vector<unsigned long long> c((size_t)blocks * waves);
            hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
            std::sort(c.begin(), c.end());
            const double per = (double)c[c.size() / 2] / (iters * 16.0);
            printf("v_mfma_f32_16x16x%d_bf16, %d wavefront(s) per SIMD: %.2f cycles per instruction per wavefront = %.2f per SIMD "
                   "(%.0f FLOP per cycle and SIMD)\n", shape, waves / 4, per, per / (waves / 4), 16 * 16 * shape * 2 / (per / (waves / 4)));
        }
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
