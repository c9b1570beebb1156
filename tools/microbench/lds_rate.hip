// lds_rate.hip -- cycles of the CU's LDS pipe per ds_read_b128 wave-instruction (lane-linear, conflict-free 16 B per lane)
// and per global_load_lds_dwordx4 piece, with 8 wavefronts per CU (two 4-wavefront workgroups) issuing nothing else.
// Question (round 4): is the bf16x3 layer loop -- per CU and 16 matrix-pipe cycles 4 MFMAs, 2 ds_read_b128 and half a DMA
// piece -- bound by the LDS pipe itself?   Build: hipcc -O3 --offload-arch=gfx950 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: 16 ds_read_b128 per iteration; 1: 8 global_load_lds_dwordx4 per iteration; 2: both
__global__ __launch_bounds__(256, 2) void k(const float4 *__restrict__ table, float *out, unsigned long long *cyc, int iters)
{
    __shared__ float4 buf[2][2048];
    const int tid = threadIdx.x, lane = tid & 63, wave_base = tid & ~63;
    for (int i = tid; i < 4096; i += 256) (&buf[0][0])[i] = make_float4(1.f, 2.f, 3.f, (float)i);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    const float4 *src = buf[0] + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
            float4 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = src[r * 64];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(table) +
                                                                       (unsigned)(((it & 63) * 2048 + r * 256 + tid) * 16)),
                    (__attribute__((address_space(3))) void *)(buf[1] + r * 256 + wave_base), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the pieces have landed before the next round reuses the slots
        }
        asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE> static double run(const float4 *table, float *out, unsigned long long *cyc, int iters)
{
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, table, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(2048);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    return (double)c[c.size() / 2] / iters;
}

int main()
{
    float4 *table; float *out; unsigned long long *cyc;
    hipMalloc(&table, (size_t)64 * 2048 * 16 + 4096); hipMemset(table, 0, (size_t)64 * 2048 * 16 + 4096);
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 2048 * 8);
    const int iters = 20000;
    const double a = run<0>(table, out, cyc, iters), b = run<1>(table, out, cyc, iters), c = run<2>(table, out, cyc, iters);
    printf("# 8 wavefronts per CU; s_memtime ticks per iteration and wavefront (tick ~ shader clock)\n");
    printf("16 x ds_read_b128 per iteration:            %8.1f ticks = %.2f per wave-instruction = %.2f LDS cycles per instruction and CU\n", a, a / 16, a / 16 / 8);
    printf("8 x global_load_lds_dwordx4 per iteration:  %8.1f ticks = %.2f per piece and wavefront = %.2f per piece and CU\n", b, b / 8, b / 8 / 8);
    printf("both:                                       %8.1f ticks (sum of the two: %.1f)\n", c, a + b);
    return 0;
}
