// valu_issue.hip -- the vector-ALU issue rate of one gfx950 SIMD with 1 / 2 / 4 resident wavefronts.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_issue.hip -o tools/microbench/valu_issue
//
// Question: bench.py prices `roofline_valu` against one wave64 vector instruction per 4 cycles
// and SIMD.  Is that the SIMD's rate, or only what ONE wavefront reaches (with two or more wavefronts interleaving
// at one instruction per 2 cycles)?  Each variant issues 64 independent instructions per loop iteration (16 chains,
// so no instruction waits for its own result) from every wavefront of a 256-workgroup grid (one workgroup per CU,
// 4 / 8 / 16 wavefronts = 1 / 2 / 4 per SIMD) and reports shader cycles (s_memtime) per instruction per wavefront
// and per SIMD.  The instructions are pinned with inline asm so the compiler cannot fuse, pack or drop them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

enum { FMA32, ADD32, CNDMASK, MOV, PKFMA, FMA64, MUL64, EXP32, RCP32, MAXI32, FMA32_DPP, MIX_FMA_CND };

template <int V>
__global__ void k(float *out, unsigned long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    float x[16];
    f2 p[16];
    double d[16];
    for (int i = 0; i < 16; ++i) { x[i] = 1.0f + lane * 1e-3f + i; p[i] = (f2){x[i], x[i] + 0.5f}; d[i] = x[i]; }
    float a = 1.0f + 1e-7f * lane, b = 1e-9f * lane;
    f2 a2 = (f2){a, a}, b2 = (f2){b, b};
    double ad = a, bd = b;
    const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)iters;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (V == FMA32)        asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (V == ADD32)   asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if (V == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(mask));
                else if (V == MOV)     asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 15]));
                else if (V == PKFMA)   asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
                else if (V == FMA64)   asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d[i]) : "v"(ad), "v"(bd));
                else if (V == MUL64)   asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(ad));
                else if (V == EXP32)   asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                else if (V == RCP32)   asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                else if (V == MAXI32)  asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (V == FMA32_DPP) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(b));
                else if (V == MIX_FMA_CND) {
                    if (i & 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(mask));
                    else       asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i] + p[i][0] + p[i][1] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
static void run(const char *what, double flop_per_lane_inst)
{
    const int iters = 2048, blocks = 256;
    for (int waves : {4, 8, 16}) {
        float *out; unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * blocks * waves * 64);
        hipMalloc(&cyc, sizeof(unsigned long long) * blocks * waves);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * waves);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per = (double)h[h.size() / 2] / (iters * 64.0);
        const double per_simd = per / (waves / 4);
        printf("%-44s %d wave(s)/SIMD: %6.2f cycles per instruction per wave = %5.2f per instruction on the SIMD", what,
               waves / 4, per, per_simd);
        if (flop_per_lane_inst > 0)
            printf("  (%.1f TFLOP/s at 2.4 GHz x 1024 SIMDs)", flop_per_lane_inst * 64 / per_simd * 2.4e9 * 1024 / 1e12);
        printf("\n");
        hipFree(out); hipFree(cyc);
    }
}

int main()
{
    run<FMA32>("v_fma_f32 (VOP3, 16 independent chains)", 2);
    run<ADD32>("v_add_f32 (VOP2)", 1);
    run<CNDMASK>("v_cndmask_b32 (select, SGPR-pair mask)", 0);
    run<MOV>("v_mov_b32", 0);
    run<MAXI32>("v_max_i32", 0);
    run<FMA32_DPP>("v_add_f32 with DPP quad_perm", 1);
    run<MIX_FMA_CND>("v_fma_f32 / v_cndmask_b32 alternating", 0);
    run<PKFMA>("v_pk_fma_f32 (2 x float32 per lane)", 4);
    run<FMA64>("v_fma_f64", 2);
    run<MUL64>("v_mul_f64", 1);
    run<EXP32>("v_exp_f32 (transcendental)", 0);
    run<RCP32>("v_rcp_f32 (transcendental)", 0);
    return 0;
}
