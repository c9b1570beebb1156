// mfma_issue.hip -- how fast does ONE wavefront per SIMD issue independent v_mfma_f32_16x16x4_f32?
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_issue.hip -o tools/microbench/mfma_issue
// Each variant runs a loop of 48 MFMAs per iteration for 512 iterations and reports shader cycles per MFMA
// (s_memtime around the loop, median over wavefronts), with 1, 2 or 3 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int VARIANT>
__global__ void k(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float4 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_float4(i * 1e-3f, 1.0f, 0.5f, 0.25f);
    __syncthreads();
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane * 1e-3f, b = 0.5f + lane * 1e-3f;
    float x[20];
    for (int i = 0; i < 20; ++i) x[i] = lane - 30.0f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) {                 // 12 independent accumulators, round robin, nothing else
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = MFMA(a, b, acc[i]);
        } else if (VARIANT == 1) {          // + 20 integer max beside them
#pragma unroll
            for (int i = 0; i < 20; ++i) { int v = __builtin_bit_cast(int, x[i]); x[i] = __builtin_bit_cast(float, (v > 0 ? v : 0) + it); }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = MFMA(x[(r * 12 + i) % 20], b, acc[i]);
        } else if (VARIANT == 2) {          // + 5 ds_read_b128 a whole iteration ahead (operands from LDS)
            float4 w[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) w[i] = lds[((it * 5 + i) * 64 + lane) & 2047];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = MFMA(a, b, acc[i]);
            a += w[0].x + w[1].y + w[2].z + w[3].w + w[4].x;
        } else if (VARIANT == 3) {          // A operand changes every MFMA (distinct registers), B fixed
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = MFMA(x[(r * 12 + i) % 20], x[(r * 7 + i) % 20], acc[i]);
        } else if (VARIANT == 4) {          // 2 accumulators only (dependent every second MFMA)
#pragma unroll
            for (int r = 0; r < 24; ++r) { acc[0] = MFMA(a, b, acc[0]); acc[1] = MFMA(a, b, acc[1]); }
        } else if (VARIANT == 6) {          // every MFMA writes another register block than its srcC (ping-pong sets)
            f32x4 alt[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) alt[i] = MFMA(a, b, acc[i]);
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = MFMA(a, b, alt[i]);
#pragma unroll
            for (int i = 0; i < 12; ++i) alt[i] = MFMA(a, b, acc[i]);
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = MFMA(a, b, alt[i]);
        } else if (VARIANT == 7) {          // 5 chains start from ONE shared srcC block each round (a(n) = U + ...)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = MFMA(a, b, r == 0 ? acc[11] : acc[i]);
#pragma unroll
                for (int i = 5; i < 10; ++i) acc[i] = MFMA(a, b, acc[i]);
                acc[10] = MFMA(a, b, acc[10]);
                acc[11] = MFMA(a, b, acc[11]);
            }
        } else if (VARIANT == 8) {          // B operand = result of a v_max_i32 on the previous round's accumulators
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const float f_ = acc[(i + 6) % 12][r]; const int v = __builtin_bit_cast(int, f_);
                    acc[i] = MFMA(a, __builtin_bit_cast(float, v > 0 ? v : 0), acc[i]);
                }
        } else if (VARIANT == 9 || VARIANT == 10) {   // 20 v_max_i32 READ MFMA results (nothing reads theirs back)
            int keep[20];
            if (VARIANT == 9) {             // all together, ahead of the MFMAs
#pragma unroll
                for (int i = 0; i < 20; ++i) { const float f_ = acc[i % 5][i / 5]; const int v = __builtin_bit_cast(int, f_); keep[i] = v > 0 ? v : 0; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc[i] = MFMA(a, b, acc[i]);
            } else {                        // one after every second MFMA
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        acc[i] = MFMA(a, b, acc[i]);
                        const int n = r * 12 + i;
                        if ((n & 1) && n / 2 < 20) {
                            const float f_ = acc[(i + 6) % 12][r]; const int v = __builtin_bit_cast(int, f_); keep[n / 2] = v > 0 ? v : 0;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
#pragma unroll
            for (int i = 0; i < 20; ++i) x[i] = __builtin_bit_cast(float, keep[i] ^ __builtin_bit_cast(int, x[i]));
        } else if (VARIANT >= 11 && VARIANT <= 13) {  // SARL's ratio: 16 bytes of A operand per lane per 4 MFMAs
            float acc_w = 0.f;
            if (VARIANT == 11) {            // 12 ds_read_b128 per 48 MFMAs, consumed an iteration later
                float4 w[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) w[i] = lds[((it * 12 + i) * 64 + lane) & 2047];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc[i] = MFMA(a, b, acc[i]);
#pragma unroll
                for (int i = 0; i < 12; ++i) acc_w += w[i].x + w[i].w;
            } else if (VARIANT == 12) {     // the same bytes as 24 ds_read_b64
                float2 w[24];
                const float2 *l2 = reinterpret_cast<const float2 *>(lds);
#pragma unroll
                for (int i = 0; i < 24; ++i) w[i] = l2[((it * 24 + i) * 64 + lane) & 4095];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc[i] = MFMA(a, b, acc[i]);
#pragma unroll
                for (int i = 0; i < 24; ++i) acc_w += w[i].x + w[i].y;
            } else {                        // the same bytes as 12 global_load_dwordx4 from an L2-resident table
                float4 w[12];
                const float4 *g4 = reinterpret_cast<const float4 *>(out);
#pragma unroll
                for (int i = 0; i < 12; ++i) w[i] = g4[((it * 12 + i) * 64 + lane) & 4095];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc[i] = MFMA(a, b, acc[i]);
#pragma unroll
                for (int i = 0; i < 12; ++i) acc_w += w[i].x + w[i].w;
            }
            a += acc_w * 1e-30f;
        } else if (VARIANT == 5) {          // srcC from another register than vdst once per chain of 4
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                f32x4 t = MFMA(a, b, acc[(i + 5) % 12]);
                t = MFMA(a, b, t); t = MFMA(a, b, t); acc[i] = MFMA(a, b, t);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 20; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + a;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
static void run(const char *what)
{
    const int iters = 512, blocks = 256;
    for (int waves : {4, 8}) {
        float *out; unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * blocks * waves * 64);
        hipMalloc(&cyc, sizeof(unsigned long long) * blocks * waves);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * waves);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per = (double)h[h.size() / 2] / (iters * 48.0);
        printf("%-62s %d wave(s)/SIMD: %6.1f cycles per MFMA per wave = %5.1f per MFMA on the pipe\n", what, waves / 4, per,
               per / (waves / 4));
        hipFree(out); hipFree(cyc);
    }
}

int main()
{
    run<0>("12 accumulators round robin, MFMAs only");
    run<1>("+ 20 v_max_i32 per 48 MFMAs, A operand from them");
    run<2>("+ 5 ds_read_b128 per 48 MFMAs issued an iteration ahead");
    run<3>("A and B operands from 20 different registers");
    run<4>("2 accumulators (dependent every second MFMA)");
    run<5>("chains of 4 whose first srcC is another chain's result");
    run<6>("every MFMA writes another register block than its srcC");
    run<7>("5 chains restart from one shared srcC block each round");
    run<8>("B operand = v_max_i32 of another accumulator's element");
    run<11>("12 ds_read_b128 per 48 MFMAs (one per 4: SARL's ratio)");
    run<12>("24 ds_read_b64 per 48 MFMAs (same bytes)");
    run<13>("12 global_load_dwordx4 per 48 MFMAs (same bytes, L2)");
    run<9>("20 v_max_i32 read MFMA results, all ahead of the 48 MFMAs");
    run<10>("20 v_max_i32 read MFMA results, one per second MFMA");
    return 0;
}
