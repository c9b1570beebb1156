// dma_issue.hip -- what a wavefront of the bf16x3 look-ahead pays for moving its weights (gfx950, round 4).
// The kernel's what-if builds (profiles/r04_sarl_x3_whatif.txt) say the weight stream costs ~21 % of a step whatever
// its bytes.  This isolates it: every wavefront runs the layer loop's shape -- groups of 6 dependent
// v_mfma_f32_16x16x32_bf16 with 3 ds_read_b128 per group, a workgroup barrier every 8 groups (one "chunk") -- and moves
// 6 KiB per wavefront and chunk (a 24-KiB chunk shared by 4 wavefronts) from an L2-resident table in one of these ways:
//   0  nothing
//   1  6 x global_load_lds_dwordx4 behind the chunk's first MFMA (the kernel's form)
//   2  6 x global_load_dwordx4 into registers behind the first MFMA, 6 x ds_write_b128 behind the fifth group's
//   3  as 1 with 4 bytes per lane (same count, a quarter of the bytes)
//   4  as 1, one piece per group (spread)
//   5  3 x global_load_lds_dwordx4 (half the count: what an 8-wavefront workgroup or two tiles per wavefront would issue)
// 4-wavefront workgroups, 2 per CU (two wavefronts per SIMD), as the kernel.
// Build: hipcc -O3 --offload-arch=gfx950 dma_issue.hip -o dma_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kChunkRows = 1536;          // float4 rows per chunk (24 KiB)
constexpr int kGroups = 8;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float4 *__restrict__ table, int table_rows, float *out,
                                           unsigned long long *cyc, int chunks)
{
    __shared__ float4 stage[2][kChunkRows + 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_base = tid & ~63;
    bf16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)(0.5f + i * 1e-2f + lane * 1e-3f);
    for (int i = tid; i < 2 * (kChunkRows + 256); i += 256) (&stage[0][0])[i] = make_float4(1.f, 2.f, 3.f, (float)i);
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    float4 held[6];
    for (int i = 0; i < 6; ++i) held[i] = make_float4(0, 0, 0, 0);
    unsigned row = (blockIdx.x * 977u) % (unsigned)(table_rows - kChunkRows);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int c = 0; c < chunks; ++c) {
        float4 *dst = stage[(c + 1) & 1];
        const float4 *src = stage[c & 1];
        const float4 *from = table + row;
        row += kChunkRows;
        if (row >= (unsigned)(table_rows - kChunkRows)) row -= (unsigned)(table_rows - kChunkRows);
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            const float4 w0 = src[(g * 192 + lane) % kChunkRows], w1 = src[(g * 192 + 64 + lane) % kChunkRows],
                         w2 = src[(g * 192 + 128 + lane) % kChunkRows];
            if (g == kGroups - 1) __syncthreads();
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2), b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1 || MODE == 3 || MODE == 5) {
                if (g == 0) {
#pragma unroll
                    for (int r = 0; r < (MODE == 5 ? 3 : 6); ++r) {
                        const auto *gp = (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(from) +
                                                                                            (unsigned)((r * 256 + tid) * 16));
                        auto *lp = (__attribute__((address_space(3))) void *)(dst + r * 256 + wave_base);
                        if constexpr (MODE == 3) __builtin_amdgcn_global_load_lds(gp, lp, 4, 0, 0);
                        else                     __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
                    }
                }
            } else if (MODE == 4) {
                if (g < 6)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(from) +
                                                                           (unsigned)((g * 256 + tid) * 16)),
                        (__attribute__((address_space(3))) void *)(dst + g * 256 + wave_base), 16, 0, 0);
            } else if (MODE == 2) {
                if (g == 0) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) held[r] = from[r * 256 + tid];
                }
                if (g == 5) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) dst[r * 256 + tid] = held[r];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3] + held[0].x;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

// The alternative shape: ONE wavefront per SIMD carrying TWO 16-pair tiles (each A piece read once feeds 12 MFMAs on
// two independent accumulators), one 4-wavefront workgroup per CU: pieces, ds_reads and barriers per MFMA all halved.
// DMA = 0: no weight movement, 1: 6 pieces per wavefront and chunk (the same 24-KiB chunk now serves 8 tiles).
template <int DMA>
__global__ __launch_bounds__(256, 1) void k2(const float4 *__restrict__ table, int table_rows, float *out,
                                            unsigned long long *cyc, int chunks)
{
    __shared__ float4 stage[2][kChunkRows + 256];
    __shared__ float4 pad_to_one_workgroup_per_cu[4096];         // 64 KiB more: two workgroups do not fit a CU
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_base = tid & ~63;
    bf16x8 b0, b1;
    for (int i = 0; i < 8; ++i) { b0[i] = (__bf16)(0.5f + i * 1e-2f + lane * 1e-3f); b1[i] = (__bf16)(0.25f + i * 2e-2f + lane * 1e-3f); }
    for (int i = tid; i < 2 * (kChunkRows + 256); i += 256) (&stage[0][0])[i] = make_float4(1.f, 2.f, 3.f, (float)i);
    if (tid == 0) pad_to_one_workgroup_per_cu[blockIdx.x & 4095] = make_float4(0, 0, 0, 0);
    __syncthreads();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    unsigned row = (blockIdx.x * 977u) % (unsigned)(table_rows - kChunkRows);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int c = 0; c < chunks; ++c) {
        float4 *dst = stage[(c + 1) & 1];
        const float4 *src = stage[c & 1];
        const float4 *from = table + row;
        row += kChunkRows;
        if (row >= (unsigned)(table_rows - kChunkRows)) row -= (unsigned)(table_rows - kChunkRows);
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            const float4 w0 = src[(g * 192 + lane) % kChunkRows], w1 = src[(g * 192 + 64 + lane) % kChunkRows],
                         w2 = src[(g * 192 + 128 + lane) % kChunkRows];
            if (g == kGroups - 1) __syncthreads();
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2), b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2), b1, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (DMA && g == 0) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(from) +
                                                                           (unsigned)((r * 256 + tid) * 16)),
                        (__attribute__((address_space(3))) void *)(dst + r * 256 + wave_base), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#define TWO(w) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), b0, acc0, 0, 0, 0); \
               acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), b1, acc1, 0, 0, 0);
            TWO(w0) TWO(w1) TWO(w1) TWO(w0) TWO(w0)
#undef TWO
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = acc0[0] + acc0[1] + acc1[2] + acc1[3] + pad_to_one_workgroup_per_cu[tid].x;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int DMA> static double run2(const float4 *table, int rows, float *out, unsigned long long *cyc, int chunks)
{
    const int blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k2<DMA>, dim3(blocks), dim3(256), 0, 0, table, rows, out, cyc, chunks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c((size_t)blocks * 4);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    return (double)c[c.size() / 2] / chunks;
}

// Where the two-wavefront loop itself loses its 0.67: the kernel's shape (mode 1) with one ingredient changed at a time.
//   V = 1 no workgroup barrier, 2 no ds_reads (pieces stay in registers), 3 two accumulator chains per wavefront
//   (alternating MFMAs, summed per chunk), 4 = 1 + 2 + 3, 5 = reads for group g + 1 issued behind group g's first MFMA
//   (NOT comparable with the kernel's ring: here the next chunk's first reads come from the buffer the DMA is filling,
//   so the compiler puts a vmcnt(0) right behind the burst; kept as a record of that trap)
template <int V>
__global__ __launch_bounds__(256, 2) void k3(const float4 *__restrict__ table, int table_rows, float *out,
                                            unsigned long long *cyc, int chunks)
{
    __shared__ float4 stage[2][kChunkRows + 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_base = tid & ~63;
    bf16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)(0.5f + i * 1e-2f + lane * 1e-3f);
    for (int i = tid; i < 2 * (kChunkRows + 256); i += 256) (&stage[0][0])[i] = make_float4(1.f, 2.f, 3.f, (float)i);
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    float4 w0 = stage[0][lane], w1 = stage[0][64 + lane], w2 = stage[0][128 + lane];
    float4 n0 = w0, n1 = w1, n2 = w2;
    unsigned row = (blockIdx.x * 977u) % (unsigned)(table_rows - kChunkRows);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int c = 0; c < chunks; ++c) {
        float4 *dst = stage[(c + 1) & 1];
        const float4 *src = stage[c & 1];
        const float4 *from = table + row;
        row += kChunkRows;
        if (row >= (unsigned)(table_rows - kChunkRows)) row -= (unsigned)(table_rows - kChunkRows);
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            if (V == 5) { w0 = n0; w1 = n1; w2 = n2; }
            else if (V != 2 && V != 4) {
                w0 = src[(g * 192 + lane) % kChunkRows]; w1 = src[(g * 192 + 64 + lane) % kChunkRows];
                w2 = src[(g * 192 + 128 + lane) % kChunkRows];
            }
            if (g == kGroups - 1 && V != 1 && V != 4) __syncthreads();
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2), b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (V == 5) {
                const float4 *nsrc = g + 1 < kGroups ? src : dst;
                const int ng = (g + 1) % kGroups;
                n0 = nsrc[(ng * 192 + lane) % kChunkRows]; n1 = nsrc[(ng * 192 + 64 + lane) % kChunkRows];
                n2 = nsrc[(ng * 192 + 128 + lane) % kChunkRows];
            }
            if (g == 0) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(from) +
                                                                           (unsigned)((r * 256 + tid) * 16)),
                        (__attribute__((address_space(3))) void *)(dst + r * 256 + wave_base), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (V == 3 || V == 4) {
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), b, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc2, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), b, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc2[2] + acc2[3] + n0.x;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int V> static double run3(const float4 *table, int rows, float *out, unsigned long long *cyc, int chunks)
{
    const int blocks = 512;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k3<V>, dim3(blocks), dim3(256), 0, 0, table, rows, out, cyc, chunks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c((size_t)blocks * 4);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    return (double)c[c.size() / 2] / chunks;
}

template <int MODE> static double run(const float4 *table, int rows, float *out, unsigned long long *cyc, int chunks)
{
    const int blocks = 512;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, table, rows, out, cyc, chunks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c((size_t)blocks * 4);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    return (double)c[c.size() / 2] / chunks;        // 100 MHz ticks per chunk
}

int main()
{
    const int rows = 150000, chunks = 4000;         // 2.4 MB table: L2-resident, like the packed weights
    float4 *table; float *out; unsigned long long *cyc;
    hipMalloc(&table, (size_t)rows * 16); hipMemset(table, 0, (size_t)rows * 16);
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 4 * 8);
    const char *what[6] = {"no weight movement", "6 x global_load_lds_dwordx4 in one burst", "6 x global_load_dwordx4 + 6 x ds_write_b128",
                           "6 x global_load_lds_dword (a quarter of the bytes)", "6 x global_load_lds_dwordx4, one per group",
                           "3 x global_load_lds_dwordx4 (half the pieces)"};
    double t[6];
    t[0] = run<0>(table, rows, out, cyc, chunks); t[1] = run<1>(table, rows, out, cyc, chunks);
    t[2] = run<2>(table, rows, out, cyc, chunks); t[3] = run<3>(table, rows, out, cyc, chunks);
    t[4] = run<4>(table, rows, out, cyc, chunks); t[5] = run<5>(table, rows, out, cyc, chunks);
    printf("# 48 MFMAs (768 pipe cycles) + 24 ds_read_b128 + 1 barrier per chunk and wavefront; two wavefronts per SIMD\n");
    printf("# time per chunk in 10-ns ticks of s_memtime (x ~24 = shader cycles at 2.4 GHz); ideal = 2 x 768 cycles = 64 ticks\n");
    for (int m = 0; m < 6; ++m)
        printf("mode %d  %-55s %7.2f ticks per chunk  (+%.2f = %+.0f cycles per piece)\n", m, what[m], t[m], t[m] - t[0],
               (t[m] - t[0]) * 24.0 / (m == 5 ? 3 : 6) / 2.0);
    printf("# the kernel's shape (two wavefronts per SIMD, 6 pieces per chunk) with one ingredient changed at a time\n");
    printf("as mode 1 (reads right before use)          %7.2f\n", run3<0>(table, rows, out, cyc, chunks));
    printf("no workgroup barrier                        %7.2f\n", run3<1>(table, rows, out, cyc, chunks));
    printf("no ds_reads                                 %7.2f\n", run3<2>(table, rows, out, cyc, chunks));
    printf("two accumulator chains per wavefront        %7.2f\n", run3<3>(table, rows, out, cyc, chunks));
    printf("all three                                   %7.2f\n", run3<4>(table, rows, out, cyc, chunks));
    printf("reads one group ahead (the kernel's ring)   %7.2f\n", run3<5>(table, rows, out, cyc, chunks));
    const double u0 = run2<0>(table, rows, out, cyc, chunks), u1 = run2<1>(table, rows, out, cyc, chunks);
    printf("# two tiles per wavefront, one wavefront per SIMD: the same 96 MFMAs per SIMD and chunk (ideal 1536)\n");
    printf("two-tile, no weight movement                                  %7.2f ticks per chunk\n", u0);
    printf("two-tile, 6 x global_load_lds_dwordx4 in one burst            %7.2f ticks per chunk  (+%.2f)\n", u1, u1 - u0);
    return 0;
}
