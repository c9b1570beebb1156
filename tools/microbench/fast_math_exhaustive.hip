// fast_math_exhaustive.hip -- which short float32 sqrt / reciprocal / division sequences return the CORRECTLY ROUNDED
// result on gfx950, and on which exponent ranges?   (shorten the ORCA dependent chain.)
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/microbench/fast_math_exhaustive.hip -o tools/microbench/fast_math_exhaustive
//
// hipcc expands `a / b` and `sqrtf(x)` into the IEEE sequences (v_div_scale / v_div_fmas / v_div_fixup, 11 instructions;
// scale + v_sqrt + +-1 ulp fix-up + class test, ~15): every ORCA step of the latency-bound env kernels walks through
// ~6 quotients and ~4 roots of them one after the other.  The single-operand candidates below are compared with the
// IEEE result for ALL 2^32 bit patterns (a proof by exhaustion for this hardware); the two-operand division for
// 2^34 random mantissa pairs per exponent offset plus the corner mantissas.  Output: mismatches per candidate and
// biased exponent of the operand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned bits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float fbits(unsigned u) { return __builtin_bit_cast(float, u); }

// ---- reciprocal candidates
__device__ __forceinline__ float rcp3(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float rcp5(float x)
{
    const float r1 = rcp3(x);
    const float e = __builtin_fmaf(-x, r1, 1.0f);
    return __builtin_fmaf(e, r1, r1);
}
// ---- sqrt candidates
__device__ __forceinline__ float sqrt_fix(float x)            // the compiler's core without scaling / class test: 9
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = fbits(bits(s) - 1u), hi = fbits(bits(s) + 1u);
    const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
    float r = rl <= 0.0f ? lo : s;
    r = rh > 0.0f ? hi : r;
    return r;
}
__device__ __forceinline__ float sqrt_rsq5(float x)           // v_rsq, 2 mul, 2 fma
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_rsq8(float x)           // Markstein: one coupled refinement of g and h first
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_s5(float x)             // v_sqrt, v_rcp, mul, 2 fma
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(s);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
// ---- division candidates
__device__ __forceinline__ float div6(float a, float b)
{
    const float r1 = rcp3(b);
    const float q0 = a * r1;
    const float rem = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(rem, r1, q0);
}
__device__ __forceinline__ float div8(float a, float b)       // the IEEE expansion without v_div_scale / v_div_fixup
{
    const float r1 = rcp3(b);
    const float q0 = a * r1;
    const float rem0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(rem0, r1, q0);
    const float rem1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(rem1, r1, q1);
}

enum { C_RCP3, C_RCP5, C_SQRT_FIX, C_SQRT_RSQ5, C_SQRT_RSQ8, C_SQRT_S5, N_SINGLE };

// same bits, or both NaN
__device__ __forceinline__ bool same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

__global__ void single_kernel(unsigned long long *miss /* [N_SINGLE][512] */, unsigned base)
{
    const unsigned u = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = fbits(u);
    const int bucket = u >> 23;                     // sign + biased exponent
    const float want_r = 1.0f / x, want_s = sqrtf(x);
    const float got[N_SINGLE] = {rcp3(x), rcp5(x), sqrt_fix(x), sqrt_rsq5(x), sqrt_rsq8(x), sqrt_s5(x)};
#pragma unroll
    for (int c = 0; c < N_SINGLE; ++c) {
        const bool ok = same(got[c], c < C_SQRT_FIX ? want_r : want_s);
        if (!ok) atomicAdd(&miss[c * 512 + bucket], 1ull);
    }
}

__device__ __forceinline__ unsigned long long mix(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// a = 2^ea * (1 + ma / 2^23), b = 2^eb * (1 + mb / 2^23): random or corner mantissas, given biased exponents
__global__ void div_kernel(unsigned long long *miss /* [2] */, unsigned long long seed, int ea, int eb, int corner)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long r = mix(seed ^ mix(i));
    unsigned ma = (unsigned)(r & 0x7fffff), mb = (unsigned)((r >> 23) & 0x7fffff);
    if (corner == 1) mb = 0x7fffff - (unsigned)(i & 15);                  // divisor mantissa (nearly) all ones
    if (corner == 2) { mb = (unsigned)(i & 0x3ff); ma = 0x7fffff - (unsigned)((i >> 10) & 0x3ff); }
    if (corner == 3) { mb = 0x7fffff - (unsigned)(i & 0x3ff); ma = (unsigned)((i >> 10) & 0x3ff); }
    const unsigned sa = (unsigned)((r >> 46) & 1) << 31, sb = (unsigned)((r >> 47) & 1) << 31;
    const float a = fbits(sa | ((unsigned)ea << 23) | ma), b = fbits(sb | ((unsigned)eb << 23) | mb);
    const float want = a / b;
    if (!same(div6(a, b), want)) atomicAdd(&miss[0], 1ull);
    if (!same(div8(a, b), want)) atomicAdd(&miss[1], 1ull);
}

int main(int argc, char **argv)
{
    const char *names[N_SINGLE] = {"rcp3  (v_rcp + 2 fma)", "rcp5  (v_rcp + 4 fma)", "sqrt_fix (v_sqrt + +-1ulp fix, 9)",
                                   "sqrt_rsq5 (v_rsq, 2 mul, 2 fma)", "sqrt_rsq8 (Markstein, 8)", "sqrt_s5 (v_sqrt, v_rcp, mul, 2 fma)"};
    unsigned long long *d_miss;
    CHECK(hipMalloc(&d_miss, sizeof(unsigned long long) * N_SINGLE * 512));
    CHECK(hipMemset(d_miss, 0, sizeof(unsigned long long) * N_SINGLE * 512));
    for (unsigned chunk = 0; chunk < 256; ++chunk)                        // 256 x 2^24 = all 2^32 bit patterns
        hipLaunchKernelGGL(single_kernel, dim3(1 << 16), dim3(256), 0, 0, d_miss, chunk << 24);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> miss(N_SINGLE * 512);
    CHECK(hipMemcpy(miss.data(), d_miss, miss.size() * 8, hipMemcpyDeviceToHost));
    printf("single-operand candidates against the IEEE expansion, ALL 2^32 bit patterns; mismatches by biased exponent\n");
    printf("(positive operands: exponent 0 = zero / denormals, 255 = inf / NaN; a run 'e1..e2: n' lists buckets with n > 0)\n");
    for (int c = 0; c < N_SINGLE; ++c) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            if (sgn == 1 && c >= C_SQRT_FIX) continue;                    // sqrt of negatives: NaN either way
            unsigned long long total = 0;
            int first_ok = -1, last_ok = -1, best_lo = -1, best_hi = -1;
            for (int e = 0; e < 256; ++e) {
                const unsigned long long m = miss[c * 512 + sgn * 256 + e];
                total += m;
                if (m == 0) { if (first_ok < 0) first_ok = e; last_ok = e; if (last_ok - first_ok > best_hi - best_lo) { best_lo = first_ok; best_hi = last_ok; } }
                else first_ok = -1;
            }
            printf("%-38s %s: %llu mismatches; exact for biased exponents %d..%d", names[c], sgn ? "x < 0" : "x > 0", total, best_lo, best_hi);
            printf("; failing:");
            int shown = 0;
            for (int e = 0; e < 256 && shown < 12; ++e)
                if (miss[c * 512 + sgn * 256 + e]) { printf(" %d:%llu", e, miss[c * 512 + sgn * 256 + e]); ++shown; }
            printf("\n");
        }
    }
    // two-operand division: 2^30 random mantissa pairs for each of several exponent offsets + corner mantissas
    unsigned long long *d_dm;
    CHECK(hipMalloc(&d_dm, 16));
    const int offs[][2] = {{127, 127}, {127, 126}, {120, 130}, {140, 110}, {100, 150}, {127, 110}, {90, 127}, {160, 127},
                           {60, 127}, {40, 127}, {30, 127}, {24, 127}, {200, 127}, {127, 30}, {127, 220}};
    printf("division a / b: 2^30 random mantissa pairs per exponent pair (+ corner mantissas: divisor all ones etc.)\n");
    for (auto &o : offs) {
        unsigned long long tot[2] = {0, 0};
        for (int corner = 0; corner < 4; ++corner) {
            CHECK(hipMemset(d_dm, 0, 16));
            const int blocks = corner == 0 ? (1 << 22) : (1 << 12);
            hipLaunchKernelGGL(div_kernel, dim3(blocks), dim3(256), 0, 0, d_dm, 0x1234567ull + o[0] * 1000 + o[1], o[0], o[1], corner);
            CHECK(hipDeviceSynchronize());
            unsigned long long m[2];
            CHECK(hipMemcpy(m, d_dm, 16, hipMemcpyDeviceToHost));
            tot[0] += m[0]; tot[1] += m[1];
        }
        printf("  biased exponents a %3d  b %3d:  div6 (v_rcp + 5) %llu mismatches   div8 (IEEE without scale / fixup) %llu\n",
               o[0], o[1], tot[0], tot[1]);
    }
    return 0;
}
