// split_bf16.hip -- a float32 layer chain on the bf16 matrix pipe: x = hi + mid + lo (three bf16 pieces, exact to
// 2^-24), six of the nine cross products, float32 accumulation -- against the same chain on v_mfma_f32_16x16x4_f32.
// Build: hipcc -O3 --offload-arch=gfx950 -w tools/microbench/split_bf16.hip -o tools/microbench/split_bf16
//
// Workload: one wavefront carries 16 columns ("pairs") through LAYERS layers y = relu(W x), W 112 x 112 (7 tiles of 16
// features, as the 100-wide layers of the value network after padding), activations chained in registers exactly as
// mfma_chain.hpp does (an output tile's four accumulators are the next layer's B operand), weight fragments read from
// LDS.  Reports shader cycles per layer (one and two wavefronts per SIMD) and the error of either path against a
// float64 evaluation of the same float32 weights and inputs.
//
// Layouts.  16x16x4 f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[4 (l >> 4) + r][l & 15] in register r.
// 16x16x32 bf16: a lane holds 8 consecutive k of row i = l & 15 (A) / column j = l & 15 (B), k = 8 (l >> 4) + s; D as
// above.  With k slot 8 q + s carrying feature 32 m + 4 q + s (s < 4) or 32 m + 16 + 4 q + s - 4 (s >= 4), the B operand
// of input tile pair m is the lane's own accumulators of output tiles 2 m and 2 m + 1 of the previous layer: the
// register chaining carries over, the weights are permuted on the host to match.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int F = 112, NTILE = 7, NPAIR = 4, LAYERS = 8;

__device__ __forceinline__ float relu1(float x) { const int b = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, b > 0 ? b : 0); }

// ---- float32 MFMA path: wf[n][t][lane] float4 = A[i][k-step r] of (output tile n, input tile t) ----
__global__ void chain_f32(const float4 *wf, const float *x0, float *out, unsigned long long *cyc)
{
    extern __shared__ float4 lds[];
    for (int i = threadIdx.x; i < NTILE * NTILE * 64; i += blockDim.x) lds[i] = wf[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const long col0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
    f32x4 act[NTILE];
    for (int t = 0; t < NTILE; ++t)
        for (int r = 0; r < 4; ++r) act[t][r] = x0[(col0 + j) * F + 16 * t + 4 * q + r];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int l = 0; l < LAYERS; ++l) {
        int ln = lane;
        asm volatile("" : "+v"(ln));          // the weight reads belong to the layer (every layer has its own in real use)
        f32x4 nxt[NTILE];
#pragma unroll
        for (int n = 0; n < NTILE; ++n) {
            f32x4 a = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                const float4 w = lds[(n * NTILE + t) * 64 + ln];
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, act[t][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, act[t][1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, act[t][2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, act[t][3], a, 0, 0, 0);
            }
            for (int r = 0; r < 4; ++r) nxt[n][r] = relu1(a[r]);
        }
#pragma unroll
        for (int n = 0; n < NTILE; ++n) act[n] = nxt[n];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < NTILE; ++t)
        for (int r = 0; r < 4; ++r) out[(col0 + j) * F + 16 * t + 4 * q + r] = act[t][r];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- split path: wb[piece][n][m][lane] bf16x8 = A[i][8 q + s] of (output tile n, input tile pair m) ----
struct Split { bf16x8 hi, mid, lo; };
__device__ __forceinline__ Split split8(const f32x4 a, const f32x4 b)
{
    Split s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? a[i] : b[i - 4];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        s.hi[i] = h; s.mid[i] = m; s.lo[i] = (__bf16)r2;
    }
    return s;
}

template <int TERMS>
__global__ void chain_split(const bf16x8 *wb, const float *x0, float *out, unsigned long long *cyc)
{
    extern __shared__ float4 lds[];
    bf16x8 *lw = reinterpret_cast<bf16x8 *>(lds);
    for (int i = threadIdx.x; i < 3 * NTILE * NPAIR * 64; i += blockDim.x) lw[i] = wb[i];
    __syncthreads();
    const bf16x8 *Wh = lw, *Wm = lw + NTILE * NPAIR * 64, *Wl = lw + 2 * NTILE * NPAIR * 64;
    const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const long col0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
    f32x4 act[2 * NPAIR];
    for (int t = 0; t < 2 * NPAIR; ++t)
        for (int r = 0; r < 4; ++r) act[t][r] = t < NTILE ? x0[(col0 + j) * F + 16 * t + 4 * q + r] : 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int l = 0; l < LAYERS; ++l) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        Split b[NPAIR];
#pragma unroll
        for (int m = 0; m < NPAIR; ++m) b[m] = split8(act[2 * m], act[2 * m + 1]);
        f32x4 nxt[2 * NPAIR];
        nxt[2 * NPAIR - 1] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < NTILE; ++n) {
            f32x4 a = {0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < NPAIR; ++m) {
                const bf16x8 ah = Wh[(n * NPAIR + m) * 64 + ln], am = Wm[(n * NPAIR + m) * 64 + ln];
                // smallest terms first
                if (TERMS >= 6) {
                    const bf16x8 al = Wl[(n * NPAIR + m) * 64 + ln];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b[m].hi, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b[m].lo, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, b[m].mid, a, 0, 0, 0);
                }
                if (TERMS >= 3) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, b[m].hi, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b[m].mid, a, 0, 0, 0);
                }
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b[m].hi, a, 0, 0, 0);
            }
            for (int r = 0; r < 4; ++r) nxt[n][r] = relu1(a[r]);
        }
#pragma unroll
        for (int n = 0; n < 2 * NPAIR; ++n) act[n] = nxt[n];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < NTILE; ++t)
        for (int r = 0; r < 4; ++r) out[(col0 + j) * F + 16 * t + 4 * q + r] = act[t][r];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static unsigned short bf16_rne(float x)
{
    unsigned int u; std::memcpy(&u, &x, 4);
    const unsigned int lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (unsigned short)(u >> 16);
}
static float bf16_to_f(unsigned short h) { unsigned int u = (unsigned int)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

int main()
{
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> W(F * F);
    for (auto &w : W) w = nd(rng) * 1.4f / std::sqrt((float)F);          // keeps the activations' scale through ReLU layers
    const int blocks = 256;
    for (int waves : {4, 8}) {
        const long cols = (long)blocks * waves * 16;
        std::vector<float> X(cols * F);
        for (auto &x : X) x = nd(rng);
        // float64 reference
        std::vector<double> ref(X.begin(), X.end()), tmp(cols * F);
        const long chk = std::min<long>(cols, 256);                        // columns checked
        for (int l = 0; l < LAYERS; ++l) {
            for (long c = 0; c < chk; ++c)
                for (int o = 0; o < F; ++o) {
                    double s = 0;
                    for (int i = 0; i < F; ++i) s += (double)W[o * F + i] * ref[c * F + i];
                    tmp[c * F + o] = s > 0 ? s : 0;
                }
            for (long c = 0; c < chk * F; ++c) ref[c] = tmp[c];
        }
        // fragments
        std::vector<float> wf((size_t)NTILE * NTILE * 64 * 4);
        for (int n = 0; n < NTILE; ++n) for (int t = 0; t < NTILE; ++t) for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r)
            wf[(((size_t)n * NTILE + t) * 64 + l) * 4 + r] = W[(16 * n + (l & 15)) * F + 16 * t + 4 * (l >> 4) + r];
        std::vector<unsigned short> wb((size_t)3 * NTILE * NPAIR * 64 * 8);
        for (int n = 0; n < NTILE; ++n) for (int m = 0; m < NPAIR; ++m) for (int l = 0; l < 64; ++l) for (int s = 0; s < 8; ++s) {
            const int qq = l >> 4;
            const int feat = s < 4 ? 32 * m + 4 * qq + s : 32 * m + 16 + 4 * qq + (s - 4);
            const float w = feat < F ? W[(16 * n + (l & 15)) * F + feat] : 0.0f;
            const unsigned short h = bf16_rne(w); const float r1 = w - bf16_to_f(h);
            const unsigned short mm = bf16_rne(r1); const float r2 = r1 - bf16_to_f(mm);
            const size_t at = (((size_t)n * NPAIR + m) * 64 + l) * 8 + s, piece = (size_t)NTILE * NPAIR * 64 * 8;
            wb[at] = h; wb[piece + at] = mm; wb[2 * piece + at] = bf16_rne(r2);
        }
        float *dwf, *dx, *dout; unsigned short *dwb; unsigned long long *dcyc;
        hipMalloc(&dwf, wf.size() * 4); hipMalloc(&dwb, wb.size() * 2); hipMalloc(&dx, X.size() * 4);
        hipMalloc(&dout, X.size() * 4); hipMalloc(&dcyc, (size_t)blocks * waves * 8);
        hipMemcpy(dwf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dwb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
        auto report = [&](const char *what) {
            hipDeviceSynchronize();
            std::vector<float> o(chk * F); std::vector<unsigned long long> c((size_t)blocks * waves);
            hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
            std::sort(c.begin(), c.end());
            double emax = 0, vmax = 0;
            for (long i = 0; i < chk * F; ++i) { emax = std::max(emax, std::fabs((double)o[i] - ref[i])); vmax = std::max(vmax, std::fabs(ref[i])); }
            printf("%-44s %d wave(s)/SIMD: %8.0f cycles per layer per wave = %7.0f on the SIMD; max |err| %.2e (values up to %.1f)\n",
                   what, waves / 4, (double)c[c.size() / 2] / LAYERS, (double)c[c.size() / 2] / LAYERS / (waves / 4), emax, vmax);
        };
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(chain_f32, dim3(blocks), dim3(waves * 64), NTILE * NTILE * 64 * 16, 0, (const float4 *)dwf, dx, dout, dcyc);
        report("float32 MFMA (196 x 16x16x4 per layer)");
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(chain_split<6>, dim3(blocks), dim3(waves * 64), 3 * NTILE * NPAIR * 64 * 16, 0, (const bf16x8 *)dwb, dx, dout, dcyc);
        report("3 x bf16 split, 6 products (168 x 16x16x32)");
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(chain_split<3>, dim3(blocks), dim3(waves * 64), 3 * NTILE * NPAIR * 64 * 16, 0, (const bf16x8 *)dwb, dx, dout, dcyc);
        report("2 x bf16 split, 3 products (84 x 16x16x32)");
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(chain_split<1>, dim3(blocks), dim3(waves * 64), 3 * NTILE * NPAIR * 64 * 16, 0, (const bf16x8 *)dwb, dx, dout, dcyc);
        report("plain bf16 (28 x 16x16x32)");
        hipFree(dwf); hipFree(dwb); hipFree(dx); hipFree(dout); hipFree(dcyc);
    }
    return 0;
}
