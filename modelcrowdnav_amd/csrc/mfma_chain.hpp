// mfma_chain.hpp -- register-chained float32 MFMA layers for gfx950.
//
// v_mfma_f32_16x16x4_f32 computes D[i][j] += A[i][k] B[k][j]; lane l holds A[l&15][l>>4], B[l>>4][l&15]
// and D[4(l>>4)+r][l&15] in accumulator register r.  With the batch item ("pair", pedestrian, ...) on j and
// the feature on i/k, register r of an output tile is, as it stands, the B operand of the next layer's
// k-step over features {r, 4+r, 8+r, 12+r}.  Layers therefore chain in registers; the weights are the
// only thing that moves, as A operands pre-permuted on the host (mcn_pack_linear) so that one coalesced
// 16-byte load per lane feeds four MFMAs.  float32 MFMA is an exact k-ordered fmaf chain.
#pragma once
#include <hip/hip_runtime.h>

namespace mcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[n] = act( bias[n] (+ init) + sum_t in[t] x W[n][t] ), two accumulators in flight per step so
// back-to-back dependent MFMAs (40-cycle latency vs 32-cycle issue) never stall the pipe.
template <int KT, int NT, bool RELU>
__device__ __forceinline__ void dense(const f32x4 (&in)[KT], f32x4 (&out)[NT], const float4 *__restrict__ wf,
                                      const float4 *__restrict__ bf, int lane)
{
#pragma unroll
    for (int n = 0; n < NT; n += 2) {
        constexpr int dummy = 0; (void)dummy;
        const bool two = (n + 1 < NT);
        f32x4 a0, a1 = {0, 0, 0, 0};
        { const float4 b = bf[n * 64 + lane]; a0 = (f32x4){b.x, b.y, b.z, b.w}; }
        if (two) { const float4 b = bf[(n + 1) * 64 + lane]; a1 = (f32x4){b.x, b.y, b.z, b.w}; }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float4 w0 = wf[(n * KT + t) * 64 + lane];
            float4 w1 = make_float4(0, 0, 0, 0);
            if (two) w1 = wf[((n + 1) * KT + t) * 64 + lane];
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, in[t][0], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, in[t][0], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, in[t][1], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, in[t][1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, in[t][2], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, in[t][2], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, in[t][3], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, in[t][3], a1, 0, 0, 0);
        }
        if (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { a0[r] = fmaxf(a0[r], 0.0f); a1[r] = fmaxf(a1[r], 0.0f); }
        }
        out[n] = a0;
        if (two) out[n + 1] = a1;
    }
}

// same, but accumulates onto out[] (used to fold the global-state half of attention layer 0)
template <int KT, int NT, bool RELU>
__device__ __forceinline__ void dense_acc(const f32x4 (&in)[KT], const f32x4 (&init)[NT], f32x4 (&out)[NT],
                                          const float4 *__restrict__ wf, int lane)
{
#pragma unroll
    for (int n = 0; n < NT; n += 2) {
        const bool two = (n + 1 < NT);
        f32x4 a0 = init[n], a1 = {0, 0, 0, 0};
        if (two) a1 = init[n + 1];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float4 w0 = wf[(n * KT + t) * 64 + lane];
            float4 w1 = make_float4(0, 0, 0, 0);
            if (two) w1 = wf[((n + 1) * KT + t) * 64 + lane];
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, in[t][0], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, in[t][0], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, in[t][1], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, in[t][1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, in[t][2], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, in[t][2], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, in[t][3], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, in[t][3], a1, 0, 0, 0);
        }
        if (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { a0[r] = fmaxf(a0[r], 0.0f); a1[r] = fmaxf(a1[r], 0.0f); }
        }
        out[n] = a0;
        if (two) out[n + 1] = a1;
    }
}


}  // namespace mcn
