// world_attn.hip -- AttentionWorld one-step world model for E scenes in one launch (gfx950).
//
// Replaces crowd_nav/policy/world_model.py:54-106 (AttentionWorld.forward: per pedestrian mlp1 4 -> 150 -> 100, mean over
// the scene as global state, attention 200 -> 100 -> 100 -> 1 with the un-stabilised masked softmax exp(s) (s != 0) / sum,
// mlp2 100 -> 100 -> 50 pooled with those weights, then PER PEDESTRIAN mlp3 on [own state(4), pooled(50)] 54 -> 150 ->
// 100 -> 100 -> 2; no output non-linearity, :104-105) with the call convention of model_crowd_sim.py:401-407.
//
// The value network's scheme (sarl_value.hip, mfma_chain.hpp): one wavefront carries 16 scenes through the network in
// registers, pedestrians one after the other, weights staged once per 4-wavefront workgroup through LDS.  The same two
// algebraic savings: the global-state half of attention.0 is the accumulator init of the per-pedestrian half; mlp2's
// last layer is linear and the weights sum to one, so it runs once per scene on the weighted hidden activations.  And
// one more: mlp3.0 sees [state_i, pooled] -- its pooled half (50 of 54 columns) is the same for every pedestrian of
// the scene, so it is computed once and only the 4 state columns (one MFMA k-step per output tile) run per pedestrian.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "mfma_chain.hpp"

namespace mcn {

constexpr int W4 = 1, W150 = 10, W100 = 7, W50 = 4, W2 = 1;

struct AttnWorldParams {
    const float4 *w_m1a, *b_m1a, *w_m1b, *b_m1b, *w_m2a, *b_m2a, *w_m2b, *b_m2b;
    const float4 *w_ata, *b_ata, *w_atg, *w_atb, *b_atb, *w_atc, *b_atc;
    const float4 *w_m3p, *b_m3p, *w_m3s, *w_m3b, *b_m3b, *w_m3c, *b_m3c, *w_m3d, *b_m3d;
    const double *hpos, *hvel;       // [E*N][2]
    const int32_t *hcount;           // [E] or NULL
    float4 *workspace;               // [wavefronts][N][W100][64]
    double *out_vel;                 // [E*N][2]
    int E, N;
};

__global__ __launch_bounds__(kStageThreads, MCN_STAGE_WAVES >= 8 ? 1 : 2) void attn_world_kernel(const AttnWorldParams p)
{
    __shared__ float4 s_stage[2 * (kStageFloat4 + kStageBias)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a fresh opaque copy of the thread id per section: the per-lane staging addresses of a section's layers are
    // re-derived there instead of being hoisted out of the pedestrian loops as dozens of live 64-bit values
    auto stage_here = [&]() { int t = tid; asm volatile("" : "+v"(t)); return WeightStage{s_stage, t}; };
    constexpr int kWaves = kStageThreads / 64;
    const long gw = (long)blockIdx.x * kWaves + wave;
    const int j = lane & 15, q = lane >> 4;
    const int N = p.N;
    long e = gw * 16 + j;
    const bool valid = e < p.E;
    if (!valid) e = p.E - 1;                      // no early exit: the staging barriers are collective
    int ne = N;
    if (p.hcount) { ne = p.hcount[e]; ne = ne < 1 ? 1 : (ne > N ? N : ne); }
    float4 *ws = p.workspace + gw * (long)N * W100 * 64;

    // the pedestrian's own state as the one input tile, packed "q first": feature c in register 0 of lane group c
    auto state_tile = [&](int i) {
        const double2 ps = reinterpret_cast<const double2 *>(p.hpos)[e * N + i];
        const double2 vl = reinterpret_cast<const double2 *>(p.hvel)[e * N + i];
        const float f = q == 0 ? (float)ps.x : (q == 1 ? (float)ps.y : (q == 2 ? (float)vl.x : (float)vl.y));
        return (f32x4){f, 0.0f, 0.0f, 0.0f};
    };

    // ---- pass 1: mlp1 per pedestrian, global-state sum ----
    f32x4 gsum[W100];
#pragma unroll
    for (int t = 0; t < W100; ++t) gsum[t] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < N; ++i) {
        const WeightStage S = stage_here();
        f32x4 x[W4];
        x[0] = state_tile(i);
        f32x4 h1[W150];
        dense_staged<W4, W150, true, false, 1>(x, nullptr, h1, p.w_m1a, p.b_m1a, S, lane);
        f32x4 h2[W100];
        dense_staged<W150, W100, true, false, 2>(h1, nullptr, h2, p.w_m1b, p.b_m1b, S, lane);
#pragma unroll
        for (int t = 0; t < W100; ++t) {
            ws[(i * W100 + t) * 64 + lane] = make_float4(h2[t][0], h2[t][1], h2[t][2], h2[t][3]);
            if (i < ne) gsum[t] += h2[t];
        }
    }
    const float fn = (float)ne;
#pragma unroll
    for (int t = 0; t < W100; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) gsum[t][r] = gsum[t][r] / fn;
    f32x4 gat[W100];
    { const WeightStage S = stage_here();
      dense_staged<W100, W100, false, false, 1>(gsum, nullptr, gat, p.w_atg, p.b_ata, S, lane); }

    // ---- pass 2: attention score, mlp2's first layer, weighted sum ----
    f32x4 racc[W100];
#pragma unroll
    for (int t = 0; t < W100; ++t) racc[t] = (f32x4){0, 0, 0, 0};
    float denom = 0.0f;
    for (int i = 0; i < N; ++i) {
        const WeightStage S = stage_here();
        f32x4 h2[W100];
#pragma unroll
        for (int t = 0; t < W100; ++t) {
            const float4 v = ws[(i * W100 + t) * 64 + lane];
            h2[t] = (f32x4){v.x, v.y, v.z, v.w};
        }
        f32x4 a1[W100];
        dense_staged<W100, W100, true, true, 1>(h2, gat, a1, p.w_ata, nullptr, S, lane);
        f32x4 a2[W100];
        dense_staged<W100, W100, true, false, 1>(a1, nullptr, a2, p.w_atb, p.b_atb, S, lane);
        f32x4 sc[1];
        dense_staged<W100, 1, false, false, 1>(a2, nullptr, sc, p.w_atc, p.b_atc, S, lane);
        const float s = __shfl(sc[0][0], j);
        const float es = (s != 0.0f && i < ne) ? expf(s) : 0.0f;       // exp(s) * (s != 0), world_model.py:91
        denom += es;
        f32x4 m1[W100];
        dense_staged<W100, W100, true, false, 1>(h2, nullptr, m1, p.w_m2a, p.b_m2a, S, lane);
#pragma unroll
        for (int t = 0; t < W100; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) racc[t][r] = i < ne ? racc[t][r] + es * m1[t][r] : racc[t][r];
    }
#pragma unroll
    for (int t = 0; t < W100; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) racc[t][r] = racc[t][r] / denom;
    f32x4 pooled[W50];
    // the pooled half of mlp3.0 (+ its bias): the same for every pedestrian of the scene
    f32x4 m3pool[W150];
    { const WeightStage S = stage_here();
      dense_staged<W100, W50, false, false, 1>(racc, nullptr, pooled, p.w_m2b, p.b_m2b, S, lane);
      dense_staged<W50, W150, false, false, 1>(pooled, nullptr, m3pool, p.w_m3p, p.b_m3p, S, lane); }

    // ---- pass 3: mlp3 per pedestrian ----
    for (int i = 0; i < N; ++i) {
        const WeightStage S = stage_here();
        f32x4 x[W4];
        x[0] = state_tile(i);
        f32x4 v1[W150];
        dense_staged<W4, W150, true, true, 1>(x, m3pool, v1, p.w_m3s, nullptr, S, lane);
        f32x4 v2[W100];
        dense_staged<W150, W100, true, false, 2>(v1, nullptr, v2, p.w_m3b, p.b_m3b, S, lane);
        f32x4 v3[W100];
        dense_staged<W100, W100, true, false, 1>(v2, nullptr, v3, p.w_m3c, p.b_m3c, S, lane);
        f32x4 o[W2];
        dense_staged<W100, W2, false, false, 1>(v3, nullptr, o, p.w_m3d, p.b_m3d, S, lane);
        // natural output order: (vx, vy) are rows 0 and 1 = registers 0 and 1 of lane group 0
        if (valid && q == 0 && i < ne)
            reinterpret_cast<double2 *>(p.out_vel)[e * N + i] = make_double2((double)o[0][0], (double)o[0][1]);
    }
}

long attn_world_workspace_float4s(int E, int N)
{
    constexpr int kWaves = kStageThreads / 64;
    const long waves = ((long)E + 15) / 16;
    const long wpad = (waves + kWaves - 1) / kWaves * kWaves;
    return wpad * (long)N * W100 * 64;
}

int launch_attn_world(const mcn_attn_world_net *net, const double *hpos, const double *hvel, const int32_t *hcount,
                      void *workspace, double *out_vel, int E, int N, hipStream_t stream)
{
    AttnWorldParams p;
    const float4 *const *src = reinterpret_cast<const float4 *const *>(net);
    const float4 **dst = reinterpret_cast<const float4 **>(&p);
    static_assert(sizeof(mcn_attn_world_net) == 24 * sizeof(void *), "fragment table must mirror the C struct");
    for (int k = 0; k < 24; ++k) dst[k] = src[k];
    p.hpos = hpos; p.hvel = hvel; p.hcount = hcount;
    p.workspace = reinterpret_cast<float4 *>(workspace); p.out_vel = out_vel; p.E = E; p.N = N;
    constexpr int kWaves = kStageThreads / 64;
    const long waves = ((long)E + 15) / 16;
    const int blocks = (int)((waves + kWaves - 1) / kWaves);
    hipLaunchKernelGGL(attn_world_kernel, dim3(blocks), dim3(kStageThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
}

}  // namespace mcn
