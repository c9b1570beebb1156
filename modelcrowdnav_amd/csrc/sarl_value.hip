// sarl_value.hip -- SARL 81-action one-step look-ahead, fused, for gfx950 (MI355X).
//
// Replaces, per environment and candidate action (crowd_nav/policy/multi_human_rl.py:35-52):
//   propagate (cadrl.py:104-129), compute_reward (multi_human_rl.py:65-88), the [N,14] -> [N,13]
//   agent-centric transform (cadrl.py:217-252), ValueNetwork.forward (sarl.py:28-65: mlp1, mlp2,
//   global-state attention, un-stabilised masked softmax, pooling, mlp3) and
//   value = reward + gamma^(dt*v_pref) * V, followed by the strict-'>' argmax (:53-55).
//
// Structure.  A "pair" is one (environment, action).  One wavefront owns 16 consecutive pairs and
// carries them through the whole network with every activation in registers:
//
//   v_mfma_f32_16x16x4_f32 computes D[i][j] += A[i][k] B[k][j] with lane l holding A[l&15][l>>4],
//   B[l>>4][l&15] and D[4(l>>4)+r][l&15] in accumulator register r.  We put the PAIR on j (the lane's
//   low 4 bits) and the FEATURE on i/k.  A layer's output tile (16 features x 16 pairs) is then
//   already, register by register, a valid B operand of the next layer: register r of lane l holds
//   feature 4(l>>4)+r of pair l&15, i.e. B[k=l>>4][j] of the k-step that sums features {r, 4+r, 8+r,
//   12+r}.  So layers chain with no LDS round trip, no transposes and no barriers; only the weights
//   move, as A operands, pre-permuted on the host into exactly that order (one coalesced 16-B load per
//   lane feeds four MFMAs).  The N humans of a pair are processed one after the other by the same
//   lanes, which makes the reductions over humans (global-state mean, softmax denominator, pooled
//   feature) plain per-lane register arithmetic.
//
//   pass 1, per human: features -> mlp1 (13->150->100); park mlp1's output in the wavefront's workspace slot,
//                      accumulate the mean.  The grid is PERSISTENT (at most kSarlMaxBlocks workgroups, two per CU,
//                      each walking over its share of the 16-pair tiles), so the workspace is indexed by resident
//                      wavefront, not by tile: 2 048 slots x N x 7 KiB = 72 MB at N = 5 (143 MB at N = 10) whatever
//                      the batch -- it lives in the 256 MB Infinity Cache between the two passes instead of making a
//                      round trip through HBM (0.74 / 1.49 GB per look-ahead when it was indexed by tile).
//   pass 2, per human: attention (200->100->100->1) with the global half folded into the accumulator
//                      init, exp, mlp2's first layer (100->100), acc += e * relu(.).
//   tail:              mlp2's second, linear layer (100->50) once on acc / sum(e); mlp3 (56->150->100->100->1),
//                      value, store.
//
// Arithmetic: float32 MFMA is an exact k-ordered fmaf chain (no TF32), so values match the
// reference's float32 network to summation-order noise (~1e-6); rewards are float64 in the
// reference's operation order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/mcn.h"
#include "mfma_chain.hpp"
#include "fast_f32.hpp"

// x3 kernel structure switches (A/B): split a layer's output inside the layer, tile pair by tile pair (1), or as one block
// after it (0); keep the global half of attention.0 in registers (0) or park it in the workspace (1)
#ifndef MCN_X3_INLOOP_SPLIT
#define MCN_X3_INLOOP_SPLIT 0
#endif
#ifndef MCN_X3_M2A_EARLY
#define MCN_X3_M2A_EARLY 1
#endif
#ifndef MCN_X3_GAT_MEM
#define MCN_X3_GAT_MEM 0
#endif

namespace mcn {

constexpr bool kX3In = MCN_X3_INLOOP_SPLIT != 0, kX3GatMem = MCN_X3_GAT_MEM != 0, kM2aEarly = MCN_X3_M2A_EARLY != 0;
// a layer's output tiles -> the next layer's input blocks, after the layer (when it did not split them itself)
template <int NT, int NB>
__device__ __forceinline__ void split_after(const f32x4 (&t)[NT], X3 (&o)[NB])
{
    const f32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < NB; ++m) o[m] = split8(t[2 * m], 2 * m + 1 < NT ? t[2 * m + 1] : z);
}

bool tuning_sarl_x3();              // mcn_api.hip: mcn_tuning.sarl_x3 (-1 / 1: use the x3 fragments when given, 0: never)

// tiles of 16 features
constexpr int T13 = 1, T150 = 10, T100 = 7, T50 = 4, T56 = 5 /* pooled 4 + self 1 */, T1 = 1;

struct SarlFrags {
    // weight fragments [NT][KT][64 lanes] float4 and bias fragments [NT][64] float4, see pack order in
    // modelcrowdnav_amd/policy/sarl.py (_pack_plan, pack_value_network)
    const float4 *w_m1a, *b_m1a;   // 13 -> 150
    const float4 *w_m1b, *b_m1b;   // 150 -> 100
    const float4 *w_m2a, *b_m2a;   // 100 -> 100
    const float4 *w_m2b, *b_m2b;   // 100 -> 50
    const float4 *w_ata, *b_ata;   // attention layer 0, local half (100 -> 100), bias = attention.0.bias
    const float4 *w_atg;           // attention layer 0, global half (100 -> 100), no bias
    const float4 *w_atb, *b_atb;   // 100 -> 100
    const float4 *w_atc, *b_atc;   // 100 -> 1
    const float4 *w_m3a, *b_m3a;   // 56 -> 150   (K tiles: pooled x4, self x1)
    const float4 *w_m3b, *b_m3b;   // 150 -> 100
    const float4 *w_m3c, *b_m3c;   // 100 -> 100
    const float4 *w_m3d, *b_m3d;   // 100 -> 1
};

// bf16x3 weight fragments (mcn_pack_x3 of the float32 fragments above; biases stay the float32 ones): NULL table = float32 MFMA
struct SarlX3 {
    const float4 *w_m1a, *w_m1b, *w_m2a, *w_m2b, *w_ata, *w_atg, *w_atb, *w_atc, *w_m3a, *w_m3b, *w_m3c, *w_m3d;
};
// 32-feature input blocks of the x3 layers
constexpr int B13 = 1, B150 = 5, B100 = 4, B56 = 3;
constexpr int kWsRowsF32 = T100, kWsRowsX3 = B100 * 3;      // 16-byte workspace rows per human and lane: float32 tiles / x3 pieces

struct SarlParams {
    SarlFrags f;
    SarlX3 x;
    // state (same buffers as mcn_env_state)
    const double *rpos, *rvel, *rgoal, *rrad, *rvpref, *rtheta;   // [E][2] / [E]
    const double *hpos, *hvel, *hrad;                             // [E*N][2] / [E*N]
    const int32_t *hcount;                                        // [E] or NULL: humans the policy sees
    // query_env = true (multi_human_rl.py:37-38): the humans' next states and the rewards come from the env's own
    // one-step look-ahead instead of constant-velocity propagation + compute_reward; all three NULL otherwise
    const double *next_hpos, *next_hvel;                          // [E*N][2]
    const double *reward_in;                                      // [E*A]
    const double *actions;                                // [A][2]
    float4 *workspace;                                    // [resident waves][N][T100][64] float4
    long ngroups;                                         // 4-tile groups (one per workgroup pass) = ceil(tiles / 4)
    double *values;                                       // [E*A]
    float *attention;                                     // [E*A*N] or NULL
    int E, N, A, kinematics;
    double dt, gamma_pow;
};

__device__ __forceinline__ double norm2d(double x0, double x1) { return sqrt(fma(x1, x1, x0 * x0)); }

// Diagnostic build only (tools/sarl_phases.py): shader cycles each resident wavefront spends in each phase of a tile,
// summed over the tiles it walks.  [wavefront][16]: see PHASES in the tool.
#ifdef MCN_DIAG
__device__ unsigned long long g_sarl_phase[2048 * 16];
#define SARL_T0() unsigned long long sp_t_ = __builtin_amdgcn_s_memtime()
#define SARL_PHASE(k_)                                                                                          \
    do {                                                                                                        \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();                                             \
        const int w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                                     \
        if ((threadIdx.x & 63) == 0 && w_ < 2048) g_sarl_phase[w_ * 16 + (k_)] += n_ - sp_t_;                   \
        sp_t_ = n_;                                                                                             \
    } while (0)
int read_sarl_phases(void *dst, size_t bytes, int reset)
{
    if (bytes > sizeof(g_sarl_phase)) bytes = sizeof(g_sarl_phase);
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sarl_phase), bytes) != hipSuccess) return -1;
    if (reset) {
        void *p_ = nullptr;
        if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_sarl_phase)) != hipSuccess || hipMemset(p_, 0, sizeof(g_sarl_phase)) != hipSuccess) return -1;
    }
    return (int)(bytes / 8);
}
#else
#define SARL_T0()
#define SARL_PHASE(k_)
#endif

constexpr int kSarlWaves = kStageThreads / 64;     // 8 wavefronts share one LDS weight stage (2 per SIMD)

// MCN_SARL_FLOW = 1 (default): the layers of a tile (and of the next tile) form ONE weight stream -- each layer requests
// the first staged chunk of the layer that follows it (mfma_chain.hpp: dense_flow); 0: every layer starts with its own
// DMA + wait + barrier (dense_staged), the form of rounds 1-3, kept for A/B runs.
#ifndef MCN_SARL_FLOW
#define MCN_SARL_FLOW 1
#endif
#if MCN_SARL_FLOW
#define SARL_LAYER(KT, NT, RELU, INIT, L1, L2, in, init, out, w, b, next) \
    dense_flow<KT, NT, RELU, INIT, L1, L2>(in, init, out, w, b, F, lane, next)
#else
#define SARL_LAYER(KT, NT, RELU, INIT, L1, L2, in, init, out, w, b, next) \
    dense_staged<KT, NT, RELU, INIT, L1, L2>(in, init, out, w, b, S, lane)
#endif

// USE_X3 = false: float32 MFMA layers (v_mfma_f32_16x16x4_f32); USE_X3 = true: the same layers on the bf16 matrix pipe with every
// operand split into three bfloat16 pieces (mfma_chain.hpp: dense_flow_x3) -- float32-accurate, ~2.7 x fewer pipe cycles.
template <bool USE_X3>
__global__ __launch_bounds__(kSarlWaves * 64, kSarlWaves >= 8 ? 1 : 2) void sarl_value_kernel(const SarlParams p)
{
    __shared__ float4 s_stage[2 * (kStageFloat4 + kStageBias)];
    const long npairs = (long)p.E * p.A;
    const int N = p.N;
#if MCN_SARL_FLOW
    // chunk 0 of every layer, as the layer before it requests it
#define SARL_FIRST(KT, KB, NT, INIT, name, bias)                                                              \
    (USE_X3 ? first_chunk_x3<KB, NT, INIT>(p.x.w_##name, bias) : first_chunk<KT, NT, INIT>(p.f.w_##name, bias))
    const NextChunk d_m1a = SARL_FIRST(T13, B13, T150, false, m1a, p.f.b_m1a), d_m1b = SARL_FIRST(T150, B150, T100, false, m1b, p.f.b_m1b);
    const NextChunk d_atg = SARL_FIRST(T100, B100, T100, false, atg, p.f.b_ata), d_ata = SARL_FIRST(T100, B100, T100, true, ata, nullptr);
    const NextChunk d_atb = SARL_FIRST(T100, B100, T100, false, atb, p.f.b_atb), d_atc = SARL_FIRST(T100, B100, T1, false, atc, p.f.b_atc);
    const NextChunk d_m2a = SARL_FIRST(T100, B100, T100, false, m2a, p.f.b_m2a), d_m2b = SARL_FIRST(T100, B100, T50, false, m2b, p.f.b_m2b);
    const NextChunk d_m3a = SARL_FIRST(T56, B56, T150, false, m3a, p.f.b_m3a), d_m3b = SARL_FIRST(T150, B150, T100, false, m3b, p.f.b_m3b);
    const NextChunk d_m3c = SARL_FIRST(T100, B100, T100, false, m3c, p.f.b_m3c), d_m3d = SARL_FIRST(T100, B100, T1, false, m3d, p.f.b_m3d);
#undef SARL_FIRST
    const NextChunk d_none = {nullptr, 0, nullptr, 0, 0};
    WeightFlow F{s_stage, (int)threadIdx.x, 0};
    if ((long)blockIdx.x < p.ngroups) flow_stage_first(F, d_m1a, 0);         // the very first layer of this workgroup
    __syncthreads();
#endif
    // every wavefront of the workgroup runs the same number of passes (the weight-staging barriers are collective)
#pragma unroll 1
  for (long grp = blockIdx.x; grp < p.ngroups; grp += gridDim.x) {
    // the thread id is made opaque once per pass: everything derived from it (the per-lane addresses of every weight
    // chunk of every layer) would otherwise be hoisted out of this loop and live -- and spill -- across it
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const WeightStage S{s_stage, tid}; (void)S;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    X3 no4[B100], no2[2], no1[1];       // (output-pieces argument of the x3 layers that hand on float32 tiles only)
    // this wavefront's workspace slot: by resident wavefront (persistent grid), reused for every tile it walks over
    constexpr int WSR = USE_X3 ? kWsRowsX3 : kWsRowsF32;                   // 16-byte rows per human and lane
    // slot of this wavefront: N humans x WSR rows, then (x3) T100 rows for the global half of attention.0
    float4 *const ws = p.workspace + ((long)blockIdx.x * kSarlWaves + wave) * ((long)N * kWsRowsX3 + T100) * 64;
    const long pair0 = (grp * kSarlWaves + wave) * 16;
    SARL_T0();
    // no early exit: every wavefront of the workgroup takes part in the weight staging barriers
    long pair = pair0 + j;
    const bool valid = pair < npairs;
    if (!valid) pair = npairs - 1;
    const int e = (int)(pair / p.A), a = (int)(pair - (long)e * p.A);
    // pedestrians this pair's env shows to the policy; the loops below stay N long for the whole workgroup (they
    // contain the weight-staging barriers) and absent slots are masked out of every reduction
    int ne = N;
    if (p.hcount) { ne = p.hcount[e]; ne = ne < 1 ? 1 : (ne > N ? N : ne); }
    const double dt = p.dt;

    // ---- robot after the candidate action (cadrl.py:104-129), float64 like the reference ----
    const double2 rp = reinterpret_cast<const double2 *>(p.rpos)[e];
    const double2 rg = reinterpret_cast<const double2 *>(p.rgoal)[e];
    const double2 ra = make_double2(p.rrad[e], p.rvpref[e]);               // radius, v_pref
    const double2 ac = reinterpret_cast<const double2 *>(p.actions)[a];
    double npx, npy, nvx, nvy, nth;
    if (p.kinematics == MCN_KIN_UNICYCLE) {
        nth = p.rtheta[e] + ac.y;
        nvx = ac.x * cos(nth); nvy = ac.x * sin(nth);
        npx = rp.x + nvx * dt; npy = rp.y + nvy * dt;
    } else {
        nth = p.rtheta ? p.rtheta[e] : 0.0;
        nvx = ac.x; nvy = ac.y;
        npx = rp.x + ac.x * dt; npy = rp.y + ac.y * dt;
    }
    // ---- self part of the rotated state (cadrl.py:223-240), float32 like torch.Tensor(...) ----
    const float spx = (float)npx, spy = (float)npy, svx = (float)nvx, svy = (float)nvy;
    const float srad = (float)ra.x, sgx = (float)rg.x, sgy = (float)rg.y, svpref = (float)ra.y;
    const float gdx = sgx - spx, gdy = sgy - spy;
    const float dg = sqrtf(gdx * gdx + gdy * gdy);
    // cos/sin of rot = atan2(gdy, gdx) without the round trip through the angle
    const float cr = dg > 0.0f ? gdx / dg : 1.0f;
    const float sr = dg > 0.0f ? gdy / dg : 0.0f;
    const float f_theta = (p.kinematics == MCN_KIN_UNICYCLE) ? ((float)nth - atan2f(gdy, gdx)) : 0.0f;
    const float f_vx = svx * cr + svy * sr;
    const float f_vy = svy * cr - svx * sr;

    // ---- pass 1: mlp1 per human, global-state sum, reward ----
    f32x4 gsum[T100];
#pragma unroll
    for (int t = 0; t < T100; ++t) gsum[t] = (f32x4){0, 0, 0, 0};
    double dmin = INFINITY;
    SARL_PHASE(0);                      // tile set-up: robot state, self features
    for (int i = 0; i < N; ++i) {
        // per-pass opaque copy of the thread id: the staging addresses of this pass's layers are re-derived here
        // (a few integer instructions) instead of being hoisted out of the loop as ~30 live 64-bit values
        int tid_i = tid;
        asm volatile("" : "+v"(tid_i));
        const WeightStage S{s_stage, tid_i};
        (void)S;
        const long ha = (long)e * N + i;
        const double2 hp = reinterpret_cast<const double2 *>(p.hpos)[ha];
        const double2 hv = reinterpret_cast<const double2 *>(p.hvel)[ha];
        const double hr = p.hrad[ha];
        double qx = hp.x + hv.x * dt, qy = hp.y + hv.y * dt;            // constant-velocity propagate
        double nhvx = hv.x, nhvy = hv.y;
        if (p.next_hpos) {                                              // the env's look-ahead states instead
            const double2 np_ = reinterpret_cast<const double2 *>(p.next_hpos)[ha];
            const double2 nv_ = reinterpret_cast<const double2 *>(p.next_hvel)[ha];
            qx = np_.x; qy = np_.y; nhvx = nv_.x; nhvy = nv_.y;
        }
        const double d = norm2d(npx - qx, npy - qy) - ra.x - hr;        // multi_human_rl.py:70
        dmin = i < ne ? fmin(dmin, d) : dmin;
        const float hx = (float)qx, hy = (float)qy, hvx = (float)nhvx, hvy = (float)nhvy, hrad = (float)hr;
        const float ox = hx - spx, oy = hy - spy;
        float feat[16];
        feat[0] = dg; feat[1] = svpref; feat[2] = f_theta; feat[3] = srad; feat[4] = f_vx; feat[5] = f_vy;
        feat[6] = ox * cr + oy * sr;
        feat[7] = oy * cr - ox * sr;
        feat[8] = hvx * cr + hvy * sr;
        feat[9] = hvy * cr - hvx * sr;
        feat[10] = hrad;
        { const float ax_ = spx - hx, ay_ = spy - hy; feat[11] = sqrt_f32(ax_ * ax_ + ay_ * ay_); }   // == sqrtf (fast_f32.hpp)
        feat[12] = srad + hrad;
        feat[13] = feat[14] = feat[15] = 0.0f;
        f32x4 x[T13];
#pragma unroll
        for (int r = 0; r < 4; ++r)      // register r of lane group q carries feature 4q + r
            x[0][r] = q == 0 ? feat[r] : (q == 1 ? feat[4 + r] : (q == 2 ? feat[8 + r] : feat[12 + r]));
        SARL_PHASE(1);                  // per human: state loads, float64 distance, rotated features
        f32x4 h1[T150];
        f32x4 h2[T100];
        const f32x4 zero4 = {0, 0, 0, 0};
        if constexpr (USE_X3) {
            const X3 xin[B13] = {split8(x[0], zero4)};
            X3 h1p[B150];
            dense_flow_x3<B13, T150, true, false, kX3In>(xin, nullptr, h1, h1p, p.x.w_m1a, reinterpret_cast<const float4 *>(p.f.b_m1a), F, lane, d_m1b);
            if (!kX3In) split_after(h1, h1p);
            SARL_PHASE(2);              // mlp1.0
            X3 h2p[B100];
            auto add_to_mean = [&](int n, const f32x4 &v) { if (i < ne) gsum[n] += v; };       // global-state sum, tile by tile
            dense_flow_x3<B150, T100, true, false, kX3In, false>(h1p, nullptr, h2, h2p, p.x.w_m1b, reinterpret_cast<const float4 *>(p.f.b_m1b),
                                                                F, lane, (i + 1 < N ? d_m1a : d_atg), add_to_mean);
            if (!kX3In) split_after(h2, h2p);
            SARL_PHASE(3);              // mlp1.2
            // the workspace keeps mlp1's output already split: pass 2 reads the pieces twice (attention.0, mlp2.0)
#pragma unroll
            for (int m = 0; m < B100; ++m) {
                if (MCN_X3_WHATIF & 32) break;
                ws[(i * WSR + 3 * m + 0) * 64 + lane] = __builtin_bit_cast(float4, h2p[m].hi);
                ws[(i * WSR + 3 * m + 1) * 64 + lane] = __builtin_bit_cast(float4, h2p[m].mid);
                ws[(i * WSR + 3 * m + 2) * 64 + lane] = __builtin_bit_cast(float4, h2p[m].lo);
            }
        } else {
        SARL_LAYER(T13, T150, true, false, 4, 4, x, nullptr, h1, p.f.w_m1a, p.f.b_m1a, d_m1b);
        SARL_PHASE(2);                  // mlp1.0
        SARL_LAYER(T150, T100, true, false, 2, 4, h1, nullptr, h2, p.f.w_m1b, p.f.b_m1b, (i + 1 < N ? d_m1a : d_atg));
        SARL_PHASE(3);                  // mlp1.2
#pragma unroll
        for (int t = 0; t < T100; ++t) {
            ws[(i * T100 + t) * 64 + lane] = make_float4(h2[t][0], h2[t][1], h2[t][2], h2[t][3]);
            if (i < ne) gsum[t] += h2[t];
        }
        }
        SARL_PHASE(4);                  // workspace store, global-state sum
    }
    // reward ladder of MultiHumanRL.compute_reward with its hard-coded constants
    const bool reach = norm2d(npx - rg.x, npy - rg.y) < ra.x;
    double reward;
    if (dmin < 0) reward = -0.25;
    else if (reach) reward = 1;
    else if (dmin < 0.2) reward = (dmin - 0.2) * 0.5 * dt;
    else reward = 0;
    if (p.reward_in) reward = p.reward_in[pair];

    // global state = mean over humans (sarl.py:41); its contribution to attention layer 0 is the same for
    // every human of the pair, so it becomes the accumulator init of that layer
    // (one correctly rounded reciprocal + 28 multiplies instead of 28 IEEE divisions of ~11 instructions each: vector
    //  instructions are paid beside float32 MFMAs; the mean moves by <= 1 ulp, far inside the 1e-5 bar)
    const float inv_n = 1.0f / (float)ne;
#pragma unroll
    for (int t = 0; t < T100; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) gsum[t][r] = gsum[t][r] * inv_n;
    f32x4 gat[T100];
    const f32x4 zero4 = {0, 0, 0, 0};
    if constexpr (USE_X3) {
        X3 gp[B100];
#pragma unroll
        for (int m = 0; m < B100; ++m) gp[m] = split8(gsum[2 * m], 2 * m + 1 < T100 ? gsum[2 * m + 1] : zero4);
        dense_flow_x3<B100, T100, false, false>(gp, nullptr, gat, no4, p.x.w_atg, reinterpret_cast<const float4 *>(p.f.b_ata), F, lane, kM2aEarly ? d_m2a : d_ata);
        if (kX3GatMem) {
            // parked in the workspace: the per-human attention.0 layers fetch it tile by tile as their accumulator start
            // (28 registers that would otherwise be live across the whole second pass)
#pragma unroll
            for (int t = 0; t < T100; ++t) ws[(N * WSR + t) * 64 + lane] = make_float4(gat[t][0], gat[t][1], gat[t][2], gat[t][3]);
        }
    } else {
    SARL_LAYER(T100, T100, false, false, 1, 4, gsum, nullptr, gat, p.f.w_atg, p.f.b_ata, d_ata);
    }
    SARL_PHASE(5);                      // reward ladder, mean, global half of attention.0

    // ---- pass 2: attention score, mlp2, pooling ----
    // mlp2's last layer is linear (sarl.py:31, cadrl.py:11-19: no ReLU after the last Linear) and the attention
    // weights sum to one, so  sum_i w_i (W r_i + b) = W (sum_i w_i r_i) + b  with r_i = relu(mlp2.0(h_i)): the
    // weighted sum is taken over the 100-wide hidden activations and mlp2.2 runs ONCE per pair instead of once
    // per human (100 MFMAs per human fewer; same value to float32 summation-order noise, ~1e-7).
    f32x4 racc[T100];
#pragma unroll
    for (int t = 0; t < T100; ++t) racc[t] = (f32x4){0, 0, 0, 0};
    float denom = 0.0f;
    for (int i = 0; i < N; ++i) {
        int tid_i = tid;
        asm volatile("" : "+v"(tid_i));
        const WeightStage S{s_stage, tid_i};
        (void)S;
        f32x4 h2[T100];
        f32x4 a1[T100];
        f32x4 a2[T100];
        f32x4 sc[T1];
        auto load_pieces = [&](X3 (&dst)[B100]) {          // mlp1's output of human i, as pass 1 split it
#pragma unroll
            for (int m = 0; m < B100; ++m) {
                if (MCN_X3_WHATIF & 32) { dst[m].hi = dst[m].mid = dst[m].lo = __builtin_bit_cast(bf16x8, make_float4(1.f, 2.f, 3.f, (float)i)); continue; }
                dst[m].hi = __builtin_bit_cast(bf16x8, ws[(i * WSR + 3 * m + 0) * 64 + lane]);
                dst[m].mid = __builtin_bit_cast(bf16x8, ws[(i * WSR + 3 * m + 1) * 64 + lane]);
                dst[m].lo = __builtin_bit_cast(bf16x8, ws[(i * WSR + 3 * m + 2) * 64 + lane]);
            }
        };
        f32x4 m1[T100];
        if constexpr (USE_X3) {
            X3 hp[B100];
            load_pieces(hp);
            SARL_PHASE(6);              // workspace load
            if constexpr (kM2aEarly) {
                // mlp2.0 of this human FIRST, while its mlp1 pieces are in registers for attention.0 anyway: its 28
                // output registers wait for the score instead of the 12 KiB of pieces being read a second time
                dense_flow_x3<B100, T100, true, false, false, false>(hp, nullptr, m1, no4, p.x.w_m2a, reinterpret_cast<const float4 *>(p.f.b_m2a),
                                                                    F, lane, d_ata);
            }
            X3 ap[B100], bp[B100];
            dense_flow_x3<B100, T100, true, true, kX3In, kX3GatMem>(hp, kX3GatMem ? reinterpret_cast<const f32x4 *>(ws + (N * WSR) * 64 + lane) : gat,
                                                                    a1, ap, p.x.w_ata, nullptr, F, lane, d_atb);
            if (!kX3In) split_after(a1, ap);
            SARL_PHASE(7);              // attention.0
            dense_flow_x3<B100, T100, true, false, kX3In>(ap, nullptr, a2, bp, p.x.w_atb, reinterpret_cast<const float4 *>(p.f.b_atb), F, lane, d_atc);
            if (!kX3In) split_after(a2, bp);
            SARL_PHASE(8);              // attention.2
            dense_flow_x3<B100, T1, false, false>(bp, nullptr, sc, no1, p.x.w_atc, reinterpret_cast<const float4 *>(p.f.b_atc), F, lane,
                                                  kM2aEarly ? (i + 1 < N ? d_m2a : d_m2b) : d_m2a);
        } else {
#pragma unroll
        for (int t = 0; t < T100; ++t) {
            const float4 v = ws[(i * T100 + t) * 64 + lane];
            h2[t] = (f32x4){v.x, v.y, v.z, v.w};
        }
        SARL_PHASE(6);                  // workspace load
        SARL_LAYER(T100, T100, true, true, 1, 4, h2, gat, a1, p.f.w_ata, nullptr, d_atb);
        SARL_PHASE(7);                  // attention.0
        SARL_LAYER(T100, T100, true, false, 1, 4, a1, nullptr, a2, p.f.w_atb, p.f.b_atb, d_atc);
        SARL_PHASE(8);                  // attention.2
        SARL_LAYER(T100, T1, false, false, 1, 4, a2, nullptr, sc, p.f.w_atc, p.f.b_atc, d_m2a);
        }
        SARL_PHASE(9);                  // attention.4
        // score of pair j sits in lane j (q = 0), register 0; broadcast to the pair's four lanes
        const float s = __shfl(sc[0][0], j);
        const float es = (s != 0.0f && i < ne) ? expf(s) : 0.0f;   // exp(s) * (s != 0), sarl.py:52; absent: 0
        if (p.attention && valid && q == 0) p.attention[pair * N + i] = es;   // normalised by the host view
        denom += es;
        SARL_PHASE(10);                 // exp, attention output
        if constexpr (USE_X3 && kM2aEarly) {
#pragma unroll
            for (int t = 0; t < T100; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) racc[t][r] = i < ne ? __builtin_fmaf(es, m1[t][r], racc[t][r]) : racc[t][r];
        } else if constexpr (USE_X3) {
            X3 hp[B100];
            load_pieces(hp);            // (read again rather than kept: 48 registers across the attention layers)
            auto pool = [&](int n, const f32x4 &v) {        // weighted sum of the hidden activations, tile by tile
#pragma unroll
                for (int r = 0; r < 4; ++r) racc[n][r] = i < ne ? __builtin_fmaf(es, v[r], racc[n][r]) : racc[n][r];
            };
            dense_flow_x3<B100, T100, true, false, false, false>(hp, nullptr, m1, no4, p.x.w_m2a, reinterpret_cast<const float4 *>(p.f.b_m2a),
                                                                F, lane, (i + 1 < N ? d_ata : d_m2b), pool);
        } else {
        SARL_LAYER(T100, T100, true, false, 1, 4, h2, nullptr, m1, p.f.w_m2a, p.f.b_m2a, (i + 1 < N ? d_ata : d_m2b));
        }
        SARL_PHASE(11);                 // mlp2.0
        if constexpr (!USE_X3) {
#pragma unroll
        for (int t = 0; t < T100; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) racc[t][r] = i < ne ? __builtin_fmaf(es, m1[t][r], racc[t][r]) : racc[t][r];
        }
        SARL_PHASE(12);                 // weighted accumulation
    }

    // ---- tail: mlp3 on [self(6), pooled(50)] ----
    f32x4 jin[T56];
    {
        // weights = exp(s) (s != 0) / sum (sarl.py:52-53): normalise the accumulated hidden activations, then mlp2.2
        const float inv_d = 1.0f / denom;            // 0 / 0 stays NaN (0 * inf), as in the reference's softmax
#pragma unroll
        for (int t = 0; t < T100; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) racc[t][r] = racc[t][r] * inv_d;
        f32x4 pooled[T50];
        if constexpr (USE_X3) {
            X3 rp3[B100];
#pragma unroll
            for (int m = 0; m < B100; ++m) rp3[m] = split8(racc[2 * m], 2 * m + 1 < T100 ? racc[2 * m + 1] : zero4);
            dense_flow_x3<B100, T50, false, false>(rp3, nullptr, pooled, no2, p.x.w_m2b, reinterpret_cast<const float4 *>(p.f.b_m2b), F, lane, d_m3a);
        } else {
        SARL_LAYER(T100, T50, false, false, 1, 4, racc, nullptr, pooled, p.f.w_m2b, p.f.b_m2b, d_m3a);
        }
#pragma unroll
        for (int t = 0; t < T50; ++t) jin[t] = pooled[t];
    }
    {
        // the 6 self features, packed "q first": feature j sits in register j/4 of lane group j%4 (two k-steps)
        const float s0 = q == 0 ? dg : (q == 1 ? svpref : (q == 2 ? f_theta : srad));
        const float s1 = q == 0 ? f_vx : (q == 1 ? f_vy : 0.0f);
        jin[T50] = (f32x4){s0, s1, 0.0f, 0.0f};
    }
    SARL_PHASE(13);                     // normalise, mlp2.2, self tile
    f32x4 v1[T150];
    f32x4 v2[T100];
    f32x4 v3[T100];
    f32x4 vo[T1];
    if constexpr (USE_X3) {
        X3 jp[B56];                     // input blocks: (pooled tiles 0, 1), (pooled tiles 2, 3), (self tile, -)
        jp[0] = split8(jin[0], jin[1]); jp[1] = split8(jin[2], jin[3]); jp[2] = split8(jin[4], zero4);
        X3 vp[B150];
        dense_flow_x3<B56, T150, true, false, kX3In>(jp, nullptr, v1, vp, p.x.w_m3a, reinterpret_cast<const float4 *>(p.f.b_m3a), F, lane, d_m3b);
        if (!kX3In) split_after(v1, vp);
        X3 wp[B100], xp[B100];
        dense_flow_x3<B150, T100, true, false, kX3In>(vp, nullptr, v2, wp, p.x.w_m3b, reinterpret_cast<const float4 *>(p.f.b_m3b), F, lane, d_m3c);
        if (!kX3In) split_after(v2, wp);
        dense_flow_x3<B100, T100, true, false, kX3In>(wp, nullptr, v3, xp, p.x.w_m3c, reinterpret_cast<const float4 *>(p.f.b_m3c), F, lane, d_m3d);
        if (!kX3In) split_after(v3, xp);
        dense_flow_x3<B100, T1, false, false>(xp, nullptr, vo, no1, p.x.w_m3d, reinterpret_cast<const float4 *>(p.f.b_m3d), F, lane,
                                             (grp + gridDim.x < p.ngroups ? d_m1a : d_none));
    } else {
    SARL_LAYER(T56, T150, true, false, 2, 1, jin, nullptr, v1, p.f.w_m3a, p.f.b_m3a, d_m3b);
    SARL_LAYER(T150, T100, true, false, 2, 4, v1, nullptr, v2, p.f.w_m3b, p.f.b_m3b, d_m3c);
    SARL_LAYER(T100, T100, true, false, 1, 4, v2, nullptr, v3, p.f.w_m3c, p.f.b_m3c, d_m3d);
    SARL_LAYER(T100, T1, false, false, 1, 4, v3, nullptr, vo, p.f.w_m3d, p.f.b_m3d, (grp + gridDim.x < p.ngroups ? d_m1a : d_none));
    }
    if (valid && q == 0) {
        // value = reward + gamma^(dt * v_pref) * V   (multi_human_rl.py:52, Python float arithmetic)
        p.values[pair] = reward + p.gamma_pow * (double)vo[0][0];
    }
    if (p.attention && valid && q == 0) {
        for (int i = 0; i < N; ++i) p.attention[pair * N + i] /= denom;
    }
    SARL_PHASE(14);                     // mlp3, value store
  }
}

// Strict-'>' argmax over the A candidate values of each env (multi_human_rl.py:53-55: the first maximum
// wins), one wavefront per env.  Also reports reach_destination (policy.py:43-49), for which the reference
// returns the zero action without evaluating anything.
__global__ __launch_bounds__(64) void sarl_argmax_kernel(const double *__restrict__ values, const double *rpos,
                                                       const double *rgoal, const double *rrad, int E, int A,
                                                       int32_t *__restrict__ best, double *__restrict__ best_val,
                                                       const double *__restrict__ actions, double *__restrict__ action_out,
                                                       double epsilon, unsigned long long seed)
{
    const int e = blockIdx.x;
    const int lane = threadIdx.x;
    double bv = -INFINITY; int bi = -1;
    for (int k = lane; k < A; k += 64) {
        const double v = values[(long)e * A + k];
        if (v > bv) { bv = v; bi = k; }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        const bool take = (oi >= 0) && (bi < 0 || ov > bv || (ov == bv && oi < bi));
        if (take) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        const double2 rp = reinterpret_cast<const double2 *>(rpos)[e];
        const double2 rg = reinterpret_cast<const double2 *>(rgoal)[e];
        // numpy norm((py - gy, px - gx)): dot = fma(x1, x1, x0 * x0) with x0 = py-gy, x1 = px-gx
        const bool reached = norm2d(rp.y - rg.y, rp.x - rg.x) < rrad[e];
        // epsilon-greedy (multi_human_rl.py:27-29, phase 'train'): with probability epsilon the env takes a uniformly
        // drawn table row instead (best = -2); a robot on its goal returns before the draw (:22-23).  Counter-based
        // stream: two 64-bit mixes of (seed, env) -- the caller passes a fresh seed per call.
        int choice = bi;
        bool explore = false;
        if (epsilon > 0.0 && !reached) {
            auto mix = [](unsigned long long z) {                      // splitmix64 finaliser
                z += 0x9E3779B97F4A7C15ull;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                return z ^ (z >> 31);
            };
            const unsigned long long r1 = mix(seed ^ mix((unsigned long long)e));
            const unsigned long long r2 = mix(r1);
            explore = (double)(r1 >> 11) * (1.0 / 9007199254740992.0) < epsilon;
            if (explore) choice = (int)(((r2 >> 32) * (unsigned long long)A) >> 32);
        }
        best[e] = reached ? -1 : (explore ? -2 : bi);
        best_val[e] = bv;
        if (action_out) {
            // what MultiHumanRL.predict returns: the table row of the best value, the zero action on the goal
            // (multi_human_rl.py:22-23) -- and also when every value is NaN (bi < 0: the host raises, as the reference)
            const bool zero = reached || choice < 0;
            const double2 act = reinterpret_cast<const double2 *>(actions)[zero ? 0 : choice];
            reinterpret_cast<double2 *>(action_out)[e] = zero ? make_double2(0.0, 0.0) : act;
        }
    }
}

// Persistent grid: two 4-wave workgroups per CU on the 256 CUs of an MI355X (fewer CUs: more passes, same result).
#ifndef MCN_SARL_MAX_BLOCKS
#define MCN_SARL_MAX_BLOCKS ((kSarlWaves >= 8 ? 1 : 2) * 256)
#endif
constexpr int kSarlMaxBlocks = MCN_SARL_MAX_BLOCKS;     // A/B: a huge value = one workgroup per 4-tile group, workspace by tile

int launch_sarl(SarlParams &p, int32_t *best, double *best_val, double *action_out, double epsilon,
                unsigned long long seed, hipStream_t stream)
{
    const long npairs = (long)p.E * p.A;
    const long waves = (npairs + 15) / 16;
    p.ngroups = (waves + kSarlWaves - 1) / kSarlWaves;
    const int blocks = (int)(p.ngroups < kSarlMaxBlocks ? p.ngroups : kSarlMaxBlocks);
#if MCN_SARL_FLOW
    if (p.x.w_m1a) hipLaunchKernelGGL(sarl_value_kernel<true>, dim3(blocks), dim3(kSarlWaves * 64), 0, stream, p);
    else
#endif
    hipLaunchKernelGGL(sarl_value_kernel<false>, dim3(blocks), dim3(kSarlWaves * 64), 0, stream, p);
    if (best) {
        hipLaunchKernelGGL(sarl_argmax_kernel, dim3(p.E), dim3(64), 0, stream, p.values, p.rpos, p.rgoal, p.rrad,
                           p.E, p.A, best, best_val, p.actions, action_out, epsilon, seed);
    }
    return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
}

int launch_sarl_c(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int A, double dt,
                  double gamma_pow, int kinematics, void *workspace, double *values, int32_t *best, double *best_val,
                  float *attention, const double *next_hpos, const double *next_hvel, const double *reward_in,
                  double *action_out, double epsilon, unsigned long long seed, int E, int N, hipStream_t stream)
{
    SarlParams p;
    const float4 *const *src = reinterpret_cast<const float4 *const *>(net);
    const float4 **dst = reinterpret_cast<const float4 **>(&p.f);
    static_assert(sizeof(SarlFrags) + sizeof(void *) == sizeof(mcn_sarl_net), "fragment tables must mirror the C struct");
    static_assert(sizeof(SarlX3) == sizeof(mcn_sarl_x3), "x3 fragment tables must mirror the C struct");
    for (size_t k = 0; k < sizeof(SarlFrags) / sizeof(float *); ++k) dst[k] = src[k];
    memset(&p.x, 0, sizeof(p.x));
    if (net->x3 && tuning_sarl_x3()) memcpy(&p.x, net->x3, sizeof(p.x));
    p.rpos = st->rpos; p.rvel = st->rvel; p.rgoal = st->rgoal; p.rrad = st->rrad; p.rvpref = st->rvpref; p.rtheta = st->rtheta;
    p.hpos = st->hpos; p.hvel = st->hvel; p.hrad = st->hrad; p.hcount = st->hcount;
    p.next_hpos = next_hpos; p.next_hvel = next_hvel; p.reward_in = reward_in;
    p.actions = actions; p.workspace = reinterpret_cast<float4 *>(workspace);
    p.values = values; p.attention = attention;
    p.E = E; p.N = N; p.A = A; p.kinematics = kinematics; p.dt = dt; p.gamma_pow = gamma_pow;
    return launch_sarl(p, best, best_val, action_out, epsilon, seed, stream);
}

long sarl_workspace_float4s(int E, int N, int A)
{
    const long waves = ((long)E * A + 15) / 16;
    long groups = (waves + kSarlWaves - 1) / kSarlWaves;
    if (groups > kSarlMaxBlocks) groups = kSarlMaxBlocks;          // one slot per RESIDENT wavefront
    return groups * kSarlWaves * ((long)N * kWsRowsX3 + T100) * 64;           // per wavefront: N humans x 12 rows + 7 (x3 layout)
}

}  // namespace mcn
