// sgan_step.hip -- Social-GAN one-step pedestrian prediction for gfx950 (MI355X).
//
// Replaces SGANWorld.forward (crowd_nav/policy/world_model.py:234-268) for E scenes x N pedestrians:
//   * the text-file ring buffer (append frame, re-read, drop oldest, rewrite: :238-248) becomes a
//     device-resident ring hist[E][8][N][2] of positions rounded to 1e-4 (np.around(..., 4), :169,192);
//   * TrajectoryGenerator.forward (sgan/models.py:501-553): Encoder (:28-71: Linear(2,16) + LSTM(16,32) over
//     8 relative displacements), PoolHiddenNet (:167-232: per ordered pair Linear(2,16) (+) h -> 512 -> 8,
//     ReLU, max over partners), mlp_decoder_context (32[+8] -> 64 -> 24, ReLU), add_noise (:454-490, one
//     8-vector per scene), Decoder (:127-164) for seq_len 1: Linear(2,16), LSTM cell, Linear(32,2);
//   * relative_to_abs (sgan/utils.py:85-98) and velocity = (pred - last) / time_step (world_model.py:258-268).
//
// Two launches (the pooling needs every pedestrian's encoder state of the scene):
//   sgan_encode_kernel  one wavefront = 16 pedestrians; 8 LSTM steps chained in registers (mfma_chain.hpp)
//   sgan_decode_kernel  one wavefront = 16 pedestrians; loops over the N partners of the scene with the
//                       running max kept per lane, then context MLP, noise, decoder cell, output.
// All network arithmetic is float32 (MFMA fmaf chains); positions / velocities are float64 like the env.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "mfma_chain.hpp"

namespace mcn {

struct SganFrags {
    const float4 *w_eemb, *b_eemb;     // encoder.spatial_embedding          2 -> 16
    const float4 *w_elstm, *b_elstm;   // encoder.encoder  [W_ih | W_hh]     48 -> 128, bias = b_ih + b_hh
    const float4 *w_pemb, *b_pemb;     // pool_net.spatial_embedding          2 -> 16
    const float4 *w_p1, *b_p1;         // pool_net.mlp_pre_pool.0            48 -> 512
    const float4 *w_p2, *b_p2;         // pool_net.mlp_pre_pool.2           512 -> 8
    const float4 *w_c1, *b_c1;         // mlp_decoder_context.0         32(+8) -> 64
    const float4 *w_c2, *b_c2;         // mlp_decoder_context.2              64 -> 24
    const float4 *w_demb, *b_demb;     // decoder.spatial_embedding           2 -> 16
    const float4 *w_dlstm, *b_dlstm;   // decoder.decoder  [W_ih | W_hh]     48 -> 128
    const float4 *w_h2p, *b_h2p;       // decoder.hidden2pos                 32 -> 2
};

struct SganParams {
    SganFrags f;
    double *hist;             // [E][8][N][2] rounded positions (ring)
    const double *cur_pos;    // [E*N][2] frame to push, or NULL
    const float *noise;       // [E][8]
    const int32_t *hcount;    // [E] or NULL: pedestrians present per scene
    float *henc;              // [E*N][32] encoder final hidden state
    float *last;              // [E*N][4]  last_pos.xy, last_rel.xy (float32)
    double *out_vel;          // [E*N][2]
    float *out_rel;           // [E*N][2] pred_rel (float32) or NULL
    int E, N, pooling, push_slot, oldest;
    double time_step;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// numpy.around(x, 4): rint(x * 1e4) / 1e4 in float64 (round-half-even)
__device__ __forceinline__ double round4(double x) { return rint(x * 10000.0) / 10000.0; }

// LSTM cell on gate tiles [i i f f g g o o] (PyTorch gate order), state tiles h[2], c[2]
__device__ __forceinline__ void lstm_update(const f32x4 (&g)[8], f32x4 (&h)[2], f32x4 (&c)[2])
{
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoidf_(g[t][r]), fg = sigmoidf_(g[2 + t][r]);
            const float gg = tanhf(g[4 + t][r]), og = sigmoidf_(g[6 + t][r]);
            const float cn = fg * c[t][r] + ig * gg;
            c[t][r] = cn;
            h[t][r] = og * tanhf(cn);
        }
}

// put a 2-vector into slots 0,1 of an input tile (feature 4q + r lives in register r of lane group q)
__device__ __forceinline__ f32x4 tile_xy(float x, float y, int q)
{
    f32x4 v = {0, 0, 0, 0};
    if (q == 0) { v[0] = x; v[1] = y; }
    return v;
}

constexpr int kSganWaves = kStageThreads / 64;      // the decode kernel shares one LDS weight stage per workgroup

__global__ __launch_bounds__(kSganWaves * 64) void sgan_encode_kernel(const SganParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const long nped = (long)p.E * p.N;
    const long ped0 = ((long)blockIdx.x * kSganWaves + wave) * 16;
    if (ped0 >= nped) return;
    long ped = ped0 + j;
    const bool valid = ped < nped;
    if (!valid) ped = nped - 1;
    const int e = (int)(ped / p.N), i = (int)(ped - (long)e * p.N);
    const int N = p.N;
    double2 *hist = reinterpret_cast<double2 *>(p.hist);
    auto slot = [&](int s) -> double2 & { return hist[((long)e * 8 + s) * N + i]; };

    if (p.cur_pos) {       // push the newest frame over the oldest one (world_model.py:238-248)
        const double2 cp = reinterpret_cast<const double2 *>(p.cur_pos)[ped];
        const double2 rounded = make_double2(round4(cp.x), round4(cp.y));
        if (valid && q == 0) slot(p.push_slot) = rounded;
        // all four lane groups of a pedestrian need the value now; they may not see lane group 0's store yet
        // so read it from the register instead
        (void)rounded;
    }
    f32x4 h[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, c[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double2 prev = make_double2(0, 0);
    float lrx = 0, lry = 0;
    for (int t = 0; t < 8; ++t) {
        const int s = (p.oldest + t) & 7;
        double2 cur;
        if (p.cur_pos && s == p.push_slot) {
            const double2 cp = reinterpret_cast<const double2 *>(p.cur_pos)[ped];
            cur = make_double2(round4(cp.x), round4(cp.y));
        } else {
            cur = slot(s);
        }
        // relative displacement in float64 on the rounded values, then float32 (world_model.py:190-206)
        const float rx = t == 0 ? 0.0f : (float)(cur.x - prev.x);
        const float ry = t == 0 ? 0.0f : (float)(cur.y - prev.y);
        prev = cur; lrx = rx; lry = ry;
        f32x4 xin[1] = {tile_xy(rx, ry, q)};
        f32x4 emb[1];
        dense<1, 1, false>(xin, emb, p.f.w_eemb, p.f.b_eemb, lane);
        f32x4 cat[3] = {emb[0], h[0], h[1]};
        f32x4 g[8];
        dense<3, 8, false>(cat, g, p.f.w_elstm, p.f.b_elstm, lane);
        lstm_update(g, h, c);
    }
    if (valid) {
        float4 *dst = reinterpret_cast<float4 *>(p.henc + ped * 32);
        dst[q] = make_float4(h[0][0], h[0][1], h[0][2], h[0][3]);           // features 4q .. 4q+3
        dst[4 + q] = make_float4(h[1][0], h[1][1], h[1][2], h[1][3]);       // features 16+4q ..
        if (q == 0)
            reinterpret_cast<float4 *>(p.last)[ped] = make_float4((float)prev.x, (float)prev.y, lrx, lry);
    }
}

__global__ __launch_bounds__(kSganWaves * 64, 2) void sgan_decode_kernel(const SganParams p)
{
    __shared__ float4 s_stage[2 * (kStageFloat4 + kStageBias)];
    const WeightStage S{s_stage, (int)threadIdx.x};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const long nped = (long)p.E * p.N;
    const long ped0 = ((long)blockIdx.x * kSganWaves + wave) * 16;
    // no early exit: every wavefront of the workgroup takes part in the weight-staging barriers
    long ped = ped0 + j;
    const bool valid = ped < nped;
    if (!valid) ped = nped - 1;
    const int N = p.N;
    const int e = (int)(ped / N);
    const float4 mine = reinterpret_cast<const float4 *>(p.last)[ped];       // last_pos.xy, last_rel.xy
    f32x4 hi[2];
    {
        const float4 *src = reinterpret_cast<const float4 *>(p.henc + ped * 32);
        const float4 a = src[q], b = src[4 + q];
        hi[0] = (f32x4){a.x, a.y, a.z, a.w};
        hi[1] = (f32x4){b.x, b.y, b.z, b.w};
    }
    f32x4 ctx[2];
    if (p.pooling) {
        f32x4 pool = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int ne = N;                                      // partners in this pedestrian's scene (loop stays N long:
        if (p.hcount) { ne = p.hcount[e]; ne = ne < 1 ? 1 : (ne > N ? N : ne); }     // it holds the staging barriers)
        for (int k = 0; k < N; ++k) {
            const long other = (long)e * N + k;
            const float4 theirs = reinterpret_cast<const float4 *>(p.last)[other];
            const float4 *src = reinterpret_cast<const float4 *>(p.henc + other * 32);
            const float4 a = src[q], b = src[4 + q];
            f32x4 xin[1] = {tile_xy(theirs.x - mine.x, theirs.y - mine.y, q)};       // P_k - P_i (models.py:221)
            f32x4 emb[1];
            dense_staged<1, 1, false, false>(xin, nullptr, emb, p.f.w_pemb, p.f.b_pemb, S, lane);
            f32x4 cat[3] = {emb[0], (f32x4){a.x, a.y, a.z, a.w}, (f32x4){b.x, b.y, b.z, b.w}};
            f32x4 hid[32];
            dense_staged<3, 32, true, false>(cat, nullptr, hid, p.f.w_p1, p.f.b_p1, S, lane);
            f32x4 o[1];
            dense_staged<32, 1, true, false>(hid, nullptr, o, p.f.w_p2, p.f.b_p2, S, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) pool[r] = k < ne ? fmaxf(pool[r], o[0][r]) : pool[r];
        }
        f32x4 cin[3] = {hi[0], hi[1], pool};
        f32x4 c1[4];
        dense<3, 4, true>(cin, c1, p.f.w_c1, p.f.b_c1, lane);
        dense<4, 2, true>(c1, ctx, p.f.w_c2, p.f.b_c2, lane);
    } else {
        f32x4 c1[4];
        dense<2, 4, true>(hi, c1, p.f.w_c1, p.f.b_c1, lane);
        dense<4, 2, true>(c1, ctx, p.f.w_c2, p.f.b_c2, lane);
    }
    // decoder_h = [context(24), noise(8)] (add_noise, 'global' mix: one vector per scene)
    f32x4 dh[2] = {ctx[0], ctx[1]}, dc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (q >= 2) {
        const float4 z = reinterpret_cast<const float4 *>(p.noise + (long)e * 8)[q - 2];
        dh[1] = (f32x4){z.x, z.y, z.z, z.w};
    }
    f32x4 xin[1] = {tile_xy(mine.z, mine.w, q)};
    f32x4 demb[1];
    dense<1, 1, false>(xin, demb, p.f.w_demb, p.f.b_demb, lane);
    f32x4 cat[3] = {demb[0], dh[0], dh[1]};
    f32x4 g[8];
    dense<3, 8, false>(cat, g, p.f.w_dlstm, p.f.b_dlstm, lane);
    lstm_update(g, dh, dc);
    f32x4 out[1];
    dense<2, 1, false>(dh, out, p.f.w_h2p, p.f.b_h2p, lane);
    if (valid && q == 0) {
        const float rx = out[0][0], ry = out[0][1];
        const float ax = rx + mine.x, ay = ry + mine.y;                 // relative_to_abs, float32
        if (p.out_rel) reinterpret_cast<float2 *>(p.out_rel)[ped] = make_float2(rx, ry);
        reinterpret_cast<double2 *>(p.out_vel)[ped] =
            make_double2(((double)ax - (double)mine.x) / p.time_step, ((double)ay - (double)mine.y) / p.time_step);
    }
}

int launch_sgan(const mcn_sgan_net *net, double *hist, int push_slot, int oldest, const double *cur_pos,
                const float *noise, const int32_t *hcount, void *workspace, double *out_vel, float *out_rel,
                double time_step, int E, int N, hipStream_t stream)
{
    SganParams p;
    static_assert(sizeof(SganFrags) == 20 * sizeof(void *), "fragment table size");
    const float4 *const *src = reinterpret_cast<const float4 *const *>(net);
    const float4 **dst = reinterpret_cast<const float4 **>(&p.f);
    for (int k = 0; k < 20; ++k) dst[k] = src[k];
    p.hist = hist; p.cur_pos = cur_pos; p.noise = noise; p.hcount = hcount;
    p.henc = reinterpret_cast<float *>(workspace);
    p.last = p.henc + (size_t)E * N * 32;
    p.out_vel = out_vel; p.out_rel = out_rel;
    p.E = E; p.N = N; p.pooling = net->pooling; p.push_slot = push_slot; p.oldest = oldest; p.time_step = time_step;
    const long tiles = ((long)E * N + 15) / 16;
    const int blocks = (int)((tiles + kSganWaves - 1) / kSganWaves);
    hipLaunchKernelGGL(sgan_encode_kernel, dim3(blocks), dim3(kSganWaves * 64), 0, stream, p);
    hipLaunchKernelGGL(sgan_decode_kernel, dim3(blocks), dim3(kSganWaves * 64), 0, stream, p);
    return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
}

}  // namespace mcn
