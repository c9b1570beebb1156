// world_mlp.hip -- MlpWorld one-step world model for E scenes in one launch (gfx950).
//
// Replaces crowd_nav/policy/world_model.py:22-42 (MlpWorld.forward in eval mode: Linear(4N,128) ReLU [Dropout]
// Linear(128,64) ReLU [Dropout] Linear(64,12) ReLU Linear(12,2N) Tanh) together with the call convention of
// crowd_sim/envs/model_crowd_sim.py:401-407 -- the scene's observable states [px, py, vx, vy] x N as one float32 row,
// the returned row reshaped to N velocities -- for a whole batch of scenes resident in HBM.
//
// Same register-chained float32 MFMA scheme as the other network kernels (mfma_chain.hpp): scene on the MFMA column
// index, feature on the row index, a layer's output tile is the next layer's B operand.  The whole network (46-64 KiB
// of pre-permuted fragments) is LDS-resident per 4-wavefront workgroup; no staging, one barrier after the fill.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "mfma_chain.hpp"

namespace mcn {

struct MlpWorldParams {
    const float4 *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
    const double *hpos, *hvel;      // [E*N][2]
    double *out_vel;                // [E*N][2]
    int E, N;
};

// KT1 input tiles (4N features), NT4 output tiles (2N features)
template <int KT1, int NT4>
__global__ __launch_bounds__(256) void mlp_world_kernel(const MlpWorldParams p)
{
    constexpr int T128 = 8, T64 = 4, T12 = 1;
    __shared__ float4 s_w1[T128 * KT1 * 64], s_w2[T64 * T128 * 64], s_w3[T12 * T64 * 64], s_w4[NT4 * T12 * 64];
    __shared__ float4 s_b1[T128 * 4], s_b2[T64 * 4], s_b3[T12 * 4], s_b4[NT4 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    lds_fill_layer<256>(s_w1, s_b1, p.w1, p.b1, KT1, T128, tid);
    lds_fill_layer<256>(s_w2, s_b2, p.w2, p.b2, T128, T64, tid);
    lds_fill_layer<256>(s_w3, s_b3, p.w3, p.b3, T64, T12, tid);
    lds_fill_layer<256>(s_w4, s_b4, p.w4, p.b4, T12, NT4, tid);
    __syncthreads();
    const int j = lane & 15, q = lane >> 4;
    const int N = p.N, F = 4 * N;
    const long e = ((long)blockIdx.x * 4 + wave) * 16 + j;
    const bool valid = e < p.E;
    const long eb = valid ? e : p.E - 1;
    // input row: feature f = 4 h + c, c = (px, py, vx, vy) -- torch.Tensor([current_s]) (float32), model_crowd_sim.py:404
    f32x4 x[KT1];
#pragma unroll
    for (int t = 0; t < KT1; ++t) {
        const int h = 4 * t + q;                            // features 16 t + 4 q + r  <=>  human 4 t + q, component r
        f32x4 v = {0, 0, 0, 0};
        if (4 * h < F) {
            const double2 ps = reinterpret_cast<const double2 *>(p.hpos)[eb * N + h];
            const double2 vl = reinterpret_cast<const double2 *>(p.hvel)[eb * N + h];
            v = (f32x4){(float)ps.x, (float)ps.y, (float)vl.x, (float)vl.y};
        }
        x[t] = v;
    }
    f32x4 h1[T128], h2[T64], h3[T12], o[NT4];
    dense_lds<KT1, T128, true>(x, h1, s_w1, s_b1, lane);
    dense_lds<T128, T64, true>(h1, h2, s_w2, s_b2, lane);
    dense_lds<T64, T12, true>(h2, h3, s_w3, s_b3, lane);
    dense_lds<T12, NT4, false, 3>(h3, o, s_w4, s_b4, lane);       // 12 inputs packed "q first": 3 k-steps
    if (valid) {
#pragma unroll
        for (int n = 0; n < NT4; ++n) {
            // output features 16 n + 4 q + r = 2 human + component: registers (0,1) and (2,3) are two humans' velocities
            const int f0 = 16 * n + 4 * q;
            if (f0 < 2 * N)
                reinterpret_cast<double2 *>(p.out_vel)[eb * N + f0 / 2] = make_double2((double)tanhf(o[n][0]), (double)tanhf(o[n][1]));
            if (f0 + 2 < 2 * N)
                reinterpret_cast<double2 *>(p.out_vel)[eb * N + f0 / 2 + 1] = make_double2((double)tanhf(o[n][2]), (double)tanhf(o[n][3]));
        }
    }
}

int launch_mlp_world(const mcn_mlp_world_net *net, const double *hpos, const double *hvel, double *out_vel, int E, int N,
                     hipStream_t stream)
{
    MlpWorldParams p;
    p.w1 = reinterpret_cast<const float4 *>(net->w1); p.b1 = reinterpret_cast<const float4 *>(net->b1);
    p.w2 = reinterpret_cast<const float4 *>(net->w2); p.b2 = reinterpret_cast<const float4 *>(net->b2);
    p.w3 = reinterpret_cast<const float4 *>(net->w3); p.b3 = reinterpret_cast<const float4 *>(net->b3);
    p.w4 = reinterpret_cast<const float4 *>(net->w4); p.b4 = reinterpret_cast<const float4 *>(net->b4);
    p.hpos = hpos; p.hvel = hvel; p.out_vel = out_vel; p.E = E; p.N = N;
    const int blocks = (E + 63) / 64;
    if (N <= 4)      hipLaunchKernelGGL((mlp_world_kernel<1, 1>), dim3(blocks), dim3(256), 0, stream, p);
    else if (N <= 8) hipLaunchKernelGGL((mlp_world_kernel<2, 1>), dim3(blocks), dim3(256), 0, stream, p);
    else             hipLaunchKernelGGL((mlp_world_kernel<3, 2>), dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
}

}  // namespace mcn
