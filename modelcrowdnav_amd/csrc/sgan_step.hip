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
// Three launches (the pooling needs every pedestrian's encoder state of the scene, the decoder every pooled vector):
//   sgan_encode_kernel  one wavefront = 16 pedestrians; 8 LSTM steps chained in registers (mfma_chain.hpp)
//   sgan_pool_kernel    one wavefront = 16 partners x 5 pedestrians of their scenes; pool-net weights LDS-resident
//   sgan_decode_kernel  one wavefront = 16 pedestrians; context MLP, noise, decoder cell, output.
// All network arithmetic is float32 (MFMA fmaf chains); positions / velocities are float64 like the env.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "mfma_chain.hpp"

namespace mcn {

struct SganFrags {                     // spatial embeddings folded into the layer they feed (include/mcn.h)
    const float4 *w_elstm, *b_elstm;   // encoder.encoder  [W_ih W_se | W_hh]   2 + 32 -> 128
    const float4 *w_p1, *b_p1;         // pool_net.mlp_pre_pool.0 [W_e W_se | W_h]  2 + 32 -> 512
    const float4 *w_p2, *b_p2;         // pool_net.mlp_pre_pool.2           512 -> 8
    const float4 *w_c1, *b_c1;         // mlp_decoder_context.0         32(+8) -> 64
    const float4 *w_c2, *b_c2;         // mlp_decoder_context.2              64 -> 24
    const float4 *w_dlstm, *b_dlstm;   // decoder.decoder  [W_ih W_se | W_hh]   2 + 32 -> 128
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
    float *pool;              // [E*N][8]  pooled features (zeroed by the encoder, atomic max by the pool kernel)
    double *out_vel;          // [E*N][2]
    float *out_rel;           // [E*N][2] pred_rel (float32) or NULL
    int E, N, pooling, push_slot, oldest;
    double time_step;
};

// sigmoid / tanh of the LSTM cells on the hardware transcendentals.  The float32 MFMA shares its SIMD's issue with
// every other vector instruction (see sgan_pool_kernel), so the cell arithmetic is paid in full beside the gates'
// matrix work: libm's expf / tanhf and IEEE division are ~150 instructions per cell element, these forms 26.
// v_exp_f32 and v_rcp_f32 are 1-ulp instructions; with the rounding of the scaled argument the absolute error of
// either function is ~2e-7.  exp2 overflowing to +inf gives rcp = 0, underflowing to 0 gives rcp(1) = 1: both limits
// come out exact without clamps.
__device__ __forceinline__ float sigmoidf_(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x));
}
__device__ __forceinline__ float tanhf_(float x)                // 1 - 2 / (1 + e^2x)
{
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * x)), 1.0f);
}

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
            const float gg = tanhf_(g[4 + t][r]), og = sigmoidf_(g[6 + t][r]);
            const float cn = fg * c[t][r] + ig * gg;
            c[t][r] = cn;
            h[t][r] = og * tanhf_(cn);
        }
}

// a displacement as input tile 0 of a folded layer: x in slot 0, y in slot 4 (slot 4q + r = register r of lane
// group q), i.e. both in k-step 0
__device__ __forceinline__ f32x4 tile_xy(float x, float y, int q)
{
    f32x4 v = {0, 0, 0, 0};
    v[0] = q == 0 ? x : (q == 1 ? y : 0.0f);
    return v;
}

constexpr int kSganWaves = 4;

#ifndef MCN_ENC_MINWAVES
#define MCN_ENC_MINWAVES 1
#endif
__global__ __launch_bounds__(kSganWaves * 64, MCN_ENC_MINWAVES) void sgan_encode_kernel(const SganParams p)
{
    // encoder weights resident in LDS (24 KiB): the 8 LSTM steps re-read them from there, registers stay few enough
    // for five wavefronts per SIMD, whose cell arithmetic (vector ALU) overlaps the other wavefronts' MFMAs
    __shared__ float4 s_wl[8 * 3 * 64], s_bl[8 * 4];
    lds_fill_layer<kSganWaves * 64>(s_wl, s_bl, p.f.w_elstm, p.f.b_elstm, 3, 8, threadIdx.x);
    __syncthreads();
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
    double2 pushed = make_double2(0, 0);
    if (p.cur_pos) {       // push the newest frame over the oldest one (world_model.py:238-248)
        const double2 cp = reinterpret_cast<const double2 *>(p.cur_pos)[ped];
        pushed = make_double2(round4(cp.x), round4(cp.y));
        if (valid && q == 0) hist[((long)e * 8 + p.push_slot) * N + i] = pushed;
    }
    // frame t of the window; the pushed frame comes from the register (the store above may not be visible yet)
    auto frame = [&](int t) -> double2 {
        const int s = (p.oldest + t) & 7;
        if (p.cur_pos && s == p.push_slot) return pushed;
        return hist[((long)e * 8 + s) * N + i];
    };
    f32x4 h[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, c[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double2 prev = make_double2(0, 0), cur = frame(0);
    float lrx = 0, lry = 0;
#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        const double2 nxt = frame(t < 7 ? t + 1 : 7);            // in flight during this step's arithmetic
        // relative displacement in float64 on the rounded values, then float32 (world_model.py:190-206)
        const float rx = t == 0 ? 0.0f : (float)(cur.x - prev.x);
        const float ry = t == 0 ? 0.0f : (float)(cur.y - prev.y);
        prev = cur; cur = nxt; lrx = rx; lry = ry;
        int ln = lane;
#ifndef MCN_ENC_HOIST
        asm volatile("" : "+v"(ln));        // the LDS reads stay inside the loop (hoisted, they pin 100 registers)
#endif
        f32x4 cat[3] = {tile_xy(rx, ry, q), h[0], h[1]};
        f32x4 g[8];
        dense_lds<3, 8, false, 1>(cat, g, s_wl, s_bl, ln);
        lstm_update(g, h, c);
    }
    if (valid) {
        float4 *dst = reinterpret_cast<float4 *>(p.henc + ped * 32);
        dst[q] = make_float4(h[0][0], h[0][1], h[0][2], h[0][3]);           // features 4q .. 4q+3
        dst[4 + q] = make_float4(h[1][0], h[1][1], h[1][2], h[1][3]);       // features 16+4q ..
        if (q == 0)
            reinterpret_cast<float4 *>(p.last)[ped] = make_float4((float)prev.x, (float)prev.y, lrx, lry);
        if (p.pooling && q < 2) reinterpret_cast<float4 *>(p.pool)[ped * 2 + q] = make_float4(0, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// PoolHiddenNet (sgan/models.py:167-232).  For the ordered pair (i, k) the first layer sees [emb(P_k - P_i), h_k]:
// its h_k half (+ bias) does not depend on i, and emb is linear, folded into the layer by the packer.  A wavefront
// carries 16 PARTNERS k on the MFMA column index and walks over kPoolIC pedestrians i of the partners' scenes; per
// 16-feature tile n of the 512-wide hidden layer
//     U(n)  = b1' + W1h[n] h_k                      once per partner                           8 MFMAs
//     a_i   = U(n) + W1r[n] (P_k - P_i)             per pair: x and y share one k-step         1 MFMA
//     o_i  += W2[:, n] relu(a_i)                    per pair, 8 outputs in a 16-row tile       4 MFMAs
// i.e. 8 + 5 kPoolIC MFMAs instead of the 16 kPoolIC of the layer as written, and the hidden layer never leaves the
// accumulators.  All pool-net weights (128 KiB of A-operand fragments) sit in LDS for the whole kernel: one
// workgroup per CU, no staging barriers.  The float32 matrix instruction does not overlap with other vector work
// of its SIMD (measured: tools/microbench/mfma_issue.hip -- every VALU / LDS instruction beside the MFMAs adds its
// issue cycles, with one or with two wavefronts per SIMD), so the loop carries nothing but the MFMAs, one integer max
// per hidden element (ReLU) and five LDS reads per tile, issued a tile ahead.
// The max over partners is a segmented DPP scan over the 16 columns (scenes are contiguous) followed by one atomic
// max per scene part: ReLU outputs are >= +0, so their bit patterns order like unsigned integers and 0 is neutral.
// A work unit is (16-partner tile, kPoolIC pedestrians i): E x N = 4096 x 10 gives 5 120 units = 5 per SIMD.
#ifndef MCN_POOL_IC
#define MCN_POOL_IC 5
#endif
#ifndef MCN_POOL_WAVES
#define MCN_POOL_WAVES 16
#endif
constexpr int kPoolIC = MCN_POOL_IC;
constexpr int kPoolWaves = MCN_POOL_WAVES;

// Diagnostic build only (tools/pool_clock.py): shader-clock and 100 MHz real-time counters of every wavefront, to a
// buffer nothing else reads.  [0] shader clock, [1] real time at entry, [2] after the LDS fill, [3] after the first
// unit, [4] / [5] around its hidden-tile loop, [6] shader clock, [7] real time at exit.
#ifdef MCN_DIAG
__device__ unsigned long long g_pool_clock[256 * kPoolWaves * 8];
#define POOL_CLOCK(slot)                                                                                         \
    do {                                                                                                         \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) {                                                       \
            unsigned long long *d_ = g_pool_clock + (blockIdx.x * kPoolWaves + (threadIdx.x >> 6)) * 8;          \
            if ((slot) == 0 || (slot) == 6) d_[(slot)] = __builtin_amdgcn_s_memtime();                           \
            d_[(slot) == 0 ? 1 : ((slot) == 6 ? 7 : (slot))] = __builtin_amdgcn_s_memrealtime();                 \
        }                                                                                                        \
    } while (0)
int read_pool_clock(void *dst, size_t bytes)
{
    if (bytes > sizeof(g_pool_clock)) bytes = sizeof(g_pool_clock);
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pool_clock), bytes) == hipSuccess ? (int)(bytes / 8) : -1;
}
#else
#define POOL_CLOCK(slot)
#endif

template <int CTRL>
__device__ __forceinline__ int dpp_row(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

__device__ __forceinline__ f32x4 as_tile(const float4 v) { return (f32x4){v.x, v.y, v.z, v.w}; }

__global__ __launch_bounds__(kPoolWaves * 64) void sgan_pool_kernel(const SganParams p)
{
    __shared__ float4 s_w1[32 * 3 * 64];      // mlp_pre_pool.0 fragments [n][displacement, h lo, h hi][lane]  96 KiB
    __shared__ float4 s_w2[32 * 64];          // mlp_pre_pool.2 fragments [k tile][lane]                       32 KiB
    __shared__ float4 s_b1[32 * 4];           // biases by (tile, lane group): features 16n + 4q .. + 3
    __shared__ float4 s_b2[4];
    __shared__ int s_rank[4];
    const int tid = threadIdx.x;
    POOL_CLOCK(0);
    if (tid < 4) s_rank[tid] = 0;
    __syncthreads();
    for (int i = tid; i < 32 * 3 * 64; i += kPoolWaves * 64) s_w1[i] = p.f.w_p1[i];
    for (int i = tid; i < 32 * 64; i += kPoolWaves * 64) s_w2[i] = p.f.w_p2[i];
    if (tid < 128) s_b1[tid] = p.f.b_p1[(tid >> 2) * 64 + (tid & 3) * 16];      // column 0 of lane group tid & 3
    if (tid < 4) s_b2[tid] = p.f.b_p2[tid * 16];

    const int lane = tid & 63;
    // slot of this wavefront in the workgroup's unit order: rank among the wavefronts of its SIMD first, so that a
    // last, partial pass over the units lands evenly on the four matrix pipes whatever the wave -> SIMD placement is
    const int simd = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));         // HW_ID.SIMD_ID
    int rank = 0;
    if (lane == 0) rank = atomicAdd(&s_rank[simd], 1);
    rank = __builtin_amdgcn_readfirstlane(rank);
    __syncthreads();
    int slot = 0;                             // wavefronts ordered by (rank, simd): dense 0 .. kPoolWaves-1 however
#pragma unroll                                // many of them each SIMD received
    for (int k = 0; k < 4; ++k) {
        const int cnt = s_rank[k];
        slot += (cnt < rank ? cnt : rank) + ((k < simd && cnt > rank) ? 1 : 0);
    }
    const int j = lane & 15, q = lane >> 4;
    const int N = p.N;
    const long nped = (long)p.E * N;
    const int upt = (N + kPoolIC - 1) / kPoolIC;                   // units per 16-partner tile
    const long nunits = ((nped + 15) / 16) * upt;
    const float4 *last4 = reinterpret_cast<const float4 *>(p.last);
    const float4 *henc4 = reinterpret_cast<const float4 *>(p.henc);
    unsigned int *pool = reinterpret_cast<unsigned int *>(p.pool);

    struct TileW { float4 h0, h1, v; float e; f32x4 b; };          // LDS operands of one hidden tile
    auto tile_w = [&](int n) {
        TileW w;
        w.h0 = s_w1[(n * 3 + 1) * 64 + lane]; w.h1 = s_w1[(n * 3 + 2) * 64 + lane];
        w.e = reinterpret_cast<const float *>(s_w1 + (n * 3) * 64 + lane)[0];        // k-step 0 of the displacement tile
        w.v = s_w2[n * 64 + lane]; w.b = as_tile(s_b1[n * 4 + q]);
        return w;
    };
    auto el = [](const float4 &v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); };

    POOL_CLOCK(2);
    int pass_ = 0; (void)pass_;
    for (long u = blockIdx.x + (long)gridDim.x * slot; u < nunits; u += (long)gridDim.x * kPoolWaves) {
        const long tile = u / upt;
        const int i0 = (int)(u - tile * upt) * kPoolIC;
        long k = tile * 16 + j;
        const bool kvalid = k < nped;
        if (!kvalid) k = nped - 1;
        const int e = (int)(k / N), kk = (int)(k - (long)e * N);
        int ne = N;
        if (p.hcount) { ne = p.hcount[e]; ne = ne < 1 ? 1 : (ne > N ? N : ne); }
        const bool partner = kvalid && kk < ne;                    // this column is a partner the scene really has
        const float4 own = last4[k];
        const f32x4 hk[2] = {as_tile(henc4[k * 8 + q]), as_tile(henc4[k * 8 + 4 + q])};
        float rel[kPoolIC];                                        // B operand of the displacement k-step
        f32x4 o[kPoolIC];
#pragma unroll
        for (int m = 0; m < kPoolIC; ++m) {
            const int ii = i0 + m < N ? i0 + m : N - 1;
            const float4 their = last4[(long)e * N + ii];
            rel[m] = tile_xy(own.x - their.x, own.y - their.y, q)[0];               // P_k - P_i (models.py:221)
            o[m] = as_tile(s_b2[q]);
        }
        if (pass_ == 0) POOL_CLOCK(4);
        // Software-pipelined over the 32 hidden tiles so that no MFMA waits for an operand: beside tile n's a_i go
        // o_i += W2[n-1] relu(a_i(n-1)) and the chain U(n+1); the LDS operands of tile n + 1 are read a tile ahead.
        TileW wc = tile_w(0);
        f32x4 U = wc.b;
#pragma unroll
        for (int s = 0; s < 4; ++s) U = __builtin_amdgcn_mfma_f32_16x16x4f32(el(wc.h0, s), hk[0][s], U, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) U = __builtin_amdgcn_mfma_f32_16x16x4f32(el(wc.h1, s), hk[1][s], U, 0, 0, 0);
        f32x4 a[kPoolIC];
        float4 vprev = wc.v;
        {   // tile 0: a(0) and the chain U(1)
            const TileW wn = tile_w(1);
            f32x4 Un = wn.b;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < kPoolIC) a[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc.e, rel[c], U, 0, 0, 0);
                Un = __builtin_amdgcn_mfma_f32_16x16x4f32(el(c < 4 ? wn.h0 : wn.h1, c & 3), hk[c >> 2][c & 3], Un, 0, 0, 0);
            }
#pragma unroll
            for (int m = 8; m < kPoolIC; ++m) a[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc.e, rel[m], U, 0, 0, 0);
            U = Un; wc = wn;
        }
#pragma unroll 1
        for (int n = 1; n < 32; ++n) {
            const TileW wn = tile_w(n < 31 ? n + 1 : 31);          // (tile 31 computes a U nobody reads: 8 of 1 080 MFMAs)
            __builtin_amdgcn_sched_barrier(0);                     // the reads are issued here, a tile ahead of their use
            f32x4 r[kPoolIC];
#pragma unroll
            for (int m = 0; m < kPoolIC; ++m)
#pragma unroll
                for (int c = 0; c < 4; ++c) r[m][c] = relu_f32(a[m][c]);
            f32x4 Un = wn.b;
            auto o_mm = [&](int s, int m) { o[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(el(vprev, s), r[m][s], o[m], 0, 0, 0); };
            auto u_mm = [&](int c) {       // c-th product of the chain b1' + W1h h_k (h tile c / 4, k-step c % 4)
                Un = __builtin_amdgcn_mfma_f32_16x16x4f32(el(c < 4 ? wn.h0 : wn.h1, c & 3), hk[c >> 2][c & 3], Un, 0, 0, 0);
            };
            // issue order: every accumulator rests for two or more MFMAs between its products, and the chain U(n+1)
            // starts ten MFMAs behind the LDS reads it depends on
#pragma unroll
            for (int m = 0; m < kPoolIC; ++m) a[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc.e, rel[m], U, 0, 0, 0);
#pragma unroll
            for (int m = 0; m < kPoolIC; ++m) o_mm(0, m);
            __builtin_amdgcn_sched_barrier(0);
            static_assert(kPoolIC >= 5, "the interleave below names five o accumulators");
            u_mm(0); o_mm(1, 0); o_mm(1, 1); u_mm(1); o_mm(1, 2); o_mm(1, 3); u_mm(2); o_mm(1, 4);
#pragma unroll
            for (int m = 5; m < kPoolIC; ++m) o_mm(1, m);
            o_mm(2, 0); u_mm(3); o_mm(2, 1); o_mm(2, 2); u_mm(4); o_mm(2, 3); o_mm(2, 4); u_mm(5);
#pragma unroll
            for (int m = 5; m < kPoolIC; ++m) o_mm(2, m);
            o_mm(3, 0); o_mm(3, 1); u_mm(6); o_mm(3, 2); o_mm(3, 3); u_mm(7); o_mm(3, 4);
#pragma unroll
            for (int m = 5; m < kPoolIC; ++m) o_mm(3, m);
            __builtin_amdgcn_sched_barrier(0);
            U = Un; vprev = wc.v; wc = wn;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < kPoolIC; ++m)
                o[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(el(vprev, s), relu_f32(a[m][s]), o[m], 0, 0, 0);
        if (pass_ == 0) POOL_CLOCK(5);
        // max over the partners of each scene part held by this tile, then one atomic per (scene part, i, feature);
        // non-negative floats compare like integers
        const bool seg_end = j == 15 || kk == N - 1 || k == nped - 1;
#pragma unroll
        for (int m = 0; m < kPoolIC; ++m) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float oc = o[m][c];                          // (bit-casting the vector element expression itself reads element 0)
                int v = __builtin_bit_cast(int, oc);
                v = (partner && v > 0) ? v : 0;                    // ReLU; absent partners contribute the neutral 0
                int g;
                g = dpp_row<0x111>(v); v = kk >= 1 ? (g > v ? g : v) : v;   // row_shr:1 (own value where the row ends)
                g = dpp_row<0x112>(v); v = kk >= 2 ? (g > v ? g : v) : v;
                g = dpp_row<0x114>(v); v = kk >= 4 ? (g > v ? g : v) : v;
                g = dpp_row<0x118>(v); v = kk >= 8 ? (g > v ? g : v) : v;
                if (seg_end && q < 2 && i0 + m < N)
                    atomicMax(pool + ((long)e * N + i0 + m) * 8 + 4 * q + c, (unsigned int)v);
            }
        }
        if (pass_ == 0) POOL_CLOCK(3);
        ++pass_;
    }
    POOL_CLOCK(6);
}

__global__ __launch_bounds__(kSganWaves * 64) void sgan_decode_kernel(const SganParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const long nped = (long)p.E * p.N;
    const long ped0 = ((long)blockIdx.x * kSganWaves + wave) * 16;
    if (ped0 >= nped) return;
    long ped = ped0 + j;
    const bool valid = ped < nped;
    if (!valid) ped = nped - 1;
    const int e = (int)(ped / p.N);
    const float4 mine = reinterpret_cast<const float4 *>(p.last)[ped];       // last_pos.xy, last_rel.xy
    f32x4 hi[2];
    {
        const float4 *src = reinterpret_cast<const float4 *>(p.henc + ped * 32);
        hi[0] = as_tile(src[q]);
        hi[1] = as_tile(src[4 + q]);
    }
    f32x4 ctx[2];
    if (p.pooling) {
        f32x4 pool = {0, 0, 0, 0};
        if (q < 2) pool = as_tile(reinterpret_cast<const float4 *>(p.pool)[ped * 2 + q]);
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
    f32x4 cat[3] = {tile_xy(mine.z, mine.w, q), dh[0], dh[1]};
    f32x4 g[8];
    dense<3, 8, false, 1>(cat, g, p.f.w_dlstm, p.f.b_dlstm, lane);
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

static int device_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    return cus;
}

int launch_sgan(const mcn_sgan_net *net, double *hist, int push_slot, int oldest, const double *cur_pos,
                const float *noise, const int32_t *hcount, void *workspace, double *out_vel, float *out_rel,
                double time_step, int E, int N, hipStream_t stream)
{
    SganParams p;
    static_assert(sizeof(SganFrags) == 14 * sizeof(void *), "fragment table size");
    const float4 *const *src = reinterpret_cast<const float4 *const *>(net);
    const float4 **dst = reinterpret_cast<const float4 **>(&p.f);
    for (int k = 0; k < 14; ++k) dst[k] = src[k];
    p.hist = hist; p.cur_pos = cur_pos; p.noise = noise; p.hcount = hcount;
    p.henc = reinterpret_cast<float *>(workspace);
    p.last = p.henc + (size_t)E * N * 32;
    p.pool = p.last + (size_t)E * N * 4;
    p.out_vel = out_vel; p.out_rel = out_rel;
    p.E = E; p.N = N; p.pooling = net->pooling; p.push_slot = push_slot; p.oldest = oldest; p.time_step = time_step;
    const long tiles = ((long)E * N + 15) / 16;
    const int blocks = (int)((tiles + kSganWaves - 1) / kSganWaves);
    hipLaunchKernelGGL(sgan_encode_kernel, dim3(blocks), dim3(kSganWaves * 64), 0, stream, p);
    if (p.pooling) {
        const long units = tiles * ((N + kPoolIC - 1) / kPoolIC);
        const long want = (units + kPoolWaves - 1) / kPoolWaves;
        const int grid = (int)(want < device_cus() ? want : device_cus());
        hipLaunchKernelGGL(sgan_pool_kernel, dim3(grid), dim3(kPoolWaves * 64), 0, stream, p);
    }
    hipLaunchKernelGGL(sgan_decode_kernel, dim3(blocks), dim3(kSganWaves * 64), 0, stream, p);
    return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
}

}  // namespace mcn
