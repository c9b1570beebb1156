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
#include <type_traits>
#include <utility>

namespace mcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ReLU as ONE vector instruction: fmaxf(x, 0) compiles to two under IEEE rules (v_max x, x to quiet a signalling NaN,
// then v_max 0, x), and so does every float form the optimiser recognises as a maximum.  As signed integers the
// negative floats (and -0) are < 0 and the others keep their order, so max(bits, 0) is the same function.
__device__ __forceinline__ float relu_f32(float x)
{
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// out[n] = act( bias[n] (+ init) + sum_t in[t] x W[n][t] ), two accumulators in flight per step so
// back-to-back dependent MFMAs (40-cycle latency vs 32-cycle issue) never stall the pipe.
// FIRST: k-steps of input tile 0 that carry anything (a 2-vector packed into slots 0 and 4 needs one).
template <int KT, int NT, bool RELU, int FIRST = 4>
__device__ __forceinline__ void dense(const f32x4 (&in)[KT], f32x4 (&out)[NT], const float4 *__restrict__ wf,
                                      const float4 *__restrict__ bf, int lane)
{
#pragma unroll
    for (int n = 0; n < NT; n += 2) {
        const bool two = (n + 1 < NT);
        f32x4 a0, a1 = {0, 0, 0, 0};
        { const float4 b = bf[n * 64 + lane]; a0 = (f32x4){b.x, b.y, b.z, b.w}; }
        if (two) { const float4 b = bf[(n + 1) * 64 + lane]; a1 = (f32x4){b.x, b.y, b.z, b.w}; }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float4 w0 = wf[(n * KT + t) * 64 + lane];
            float4 w1 = make_float4(0, 0, 0, 0);
            if (two) w1 = wf[((n + 1) * KT + t) * 64 + lane];
            const int steps = t == 0 ? FIRST : 4;
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, in[t][0], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, in[t][0], a1, 0, 0, 0);
            if (steps > 1) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, in[t][1], a0, 0, 0, 0);
                if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, in[t][1], a1, 0, 0, 0);
            }
            if (steps > 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, in[t][2], a0, 0, 0, 0);
                if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, in[t][2], a1, 0, 0, 0);
            }
            if (steps > 3) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, in[t][3], a0, 0, 0, 0);
                if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, in[t][3], a1, 0, 0, 0);
            }
        }
        if (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { a0[r] = relu_f32(a0[r]); a1[r] = relu_f32(a1[r]); }
        }
        out[n] = a0;
        if (two) out[n + 1] = a1;
    }
}

// Same layer with its fragments RESIDENT in LDS (small networks whose weights fit next to the workgroup for the whole
// kernel): w = [NT][KT][64] float4 as packed by mcn_pack_linear, bq = biases by (output tile, lane group): the float4
// of features 16n + 4q .. + 3 at bq[4n + q] (column 0 of the packed bias fragment).
template <int KT, int NT, bool RELU, int FIRST = 4>
__device__ __forceinline__ void dense_lds(const f32x4 (&in)[KT], f32x4 (&out)[NT], const float4 *w, const float4 *bq,
                                          int lane)
{
    const int q = lane >> 4;
#pragma unroll
    for (int n = 0; n < NT; n += 2) {
        const bool two = (n + 1 < NT);
        f32x4 a0, a1 = {0, 0, 0, 0};
        { const float4 b = bq[4 * n + q]; a0 = (f32x4){b.x, b.y, b.z, b.w}; }
        if (two) { const float4 b = bq[4 * n + 4 + q]; a1 = (f32x4){b.x, b.y, b.z, b.w}; }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float4 w0 = w[(n * KT + t) * 64 + lane];
            float4 w1 = make_float4(0, 0, 0, 0);
            if (two) w1 = w[((n + 1) * KT + t) * 64 + lane];
            const int steps = t == 0 ? FIRST : 4;
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, in[t][0], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, in[t][0], a1, 0, 0, 0);
            if (steps > 1) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, in[t][1], a0, 0, 0, 0);
                if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, in[t][1], a1, 0, 0, 0);
            }
            if (steps > 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, in[t][2], a0, 0, 0, 0);
                if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, in[t][2], a1, 0, 0, 0);
            }
            if (steps > 3) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, in[t][3], a0, 0, 0, 0);
                if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, in[t][3], a1, 0, 0, 0);
            }
        }
        if (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { a0[r] = relu_f32(a0[r]); a1[r] = relu_f32(a1[r]); }
        }
        out[n] = a0;
        if (two) out[n + 1] = a1;
    }
}

// copy a packed layer into LDS (call from every thread of the workgroup, barrier afterwards)
template <int THREADS>
__device__ __forceinline__ void lds_fill_layer(float4 *w_lds, float4 *bq_lds, const float4 *__restrict__ wf,
                                               const float4 *__restrict__ bf, int KT, int NT, int tid)
{
    for (int i = tid; i < NT * KT * 64; i += THREADS) w_lds[i] = wf[i];
    for (int i = tid; i < NT * 4; i += THREADS) bq_lds[i] = bf[(i >> 2) * 64 + (i & 3) * 16];
}

// ------------------------------------------------------------------------------------------------
// LDS-staged variant: the workgroup's waves all run the same layer on different batch tiles, so the
// weight fragments are fetched from L2 ONCE per workgroup into a double-buffered LDS stage (one chunk =
// the fragments of two output tiles, <= 20 KiB) and every wave reads its A operands from there with
// conflict-free ds_read_b128 (consecutive lanes, consecutive 16-B slots).  While chunk c is being
// multiplied, the loads of chunk c+1 are already in flight (issued before the MFMAs, written to the other
// buffer after them); one barrier per chunk.  Every thread of the workgroup must call this uniformly.
#ifndef MCN_STAGE_WAVES
#define MCN_STAGE_WAVES 4
#endif
constexpr int kStageThreads = 64 * MCN_STAGE_WAVES;
// output tiles per staged chunk: with 4 wavefronts per workgroup (two workgroups per CU: 2 x 2 buffers + bias tails =
// 2 x 72 KiB of the CU's 160 KiB) 4 for layers with <= 7 input tiles and 2 for the 10-tile ones, i.e. chunks of at most
// 32 KiB; with 8 wavefronts (one workgroup per CU) twice that.
constexpr int kChunkScale = MCN_STAGE_WAVES >= 8 ? 2 : 1;
__host__ __device__ constexpr int chunk_tiles(int KT) { return kChunkScale * (KT > 16 ? 1 : (KT > 7 ? 2 : 4)); }
constexpr int kStageFloat4 = kChunkScale * 32 * 64;              // max over layers of chunk_tiles(KT) * KT * 64
// bias fragments of the chunk's output tiles -- and, for the flow / x3 layers that place their biases themselves, simply
// the tail of the stage buffer: MCN_STAGE_BIAS_ROWS = 512 makes a buffer 40 KiB (two workgroups then use all 160 KiB)
#ifndef MCN_STAGE_BIAS_ROWS
#define MCN_STAGE_BIAS_ROWS 256
#endif
constexpr int kStageBias = kChunkScale * MCN_STAGE_BIAS_ROWS;

struct WeightStage {
    float4 *buf;      // LDS, 2 * (kStageFloat4 + kStageBias) float4
    int tid;          // threadIdx.x, 0 .. kStageThreads-1
};

// MCN_DENSE_PIPE = 1 (default): the chunk loop below is software-pipelined by hand.  Left to itself the compiler
// issues the ds_read_b128 pair of an 8-MFMA group right before that group's MFMAs (it reuses the registers of the
// previous fragments, so the reads cannot start earlier) and the wavefront sits out an LDS round trip (~100+ cycles)
// per 256 cycles of matrix work: a lone wavefront reaches ~75 % of the pipe and two of them still leave 10 % idle
// (round-3 phase table: 100-wide layers at 2.2 x their MFMA time).  Here the A fragments live in a two-slot register
// ring: the reads of group g + 1 are issued BEFORE the eight MFMAs of group g (256 cycles of cover), across output
// tile pairs and across staged chunks; __builtin_amdgcn_sched_barrier(0) keeps the scheduler from sinking them back.
// The chunk barrier sits before the LAST group of a chunk (all reads of the chunk have been issued and have
// returned by then), so the next chunk's first fragments are fetched under that group's MFMAs as well.
#ifndef MCN_DENSE_PIPE
#define MCN_DENSE_PIPE 1
#endif

// LAST / LAST2: k-steps actually needed in the last / second-to-last input tile (ragged tiles packed "q first",
// see mcn_pack_linear): the skipped steps would multiply zeros.
template <int KT, int NT, bool RELU, bool HAS_INIT, int LAST = 4, int LAST2 = 4>
__device__ __forceinline__ void dense_staged(const f32x4 (&in)[KT], const f32x4 *init, f32x4 (&out)[NT],
                                             const float4 *__restrict__ wf, const float4 *__restrict__ bf,
                                             const WeightStage &S, int lane)
{
    constexpr int kChunkTiles = chunk_tiles(KT);
    constexpr int CH = kChunkTiles * KT * 64;             // float4 per full chunk
    static_assert(CH <= kStageFloat4, "chunk does not fit the LDS stage");
    constexpr int TOTAL = NT * KT * 64;
    constexpr int NCH = (NT + kChunkTiles - 1) / kChunkTiles;
    constexpr int PER = (CH + kStageThreads - 1) / kStageThreads;
    constexpr int BCH = kChunkTiles * 64;                 // bias float4 per chunk
    // LDS-DMA staging: the stage image is lane-linear (thread i's float4 lands at float4 slot i), which is
    // exactly what global_load_lds writes (wave-uniform LDS base + lane * 16 B), so the weights go L2 -> LDS
    // without passing through (and pinning) VGPRs.  Biases ride along in a small tail region so that no ordinary
    // global load sits between a DMA and the barrier that retires it.
    const int wave_base = S.tid & ~63; (void)wave_base;
#if MCN_DENSE_PIPE
    // straight-line staging: every thread issues every DMA (a chunk is a whole number of 256-thread rounds); the source
    // index of the ragged last chunk / of bias slots beyond the layer is clamped, the duplicates land in stage slots
    // nobody reads.  No exec-mask branches: the DMA issue can sit between the MFMAs of a group.
    static_assert(CH % kStageThreads == 0 || NCH == 1, "a staged chunk is a whole number of DMA rounds");
    auto stage = [&](int c, int b) {
        // (opaque copy: the per-lane addresses are derived here, a few integer instructions per DMA, instead of being
        //  hoisted out of the tile loop as dozens of live 64-bit values)
        int tid_ = S.tid;
        asm volatile("" : "+v"(tid_));
        const int wave_base = tid_ & ~63;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            int i = c * CH + tid_ + k * kStageThreads;
            if ((c + 1) * CH > TOTAL || CH % kStageThreads != 0) i = i < TOTAL - 1 ? i : TOTAL - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(wf + i),
                (__attribute__((address_space(3))) void *)(S.buf + b * (kStageFloat4 + kStageBias) + k * kStageThreads + wave_base),
                16, 0, 0);
        }
        if (!HAS_INIT) {
            int i = c * BCH + tid_;
            i = i < NT * 64 - 1 ? i : NT * 64 - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(bf + i),
                (__attribute__((address_space(3))) void *)(S.buf + b * (kStageFloat4 + kStageBias) + kStageFloat4 + wave_base),
                16, 0, 0);
        }
    };
#else
    auto stage = [&](int c, int b) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = S.tid + k * kStageThreads;
            if (i < CH && c * CH + i < TOTAL)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(wf + c * CH + i),
                    (__attribute__((address_space(3))) void *)(S.buf + b * (kStageFloat4 + kStageBias) + k * kStageThreads + wave_base),
                    16, 0, 0);
        }
        if (!HAS_INIT && S.tid < BCH && c * BCH + S.tid < NT * 64)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(bf + c * BCH + S.tid),
                (__attribute__((address_space(3))) void *)(S.buf + b * (kStageFloat4 + kStageBias) + kStageFloat4 + wave_base),
                16, 0, 0);
    };
#endif
    stage(0, 0);
    __syncthreads();                                      // drains vmcnt (the DMA) and orders it before the reads
#if MCN_DENSE_PIPE
    // flattened (output tile pair, input tile) groups of the whole layer, eight MFMAs each
    constexpr int PAIRS = (kChunkTiles + 1) / 2;          // tile pairs per staged chunk
    constexpr int NP = (NT + 1) / 2;                      // tile pairs of the layer
    constexpr int NG = NP * KT;
    float4 ra[2], rb[2], ba = make_float4(0, 0, 0, 0), bb = make_float4(0, 0, 0, 0);
    rb[0] = rb[1] = make_float4(0, 0, 0, 0);
    auto issue = [&](int g) {                             // LDS -> register ring slot g & 1 (and the pair's biases)
        const int pr = g / KT, t = g - pr * KT;
        const int c = pr / PAIRS, h2 = 2 * (pr - c * PAIRS), n = 2 * pr;
        const float4 *wc = S.buf + (c & 1) * (kStageFloat4 + kStageBias);
        const float4 *w = wc + h2 * KT * 64;
        ra[g & 1] = w[t * 64 + lane];
        if (n + 1 < NT) rb[g & 1] = w[(KT + t) * 64 + lane];
        if (t == 0 && !HAS_INIT) {
            const float4 *bc = wc + kStageFloat4;
            ba = bc[h2 * 64 + lane];
            if (n + 1 < NT) bb = bc[(h2 + 1) * 64 + lane];
        }
    };
    issue(0);
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int pr = g / KT, t = g - pr * KT;
        const int c = pr / PAIRS, n = 2 * pr;
        const bool two = (n + 1 < NT);
        const bool chunk_first = (t == 0) && (pr == c * PAIRS);
        // last group that reads chunk c: the chunk's last pair (or the layer's), last input tile
        const bool chunk_last = (t == KT - 1) && ((pr == c * PAIRS + PAIRS - 1) || (pr == NP - 1));
        if (chunk_last) __syncthreads();      // every read of chunk c has been issued (and is back: the data of THIS
                                              // group is awaited here); the next chunk's DMA has landed
        if (t == 0) {
            if (HAS_INIT) { a0 = init[n]; if (two) a1 = init[n + 1]; }
            else { a0 = (f32x4){ba.x, ba.y, ba.z, ba.w}; if (two) a1 = (f32x4){bb.x, bb.y, bb.z, bb.w}; }
        }
        const float4 w0 = ra[g & 1], w1 = rb[g & 1];
        const int steps = (t == KT - 1) ? LAST : ((t == KT - 2) ? LAST2 : 4);
        // The group's first MFMA comes BEFORE the next group's reads are issued: the compiler cannot count LDS reads
        // individually while an LDS-DMA is in flight (it models global_load_lds as a FLAT access, after which every
        // LDS wait is lgkmcnt(0)), so the wait for this group's fragments must not see the next group's reads yet.
        // This group's fragments were requested seven MFMAs (~230 cycles) ago: the wait is free.
        __builtin_amdgcn_sched_barrier(0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, in[t][0], a0, 0, 0, 0);
        if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, in[t][0], a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG) issue(g + 1);
        if (chunk_first && c + 1 < NCH) stage(c + 1, (c + 1) & 1);          // DMA of the next chunk, other buffer
        __builtin_amdgcn_sched_barrier(0);
        if (steps > 1) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, in[t][1], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, in[t][1], a1, 0, 0, 0);
        }
        if (steps > 2) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, in[t][2], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, in[t][2], a1, 0, 0, 0);
        }
        if (steps > 3) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, in[t][3], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, in[t][3], a1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t == KT - 1) {
            if (RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { a0[r] = relu_f32(a0[r]); a1[r] = relu_f32(a1[r]); }
            }
            out[n] = a0;
            if (two) out[n + 1] = a1;
        }
    }
#else
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) stage(c + 1, (c + 1) & 1);
        const float4 *wc = S.buf + (c & 1) * (kStageFloat4 + kStageBias);
        const float4 *bc = wc + kStageFloat4;
#pragma unroll
        for (int h2 = 0; h2 < kChunkTiles; h2 += 2) {
            const int n = kChunkTiles * c + h2;
            if (n < NT) {
                const bool two = (n + 1 < NT);
                const float4 *w = wc + h2 * KT * 64;
                f32x4 a0, a1 = {0, 0, 0, 0};
                if (HAS_INIT) {
                    a0 = init[n];
                    if (two) a1 = init[n + 1];
                } else {
                    { const float4 b = bc[h2 * 64 + lane]; a0 = (f32x4){b.x, b.y, b.z, b.w}; }
                    if (two) { const float4 b = bc[(h2 + 1) * 64 + lane]; a1 = (f32x4){b.x, b.y, b.z, b.w}; }
                }
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 w0 = w[t * 64 + lane];
                    float4 w1 = make_float4(0, 0, 0, 0);
                    if (two) w1 = w[(KT + t) * 64 + lane];
                    const int steps = (t == KT - 1) ? LAST : ((t == KT - 2) ? LAST2 : 4);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, in[t][0], a0, 0, 0, 0);
                    if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, in[t][0], a1, 0, 0, 0);
                    if (steps > 1) {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, in[t][1], a0, 0, 0, 0);
                        if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, in[t][1], a1, 0, 0, 0);
                    }
                    if (steps > 2) {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, in[t][2], a0, 0, 0, 0);
                        if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, in[t][2], a1, 0, 0, 0);
                    }
                    if (steps > 3) {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, in[t][3], a0, 0, 0, 0);
                        if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, in[t][3], a1, 0, 0, 0);
                    }
                }
                if (RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a0[r] = relu_f32(a0[r]); a1[r] = relu_f32(a1[r]); }
                }
                out[n] = a0;
                if (two) out[n + 1] = a1;
            }
        }
        __syncthreads();
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// dense_flow: the staged layer as ONE continuous weight stream across layers (round 4).
// dense_staged starts every layer with "DMA chunk 0, wait, barrier": an L2 -> LDS round trip (~1 500 cycles) during
// which the wavefront does nothing, 40 times per 16-pair tile of the SARL look-ahead.  Here a layer also issues the DMA
// of the FIRST chunk of the layer that follows it (NextChunk) -- at the first group of its own last chunk, into the
// stage buffer that chunk does not use -- and the barrier before its last group (which every layer has anyway) is
// what makes that chunk visible.  A layer therefore begins reading at once; per tile there is one barrier per staged
// chunk and no other synchronisation.  The double buffer alternates across layers: WeightFlow.parity is which buffer
// holds chunk 0 of the layer about to run (chunk c of it sits in buffer (parity + c) & 1).
struct NextChunk {
    const float4 *w; int n_w;       // fragments of the following layer's chunk 0 (n_w float4; 0 = nothing follows)
    const float4 *b; int n_b;       // its bias fragments (n_b float4; 0 = accumulators start from `init`)
    int b_at;                       // float4 offset of the bias fragments inside the stage buffer
};
// Output tiles per staged chunk of a dense_flow layer: as chunk_tiles(), except that a one-k-tile layer (the 13-wide
// input layer: 10 output tiles x 1 KiB) is ONE chunk -- three barriers for 40 MFMAs were most of that layer's time.
// The bias fragments sit right behind the chunk's weight fragments (every layer's chunk + biases fits the 32-KiB
// stage), so the chunk size is not tied to a fixed bias region.
__host__ __device__ constexpr int flow_chunk_tiles(int KT) { return KT <= 2 ? kChunkScale * 12 : chunk_tiles(KT); }
template <int KT, int NT, bool HAS_INIT>
__device__ __forceinline__ NextChunk first_chunk(const float4 *wf, const float4 *bf)
{
    constexpr int CH = flow_chunk_tiles(KT) * KT * 64, TOTAL = NT * KT * 64, BCH = flow_chunk_tiles(KT) * 64;
    static_assert(CH + (HAS_INIT ? 0 : BCH) <= kStageFloat4 + kStageBias, "chunk + biases do not fit the LDS stage");
    return NextChunk{wf, CH < TOTAL ? CH : TOTAL, HAS_INIT ? nullptr : bf, HAS_INIT ? 0 : (BCH < NT * 64 ? BCH : NT * 64), CH};
}
struct WeightFlow {
    float4 *buf;      // LDS, 2 * (kStageFloat4 + kStageBias) float4
    int tid;          // threadIdx.x
    int parity;       // wave-uniform: buffer of the next layer's chunk 0
};
constexpr int kStageBuf = kStageFloat4 + kStageBias;

#ifndef MCN_X3_WHATIF
#define MCN_X3_WHATIF 0     // timing-only what-if bits (profiles/r04_sarl_x3_whatif.txt); any non-zero value computes garbage
#endif
#ifndef MCN_DMA_SKIP_EMPTY
#define MCN_DMA_SKIP_EMPTY 1
#endif
#ifndef MCN_LEAN_DMA
#define MCN_LEAN_DMA 1
#endif
// One workgroup-wide LDS-DMA of `n` float4 rows src[0 .. n) to dst[0 .. n) in rounds of kStageThreads rows (round 4).
// The what-if builds (profiles/r04_sarl_x3_whatif.txt) showed ISSUING the weight stream -- ~750 global_load_lds per
// wavefront and tile, each with a 64-bit vector address, a clamp, a readfirstlane and an M0 move -- to cost as much as
// the splitting and the barriers together.  Here an instruction is SGPR base + one 32-bit VGPR offset (the saddr form:
// one v_add instead of a 64-bit address), no round past the end of the rows, a clamp only in the round that straddles
// it: 5 instructions per DMA instead of 10.  `n` and the round loop fold
// at compile time at every call site.
struct DmaLane {
    unsigned off;     // this thread's byte offset inside a round: tid * 16 (VGPR)
    int wave_base;    // tid & ~63 (left in a VGPR: read into an SGPR once, the per-round M0 values were hoisted as scalars
                      // and spilled -- 569 VGPR spills)
};
__device__ __forceinline__ DmaLane dma_lane(int tid)
{
    int t = tid;
    asm volatile("" : "+v"(t));                            // (a hoisted copy per call site costs a VGPR each)
    return {static_cast<unsigned>(t) * 16u, t & ~63};
}
template <int MAX_ROUNDS>
__device__ __forceinline__ void dma_rows(const float4 *base, int first, int n, float4 *dst, const DmaLane &L,
                                         int k0 = 0, int k1 = MAX_ROUNDS)
{
    // rows base[first .. first + n); the row offset goes into the VGPR offset (one v_add per instruction, wrapping
    // unsigned so that it cannot be split off again): a scalar base per round would cost an SGPR pair per round and
    // layer -- 541 spilled SGPRs when tried
#pragma unroll
    for (int k = 0; k < MAX_ROUNDS; ++k) {
        const int left = n - k * kStageThreads;            // rows from this round's first to the end
        if (left > 0 && k >= k0 && k < k1) {               // wave-uniform (compile-time at the call sites)
            unsigned off = L.off;
            if (left < kStageThreads) {
                // the round that straddles the end: wavefronts wholly past it issue nothing (a piece costs its issue
                // slot whatever it moves), the one across it clamps
                if (MCN_DMA_SKIP_EMPTY && __builtin_amdgcn_readfirstlane(L.wave_base) >= left) continue;
                off = off < (left - 1) * 16u ? off : (left - 1) * 16u;
            }
            off += static_cast<unsigned>(first + k * kStageThreads) * 16u;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(base) + off),
                (__attribute__((address_space(3))) void *)(dst + k * kStageThreads + L.wave_base), (MCN_X3_WHATIF & 128) ? 4 : 16, 0, 0);
        }
    }
}

// DMA of a layer's chunk 0 into stage buffer `b` (0 / 1, run-time); every thread of the workgroup calls it.
__device__ __forceinline__ void flow_stage_first(const WeightFlow &F, const NextChunk &d, int b)
{
    float4 *dst = F.buf + b * kStageBuf;
#if MCN_LEAN_DMA
    const DmaLane L = dma_lane(F.tid);
    dma_rows<kStageFloat4 / kStageThreads>(d.w, 0, d.n_w, dst, L);
    dma_rows<3>(d.b, 0, d.n_b, dst + d.b_at, L);              // up to 12 bias tiles
#else
    int tid_ = F.tid;
    asm volatile("" : "+v"(tid_));
    const int wave_base = tid_ & ~63;
#pragma unroll
    for (int k = 0; k < kStageFloat4 / kStageThreads; ++k) {
        if (k * kStageThreads < d.n_w) {                                   // wave-uniform
            int i = tid_ + k * kStageThreads;
            i = i < d.n_w - 1 ? i : d.n_w - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(d.w + i),
                                             (__attribute__((address_space(3))) void *)(dst + k * kStageThreads + wave_base),
                                             16, 0, 0);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                          // up to 12 bias tiles
        if (k * kStageThreads < d.n_b) {
            int i = tid_ + k * kStageThreads;
            i = i < d.n_b - 1 ? i : d.n_b - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(d.b + i),
                                             (__attribute__((address_space(3))) void *)(dst + d.b_at + k * kStageThreads + wave_base),
                                             16, 0, 0);
        }
    }
#endif
}

// Precondition: chunk 0 of this layer has been requested into buffer F.parity AND a workgroup barrier (with vmcnt(0))
// has been passed since -- the previous layer's last barrier, or the caller's own at the very beginning.
template <int KT, int NT, bool RELU, bool HAS_INIT, int LAST = 4, int LAST2 = 4>
__device__ __forceinline__ void dense_flow(const f32x4 (&in)[KT], const f32x4 *init, f32x4 (&out)[NT],
                                           const float4 *__restrict__ wf, const float4 *__restrict__ bf,
                                           WeightFlow &F, int lane, const NextChunk &next)
{
    constexpr int kChunkTiles = flow_chunk_tiles(KT);
    constexpr int CH = kChunkTiles * KT * 64;
    constexpr int TOTAL = NT * KT * 64;
    constexpr int NCH = (NT + kChunkTiles - 1) / kChunkTiles;
    constexpr int PER = (CH + kStageThreads - 1) / kStageThreads;
    constexpr int BCH = kChunkTiles * 64;
    constexpr int BPER = (BCH + kStageThreads - 1) / kStageThreads;
    static_assert(CH + (HAS_INIT ? 0 : BCH) <= kStageFloat4 + kStageBias, "chunk + biases do not fit the LDS stage");
    static_assert(CH % kStageThreads == 0 || NCH == 1, "a staged chunk is a whole number of DMA rounds");
    float4 *const bufp[2] = {F.buf + F.parity * kStageBuf, F.buf + (F.parity ^ 1) * kStageBuf};
    auto stage = [&](int c) {                             // chunk c >= 1 of THIS layer -> buffer (parity + c) & 1
        int tid_ = F.tid;
        asm volatile("" : "+v"(tid_));
        const int wave_base = tid_ & ~63;
        float4 *dst = bufp[c & 1];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            int i = c * CH + tid_ + k * kStageThreads;
            if ((c + 1) * CH > TOTAL) i = i < TOTAL - 1 ? i : TOTAL - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wf + i),
                                             (__attribute__((address_space(3))) void *)(dst + k * kStageThreads + wave_base),
                                             16, 0, 0);
        }
        if (!HAS_INIT) {
#pragma unroll
            for (int k = 0; k < BPER; ++k) {
                int i = c * BCH + tid_ + k * kStageThreads;
                i = i < NT * 64 - 1 ? i : NT * 64 - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bf + i),
                                                 (__attribute__((address_space(3))) void *)(dst + CH + k * kStageThreads + wave_base),
                                                 16, 0, 0);
            }
        }
    };
    constexpr int PAIRS = (kChunkTiles + 1) / 2;
    constexpr int NP = (NT + 1) / 2;
    constexpr int NG = NP * KT;
    float4 ra[2], rb[2], ba = make_float4(0, 0, 0, 0), bb = make_float4(0, 0, 0, 0);
    rb[0] = rb[1] = make_float4(0, 0, 0, 0);
    auto issue = [&](int g) {
        const int pr = g / KT, t = g - pr * KT;
        const int c = pr / PAIRS, h2 = 2 * (pr - c * PAIRS), n = 2 * pr;
        const float4 *wc = bufp[c & 1];
        const float4 *w = wc + h2 * KT * 64;
        ra[g & 1] = w[t * 64 + lane];
        if (n + 1 < NT) rb[g & 1] = w[(KT + t) * 64 + lane];
        if (t == 0 && !HAS_INIT) {
            const float4 *bc = wc + CH;
            ba = bc[h2 * 64 + lane];
            if (n + 1 < NT) bb = bc[(h2 + 1) * 64 + lane];
        }
    };
    issue(0);
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int pr = g / KT, t = g - pr * KT;
        const int c = pr / PAIRS, n = 2 * pr;
        const bool two = (n + 1 < NT);
        const bool chunk_first = (t == 0) && (pr == c * PAIRS);
        const bool chunk_last = (t == KT - 1) && ((pr == c * PAIRS + PAIRS - 1) || (pr == NP - 1));
        // a one-group chunk (not used by any layer today) would have to request the following data before its barrier
        if (chunk_first && chunk_last) {
            if (c + 1 < NCH) stage(c + 1);
            else if (next.n_w > 0) flow_stage_first(F, next, (F.parity + NCH) & 1);
        }
        if (chunk_last) __syncthreads();      // reads of chunk c all issued and back; the DMAs requested so far have landed
        if (t == 0) {
            if (HAS_INIT) { a0 = init[n]; if (two) a1 = init[n + 1]; }
            else { a0 = (f32x4){ba.x, ba.y, ba.z, ba.w}; if (two) a1 = (f32x4){bb.x, bb.y, bb.z, bb.w}; }
        }
        const float4 w0 = ra[g & 1], w1 = rb[g & 1];
        const int steps = (t == KT - 1) ? LAST : ((t == KT - 2) ? LAST2 : 4);
        __builtin_amdgcn_sched_barrier(0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, in[t][0], a0, 0, 0, 0);
        if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, in[t][0], a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG) issue(g + 1);
        if (chunk_first && !chunk_last) {
            if (c + 1 < NCH) stage(c + 1);                                    // next chunk of this layer, other buffer
            else if (next.n_w > 0) flow_stage_first(F, next, (F.parity + NCH) & 1);   // chunk 0 of the layer that follows
        }
        __builtin_amdgcn_sched_barrier(0);
        if (steps > 1) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, in[t][1], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, in[t][1], a1, 0, 0, 0);
        }
        if (steps > 2) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, in[t][2], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, in[t][2], a1, 0, 0, 0);
        }
        if (steps > 3) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, in[t][3], a0, 0, 0, 0);
            if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, in[t][3], a1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t == KT - 1) {
            if (RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { a0[r] = relu_f32(a0[r]); a1[r] = relu_f32(a1[r]); }
            }
            out[n] = a0;
            if (two) out[n + 1] = a1;
        }
    }
    F.parity = (F.parity + NCH) & 1;
}

// ------------------------------------------------------------------------------------------------
// bf16x3: float32 layers on the bf16 matrix pipe (round 4).
// A float32 value is split exactly into three bfloat16 pieces, x = hi + mid + lo (8 + 8 + 8 significand bits); of the
// nine piece products of w * x six are kept (everything down to 2^-24 relative: hi hi, hi mid, mid hi, hi lo, lo hi,
// mid mid) and accumulated in float32 by v_mfma_f32_16x16x32_bf16 -- 16 cycles per instruction and 32 features deep,
// i.e. 96 cycles per 32 x 16 block against 256 for the eight v_mfma_f32_16x16x4_f32 it replaces, and, unlike the
// float32 form, the bf16 form leaves the vector ALU free for the splitting while it runs.  Measured error against a
// float64 evaluation: the same as the float32 MFMA chain's (tools/microbench/split_bf16.hip, profiles/r02_split_bf16.txt).
// Register chaining carries over: a lane's accumulators of output tiles 2m and 2m + 1 (features 4q + r of each) are,
// as they stand, the eight k-slots 8q + s of the B operand of input block m, so `split8` of two finished output tiles
// IS the next layer's operand; the weight fragments are the float32 fragments of mcn_pack_linear regrouped by
// mcn_pack_x3: per (output tile, input block) three 16-byte pieces per lane.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct X3 { bf16x8 hi, mid, lo; };
// a chunk's DMA rounds in one burst behind its first MFMA (1) or spread over its first groups (measured: no gain)
#ifndef MCN_X3_DMA_SPREAD
#define MCN_X3_DMA_SPREAD 1
#endif
// scheduling fences around the MFMA groups of an x3 layer: 0 = nothing moves across (the split / ReLU blocks of one
// wavefront then run between its MFMA groups and overlap the OTHER wavefront's MFMAs), 6 = vector / scalar ALU work may
#ifndef MCN_X3_FENCE
#define MCN_X3_FENCE 0
#endif

__device__ __forceinline__ X3 split8(const f32x4 a, const f32x4 b)
{
    X3 s;
#if MCN_X3_WHATIF & 16
    s.hi = __builtin_bit_cast(bf16x8, a); s.mid = __builtin_bit_cast(bf16x8, b); s.lo = s.hi;
    return s;
#endif
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

// staged chunk of an x3 layer: whole output tiles, as many as fit the stage with their biases; balanced over the chunks
__host__ __device__ constexpr int x3_rows(int KB) { return KB * 3 * 64; }                       // 16-byte rows per output tile
__host__ __device__ constexpr int x3_fit(int KB)          // most output tiles whose rows (rounded up to a DMA round) + biases fit
{
    int t = 1;
    while (((t + 1) * x3_rows(KB) + kStageThreads - 1) / kStageThreads * kStageThreads + (t + 1) * 64 <= kStageFloat4 + kStageBias) ++t;
    return t;
}
__host__ __device__ constexpr int x3_chunks(int KB, int NT) { return (NT + x3_fit(KB) - 1) / x3_fit(KB); }
__host__ __device__ constexpr int x3_chunk_tiles(int KB, int NT) { return (NT + x3_chunks(KB, NT) - 1) / x3_chunks(KB, NT); }
__host__ __device__ constexpr int x3_bias_at(int KB, int NT)       // biases start at the next DMA round after the chunk
{
    return (x3_chunk_tiles(KB, NT) * x3_rows(KB) + kStageThreads - 1) / kStageThreads * kStageThreads;
}
template <int KB, int NT, bool HAS_INIT>
__device__ __forceinline__ NextChunk first_chunk_x3(const float4 *wf, const float4 *bf)
{
    constexpr int CT = x3_chunk_tiles(KB, NT), CH = CT * x3_rows(KB), TOTAL = NT * x3_rows(KB), BCH = CT * 64;
    static_assert(x3_bias_at(KB, NT) + (HAS_INIT ? 0 : BCH) <= kStageFloat4 + kStageBias, "x3 chunk + biases do not fit the LDS stage");
    return NextChunk{wf, CH < TOTAL ? CH : TOTAL, HAS_INIT ? nullptr : bf, HAS_INIT ? 0 : (BCH < NT * 64 ? BCH : NT * 64),
                     x3_bias_at(KB, NT)};
}

#if MCN_X3_WHATIF & 8
#define X3_MFMA(wa, xb, acc, c0, c1, c2) (acc)
#else
#define X3_MFMA(wa, xb, acc, c0, c1, c2) __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc, c0, c1, c2)
#endif
// One layer: in = KB input blocks of 32 features (split), out = NT output tiles of 16 features (float32 accumulators,
// bias / init added, ReLU applied); SPLIT: also outp = the ceil(NT / 2) input blocks of the layer that follows, each
// split as soon as its two tiles are finished, so that the splitting's vector instructions run in the shadow of the
// next tiles' MFMAs (a bf16 MFMA holds the vector issue for 8 of its 16 cycles; the scheduling fences below let vector
// and scalar instructions -- and nothing else -- move across them).  Weight stream and hand-over of the following
// layer's first chunk as dense_flow; the A pieces are requested TWO groups ahead (a group is only 96 cycles here,
// less than an LDS round trip under load) into a three-slot register ring, and the chunk barrier sits before the
// second-to-last group of a chunk (every read of the chunk has been issued by then).
// init: accumulator start when HAS_INIT -- a register array, or (INIT_MEM) this lane's column of a [NT][64] float4 table
// in global memory (`init` then points at element [0][lane]; tile n + 1's row is fetched while tile n is multiplied).
// chunk geometry of an x3 layer.  A UNIT is one (output tile, input block): three A pieces, six MFMAs, 96 pipe cycles.
// A GROUP is up to MCN_X3_UNITS consecutive units of one chunk; the A pieces of group G + 1 are requested behind the first
// MFMA of group G, so a group's length is the cover its successor's LDS reads get.  (While an LDS-DMA is in flight every
// LDS wait the compiler emits is lgkmcnt(0): "request further ahead" does not lengthen the cover -- the wait of group G
// drains the reads of G + 1 as well -- longer groups do.)  Measured at 4096 x 5 (same box, diagnostic builds): 1 unit
// 1.62-1.63 ms per step, 2 units 1.64, 3 units 1.66-1.69: the LDS round trip is not what the layers wait for, and the
// longer ring costs registers.  Default 1.
#ifndef MCN_X3_UNITS
#define MCN_X3_UNITS 1
#endif
// TIMING-ONLY what-if builds (wrong values; tools/ab_build.sh + kbench): bit 0 no workgroup barriers, bit 1 no weight DMA,
// bit 2 no LDS reads of the A pieces (after the first group), bit 3 no MFMAs, bit 4 no splitting (split8 returns its input
// bits), bit 5 no workspace traffic between the passes.  0 in every product build.
template <int KB, int NT>
struct X3Geo {
    static constexpr int U = MCN_X3_UNITS;
    static constexpr int CT = x3_chunk_tiles(KB, NT);
    static constexpr int NCH = x3_chunks(KB, NT);
    static constexpr int UC = CT * KB;                                   // units of a full chunk
    static constexpr int GPC = (UC + U - 1) / U;                         // groups of a full chunk
    static constexpr int units_in(int c) { return ((c + 1) * CT < NT ? CT : NT - c * CT) * KB; }
    static constexpr int groups_in(int c) { return (units_in(c) + U - 1) / U; }
    static constexpr int NG = (NCH - 1) * GPC + groups_in(NCH - 1);
    static constexpr int chunk_of(int g) { return g / GPC < NCH ? g / GPC : NCH - 1; }
    static constexpr int first_of(int c) { return c * GPC; }
    static constexpr int last_of(int c) { return c * GPC + groups_in(c) - 1; }
    static constexpr int unit0(int g) { return chunk_of(g) * UC + (g - first_of(chunk_of(g))) * U; }
    static constexpr int count(int g)
    {
        return units_in(chunk_of(g)) - (g - first_of(chunk_of(g))) * U < U ? units_in(chunk_of(g)) - (g - first_of(chunk_of(g))) * U : U;
    }
};
template <class Fn, int... Gs>
__device__ __forceinline__ void static_for(Fn &&f, std::integer_sequence<int, Gs...>)
{
    (f(std::integral_constant<int, Gs>{}), ...);
}

// on_tile(n, a): called with every finished output tile (a consumer that needs each tile once -- a running sum -- does not
// keep the layer's whole float32 output alive).
struct NoTileHook { __device__ __forceinline__ void operator()(int, const f32x4 &) const {} };
template <int KB, int NT, bool RELU, bool HAS_INIT, bool SPLIT = false, bool INIT_MEM = false, class OnTile = NoTileHook>
__device__ __forceinline__ void dense_flow_x3(const X3 (&in)[KB], const f32x4 *init, f32x4 (&out)[NT],
                                              X3 (&outp)[(NT + 1) / 2],
                                              const float4 *__restrict__ wf, const float4 *__restrict__ bf,
                                              WeightFlow &F, int lane, const NextChunk &next, const OnTile &on_tile = OnTile())
{
    constexpr int CT = x3_chunk_tiles(KB, NT);
    constexpr int ROWS = x3_rows(KB);
    constexpr int CH = CT * ROWS;
    constexpr int TOTAL = NT * ROWS;
    constexpr int NCH = x3_chunks(KB, NT);
    constexpr int PER = (CH + kStageThreads - 1) / kStageThreads;
    constexpr int BOFF = x3_bias_at(KB, NT);
    constexpr int BCH = CT * 64;
    constexpr int BPER = (BCH + kStageThreads - 1) / kStageThreads;
    static_assert(BOFF + (HAS_INIT ? 0 : BCH) <= kStageFloat4 + kStageBias, "x3 chunk + biases do not fit the LDS stage");
    float4 *const bufp[2] = {F.buf + F.parity * kStageBuf, F.buf + (F.parity ^ 1) * kStageBuf};
    auto stage = [&](int c) {                             // chunk c >= 1 of THIS layer -> buffer (parity + c) & 1
        float4 *dst = bufp[c & 1];
        const DmaLane L = dma_lane(F.tid);
        const int rows = TOTAL - c * CH < CH ? TOTAL - c * CH : CH;
        dma_rows<PER>(wf, c * CH, rows, dst, L);
        if (!HAS_INIT) {
            const int brows = NT * 64 - c * BCH < BCH ? NT * 64 - c * BCH : BCH;
            dma_rows<BPER>(bf, c * BCH, brows, dst + BOFF, L);
        }
    };
    using Geo = X3Geo<KB, NT>;
    constexpr int NG = Geo::NG, U = Geo::U;
    float4 ra[2][U][3], bias[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    // (The group loop is a compile-time expansion -- static_for over integral constants -- not an unrolled run-time loop:
    //  every array index is a constant from the start, so the register promotion of the operand arrays does not depend
    //  on when the optimiser gets round to unrolling ~50 groups; as a `for` loop the output pieces ended up in scratch.)
    auto issue = [&](auto gc) {                           // LDS -> ring slot g & 1: the A pieces of the group's units
        constexpr int g = decltype(gc)::value;
        constexpr int c = Geo::chunk_of(g);
        static_for([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < Geo::count(g)) {
                constexpr int u = Geo::unit0(g) + k, n = u / KB, m = u - n * KB, h = n - c * CT;
                const float4 *w = bufp[c & 1] + (h * KB + m) * 192;
                if (MCN_X3_WHATIF & 64) {                // timing only: the same three reads at half the bytes
                    const float2 *w2 = reinterpret_cast<const float2 *>(w);
                    const float2 t0 = w2[lane], t1 = w2[128 + lane], t2 = w2[256 + lane];
                    ra[g & 1][k][0] = make_float4(t0.x, t0.y, t0.x, t0.y); ra[g & 1][k][1] = make_float4(t1.x, t1.y, t1.x, t1.y);
                    ra[g & 1][k][2] = make_float4(t2.x, t2.y, t2.x, t2.y);
                } else {
                    ra[g & 1][k][0] = w[lane]; ra[g & 1][k][1] = w[64 + lane]; ra[g & 1][k][2] = w[128 + lane];
                }
                if (m == 0 && !HAS_INIT) bias[n & 3] = bufp[c & 1][BOFF + h * 64 + lane];
            }
        }, std::make_integer_sequence<int, U>{});
    };
    auto request_next = [&](int c) {                      // what follows chunk c: this layer's next chunk, or the next layer's first
        if (MCN_X3_WHATIF & 2) return;
        if (c + 1 < NCH) stage(c + 1);
        else if (next.n_w > 0) flow_stage_first(F, next, (F.parity + NCH) & 1);
    };
    // part j of `parts` of the same request (MCN_X3_DMA_SPREAD > 1: the rounds of a chunk's DMA spread over its first
    // groups instead of one burst behind the first MFMA)
    auto request_part = [&](int c, int j, int parts) {
        if (MCN_X3_WHATIF & 2) return;
        const DmaLane L = dma_lane(F.tid);
        if (c + 1 < NCH) {
            float4 *dst = bufp[(c + 1) & 1];
            const int rows = TOTAL - (c + 1) * CH < CH ? TOTAL - (c + 1) * CH : CH;
            const int brows = HAS_INIT ? 0 : (NT * 64 - (c + 1) * BCH < BCH ? NT * 64 - (c + 1) * BCH : BCH);
            const int rw = (rows + kStageThreads - 1) / kStageThreads, rb = (brows + kStageThreads - 1) / kStageThreads;
            const int k0 = j * (rw + rb) / parts, k1 = (j + 1) * (rw + rb) / parts;
            dma_rows<PER>(wf, (c + 1) * CH, rows, dst, L, k0, k1 < rw ? k1 : rw);
            if (!HAS_INIT) dma_rows<BPER>(bf, (c + 1) * BCH, brows, dst + BOFF, L, k0 - rw, k1 - rw);
        } else if (next.n_w > 0) {
            float4 *dst = F.buf + ((F.parity + NCH) & 1) * kStageBuf;
            const int rw = (next.n_w + kStageThreads - 1) / kStageThreads, rb = (next.n_b + kStageThreads - 1) / kStageThreads;
            const int k0 = j * (rw + rb) / parts, k1 = (j + 1) * (rw + rb) / parts;
            dma_rows<kStageFloat4 / kStageThreads>(next.w, 0, next.n_w, dst, L, k0, k1 < rw ? k1 : rw);
            dma_rows<3>(next.b, 0, next.n_b, dst + next.b_at, L, k0 - rw, k1 - rw);
        }
    };
    issue(std::integral_constant<int, 0>{});
    f32x4 a = {0, 0, 0, 0};
    float4 initv[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (HAS_INIT && INIT_MEM) initv[0] = reinterpret_cast<const float4 *>(init)[0];
    auto group = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int c = Geo::chunk_of(g);
        constexpr bool at_first = g == Geo::first_of(c), at_last = g == Geo::last_of(c);
        if constexpr (at_last) {
            if constexpr (at_first) request_next(c);      // (the request must precede the barrier that publishes it)
            if (!(MCN_X3_WHATIF & 1)) __syncthreads();    // every read of chunk c has been issued and is back; DMAs landed
        }
        static_for([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < Geo::count(g)) {
                constexpr int u = Geo::unit0(g) + k, n = u / KB, m = u - n * KB;
                if constexpr (m == 0) {
                    if constexpr (HAS_INIT && INIT_MEM) {
                        a = (f32x4){initv[n & 1].x, initv[n & 1].y, initv[n & 1].z, initv[n & 1].w};
                        if constexpr (n + 1 < NT) initv[(n + 1) & 1] = reinterpret_cast<const float4 *>(init)[(n + 1) * 64];
                    } else if constexpr (HAS_INIT) a = init[n];
                    else a = (f32x4){bias[n & 3].x, bias[n & 3].y, bias[n & 3].z, bias[n & 3].w};
                }
                const bf16x8 wh = __builtin_bit_cast(bf16x8, ra[g & 1][k][0]), wm = __builtin_bit_cast(bf16x8, ra[g & 1][k][1]),
                             wl = __builtin_bit_cast(bf16x8, ra[g & 1][k][2]);
                __builtin_amdgcn_sched_barrier(MCN_X3_FENCE);
                a = X3_MFMA(wl, in[m].hi, a, 0, 0, 0);    // smallest terms first
                if constexpr (k == 0) {
                    // behind the group's FIRST MFMA (whose wait for this group's pieces must not see them yet): the next
                    // group's reads, and at a chunk's first group the DMA of what follows the chunk
                    __builtin_amdgcn_sched_barrier(MCN_X3_FENCE);
                    if constexpr (g + 1 < NG) { if (!(MCN_X3_WHATIF & 4)) issue(std::integral_constant<int, g + 1>{}); }
                    if constexpr (MCN_X3_DMA_SPREAD <= 1) {
                        if constexpr (at_first && !at_last) request_next(c);
                    } else if constexpr (!at_last) {
                        constexpr int gn = Geo::groups_in(c), j = g - Geo::first_of(c);
                        constexpr int parts = gn - 1 < MCN_X3_DMA_SPREAD ? gn - 1 : MCN_X3_DMA_SPREAD;
                        if constexpr (j < parts) request_part(c, j, parts);
                    }
                    __builtin_amdgcn_sched_barrier(MCN_X3_FENCE);
                }
                a = X3_MFMA(wh, in[m].lo, a, 0, 0, 0);
                a = X3_MFMA(wm, in[m].mid, a, 0, 0, 0);
                a = X3_MFMA(wm, in[m].hi, a, 0, 0, 0);
                a = X3_MFMA(wh, in[m].mid, a, 0, 0, 0);
                a = X3_MFMA(wh, in[m].hi, a, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(MCN_X3_FENCE);
                if constexpr (m == KB - 1) {
                    if (RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = relu_f32(a[r]);
                    }
                    out[n] = a;
                    on_tile(n, a);
                    if constexpr (SPLIT && ((n & 1) || n == NT - 1)) {
                        const f32x4 z = {0, 0, 0, 0};
                        if constexpr (n & 1) outp[n >> 1] = split8(out[n - 1], out[n]);
                        else outp[n >> 1] = split8(out[n], z);
                    }
                }
            }
        }, std::make_integer_sequence<int, U>{});
    };
    static_for(group, std::make_integer_sequence<int, NG>{});
    F.parity = (F.parity + NCH) & 1;
}

}  // namespace mcn
