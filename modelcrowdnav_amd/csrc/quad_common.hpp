// quad_common.hpp -- building blocks shared by the quad-parallel kernels (env_step_quad.hip, env_rollout_quad.hip):
// quad broadcasts, the speculative 1-D / 3-D LP candidates and the per-lane ORCA solve.
#pragma once
#include <hip/hip_runtime.h>
#include "orca_device.hpp"
#include "orca_static.hpp"

namespace mcn {

// Diagnostic build only: how often a wavefront takes each data-dependent path (tools/fixed_cost.py).
#ifdef MCN_DIAG
static __device__ unsigned int g_diag_counts[8192 * 4];
#define DIAG_COUNT(which)                                                                                        \
    do {                                                                                                         \
        const int w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                                      \
        if (w_ < 8192 && (int)(threadIdx.x & 63) == __ffsll((unsigned long long)__ballot(1)) - 1)                \
            atomicAdd(&g_diag_counts[w_ * 4 + (which)], 1u);                                                     \
    } while (0)
#else
#define DIAG_COUNT(which)
#endif

// quad broadcast of lane I (0..3) of every quad: a DPP move, no LDS traffic
template <int I>
__device__ __forceinline__ int qbi(int v) { return __builtin_amdgcn_update_dpp(0, v, I * 0x55, 0xf, 0xf, false); }
template <int I>
__device__ __forceinline__ float qbf(float v) { return __builtin_bit_cast(float, qbi<I>(__builtin_bit_cast(int, v))); }
template <int I>
__device__ __forceinline__ double qbd(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = qbi<I>((int)b), hi = qbi<I>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int I>
__device__ __forceinline__ float4 qb4(float4 v) { return make_float4(qbf<I>(v.x), qbf<I>(v.y), qbf<I>(v.z), qbf<I>(v.w)); }

__device__ __forceinline__ float4 sel4(const float4 (&L)[4], int i)
{
    const float4 a = i == 1 ? L[1] : L[0], b = i == 3 ? L[3] : L[2];
    return i >= 2 ? b : a;
}

// linearProgram1 on line `no` (run-time, 0..3) against lines [0, no): uniform code, predicated steps.
template <bool DIR>
__device__ __forceinline__ bool lp1_rt(const float4 (&L)[4], int no, float radius, float optx, float opty, float &rx, float &ry)
{
    const float4 ln = sel4(L, no);
    const float dp = dot2(ln.x, ln.y, ln.z, ln.w);
    const float disc = dp * dp + radius * radius - dot2(ln.x, ln.y, ln.x, ln.y);
    bool ok = !(disc < 0.0f);
    const float sq = sqrt_f32(disc, ok);
    float tl = -dp - sq;
    float tr = -dp + sq;
    // straight-line: every lane runs all three steps and masks them with selects (a lone wavefront per SIMD pays
    // for every exec-mask instruction; the quotient of a skipped / parallel line is computed and discarded)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float4 li = L[i];
        const float den = det2(ln.z, ln.w, li.z, li.w);
        const float num = det2(li.z, li.w, ln.x - li.x, ln.y - li.y);
        const float t = num / den;
        const bool live = i < no;
        const bool par = fabsf(den) <= kRvoEps;
        const bool cut = live & !par;
        const float ntr = fminf(tr, t), ntl = fmaxf(tl, t);
        tr = (cut & (den >= 0.0f)) ? ntr : tr;
        tl = (cut & !(den >= 0.0f)) ? ntl : tl;
        ok = ok & !(live & par & (num < 0.0f)) & !(cut & (tl > tr));
    }
    float t;
    if (DIR) {
        t = (dot2(optx, opty, ln.z, ln.w) > 0.0f) ? tr : tl;
    } else {
        t = dot2(ln.z, ln.w, optx - ln.x, opty - ln.y);
        if (t < tl) t = tl; else if (t > tr) t = tr;
    }
    rx = ln.x + t * ln.z;
    ry = ln.y + t * ln.w;
    return ok;
}

// 3-D LP candidate of line `i` (run-time, quad-uniform), computed by the four lanes of the quad TOGETHER: lane k
// projects line k on line i (RVO2 linearProgram3's inner loop, one projection per lane instead of three in a row),
// then solves the direction-optimising 1-D LP of ITS projected line against the earlier kept ones speculatively
// (as the 2-D LP of quad_orca_velocity does), and the incremental LP collapses into three compare-and-take steps.
// Lines the reference drops (parallel, same direction) are not compacted away but masked: the incremental LP over
// the kept lines in their original order is the same sequence of operations.  Same arithmetic per value as
// lp3() in orca_device.hpp / the oracle, so the same bits.  Returns whether the candidate is valid (the inner
// 2-D LP succeeded); the result is identical on the four lanes.
__device__ __forceinline__ bool lp3_candidate_quad(const float4 (&L)[4], int i, int k, float radius, float &rx, float &ry)
{
    const float4 li = sel4(L, i);
    const float4 lj = sel4(L, k);
    const float dt = det2(li.z, li.w, lj.z, lj.w);
    const bool par = fabsf(dt) <= kRvoEps;
    const int kept = ((k < i) & !(par & (dot2(li.z, li.w, lj.z, lj.w) > 0.0f))) ? 1 : 0;
    const float sc = det2(lj.z, lj.w, li.x - lj.x, li.y - lj.y) / dt;
    const float ddx = lj.z - li.z, ddy = lj.w - li.w;
    const float inv = rcp_sqrt_f32(dot2(ddx, ddy, ddx, ddy), kept != 0);   // dropped / self projections: unused
    const float4 q = make_float4(par ? 0.5f * (li.x + lj.x) : li.x + sc * li.z,
                                 par ? 0.5f * (li.y + lj.y) : li.y + sc * li.w, ddx * inv, ddy * inv);
    const float4 P0 = qb4<0>(q), P1 = qb4<1>(q), P2 = qb4<2>(q);
    const int k0 = qbi<0>(kept), k1 = qbi<1>(kept), k2 = qbi<2>(kept);
    const float ox = -li.w, oy = li.z;

    // speculative 1-D LP of projected line k against the kept projected lines 0 .. min(k, 2) - 1
    const float dp = dot2(q.x, q.y, q.z, q.w);
    const float disc = dp * dp + radius * radius - dot2(q.x, q.y, q.x, q.y);
    bool ok = !(disc < 0.0f);
    const float sq = sqrt_f32(disc, ok & (kept != 0));
    float tl = -dp - sq;
    float tr = -dp + sq;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 pj = j == 0 ? P0 : P1;
        const bool live = (j < k) & ((j == 0 ? k0 : k1) != 0);
        const float den = det2(q.z, q.w, pj.z, pj.w);
        const float num = det2(pj.z, pj.w, q.x - pj.x, q.y - pj.y);
        const float t = num / den;
        const bool parj = fabsf(den) <= kRvoEps;
        const bool cut = live & !parj;
        const float ntr = fminf(tr, t), ntl = fmaxf(tl, t);
        tr = (cut & (den >= 0.0f)) ? ntr : tr;
        tl = (cut & !(den >= 0.0f)) ? ntl : tl;
        ok = ok & !(live & parj & (num < 0.0f)) & !(cut & (tl > tr));
    }
    const float tt = (dot2(ox, oy, q.z, q.w) > 0.0f) ? tr : tl;
    const float cx = q.x + tt * q.z, cy = q.y + tt * q.w;
    const int okm = ok ? 1 : 0;

    rx = radius * ox; ry = radius * oy;
    bool failed = false;
#define MCN_LP3_INNER_TAKE(I, PI, KI)                                                          \
    {                                                                                          \
        const int ok_i = qbi<I>(okm); const float cx_i = qbf<I>(cx), cy_i = qbf<I>(cy);        \
        const bool viol = (KI != 0) & !failed & (det2(PI.z, PI.w, PI.x - rx, PI.y - ry) > 0.0f); \
        rx = (viol & (ok_i != 0)) ? cx_i : rx; ry = (viol & (ok_i != 0)) ? cy_i : ry;           \
        failed = failed | (viol & (ok_i == 0));                                                 \
    }
    MCN_LP3_INNER_TAKE(0, P0, k0) MCN_LP3_INNER_TAKE(1, P1, k1) MCN_LP3_INNER_TAKE(2, P2, k2)
#undef MCN_LP3_INNER_TAKE
    return !failed;
}

// ORCA velocity of the human owning this quad.  Lane k holds candidate neighbour k: `o` = its (px, py, vx, vy) in
// float32, `crd` its radius, `cand_valid` whether the slot is populated.  The result is identical on the four
// lanes of the quad (layout: lane = 4 * human + k).
// inv_th / inv_ts = 1 / timeHorizon, 1 / timeStep: computed by the caller once per launch and handed over as opaque
// values -- left to itself the compiler rewrites `apart ? 1 / a : 1 / b` into `1 / (apart ? a : b)`, i.e. one IEEE
// division (11 instructions) on the critical path of EVERY step instead of two before the step loop.
__device__ __forceinline__ void quad_orca_velocity(const mcn_env_cfg &c, int lane, int k, bool cand_valid,
                                                   double2 pos, double2 vel, double2 goal, double rad, double vpref,
                                                   float4 o, double crd, float inv_th, float inv_ts, float &rx, float &ry)
{
    const float fpx = (float)pos.x, fpy = (float)pos.y, fvx = (float)vel.x, fvy = (float)vel.y;
    const float frad = (float)(rad + 0.01 + c.orca_safety_space);
    const float ms = (float)vpref;
    const float prefx = (float)(goal.x - pos.x), prefy = (float)(goal.y - pos.y);
    const float orad = (float)(crd + 0.01 + c.orca_safety_space);
    const float range_sq = c.orca_neighbor_dist * c.orca_neighbor_dist;
    const float ddx = fpx - o.x, ddy = fpy - o.y;
    const float d = dot2(ddx, ddy, ddx, ddy);
    const int in = (cand_valid && (d < range_sq)) ? 1 : 0;
    const float d0 = qbf<0>(d), d1 = qbf<1>(d), d2 = qbf<2>(d), d3 = qbf<3>(d);
    const int i0 = qbi<0>(in), i1 = qbi<1>(in), i2 = qbi<2>(in), i3 = qbi<3>(in);
    const int nin = i0 + i1 + i2 + i3;
    // stable ascending order among the in-range candidates (RVO2 insertAgentNeighbor); the rest fill the
    // remaining slots in lane order so that ranks stay a permutation
    const int rank_in = ((i0 != 0) & ((d0 < d) | ((d0 == d) & (0 < k)))) + ((i1 != 0) & ((d1 < d) | ((d1 == d) & (1 < k)))) +
                        ((i2 != 0) & ((d2 < d) | ((d2 == d) & (2 < k)))) + ((i3 != 0) & ((d3 < d) | ((d3 == d) & (3 < k))));
    const int rank_out = nin + (((0 < k) & !i0) ? 1 : 0) + (((1 < k) & !i1) ? 1 : 0) + (((2 < k) & !i2) ? 1 : 0);
    const int rank = in ? rank_in : rank_out;
    int nl = nin < c.orca_max_neighbors ? nin : c.orca_max_neighbors;
    const float4 mine = orca_line_select(fpx, fpy, fvx, fvy, frad, o, orad, inv_th, inv_ts, in != 0);
    // route my half-plane to lane `rank` of the quad, then share all four
    const int dst = ((lane & ~3) | rank) << 2;
    float4 srt;
    srt.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dst, __builtin_bit_cast(int, mine.x)));
    srt.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dst, __builtin_bit_cast(int, mine.y)));
    srt.z = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dst, __builtin_bit_cast(int, mine.z)));
    srt.w = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dst, __builtin_bit_cast(int, mine.w)));
    const float4 L[4] = {qb4<0>(srt), qb4<1>(srt), qb4<2>(srt), qb4<3>(srt)};

    // speculative 1-D LPs, one per lane; then the incremental LP is four compare-and-take steps
    float cx, cy;
    const int okm = lp1_rt<false>(L, k, ms, prefx, prefy, cx, cy) ? 1 : 0;
    {
        const float pp = dot2(prefx, prefy, prefx, prefy);
        const bool clip = pp > ms * ms;
        const float inv = rcp_sqrt_f32(pp, clip);
        rx = clip ? ms * (prefx * inv) : prefx;
        ry = clip ? ms * (prefy * inv) : prefy;
    }
    int fail = nl;
#define MCN_LP2_TAKE(I)                                                                        \
    {                                                                                          \
        const int ok_i = qbi<I>(okm); const float cx_i = qbf<I>(cx), cy_i = qbf<I>(cy);        \
        const bool viol = (I < nl) & (fail == nl) & (det2(L[I].z, L[I].w, L[I].x - rx, L[I].y - ry) > 0.0f); \
        rx = (viol & (ok_i != 0)) ? cx_i : rx; ry = (viol & (ok_i != 0)) ? cy_i : ry;           \
        fail = (viol & (ok_i == 0)) ? I : fail;                                                 \
    }
    MCN_LP2_TAKE(0) MCN_LP2_TAKE(1) MCN_LP2_TAKE(2) MCN_LP2_TAKE(3)
#undef MCN_LP2_TAKE
    if (fail < nl) {
        // dense crowd: 3-D LP (RVO2 linearProgram3).  fail / nl / the running result are quad-uniform, so the
        // whole region is entered per quad and the quad broadcasts inside see all four lanes.  Per round: find
        // the next line from `i` on that the running result violates by more than `dist`, let the quad's four
        // lanes compute its candidate together, take it if valid, update `dist`.  Measured on the benchmark
        // workload: 0.65 % of the solves get here; 93 % of those need one round, 6 % two, 1 % three.
        DIAG_COUNT(0);
        float dist = 0.0f;
        int i = fail;
        for (;;) {
            int nxt = 4;
#define MCN_LP3_NEXT(I) nxt = ((I >= i) & (I < nl) & (det2(L[I].z, L[I].w, L[I].x - rx, L[I].y - ry) > dist)) ? I : nxt;
            MCN_LP3_NEXT(3) MCN_LP3_NEXT(2) MCN_LP3_NEXT(1) MCN_LP3_NEXT(0)
#undef MCN_LP3_NEXT
            if (nxt >= 4) break;
            float cx3, cy3;
            const bool ok3 = lp3_candidate_quad(L, nxt, k, ms, cx3, cy3);
            rx = ok3 ? cx3 : rx; ry = ok3 ? cy3 : ry;
            const float4 ln = sel4(L, nxt);
            dist = det2(ln.z, ln.w, ln.x - rx, ln.y - ry);
            i = nxt + 1;
        }
    }
}

}  // namespace mcn
