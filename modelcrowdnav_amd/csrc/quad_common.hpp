// quad_common.hpp -- building blocks shared by the quad-parallel kernels (env_step_quad.hip, env_rollout_quad.hip):
// quad broadcasts, the speculative 1-D / 3-D LP candidates and the per-lane ORCA solve.
#pragma once
#include <hip/hip_runtime.h>
#include "orca_device.hpp"
#include "orca_static.hpp"

namespace mcn {

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
    const float sq = sqrtf(disc);
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

// 3-D LP candidate of line `no` (run-time): project lines [0,no) on it, direction-optimising 2-D LP.
__device__ __forceinline__ bool lp3_candidate(const float4 (&L)[4], int no, float radius, float &rx, float &ry)
{
    const float4 li = sel4(L, no);
    float4 P[3];
    int m = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float4 lj = L[j];
        const float dt = det2(li.z, li.w, lj.z, lj.w);
        float qx, qy;
        bool skip = !(j < no);
        if (fabsf(dt) <= kRvoEps) {
            if (dot2(li.z, li.w, lj.z, lj.w) > 0.0f) skip = true;
            qx = 0.5f * (li.x + lj.x); qy = 0.5f * (li.y + lj.y);
        } else {
            const float sc = det2(lj.z, lj.w, li.x - lj.x, li.y - lj.y) / dt;
            qx = li.x + sc * li.z; qy = li.y + sc * li.w;
        }
        const float ddx = lj.z - li.z, ddy = lj.w - li.w;
        const float inv = 1.0f / sqrtf(dot2(ddx, ddy, ddx, ddy));
        const float4 q = make_float4(qx, qy, ddx * inv, ddy * inv);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
            if (!skip && sl == m) P[sl] = q;
        m += skip ? 0 : 1;
    }
    const float ox = -li.w, oy = li.z;
    rx = radius * ox; ry = radius * oy;
    int fail = m;
    Lp2Step<0, 3, true>::run(P, m, radius, ox, oy, rx, ry, fail);
    return fail == m;
}

// ORCA velocity of the human owning this quad.  Lane k holds candidate neighbour k: `o` = its (px, py, vx, vy) in
// float32, `crd` its radius, `cand_valid` whether the slot is populated.  The result is identical on the four
// lanes of the quad (layout: lane = 4 * human + k).
__device__ __forceinline__ void quad_orca_velocity(const mcn_env_cfg &c, int lane, int k, bool cand_valid,
                                                   double2 pos, double2 vel, double2 goal, double rad, double vpref,
                                                   float4 o, double crd, double dt, float &rx, float &ry)
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
    const float inv_th = 1.0f / c.orca_time_horizon;
    const float inv_ts = 1.0f / (float)dt;
    const float4 mine = orca_line_select(fpx, fpy, fvx, fvy, frad, o, orad, inv_th, inv_ts);
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
        const float inv = 1.0f / sqrtf(pp);
        const bool clip = pp > ms * ms;
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
        // dense crowd: 3-D LP.  Candidates of the lines from `fail` on, in parallel, combined in line order.  The
        // region is entered per quad (fail / nl are quad-uniform, so the quad broadcasts below see all four
        // lanes); with only the few lanes that need a candidate active, most inner 1-D LPs are skipped outright.
        float c3x = 0.0f, c3y = 0.0f;
        int ok3 = 0;
        if (k >= fail && k < nl) ok3 = lp3_candidate(L, k, ms, c3x, c3y) ? 1 : 0;
        float dist = 0.0f;
#define MCN_LP3_TAKE(I)                                                                        \
    {                                                                                          \
        const int ok_i = qbi<I>(ok3); const float cx_i = qbf<I>(c3x), cy_i = qbf<I>(c3y);      \
        if (fail < nl && I >= fail && I < nl && det2(L[I].z, L[I].w, L[I].x - rx, L[I].y - ry) > dist) { \
            if (ok_i) { rx = cx_i; ry = cy_i; }                                                 \
            dist = det2(L[I].z, L[I].w, L[I].x - rx, L[I].y - ry);                             \
        }                                                                                      \
    }
        MCN_LP3_TAKE(0) MCN_LP3_TAKE(1) MCN_LP3_TAKE(2) MCN_LP3_TAKE(3)
#undef MCN_LP3_TAKE
    }
}

}  // namespace mcn
