// orca_static.hpp -- register-resident ORCA solve for a compile-time candidate count.
//
// Same arithmetic, in the same order, as orca_device.hpp (which stays as the run-time-N
// path and as the rare 3-D LP fallback), restructured for the wave64 VALU:
//   * candidates, their squared distances and the resulting half-planes live in VGPRs; every
//     array index is a compile-time constant after unrolling, so nothing goes to scratch;
//   * neighbour ordering (ascending squared distance, stable, RVO2 insertAgentNeighbor) is a
//     fixed insertion-sort network of predicated swaps instead of data-dependent LDS traffic;
//   * the cut-off-circle and collision cases of the half-plane construction share one code
//     path (they differ only in the inverse time constant), halving the divergent code;
//   * the 3-D LP (only reached when the 2-D LP is infeasible, i.e. in dense crowds) copies
//     the lines to private memory and reuses the generic solver.
#pragma once
#include <hip/hip_runtime.h>
#include "orca_device.hpp"
#include "fast_f32.hpp"

namespace mcn {

// half-plane construction used by the lane-per-human solver: branchy (orca_line_merged) or straight-line
// (orca_line_select); same bits either way
#ifndef MCN_STATIC_LINE
#define MCN_STATIC_LINE orca_line_select
#endif

// Half-plane induced by one neighbour.  The cut-off-circle case and the already-colliding case differ only in
// the inverse time constant (1/timeHorizon vs 1/timeStep), so they share one code path.
__device__ __forceinline__ float4 orca_line_merged(float px, float py, float vx, float vy, float radius, float4 o,
                                                   float orad, float inv_th, float inv_ts)
{
        const float rpx = o.x - px, rpy = o.y - py;
    const float rvx = vx - o.z, rvy = vy - o.w;
    const float dist_sq = dot2(rpx, rpy, rpx, rpy);
    const float cr = radius + orad;
    const float cr_sq = cr * cr;
    const bool apart = dist_sq > cr_sq;
    const float inv = apart ? inv_th : inv_ts;          // collision case uses 1/timeStep
    const float wx = rvx - inv * rpx, wy = rvy - inv * rpy;
    const float wl_sq = dot2(wx, wy, wx, wy);
    const float dp1 = dot2(wx, wy, rpx, rpy);
    float dx, dy, ux, uy;
    if (!apart || (dp1 < 0.0f && dp1 * dp1 > cr_sq * wl_sq)) {
        const float wl = sqrtf(wl_sq);
        const float iw = 1.0f / wl;
        const float uwx = wx * iw, uwy = wy * iw;
        dx = uwy; dy = -uwx;
        const float s = cr * inv - wl;
        ux = s * uwx; uy = s * uwy;
    } else {
        const float leg = sqrtf(dist_sq - cr_sq);
        const float id = 1.0f / dist_sq;
        if (det2(rpx, rpy, wx, wy) > 0.0f) {
            dx = (rpx * leg - rpy * cr) * id;
            dy = (rpx * cr + rpy * leg) * id;
        } else {
            dx = -((rpx * leg + rpy * cr) * id);
            dy = -((-rpx * cr + rpy * leg) * id);
        }
        const float dp2 = dot2(rvx, rvy, dx, dy);
        ux = dp2 * dx - rvx; uy = dp2 * dy - rvy;
    }
    return make_float4(vx + 0.5f * ux, vy + 0.5f * uy, dx, dy);
}

// orca_line_merged as one straight-line block: the cut-off-circle and the leg projections are both
// evaluated and the result selected.  Same operations on the selected side, so the same bits.
// Returns (u.x, u.y, dir.x, dir.y): the half-plane is point = v + u / 2, direction = dir.
// Every operation is odd-symmetric under swapping the two agents (relative position and velocity change sign, the
// radius sum does not), so the pair's other half-plane is exactly (-u, -dir): env_step.hip builds each pair once.
// `used`: lanes whose half-plane is consumed (an empty candidate slot's operands must not force the IEEE path).
__device__ __forceinline__ float4 orca_u_dir(float px, float py, float vx, float vy, float radius, float4 o,
                                             float orad, float inv_th, float inv_ts, bool used = true)
{
    const float rpx = o.x - px, rpy = o.y - py;
    const float rvx = vx - o.z, rvy = vy - o.w;
    const float dist_sq = dot2(rpx, rpy, rpx, rpy);
    const float cr = radius + orad;
    const float cr_sq = cr * cr;
    const bool apart = dist_sq > cr_sq;
    const float inv = apart ? inv_th : inv_ts;          // collision case uses 1/timeStep
    const float wx = rvx - inv * rpx, wy = rvy - inv * rpy;
    const float wl_sq = dot2(wx, wy, wx, wy);
    const float dp1 = dot2(wx, wy, rpx, rpy);
    const bool circle = !apart | ((dp1 < 0.0f) & (dp1 * dp1 > cr_sq * wl_sq));
    // The block's two roots and two reciprocals behind ONE wave-uniform range guard (fast_f32.hpp): |w|^2 and the
    // squared leg inside sqrt5's range (their roots are then inside rcp3's), dist^2 -- at least the squared leg, so
    // normal -- below 2^126.  Same bits as the IEEE expansions either way.
    const float leg_sq = dist_sq - cr_sq;
    float wl, iw, leg, id;
#if MCN_FAST_F32
    // (a lane that takes the cut-off circle -- every colliding pair does -- only needs |w|, one that takes a leg only
    //  the leg and dist^2: a negative squared leg of an overlapping pair must not send the wavefront down the slow path)
    // (lane masks combined on the scalar unit: as booleans the compiler materialises them in vector registers)
    const unsigned long long m_circle = __builtin_amdgcn_ballot_w64(circle), m_used = __builtin_amdgcn_ballot_w64(used);
    const unsigned long long m_w = __builtin_amdgcn_ballot_w64(sqrt5_ok(wl_sq));
    const unsigned long long m_l = __builtin_amdgcn_ballot_w64(sqrt5_ok(leg_sq)) & __builtin_amdgcn_ballot_w64(dist_sq < 0x1p126f);
    if ((m_used & ~((m_circle & m_w) | (~m_circle & m_l))) == 0) {
        wl = sqrt5(wl_sq); iw = rcp3(wl); leg = sqrt5(leg_sq); id = rcp3(dist_sq);
    } else
#endif
    {
        wl = sqrtf(wl_sq); iw = 1.0f / wl; leg = sqrtf(leg_sq); id = 1.0f / dist_sq;
    }
    // cut-off circle
    const float uwx = wx * iw, uwy = wy * iw;
    const float sc = cr * inv - wl;
    const float cux = sc * uwx, cuy = sc * uwy;
    // legs
    const bool left = det2(rpx, rpy, wx, wy) > 0.0f;
    const float lx = (rpx * leg - rpy * cr) * id, ly = (rpx * cr + rpy * leg) * id;
    const float rx = -((rpx * leg + rpy * cr) * id), ry = -((-rpx * cr + rpy * leg) * id);
    const float gx = left ? lx : rx, gy = left ? ly : ry;
    const float dp2 = dot2(rvx, rvy, gx, gy);
    const float lux = dp2 * gx - rvx, luy = dp2 * gy - rvy;
    const float dx = circle ? uwy : gx, dy = circle ? -uwx : gy;
    const float ux = circle ? cux : lux, uy = circle ? cuy : luy;
    return make_float4(ux, uy, dx, dy);
}

__device__ __forceinline__ float4 orca_line_select(float px, float py, float vx, float vy, float radius, float4 o,
                                                   float orad, float inv_th, float inv_ts, bool used = true)
{
    const float4 ud = orca_u_dir(px, py, vx, vy, radius, o, orad, inv_th, inv_ts, used);
    return make_float4(vx + 0.5f * ud.x, vy + 0.5f * ud.y, ud.z, ud.w);
}

// 1-D LP on line NO (compile-time) against lines [0, NO).  DIR = false: closest point to (optx,opty);
// DIR = true: farthest point along the unit direction (optx,opty) (used by the 3-D LP).
template <int NO, int NL, bool DIR = false>
__device__ __forceinline__ bool lp1_s(const float4 (&L)[NL], float radius, float optx, float opty, float &rx, float &ry)
{
    const float4 ln = L[NO];
    const float dp = dot2(ln.x, ln.y, ln.z, ln.w);
    const float disc = dp * dp + radius * radius - dot2(ln.x, ln.y, ln.x, ln.y);
    if (disc < 0.0f) return false;
    const float sq = sqrt_f32(disc);
    float tl = -dp - sq;
    float tr = -dp + sq;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        // select form: the lanes that reach this point run every step anyway; a parallel line's quotient is
        // computed and discarded instead of being branched around
        const float4 li = L[i];
        const float den = det2(ln.z, ln.w, li.z, li.w);
        const float num = det2(li.z, li.w, ln.x - li.x, ln.y - li.y);
        const float t = num / den;
        const bool par = fabsf(den) <= kRvoEps;
        const float ntr = fminf(tr, t), ntl = fmaxf(tl, t);
        tr = (!par & (den >= 0.0f)) ? ntr : tr;
        tl = (!par & !(den >= 0.0f)) ? ntl : tl;
        ok = ok & !(par & (num < 0.0f)) & !(!par & (tl > tr));
    }
    // a failed lane keeps computing garbage that is discarded: once ok is false it stays false,
    // and the reference returns at the first failure without touching the result
    if (!ok) return false;
    float t;
    if constexpr (DIR) {
        t = (dot2(optx, opty, ln.z, ln.w) > 0.0f) ? tr : tl;
    } else {
        t = dot2(ln.z, ln.w, optx - ln.x, opty - ln.y);
        if (t < tl) t = tl; else if (t > tr) t = tr;
    }
    rx = ln.x + t * ln.z;
    ry = ln.y + t * ln.w;
    return true;
}

template <int I, int NL, bool DIR = false>
struct Lp2Step {
    static __device__ __forceinline__ void run(const float4 (&L)[NL], int n, float radius, float optx, float opty,
                                               float &rx, float &ry, int &fail)
    {
        if constexpr (I < NL) {
            if (I < n && fail == n) {
                const float4 li = L[I];
                if (det2(li.z, li.w, li.x - rx, li.y - ry) > 0.0f) {
                    float nx = rx, ny = ry;
                    if (lp1_s<I, NL, DIR>(L, radius, optx, opty, nx, ny)) { rx = nx; ry = ny; }
                    else fail = I;
                }
            }
            Lp2Step<I + 1, NL, DIR>::run(L, n, radius, optx, opty, rx, ry, fail);
        }
    }
};

// Register-resident 3-D LP (minimise the maximum penetration) for small line counts: step I projects lines
// [0, I) onto line I, runs the direction-optimising 2-D LP on them and updates the penetration bound.
template <int I, int NL>
struct Lp3Step {
    static __device__ __forceinline__ void run(const float4 (&L)[NL], int n, int begin, float radius, float &rx, float &ry,
                                               float &dist)
    {
        if constexpr (I < NL) {
            if (I >= begin && I < n) {
                const float4 li = L[I];
                if (det2(li.z, li.w, li.x - rx, li.y - ry) > dist) {
                    constexpr int NP = I > 0 ? I : 1;
                    float4 P[NP];
                    int m = 0;
#pragma unroll
                    for (int jj = 0; jj < I; ++jj) {
                        const float4 lj = L[jj];
                        const float dt = det2(li.z, li.w, lj.z, lj.w);
                        float qx, qy;
                        bool skip = false;
                        if (fabsf(dt) <= kRvoEps) {
                            if (dot2(li.z, li.w, lj.z, lj.w) > 0.0f) skip = true;
                            qx = 0.5f * (li.x + lj.x); qy = 0.5f * (li.y + lj.y);
                        } else {
                            const float sc = det2(lj.z, lj.w, li.x - lj.x, li.y - lj.y) / dt;
                            qx = li.x + sc * li.z; qy = li.y + sc * li.w;
                        }
                        const float ddx = lj.z - li.z, ddy = lj.w - li.w;
                        const float inv = rcp_sqrt_f32(dot2(ddx, ddy, ddx, ddy));
                        const float4 q = make_float4(qx, qy, ddx * inv, ddy * inv);
                        // compact the kept lines to the front (m is the running count): static slots via selects
#pragma unroll
                        for (int sl = 0; sl < NP; ++sl)
                            if (!skip && sl == m) P[sl] = q;
                        m += skip ? 0 : 1;
                    }
                    const float kx = rx, ky = ry;
                    const float ox = -li.w, oy = li.z;
                    rx = radius * ox; ry = radius * oy;
                    int fail = m;
                    if constexpr (I > 0) Lp2Step<0, NP, true>::run(P, m, radius, ox, oy, rx, ry, fail);
                    if (fail < m) { rx = kx; ry = ky; }
                    dist = det2(li.z, li.w, li.x - rx, li.y - ry);
                }
            }
            Lp3Step<I + 1, NL>::run(L, n, begin, radius, rx, ry, dist);
        }
    }
};

constexpr int kLp3StaticMax = 10;     // above this the unrolled code grows as NL^3: use the generic solver

// NC candidates in insertion order -> new velocity.  cpv[c] = (px,py,vx,vy), crad[c] = radius.
template <int NC>
__device__ __forceinline__ void orca_solve_static(float4 (&cpv)[NC > 0 ? NC : 1], float (&crad)[NC > 0 ? NC : 1],
                                                  float px, float py, float vx, float vy, float radius,
                                                  float max_speed, float prefx, float prefy, float neighbor_dist,
                                                  int max_neighbors, float time_horizon, float time_step,
                                                  float &outx, float &outy)
{
    constexpr int NL = NC < kMaxLines ? (NC > 0 ? NC : 1) : kMaxLines;
    const float range_sq = neighbor_dist * neighbor_dist;
    float d[NC > 0 ? NC : 1];
    int nin = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float ddx = px - cpv[c].x, ddy = py - cpv[c].y;
        const float dd = dot2(ddx, ddy, ddx, ddy);
        const bool in = dd < range_sq;
        d[c] = in ? dd : INFINITY;
        nin += in;
    }
    // stable ascending insertion-sort network (strict <, so equal keys keep insertion order)
#pragma unroll
    for (int i = 1; i < NC; ++i) {
#pragma unroll
        for (int j = i; j >= 1; --j) {
            const bool sw = d[j] < d[j - 1];
            const float td = sw ? d[j - 1] : d[j];       d[j - 1] = sw ? d[j] : d[j - 1];       d[j] = td;
            const float tr = sw ? crad[j - 1] : crad[j]; crad[j - 1] = sw ? crad[j] : crad[j - 1]; crad[j] = tr;
            const float4 a = cpv[j - 1], b = cpv[j];
            cpv[j - 1] = make_float4(sw ? b.x : a.x, sw ? b.y : a.y, sw ? b.z : a.z, sw ? b.w : a.w);
            cpv[j]     = make_float4(sw ? a.x : b.x, sw ? a.y : b.y, sw ? a.z : b.z, sw ? a.w : b.w);
        }
    }
    int nl = nin < max_neighbors ? nin : max_neighbors;
    if (nl > NL) nl = NL;

    const float inv_th = 1.0f / time_horizon;
    const float inv_ts = 1.0f / time_step;
    float4 L[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        if (k < NC) {
            L[k] = MCN_STATIC_LINE(px, py, vx, vy, radius, cpv[k], crad[k], inv_th, inv_ts);
        } else {
            L[k] = make_float4(0, 0, 1, 0);
        }
    }

    // 2-D LP: start from the preferred velocity clipped to the speed disc
    float rx, ry;
    if (dot2(prefx, prefy, prefx, prefy) > max_speed * max_speed) {
        const float inv = rcp_sqrt_f32(dot2(prefx, prefy, prefx, prefy));
        rx = max_speed * (prefx * inv); ry = max_speed * (prefy * inv);
    } else {
        rx = prefx; ry = prefy;
    }
    int fail = nl;
    Lp2Step<0, NL>::run(L, nl, max_speed, prefx, prefy, rx, ry, fail);

    if constexpr (NL <= kLp3StaticMax) {
        if (fail < nl) {
            float dist = 0.0f;
            Lp3Step<0, NL>::run(L, nl, fail, max_speed, rx, ry, dist);
        }
    } else if (fail < nl) {
        // dense-crowd fallback: minimise the maximum penetration (generic solver, private memory)
        float4 buf[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) buf[k] = L[k];
        PrivLines PL{buf};
        lp3(PL, nl, fail, max_speed, rx, ry);
    }
    outx = rx; outy = ry;
}

// The same solve from half-planes built elsewhere: Lnat[c] is candidate c's half-plane (insertion order), dd[c] its
// squared distance.  Sorting the finished lines by distance instead of the candidates before building them gives the
// same lines in the same order (stable network, same keys).
// First half: sort + 2-D LP.  On return Lnat is sorted, nl is the line count, (rx, ry) the 2-D LP's result and `fail`
// the first line it could not satisfy (== nl when it succeeded); the caller finishes with the 3-D LP (lp3_static
// below, or lp3_wave_coop in orca_coop.hpp).
template <int NC>
__device__ __forceinline__ void orca_sort_lp2(float4 (&Lnat)[NC > 0 ? NC : 1], float (&dd)[NC > 0 ? NC : 1],
                                              float max_speed, float prefx, float prefy, float neighbor_dist,
                                              int max_neighbors, float &rx, float &ry, int &fail, int &nl)
{
    constexpr int NL = NC > 0 ? NC : 1;
    static_assert(NC <= kMaxLines, "more candidates than line slots: the tail of the sorted list would be lost");
    const float range_sq = neighbor_dist * neighbor_dist;
    float d[NL];
    int nin = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const bool in = dd[c] < range_sq;
        d[c] = in ? dd[c] : INFINITY;
        nin += in;
    }
#pragma unroll
    for (int i = 1; i < NC; ++i) {
#pragma unroll
        for (int j = i; j >= 1; --j) {
            const bool sw = d[j] < d[j - 1];
            const float td = sw ? d[j - 1] : d[j];       d[j - 1] = sw ? d[j] : d[j - 1];       d[j] = td;
            const float4 a = Lnat[j - 1], b = Lnat[j];
            Lnat[j - 1] = make_float4(sw ? b.x : a.x, sw ? b.y : a.y, sw ? b.z : a.z, sw ? b.w : a.w);
            Lnat[j]     = make_float4(sw ? a.x : b.x, sw ? a.y : b.y, sw ? a.z : b.z, sw ? a.w : b.w);
        }
    }
    nl = nin < max_neighbors ? nin : max_neighbors;
    if (nl > NL) nl = NL;
    if (dot2(prefx, prefy, prefx, prefy) > max_speed * max_speed) {
        const float inv = rcp_sqrt_f32(dot2(prefx, prefy, prefx, prefy));
        rx = max_speed * (prefx * inv); ry = max_speed * (prefy * inv);
    } else {
        rx = prefx; ry = prefy;
    }
    fail = nl;
    Lp2Step<0, NL>::run(Lnat, nl, max_speed, prefx, prefy, rx, ry, fail);
}

template <int NL>
__device__ __forceinline__ void lp3_static(const float4 (&L)[NL], int nl, int fail, float max_speed, float &rx, float &ry)
{
    if (fail < nl) {
        float dist = 0.0f;
        Lp3Step<0, NL>::run(L, nl, fail, max_speed, rx, ry, dist);
    }
}

template <int NC>
__device__ __forceinline__ void orca_solve_static_lines(float4 (&Lnat)[NC > 0 ? NC : 1], float (&dd)[NC > 0 ? NC : 1],
                                                        float max_speed, float prefx, float prefy, float neighbor_dist,
                                                        int max_neighbors, float &outx, float &outy)
{
    int nl, fail;
    orca_sort_lp2<NC>(Lnat, dd, max_speed, prefx, prefy, neighbor_dist, max_neighbors, outx, outy, fail, nl);
    lp3_static<(NC > 0 ? NC : 1)>(Lnat, nl, fail, max_speed, outx, outy);
}

}  // namespace mcn
