// orca_device.hpp -- device-side ORCA velocity solve for gfx950.
//
// Replaces the rvo2 round trip of crowd_sim/envs/policy/orca.py:95-129 (one RVO2 simulator
// per human, doStep, read agent 0).  rvo2 itself is a third-party module that is neither
// vendored nor pinned by the reference; this is a from-scratch statement of the published
// ORCA algorithm (van den Berg et al. 2011) with RVO2's documented conventions: float32,
// epsilon 1e-5, neighbours ordered by squared distance (stable, capped at maxNeighbors),
// incremental 2-D LP, 3-D LP fallback.  Every float operation is written in the same order
// as the CPU checker so the two agree bit for bit (build with -ffp-contract=off).
//
// Mapping: one lane solves one agent.  Its half-planes live in LDS as float4 (point.xy,
// dir.xy) at sL[k * stride + tid]: consecutive lanes hit consecutive 16-B slots, so every
// ds_read_b128 / ds_write_b128 is conflict-free, and the data-dependent loops of the LP can
// index lines at run time without spilling to scratch.
#pragma once
#include <hip/hip_runtime.h>

namespace mcn {

constexpr float kRvoEps = 1e-5f;
constexpr int kMaxLines = 10;

// Lines of one lane in LDS, strided by the block size.
struct LdsLines {
    float4 *base;   // &sL[tid]
    int stride;     // block size
    __device__ __forceinline__ float4 get(int k) const { return base[k * stride]; }
    __device__ __forceinline__ void set(int k, float4 v) const { base[k * stride] = v; }
};
// Lines in private memory (only touched on the rare 3-D LP path).
struct PrivLines {
    float4 *p;
    __device__ __forceinline__ float4 get(int k) const { return p[k]; }
    __device__ __forceinline__ void set(int k, float4 v) const { p[k] = v; }
};

__device__ __forceinline__ float det2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }

// 1-D LP on line `no` against lines [0,no) and the speed disc.
template <class Acc>
__device__ bool lp1(const Acc &L, int no, float radius, float optx, float opty, bool dir_opt, float &rx, float &ry)
{
    const float4 ln = L.get(no);   // (p.x, p.y, d.x, d.y)
    const float dp = dot2(ln.x, ln.y, ln.z, ln.w);
    const float disc = dp * dp + radius * radius - dot2(ln.x, ln.y, ln.x, ln.y);
    if (disc < 0.0f) return false;
    const float sq = sqrtf(disc);
    float tl = -dp - sq;
    float tr = -dp + sq;
    for (int i = 0; i < no; ++i) {
        const float4 li = L.get(i);
        const float den = det2(ln.z, ln.w, li.z, li.w);
        const float num = det2(li.z, li.w, ln.x - li.x, ln.y - li.y);
        if (fabsf(den) <= kRvoEps) {
            if (num < 0.0f) return false;
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) tr = fminf(tr, t);
        else             tl = fmaxf(tl, t);
        if (tl > tr) return false;
    }
    float t;
    if (dir_opt) {
        t = (dot2(optx, opty, ln.z, ln.w) > 0.0f) ? tr : tl;
    } else {
        t = dot2(ln.z, ln.w, optx - ln.x, opty - ln.y);
        if (t < tl) t = tl; else if (t > tr) t = tr;
    }
    rx = ln.x + t * ln.z;
    ry = ln.y + t * ln.w;
    return true;
}

// Incremental 2-D LP; returns index of the first failing line or n.
template <class Acc>
__device__ int lp2(const Acc &L, int n, float radius, float optx, float opty, bool dir_opt, float &rx, float &ry)
{
    if (dir_opt) {
        rx = radius * optx; ry = radius * opty;
    } else if (dot2(optx, opty, optx, opty) > radius * radius) {
        const float inv = 1.0f / sqrtf(dot2(optx, opty, optx, opty));
        rx = radius * (optx * inv); ry = radius * (opty * inv);
    } else {
        rx = optx; ry = opty;
    }
    for (int i = 0; i < n; ++i) {
        const float4 li = L.get(i);
        if (det2(li.z, li.w, li.x - rx, li.y - ry) > 0.0f) {
            const float kx = rx, ky = ry;
            if (!lp1(L, i, radius, optx, opty, dir_opt, rx, ry)) { rx = kx; ry = ky; return i; }
        }
    }
    return n;
}

// Minimise the maximum penetration after lp2 failed at `begin`.
template <class Acc>
__device__ __noinline__ void lp3(const Acc &L, int n, int begin, float radius, float &rx, float &ry)
{
    float dist = 0.0f;
    float4 pbuf[kMaxLines];
    PrivLines P{pbuf};
    for (int i = begin; i < n; ++i) {
        const float4 li = L.get(i);
        if (det2(li.z, li.w, li.x - rx, li.y - ry) > dist) {
            int m = 0;
            for (int j = 0; j < i; ++j) {
                const float4 lj = L.get(j);
                float qx, qy;
                const float dt = det2(li.z, li.w, lj.z, lj.w);
                if (fabsf(dt) <= kRvoEps) {
                    if (dot2(li.z, li.w, lj.z, lj.w) > 0.0f) continue;
                    qx = 0.5f * (li.x + lj.x); qy = 0.5f * (li.y + lj.y);
                } else {
                    const float s = det2(lj.z, lj.w, li.x - lj.x, li.y - lj.y) / dt;
                    qx = li.x + s * li.z; qy = li.y + s * li.w;
                }
                const float ddx = lj.z - li.z, ddy = lj.w - li.w;
                const float inv = 1.0f / sqrtf(dot2(ddx, ddy, ddx, ddy));
                P.set(m++, make_float4(qx, qy, ddx * inv, ddy * inv));
            }
            const float kx = rx, ky = ry;
            if (lp2(P, m, radius, -li.w, li.z, true, rx, ry) < m) { rx = kx; ry = ky; }
            dist = det2(li.z, li.w, li.x - rx, li.y - ry);
        }
    }
}

// Half-plane induced by one neighbour (orca.py parameters: tau = timeHorizon, dt = timeStep).
__device__ __forceinline__ float4 orca_line(float px, float py, float vx, float vy, float radius,
                                            float4 o /* px,py,vx,vy */, float orad,
                                            float inv_th, float inv_ts)
{
    const float rpx = o.x - px, rpy = o.y - py;
    const float rvx = vx - o.z, rvy = vy - o.w;
    const float dist_sq = dot2(rpx, rpy, rpx, rpy);
    const float cr = radius + orad;
    const float cr_sq = cr * cr;
    float dx, dy, ux, uy;
    if (dist_sq > cr_sq) {
        const float wx = rvx - inv_th * rpx, wy = rvy - inv_th * rpy;
        const float wl_sq = dot2(wx, wy, wx, wy);
        const float dp1 = dot2(wx, wy, rpx, rpy);
        if (dp1 < 0.0f && dp1 * dp1 > cr_sq * wl_sq) {
            const float wl = sqrtf(wl_sq);
            const float inv = 1.0f / wl;
            const float uwx = wx * inv, uwy = wy * inv;
            dx = uwy; dy = -uwx;
            const float s = cr * inv_th - wl;
            ux = s * uwx; uy = s * uwy;
        } else {
            const float leg = sqrtf(dist_sq - cr_sq);
            const float inv = 1.0f / dist_sq;
            if (det2(rpx, rpy, wx, wy) > 0.0f) {
                dx = (rpx * leg - rpy * cr) * inv;
                dy = (rpx * cr + rpy * leg) * inv;
            } else {
                dx = -((rpx * leg + rpy * cr) * inv);
                dy = -((-rpx * cr + rpy * leg) * inv);
            }
            const float dp2 = dot2(rvx, rvy, dx, dy);
            ux = dp2 * dx - rvx; uy = dp2 * dy - rvy;
        }
    } else {
        const float wx = rvx - inv_ts * rpx, wy = rvy - inv_ts * rpy;
        const float wl = sqrtf(dot2(wx, wy, wx, wy));
        const float inv = 1.0f / wl;
        const float uwx = wx * inv, uwy = wy * inv;
        dx = uwy; dy = -uwx;
        const float s = cr * inv_ts - wl;
        ux = s * uwx; uy = s * uwy;
    }
    return make_float4(vx + 0.5f * ux, vy + 0.5f * uy, dx, dy);
}

// Full solve for one lane.  `Cand` provides n candidates in insertion order:
//   cand.fetch(c, float4& posvel, float& radius)
template <class Cand>
__device__ void orca_solve(const Cand &cand, int ncand, float px, float py, float vx, float vy, float radius,
                           float max_speed, float prefx, float prefy, float neighbor_dist, int max_neighbors,
                           float time_horizon, float time_step, const LdsLines &L, float &outx, float &outy)
{
    const float range_sq = neighbor_dist * neighbor_dist;
    const float inv_th = 1.0f / time_horizon;
    const float inv_ts = 1.0f / time_step;
    int nl = 0;
    for (int c = 0; c < ncand; ++c) {
        float4 o; float orad;
        cand.fetch(c, o, orad);
        const float ddx = px - o.x, ddy = py - o.y;
        const float d = dot2(ddx, ddy, ddx, ddy);
        if (!(d < range_sq)) continue;
        // slot = number of in-range candidates that sort before this one (stable by insertion order)
        int rank = 0;
        for (int k = 0; k < ncand; ++k) {
            if (k == c) continue;
            float4 ok; float rk;
            cand.fetch(k, ok, rk);
            const float ex = px - ok.x, ey = py - ok.y;
            const float dk = dot2(ex, ey, ex, ey);
            rank += (dk < range_sq) && (dk < d || (dk == d && k < c));
        }
        if (rank >= max_neighbors) continue;
        L.set(rank, orca_line(px, py, vx, vy, radius, o, orad, inv_th, inv_ts));
        ++nl;
    }
    float rx, ry;
    const int fail = lp2(L, nl, max_speed, prefx, prefy, false, rx, ry);
    if (fail < nl) lp3(L, nl, fail, max_speed, rx, ry);
    outx = rx; outy = ry;
}

}  // namespace mcn
