// orca_coop.hpp -- RVO2's linearProgram3 (dense-crowd fallback) solved by the wavefront TOGETHER.
//
// In the lane-per-human kernel the 3-D LP is the worst case of SIMT: at 10 humans per env 2.9 % of the solves need
// it, i.e. 1.7 of a wavefront's 60 humans per step, 83 % of the wavefronts -- and the register-resident form
// (orca_static.hpp: Lp3Step) then runs its unrolled O(n^3) code for one or two active lanes, once per distinct
// (first failing line, violated line) combination present in the wavefront: 1600-3200 instructions at n = 9.
// Here the lanes that need it park their problem (sorted half-planes, line count, first failing line, running
// result) in one of 8 LDS slots and the wavefront's eight OCTETS solve the parked problems side by side, each with
// its 8 lanes: per round the next violated line's candidate is computed by projecting the earlier lines one (or
// two) per lane, solving every projected line's direction-optimising 1-D LP speculatively, and taking them in order
// (the reference's incremental 2-D LP collapses to compare-and-take steps because lp1 on a line depends on the
// earlier lines, not on the running result).  Same arithmetic per value as lp3() in orca_device.hpp / the oracle.
// More than 8 problems in a wavefront take another pass.
//
// Where it is used: batches small enough to be latency-bound (one wavefront per workgroup, env_step.hip: BLOCK = 64),
// where a step costs its longest wavefront's instruction stream: 4096 envs, N = 10: 28.7 -> 24.9 us, N = 7: 14.1 ->
// 12.7 us.  Throughput-bound batches keep the unrolled form: there the cooperative pass is paid by all 64 lanes of
// every wavefront that holds a dense human while the unrolled form only runs the blocks some lane needs (2^18 envs,
// N = 10: 391 vs 307 us).
#pragma once
#include <hip/hip_runtime.h>
#include "orca_device.hpp"

namespace mcn {

constexpr int kCoopSlots = 8, kCoopLines = 12;

struct CoopLds {
    float4 line[kCoopSlots][kCoopLines];
    float4 proj[kCoopSlots][kCoopLines];
    float4 cand[kCoopSlots][kCoopLines];
    int kept[kCoopSlots][kCoopLines];
    float4 meta[kCoopSlots];        // bits of nl, bits of fail, max speed, -
    float2 start[kCoopSlots];       // running result when the 2-D LP failed
    float2 res[kCoopSlots];
};

__device__ __forceinline__ void coop_sync()
{
    // the lanes exchange through LDS; a wavefront's DS operations execute in order: ordering the compiler suffices
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Every lane of the wavefront must call this (lanes without a problem pass fail >= nl).
template <int NL>
__device__ __forceinline__ void lp3_wave_coop(CoopLds &S, const float4 (&L)[NL], int nl, int fail, float ms, float &rx, float &ry)
{
    constexpr int CPL = (NL + 7) / 8;
    static_assert(NL <= kCoopLines, "line table too small");
    const int lane = threadIdx.x & 63;
    bool need = fail < nl;
    for (;;) {
        const unsigned long long mask = __ballot(need);
        if (mask == 0ull) break;
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        const bool sel = need && before < kCoopSlots;
        if (sel) {
#pragma unroll
            for (int k2 = 0; k2 < NL; ++k2) S.line[before][k2] = L[k2];
            S.meta[before] = make_float4(__int_as_float(nl), __int_as_float(fail), ms, 0.0f);
            S.start[before] = make_float2(rx, ry);
        }
        int nprob = __popcll(mask);
        nprob = nprob < kCoopSlots ? nprob : kCoopSlots;
        coop_sync();
        const int g = lane >> 3, k = lane & 7;
        if (g < nprob) {
            const float4 mt = S.meta[g];
            const int pnl = __float_as_int(mt.x);
            const float pms = mt.z;
            float crx = S.start[g].x, cry = S.start[g].y;
            float dist = 0.0f;
            int i = __float_as_int(mt.y);
            for (;;) {
                // next line from i on that the running result violates by more than dist
                float4 li = make_float4(0, 0, 0, 0);
                for (; i < pnl; ++i) {
                    li = S.line[g][i];
                    if (det2(li.z, li.w, li.x - crx, li.y - cry) > dist) break;
                }
                if (i >= pnl) break;
                // lane k projects lines k (and k + 8), those before i, on line i
                float4 pq[CPL];
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int j = k + 8 * q;
                    const float4 lj = S.line[g][j < kCoopLines ? j : 0];
                    const float dtm = det2(li.z, li.w, lj.z, lj.w);
                    const bool par = fabsf(dtm) <= kRvoEps;
                    const int kept = ((j < i) & !(par & (dot2(li.z, li.w, lj.z, lj.w) > 0.0f))) ? 1 : 0;
                    const float sc = det2(lj.z, lj.w, li.x - lj.x, li.y - lj.y) / dtm;
                    const float ddx = lj.z - li.z, ddy = lj.w - li.w;
                    const float inv = 1.0f / sqrtf(dot2(ddx, ddy, ddx, ddy));
                    pq[q] = make_float4(par ? 0.5f * (li.x + lj.x) : li.x + sc * li.z,
                                        par ? 0.5f * (li.y + lj.y) : li.y + sc * li.w, ddx * inv, ddy * inv);
                    if (j < kCoopLines) { S.proj[g][j] = pq[q]; S.kept[g][j] = kept; }
                }
                coop_sync();
                const float ox = -li.w, oy = li.z;
                // speculative direction-optimising 1-D LP of each projected line against the kept ones before it
                float ptl[CPL], ptr_[CPL];
                bool pok[CPL];
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const float dp = dot2(pq[q].x, pq[q].y, pq[q].z, pq[q].w);
                    const float disc = dp * dp + pms * pms - dot2(pq[q].x, pq[q].y, pq[q].x, pq[q].y);
                    pok[q] = !(disc < 0.0f);
                    const float sq = sqrtf(disc);
                    ptl[q] = -dp - sq;
                    ptr_[q] = -dp + sq;
                }
                for (int jj = 0; jj + 1 < i; ++jj) {
                    const float4 pj = S.proj[g][jj];
                    const bool kj = S.kept[g][jj] != 0;
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        const int j = k + 8 * q;
                        const float den = det2(pq[q].z, pq[q].w, pj.z, pj.w);
                        const float num = det2(pj.z, pj.w, pq[q].x - pj.x, pq[q].y - pj.y);
                        const float tt = num / den;
                        const bool live = (jj < j) & (j < i) & kj;
                        const bool par = fabsf(den) <= kRvoEps;
                        const bool cut = live & !par;
                        const float ntr = fminf(ptr_[q], tt), ntl = fmaxf(ptl[q], tt);
                        ptr_[q] = (cut & (den >= 0.0f)) ? ntr : ptr_[q];
                        ptl[q] = (cut & !(den >= 0.0f)) ? ntl : ptl[q];
                        pok[q] = pok[q] & !(live & par & (num < 0.0f)) & !(cut & (ptl[q] > ptr_[q]));
                    }
                }
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int j = k + 8 * q;
                    const float tt = (dot2(ox, oy, pq[q].z, pq[q].w) > 0.0f) ? ptr_[q] : ptl[q];
                    if (j < kCoopLines)
                        S.cand[g][j] = make_float4(pq[q].x + tt * pq[q].z, pq[q].y + tt * pq[q].w, pok[q] ? 1.0f : 0.0f, 0.0f);
                }
                coop_sync();
                float cx = pms * ox, cy = pms * oy;
                bool failed = false;
                for (int jj = 0; jj < i; ++jj) {
                    const float4 pj = S.proj[g][jj], cjv = S.cand[g][jj];
                    const bool viol = (S.kept[g][jj] != 0) & !failed & (det2(pj.z, pj.w, pj.x - cx, pj.y - cy) > 0.0f);
                    const bool okj = cjv.z != 0.0f;
                    cx = (viol & okj) ? cjv.x : cx; cy = (viol & okj) ? cjv.y : cy;
                    failed = failed | (viol & !okj);
                }
                crx = failed ? crx : cx; cry = failed ? cry : cy;
                dist = det2(li.z, li.w, li.x - crx, li.y - cry);
                ++i;
                coop_sync();       // the projected-line slots are reused by the next round
            }
            if (k == 0) S.res[g] = make_float2(crx, cry);
        }
        coop_sync();
        if (sel) {
            const float2 r = S.res[before];
            rx = r.x; ry = r.y;
            need = false;
        }
        coop_sync();               // before the slots are refilled
    }
}

}  // namespace mcn
