// lp3_queue.hpp -- device work queue of the DEFERRED 3-D LP (include/mcn.h: mcn_env_out.lp3_queue).
//
// RVO2's linearProgram3 is the worst case of the lane-per-human layout: 2-3 % of the humans of a circle crossing need
// it, but a wavefront holds 60 of them, so 70-85 % of the wavefronts run its unrolled O(n^3) code for one or two
// active lanes (a third of the N = 5 kernel's instructions, two thirds of the N = 10 kernel's).  With a queue the
// step kernel stops after the 2-D LP: a human whose 2-D LP failed parks its sorted half-planes, the line count, the
// first failing line and the running result here and leaves its own integration to env_lp3_kernel, a second launch
// that finishes the parked problems densely and writes those humans' velocities / positions.  Same arithmetic on the
// same sorted lines, so the same bits.
//
// kLp3Queues sub-queues, one counter each, 256 B apart: a wavefront appends with ONE atomic (its lanes take
// consecutive slots) to the sub-queue its global wavefront index selects.  One counter for the whole grid would
// serialise tens of thousands of device-scope atomics on one address (they execute memory-side on this multi-XCD
// part, ~8 ns apiece: measured 2^20 envs x 5: +95 us); 256 counters spread them over the channels.
//
// Layout (subcap entries per sub-queue, SoA inside a sub-queue so consecutive entries are consecutive in memory):
//   [q * 256]     int count[q]         entries of the running step (reset by env_lp3_kernel's last workgroup)
//   [65536]       int done             workgroups of env_lp3_kernel that have finished
//   [66560]       int4   hdr [cap]     (human index e * N + h, nl | fail << 8, max speed bits, 0), cap = 256 * subcap
//                 float2 res [cap]     running result when the 2-D LP failed
//                 int    flag[cap]     1: integrate the human (update, env not restarted)  2: look-ahead observation
//                 float4 line[NL][cap]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcn {

#ifndef MCN_LP3_CAP_DIV
#define MCN_LP3_CAP_DIV 2
#endif
constexpr int kLp3Queues = 256;
constexpr long kLp3Header = 66560;

struct Lp3Queue {
    char *base;
    int4 *hdr;
    float2 *res;
    int *flag;
    float4 *line;
    long subcap, cap;
    __host__ __device__ __forceinline__ int *count(int q) const { return reinterpret_cast<int *>(base + (long)q * 256); }
    __host__ __device__ __forceinline__ int *done() const { return reinterpret_cast<int *>(base + 65536); }
};

// entries of one sub-queue.  The worst case -- every lane of every wavefront mapped to it (every 256th of the grid, up to
// three idle ones from rounding the grid to workgroups of four) parks a problem -- is 64 per wavefront: 527 MB for
// 2^18 envs x 10 humans, three times the env state, for a queue that 0.9-3.7 % of the humans use on average.  Bounded
// (round 4) to 1 / MCN_LP3_CAP_DIV of that, at least one wavefront's worth: a lane whose slot falls at or past subcap
// solves its 3-D LP in place instead (same arithmetic, same bits), the counter may run past subcap and env_lp3_kernel
// takes min(count, subcap) entries -- every slot below subcap was handed to exactly one lane, which filled it.
// Half, not less: the need comes in bursts (the steps in which a circle crossing meets in the middle) and a sub-queue
// sees the wavefronts of one residue class: measured at 2^18 x 10, a step takes 299 us with 1/1 and 1/2, 302 with 1/4,
// 318 with 1/8 of the worst case (in-place solving is what the queue exists to avoid).
__host__ __device__ __forceinline__ long lp3_subcap(long E, int N)
{
    const long waves = (E + (64 / N) - 1) / (64 / N) + 3;
    const long worst = (waves + kLp3Queues - 1) / kLp3Queues * 64;
    const long bounded = (worst / MCN_LP3_CAP_DIV + 63) / 64 * 64;
    return bounded < 64 ? 64 : bounded;
}

__host__ __device__ __forceinline__ long lp3_lines_offset(long cap) { return (kLp3Header + cap * 28 + 15) & ~15L; }

__host__ __device__ __forceinline__ Lp3Queue lp3_queue_view(void *base, long E, int N)
{
    Lp3Queue q;
    q.base = reinterpret_cast<char *>(base);
    q.subcap = lp3_subcap(E, N);
    q.cap = q.subcap * kLp3Queues;
    q.hdr = reinterpret_cast<int4 *>(q.base + kLp3Header);
    q.res = reinterpret_cast<float2 *>(q.base + kLp3Header + q.cap * 16);
    q.flag = reinterpret_cast<int *>(q.base + kLp3Header + q.cap * 24);
    q.line = reinterpret_cast<float4 *>(q.base + lp3_lines_offset(q.cap));
    return q;
}

__host__ __device__ __forceinline__ long lp3_queue_bytes(long E, int N, int nl)
{
    const long cap = lp3_subcap(E, N) * kLp3Queues;
    return lp3_lines_offset(cap) + cap * 16 * nl;
}

}  // namespace mcn
