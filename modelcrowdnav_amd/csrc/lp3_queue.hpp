// lp3_queue.hpp -- device work queue of the DEFERRED 3-D LP (include/mcn.h: mcn_env_out.lp3_queue).
//
// RVO2's linearProgram3 is the worst case of the lane-per-human layout: 2-3 % of the humans of a circle crossing need
// it, but a wavefront holds 60 of them, so 70-85 % of the wavefronts run its unrolled O(n^3) code for one or two
// active lanes (a third of the N = 5 kernel's instructions, two thirds of the N = 10 kernel's).  With a queue the
// step kernel stops after the 2-D LP: a human whose 2-D LP failed parks its sorted half-planes, the line count, the
// first failing line and the running result here (one wave-aggregated atomic per wavefront) and leaves its own
// integration to env_lp3_kernel, a second launch that solves the parked problems one per lane -- every lane busy --
// and writes those humans' velocities / positions.  Same arithmetic (lp3_static on the same sorted lines), so the
// same bits.
//
// Layout (cap = E * N entries, SoA so that a wavefront's consecutive entries are consecutive in memory):
//   [0]  int count      entries of the running step (reset to 0 by env_lp3_kernel's last workgroup)
//   [4]  int done       workgroups of env_lp3_kernel that have finished
//   [64] int4   hdr [cap]   (human index e * N + h, nl | fail << 8, max speed bits, 0)
//        float2 res [cap]   running result when the 2-D LP failed
//        int    flag[cap]   1: integrate the human (update, env not restarted)  2: write the look-ahead observation
//        float4 line[NL][cap]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcn {

struct Lp3Queue {
    int *count, *done;
    int4 *hdr;
    float2 *res;
    int *flag;
    float4 *line;
    long cap;
};

__host__ __device__ __forceinline__ long lp3_queue_lines_offset(long cap)
{
    return (64 + cap * 28 + 15) & ~15L;
}

__host__ __device__ __forceinline__ Lp3Queue lp3_queue_view(void *base, long cap)
{
    char *b = reinterpret_cast<char *>(base);
    Lp3Queue q;
    q.count = reinterpret_cast<int *>(b);
    q.done = reinterpret_cast<int *>(b + 4);
    q.hdr = reinterpret_cast<int4 *>(b + 64);
    q.res = reinterpret_cast<float2 *>(b + 64 + cap * 16);
    q.flag = reinterpret_cast<int *>(b + 64 + cap * 24);
    q.line = reinterpret_cast<float4 *>(b + lp3_queue_lines_offset(cap));
    q.cap = cap;
    return q;
}

__host__ __device__ __forceinline__ long lp3_queue_bytes(long cap, int nl)
{
    return lp3_queue_lines_offset(cap) + cap * 16 * nl;
}

}  // namespace mcn
