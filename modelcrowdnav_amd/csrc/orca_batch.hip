// orca_batch.hip -- ORCA velocity for B independent agents (one lane per agent).
//
// Replaces the rvo2.PyRVOSimulator addAgent / setAgent* / doStep / getAgentVelocity(0)
// sequence of crowd_sim/envs/policy/orca.py:95-129 for agents that are not the env's own
// humans: the ORCA-driven robot of `test.py --policy orca` (crowd_nav/test.py:52,90-95) and
// stand-alone solver tests.  Candidates are staged once per lane in LDS, then the shared
// device solver (orca_device.hpp) runs on them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "orca_device.hpp"

namespace mcn {

struct LdsCand {
    const float4 *pv; const float *rad; int stride;   // &s[tid], strided by block
    __device__ __forceinline__ void fetch(int c, float4 &o, float &r) const { o = pv[c * stride]; r = rad[c * stride]; }
};

constexpr int kOrcaBlock = 64;

__global__ __launch_bounds__(kOrcaBlock) void orca_batch_kernel(const float *__restrict__ self,
                                                              const float *__restrict__ others,
                                                              const int32_t *__restrict__ n_other,
                                                              float *__restrict__ out, int B, int M, int nl_cap,
                                                              float neighbor_dist, int max_neighbors,
                                                              float time_horizon, float time_step)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *sL = reinterpret_cast<float4 *>(smem);                 // [nl_cap][BLOCK]
    float4 *sC = sL + (size_t)nl_cap * kOrcaBlock;                 // [M][BLOCK]
    float  *sR = reinterpret_cast<float *>(sC + (size_t)M * kOrcaBlock);   // [M][BLOCK]
    const int tid = threadIdx.x;
    const long b = (long)blockIdx.x * kOrcaBlock + tid;
    if (b >= B) return;
    const float *s = self + b * 8;
    int n = n_other[b];
    if (n > M) n = M;
    for (int c = 0; c < n; ++c) {
        const float *o = others + (b * M + c) * 5;
        sC[c * kOrcaBlock + tid] = make_float4(o[0], o[1], o[2], o[3]);
        sR[c * kOrcaBlock + tid] = o[4];
    }
    LdsCand cand{sC + tid, sR + tid, kOrcaBlock};
    LdsLines L{sL + tid, kOrcaBlock};
    float ox, oy;
    orca_solve(cand, n, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7],
               neighbor_dist, max_neighbors, time_horizon, time_step, L, ox, oy);
    out[b * 2] = ox; out[b * 2 + 1] = oy;
}

int launch_orca_batch(const float *self, const float *others, const int32_t *n_other, float *out,
                      int B, int M, float neighbor_dist, int max_neighbors, float time_horizon, float time_step,
                      hipStream_t stream)
{
    const int nl_cap = max_neighbors < M ? max_neighbors : (M > 0 ? M : 1);
    const size_t sm = (size_t)kOrcaBlock * (16u * nl_cap + 16u * M + 4u * M);
    const int blocks = (B + kOrcaBlock - 1) / kOrcaBlock;
    hipLaunchKernelGGL(orca_batch_kernel, dim3(blocks), dim3(kOrcaBlock), sm, stream,
                       self, others, n_other, out, B, M, nl_cap, neighbor_dist, max_neighbors, time_horizon, time_step);
    return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
}

}  // namespace mcn
