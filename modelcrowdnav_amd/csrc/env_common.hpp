// env_common.hpp -- float64 helpers shared by the env-step kernels (reference operation order, no contraction).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mcn.h"

namespace mcn {

__device__ __forceinline__ double norm2(double x0, double x1) { return sqrt(fma(x1, x1, x0 * x0)); }

// crowd_sim/envs/utils/utils.py:4-26 with (x3, y3) = (0, 0), the env's only call shape
__device__ __forceinline__ double p2s_origin(double x1, double y1, double x2, double y2)
{
    // Branch-free form with identical results: for a degenerate segment (px == py == 0) the reference returns
    // norm((0 - x1, 0 - y1)); with u forced to 0 the general formula gives norm((x1 + 0*0, y1 + 0*0)) -- the same
    // two squares under one fma and one sqrt.  Keeping one straight-line block lets the scheduler interleave this
    // long float64 dependency chain with the other independent ones around it.
    const double px = x2 - x1, py = y2 - y1;
    const bool degenerate = (px == 0) & (py == 0);
    double u = ((0.0 - x1) * px + (0.0 - y1) * py) / (px * px + py * py);
    u = degenerate ? 0.0 : u;
    if (u > 1) u = 1; else if (u < 0) u = 0;
    const double x = x1 + u * px, y = y1 + u * py;
    return norm2(x - 0.0, y - 0.0);
}

// One env's step record as two stores (16 + 8 bytes): assigning the struct member-wise makes the compiler emit
// separate byte / short / dword stores for done, info, padding and hh_count.
__device__ __forceinline__ void store_step_rec(mcn_step_rec *dst, double reward, double dmin, int done, int info, int hh)
{
    double *d = reinterpret_cast<double *>(dst);
    const unsigned long long tail = (unsigned long long)(unsigned)(done & 0xff) | ((unsigned long long)(unsigned)(info & 0xff) << 8) |
                                    ((unsigned long long)(unsigned)hh << 32);      // little-endian layout of mcn_step_rec
    reinterpret_cast<double2 *>(d)[0] = make_double2(reward, dmin);
    reinterpret_cast<unsigned long long *>(d)[2] = tail;
}

// Python's float % for a positive divisor
__device__ __forceinline__ double pymod(double a, double m)
{
    double r = fmod(a, m);
    if (r != 0 && r < 0) r += m;
    return r;
}

}  // namespace mcn
