// env_common.hpp -- float64 helpers shared by the env-step kernels (reference operation order, no contraction).
#pragma once
#include <hip/hip_runtime.h>

namespace mcn {

__device__ __forceinline__ double norm2(double x0, double x1) { return sqrt(fma(x1, x1, x0 * x0)); }

// crowd_sim/envs/utils/utils.py:4-26 with (x3, y3) = (0, 0), the env's only call shape
__device__ __forceinline__ double p2s_origin(double x1, double y1, double x2, double y2)
{
    const double px = x2 - x1, py = y2 - y1;
    if (px == 0 && py == 0) return norm2(0.0 - x1, 0.0 - y1);
    double u = ((0.0 - x1) * px + (0.0 - y1) * py) / (px * px + py * py);
    if (u > 1) u = 1; else if (u < 0) u = 0;
    const double x = x1 + u * px, y = y1 + u * py;
    return norm2(x - 0.0, y - 0.0);
}

// Python's float % for a positive divisor
__device__ __forceinline__ double pymod(double a, double m)
{
    double r = fmod(a, m);
    if (r != 0 && r < 0) r += m;
    return r;
}

}  // namespace mcn
