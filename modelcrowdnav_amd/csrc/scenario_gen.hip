// scenario_gen.hip -- device-side crowd scenarios for rollouts that need more distinct cases than a host-built pool
// can hold (the reference's train phase counts cases up to uint32max - 2000, crowd_sim.py:65).
//
// Same placement RULES as the reference (crowd_sim.py:165-215: circle crossing = random angle on the circle plus
// uniform noise, goal at the antipode; square crossing = random start on one side, random goal on the other;
// rejection against the robot and the humans placed so far with gap radius_i + radius_j + discomfort_dist; optional
// random v_pref in [0.5, 1.5) / radius in [0.3, 0.5), agent.py:39-45) but NOT the reference's random STREAM: numpy's
// MT19937 per case is replaced by a counter-based generator (splitmix64 of (seed, case, draw index)), and cos / sin
// are the device's.  Scenarios are therefore statistically equivalent, reproducible and partition-invariant (a case
// depends on its id only), but not bit-identical to CrowdSim.reset -- the parity paths keep using the host generator
// (envs/scenarios.py), which is.  One lane per case; the rejection loops are bounded.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "env_common.hpp"

namespace mcn {

struct CaseRng {
    uint64_t key, ctr;
    __device__ double next()                         // uniform in [0, 1), 53 bits
    {
        uint64_t z = key + (++ctr) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
};

__global__ __launch_bounds__(64) void scenario_pool_kernel(const mcn_scenario_cfg c, const uint64_t seed,
                                                           const int64_t first_case, const int P, const int N,
                                                           double *hpos, double *hgoal, double *hrad, double *hvpref)
{
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= P) return;
    CaseRng rng;
    rng.key = seed ^ ((uint64_t)(first_case + i) * 0xD1B54A32D192ED03ull);
    rng.ctr = 0;
    double2 *pos = reinterpret_cast<double2 *>(hpos) + i * N;
    double2 *goal = reinterpret_cast<double2 *>(hgoal) + i * N;
    double *rad = hrad + i * N, *vp = hvpref + i * N;
    constexpr int kMaxTries = 4096;                  // a crowd that cannot be placed keeps its last draw
    for (int h = 0; h < N; ++h) {
        double v_pref = c.human_v_pref, radius = c.human_radius;
        if (c.randomize_attributes) {                // agent.py:39-45: v_pref first, then radius
            v_pref = 0.5 + rng.next();
            radius = 0.3 + 0.2 * rng.next();
        }
        double px = 0, py = 0, gx = 0, gy = 0;
        if (c.rule == MCN_RULE_CIRCLE) {
            for (int tr = 0; tr < kMaxTries; ++tr) {
                const double angle = rng.next() * M_PI * 2;
                const double nx = (rng.next() - 0.5) * v_pref, ny = (rng.next() - 0.5) * v_pref;
                px = c.circle_radius * cos(angle) + nx;
                py = c.circle_radius * sin(angle) + ny;
                double gap = radius + c.robot_radius + c.discomfort_dist;
                bool collide = norm2(px - c.robot_start[0], py - c.robot_start[1]) < gap ||
                               norm2(px - c.robot_goal[0], py - c.robot_goal[1]) < gap;
                for (int q = 0; q < h && !collide; ++q) {
                    gap = radius + rad[q] + c.discomfort_dist;
                    collide = norm2(px - pos[q].x, py - pos[q].y) < gap || norm2(px - goal[q].x, py - goal[q].y) < gap;
                }
                if (!collide) break;
            }
            gx = -px; gy = -py;
        } else {
            const double sign = rng.next() > 0.5 ? -1.0 : 1.0;
            for (int tr = 0; tr < kMaxTries; ++tr) {
                px = rng.next() * c.square_width * 0.5 * sign;
                py = (rng.next() - 0.5) * c.square_width;
                bool collide = norm2(px - c.robot_start[0], py - c.robot_start[1]) < radius + c.robot_radius + c.discomfort_dist;
                for (int q = 0; q < h && !collide; ++q)
                    collide = norm2(px - pos[q].x, py - pos[q].y) < radius + rad[q] + c.discomfort_dist;
                if (!collide) break;
            }
            for (int tr = 0; tr < kMaxTries; ++tr) {
                gx = rng.next() * c.square_width * 0.5 * -sign;
                gy = (rng.next() - 0.5) * c.square_width;
                bool collide = norm2(gx - c.robot_goal[0], gy - c.robot_goal[1]) < radius + c.robot_radius + c.discomfort_dist;
                for (int q = 0; q < h && !collide; ++q)
                    collide = norm2(gx - goal[q].x, gy - goal[q].y) < radius + rad[q] + c.discomfort_dist;
                if (!collide) break;
            }
        }
        pos[h] = make_double2(px, py);
        goal[h] = make_double2(gx, gy);
        rad[h] = radius;
        vp[h] = v_pref;
    }
}

int launch_scenario_pool(const mcn_scenario_cfg &c, uint64_t seed, int64_t first_case, int P, int N, double *hpos,
                         double *hgoal, double *hrad, double *hvpref, hipStream_t stream)
{
    hipLaunchKernelGGL(scenario_pool_kernel, dim3((P + 63) / 64), dim3(64), 0, stream, c, seed, first_case, P, N,
                       hpos, hgoal, hrad, hvpref);
    return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
}

}  // namespace mcn
