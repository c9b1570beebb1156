// env_step_quad.hip -- latency-oriented fused CrowdSim.step for crowds with <= 4 ORCA neighbours per human
// (human_num <= 5 with an invisible robot: the reference's default configuration, env.config:22,33).
//
// Same arithmetic, bit for bit, as env_step.hip; different decomposition.  env_step.hip gives each human one
// lane, which then builds its 4 half-planes, runs the 4 steps of the incremental LP and (rarely) the 3-D LP one
// after the other -- a ~2000-instruction dependent chain that sets the latency of a small batch.  Here each
// human owns a QUAD of lanes, lane k <-> k-th candidate neighbour:
//   * every lane loads its own candidate straight from L2 and builds ONE half-plane;
//   * the distance ranking is a handful of DPP quad broadcasts; the half-planes are routed to their sorted slot
//     with one ds_permute per component and then shared by quad broadcast, so every lane holds all four;
//   * the 1-D LP on line i (RVO2 linearProgram1) depends on the lines before it and on the preferred velocity,
//     but NOT on the running result -- only the cheap violation test does.  So lane i solves line i's 1-D LP
//     speculatively, all four at once, and the incremental LP collapses into four compare-and-take steps.
//     The same holds for the 3-D LP: the candidate of line i (project lines < i, direction-optimising 2-D LP)
//     is independent of the running result, so the quad computes all candidates in parallel as well;
//   * the swept-circle test runs on lane 0 of each quad, the human-human overlap tests one pair per lane.
// Control flow is uniform across the lanes of a quad (run-time line index, predicated inner steps), so the
// parallelism is real rather than divergence.  No LDS, no barrier: operands shared inside an env are fetched
// by every lane that needs them (same-address loads coalesce) and per-env results are recomputed redundantly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "quad_common.hpp"
#include "env_step_params.hpp"
#include "env_common.hpp"

namespace mcn {
void note_dispatch(const char *family);          // mcn_api.hip: mcn_last_dispatch()

// SPLIT = true: the workgroup has two wavefronts working on the same G envs.  Wavefront 0 solves ORCA (float32),
// wavefront 1 does the float64 swept-circle / overlap tests, the reward ladder and all per-env outputs AT THE
// SAME TIME; they meet at one barrier (human velocities one way, done flag / reset case the other way through
// LDS) and wavefront 0 then integrates the humans.  The critical path becomes max(ORCA, pairwise) instead of
// their sum.  SPLIT = false: one wavefront does both in sequence.
template <int NT, int VIS, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 128 : 64) void env_step_quad_kernel(const StepParams p)
{
    __shared__ int s_dn[16], s_case[16];
#ifdef MCN_DIAG
    if (p.debug_noop) return;      // diagnostic build only: launch-floor measurement
#endif
    const int role = SPLIT ? (int)(threadIdx.x >> 6) : -1;      // 0: ORCA, 1: pairwise + ladder, -1: both
    const bool do_orca = role != 1, do_pair = role != 0;
    constexpr int NC = NT - 1 + VIS;          // candidates per human, <= 4
    constexpr int LPE = 4 * NT;               // lanes per env
    constexpr int G = 64 / LPE;               // envs per wavefront
    static_assert(NC >= 0 && NC <= 4, "quad kernel handles at most 4 ORCA neighbours");
    const int lane = threadIdx.x & 63;
    const int g = lane / LPE;
    const int r = lane - g * LPE;
    const int h = r >> 2, k = r & 3;
    // XCD-aware chunking (as env_step_kernel): workgroups are dealt round-robin over the 8 XCDs, each with a private
    // L2, and a workgroup's envs cover only part of a cache line of the per-env arrays (3 envs x 16 B at 5 humans):
    // give every XCD one contiguous range of envs so that a line is fetched into ONE L2 instead of two or three.
    // A/B switch MCN_QUAD_XCD (1 = on).
#ifndef MCN_QUAD_XCD
#define MCN_QUAD_XCD 1
#endif
    const unsigned nb_ = gridDim.x, xcd_ = blockIdx.x & 7u, idx_ = blockIdx.x >> 3;
    const unsigned qq_ = nb_ >> 3, rr__ = nb_ & 7u;
    const unsigned chunk_ = MCN_QUAD_XCD ? (xcd_ < rr__ ? xcd_ * (qq_ + 1) : rr__ * (qq_ + 1) + (xcd_ - rr__) * qq_) + idx_
                                         : blockIdx.x;
    const long e = (long)chunk_ * G + g;
    const bool active = (g < G) && (e < p.E);
    const long eb = active ? e : 0;
    const long a = eb * NT + h;
    const mcn_env_cfg &c = p.cfg;
    const double dt = c.time_step;
    const bool lead = active && r == 0 && do_pair;       // writes the per-env outputs
    const bool hlead = active && k == 0 && do_orca;      // writes the human's outputs

    // ---- loads: own human, own candidate, robot (same-address loads across a quad / env coalesce) ----
    const double2 pos = reinterpret_cast<const double2 *>(p.st.hpos)[a];
    const double2 vel = reinterpret_cast<const double2 *>(p.st.hvel)[a];
    const double2 goal = reinterpret_cast<const double2 *>(p.st.hgoal)[a];
    const double rad = p.st.hrad[a];
    const double vpref = p.st.hvpref[a];
    const double2 rpos = reinterpret_cast<const double2 *>(p.st.rpos)[eb];
    const double2 rgoal = reinterpret_cast<const double2 *>(p.st.rgoal)[eb];
    const double2 act = reinterpret_cast<const double2 *>(p.actions)[eb];
    const double rrad = p.st.rrad[eb];
    const double gtime = p.st.gtime[eb];
    double rtheta = 0;
    if (c.robot_kinematics == MCN_KIN_UNICYCLE) rtheta = p.st.rtheta[eb];
    const bool cand_h = k < NT - 1;                          // candidate is another human
    const bool cand_r = VIS && (k == NT - 1);                // candidate is the robot
    const int j = cand_h ? k + (k >= h ? 1 : 0) : h;
    const long ca = eb * NT + j;
    double2 cpos = reinterpret_cast<const double2 *>(p.st.hpos)[ca];
    double2 cvel = reinterpret_cast<const double2 *>(p.st.hvel)[ca];
    double crd = p.st.hrad[ca];
    if (cand_r) {
        cpos = rpos;
        cvel = reinterpret_cast<const double2 *>(p.st.rvel)[eb];
        crd = rrad;
    }
    int next_case = 0;
    double ep_disc = 0;
    mcn_roll_rec rs = {0, 0, 0, 0, 0, 0};
    if (lead && p.has_roll) {
        if (p.roll.state) {
            rs = p.roll.state[e];                       // one 32-byte record
            next_case = rs.next_case;
            ep_disc = p.roll.disc_table[rs.ep_steps < p.roll.disc_len ? rs.ep_steps : p.roll.disc_len - 1];
        }
    }

    // ---- K1: ORCA, one half-plane per lane ----
    double hax = 0, hay = 0;
    if (do_orca) {
    float rx, ry;
    // (opaque reciprocals: the compiler would fold `apart ? 1 / a : 1 / b` into one IEEE division after the select)
    float inv_th = 1.0f / c.orca_time_horizon, inv_ts = 1.0f / (float)dt;
    asm("" : "+v"(inv_th), "+v"(inv_ts));
    quad_orca_velocity(c, lane, k, cand_h || cand_r, pos, vel, goal, rad, vpref,
                       make_float4((float)cpos.x, (float)cpos.y, (float)cvel.x, (float)cvel.y), crd, inv_th, inv_ts, rx, ry);
    hax = (double)rx; hay = (double)ry;
    }

    // ---- K2: swept circle on lane 0 of the quad, one human-human pair per lane ----
    const int l0 = lane - r;
    double rew = 0, dmin = INFINITY, endx = 0, endy = 0, new_theta = rtheta, nrvx = 0, nrvy = 0;
    int dn = 0, inf = MCN_INFO_NOTHING;
    const double t_new = gtime + dt;
    if (do_pair) {
    double2 eff = act;
    if (c.robot_kinematics == MCN_KIN_UNICYCLE) {
        eff.x = act.x * cos(act.y + rtheta);
        eff.y = act.x * sin(act.y + rtheta);
    }
    // every lane of the quad evaluates the (same) swept distance: no branch, so this float64 chain interleaves
    // with the overlap test and the goal test below instead of serialising behind a divergent region
    double cd;
    {
        const double px = pos.x - rpos.x, py = pos.y - rpos.y;
        const double vx = vel.x - eff.x, vy = vel.y - eff.y;
        cd = p2s_origin(px, py, px + vx * dt, py + vy * dt) - rad - rrad;
    }
    int hh;
    {                                           // each unordered pair exactly once (crowd_sim.py:369-374)
        const double dx = pos.x - cpos.x, dy = pos.y - cpos.y;
        const bool counted = c.count_hh && cand_h && j > h;
        hh = (counted && (sqrt(dx * dx + dy * dy) - rad - crd) < 0) ? 1 : 0;
    }
    hh += __builtin_amdgcn_update_dpp(0, hh, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    hh += __builtin_amdgcn_update_dpp(0, hh, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    int hh_sum = 0;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int src = (l0 + 4 * q) & 63;
        dmin = fmin(dmin, __shfl(cd, src));
        hh_sum += __shfl(hh, src);
    }

    // ---- K3: ladder, recomputed by every lane of the env (cheap, avoids a broadcast) ----
    if (c.robot_kinematics == MCN_KIN_UNICYCLE) {
        const double th = rtheta + act.y;
        endx = rpos.x + cos(th) * act.x * dt;
        endy = rpos.y + sin(th) * act.x * dt;
        new_theta = pymod(rtheta + act.y, 2 * M_PI);
        nrvx = act.x * cos(new_theta); nrvy = act.x * sin(new_theta);
    } else {
        endx = rpos.x + act.x * dt; endy = rpos.y + act.y * dt;
        nrvx = act.x; nrvy = act.y;
    }
    const bool reaching = norm2(endx - rgoal.x, endy - rgoal.y) < rrad;
    if (gtime >= c.time_limit - 1)      { rew = 0; dn = 1; inf = MCN_INFO_TIMEOUT; }
    else if (dmin < 0)                  { rew = c.collision_penalty; dn = 1; inf = MCN_INFO_COLLISION; }
    else if (reaching)                  { rew = c.success_reward; dn = 1; inf = MCN_INFO_REACHGOAL; }
    else if (dmin < c.discomfort_dist)  { rew = (dmin - c.discomfort_dist) * c.discomfort_penalty_factor * dt; dn = 0; inf = MCN_INFO_DANGER; }
    else                                { rew = 0; dn = 0; inf = MCN_INFO_NOTHING; }
    if (lead) {
        store_step_rec(p.out.rec + e, rew, dmin, dn, inf, hh_sum);   // 16 + 8 bytes
    }
    }   // do_pair
    if (SPLIT) {
        // hand-off: wavefront 1 publishes the done flag (and the reset case), wavefront 0 consumes them
        if (lead) { s_dn[g] = dn; s_case[g] = next_case; }
        __syncthreads();
        if (role == 0) { dn = s_dn[g < 16 ? g : 0]; next_case = s_case[g < 16 ? g : 0]; }
    }
    if (hlead && p.out.human_act) reinterpret_cast<double2 *>(p.out.human_act)[a] = make_double2(hax, hay);

    // ---- integrate / look ahead ----
    const double npx = pos.x + hax * dt, npy = pos.y + hay * dt;
    if (!p.update) {
        if (hlead) {
            reinterpret_cast<double2 *>(p.out.nobs_pos)[a] = make_double2(npx, npy);
            reinterpret_cast<double2 *>(p.out.nobs_vel)[a] = make_double2(hax, hay);
        }
        return;
    }
    const bool do_reset = p.has_roll && p.roll.pool_hpos != nullptr;
    const int case_g = SPLIT ? next_case : __shfl(next_case, l0 & 63);
    if (hlead) {
        if (do_reset && dn) {
            const long pa = (long)case_g * NT + h;
            reinterpret_cast<double2 *>(p.st.hpos)[a]  = reinterpret_cast<const double2 *>(p.roll.pool_hpos)[pa];
            reinterpret_cast<double2 *>(p.st.hgoal)[a] = reinterpret_cast<const double2 *>(p.roll.pool_hgoal)[pa];
            p.st.hrad[a] = p.roll.pool_hrad[pa];
            p.st.hvpref[a] = p.roll.pool_hvpref[pa];
            reinterpret_cast<double2 *>(p.st.hvel)[a]  = p.roll.pool_hvel
                ? reinterpret_cast<const double2 *>(p.roll.pool_hvel)[pa] : make_double2(0, 0);
            if (p.st.human_times) p.st.human_times[a] = 0;
        } else {
            reinterpret_cast<double2 *>(p.st.hpos)[a] = make_double2(npx, npy);
            reinterpret_cast<double2 *>(p.st.hvel)[a] = make_double2(hax, hay);
            if (c.track_human_times && p.st.human_times) {
                if (p.st.human_times[a] == 0 && norm2(npx - goal.x, npy - goal.y) < rad)
                    p.st.human_times[a] = t_new;
            }
        }
    }
    if (lead) {
        if (p.has_roll) {
            const mcn_rollout &ro = p.roll;
            if (ro.state) {
                if (inf == MCN_INFO_DANGER && (ro.danger_episodes <= 0 ||
            rs.fin_count < ro.danger_episodes - ((ro.danger_short_from > 0 && e >= ro.danger_short_from - 1) ? 1 : 0))) {
                    rs.danger_count += 1; rs.danger_dist_sum += dmin;
                }
                const double ret = rs.ep_return + ep_disc * rew;
                if (dn) {
                    const int k = rs.fin_count;
                    // fin_slots == 1: keep the latest episode; otherwise keep the first fin_slots episodes
                    const bool keep = (ro.fin_slots == 1) || (k < ro.fin_slots);
                    const long rec = (long)(ro.fin_slots == 1 ? 0 : k) * p.E + e;
                    if (keep && ro.fin_return) ro.fin_return[rec] = ret;
                    if (keep && ro.fin_time)   ro.fin_time[rec] = (inf == MCN_INFO_TIMEOUT) ? c.time_limit : t_new;
                    if (keep && ro.fin_info)   ro.fin_info[rec] = (uint8_t)inf;
                    rs.fin_count = k + 1; rs.ep_return = 0; rs.ep_steps = 0;
                    if (do_reset) {
                        const int nc = next_case + ro.case_stride;
                        rs.next_case = nc >= ro.pool_size ? nc - ro.pool_size : nc;
                    }
                } else {
                    rs.ep_return = ret; rs.ep_steps += 1;
                }
                ro.state[e] = rs;                         // one 32-byte store
            }
        }
        if (do_reset && dn) {
            reinterpret_cast<double2 *>(p.st.rpos)[e]  = make_double2(p.roll.robot_start[0], p.roll.robot_start[1]);
            reinterpret_cast<double2 *>(p.st.rgoal)[e] = make_double2(p.roll.robot_goal[0], p.roll.robot_goal[1]);
            reinterpret_cast<double2 *>(p.st.rvel)[e]  = make_double2(0, 0);
            if (p.st.rtheta) p.st.rtheta[e] = p.roll.robot_theta0;
            p.st.gtime[e] = 0;
        } else {
            reinterpret_cast<double2 *>(p.st.rpos)[e] = make_double2(endx, endy);
            reinterpret_cast<double2 *>(p.st.rvel)[e] = make_double2(nrvx, nrvy);
            if (c.robot_kinematics == MCN_KIN_UNICYCLE) p.st.rtheta[e] = new_theta;
            p.st.gtime[e] = t_new;
        }
    }
}

template <int NT, int VIS>
static void launch_quad_one(const StepParams &p, hipStream_t stream)
{
    constexpr int G = 64 / (4 * NT);
    const int blocks = (p.E + G - 1) / G;
    if (p.quad_split)
        hipLaunchKernelGGL((env_step_quad_kernel<NT, VIS, true>), dim3(blocks), dim3(128), 0, stream, p);
    else
        hipLaunchKernelGGL((env_step_quad_kernel<NT, VIS, false>), dim3(blocks), dim3(64), 0, stream, p);
}

// Returns true when the quad kernel handles this problem (ORCA humans, <= 4 neighbours each).
bool launch_env_step_quad(const StepParams &p, hipStream_t stream)
{
    if (p.cfg.human_policy != MCN_HUMANS_ORCA || p.cfg.orca_max_neighbors < 4) return false;
    const int vis = p.cfg.robot_visible ? 1 : 0;
    const int nc = p.N - 1 + vis;
    if (nc > 4 || p.N < 1) return false;
    switch (p.N * 2 + vis) {
        case 2:  launch_quad_one<1, 0>(p, stream); break;
        case 3:  launch_quad_one<1, 1>(p, stream); break;
        case 4:  launch_quad_one<2, 0>(p, stream); break;
        case 5:  launch_quad_one<2, 1>(p, stream); break;
        case 6:  launch_quad_one<3, 0>(p, stream); break;
        case 7:  launch_quad_one<3, 1>(p, stream); break;
        case 8:  launch_quad_one<4, 0>(p, stream); break;
        case 9:  launch_quad_one<4, 1>(p, stream); break;
        case 10: launch_quad_one<5, 0>(p, stream); break;
        default: return false;
    }
    note_dispatch("env_step_quad_kernel");
    return true;
}

}  // namespace mcn
