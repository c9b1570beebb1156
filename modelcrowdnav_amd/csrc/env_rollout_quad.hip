// env_rollout_quad.hip -- T consecutive CrowdSim.step calls in ONE launch, for action sequences known up front
// (random / scripted robots: Explorer-style rollouts with a non-learning robot, benchmark config 2).
//
// A single step of a 4096-env batch is latency-bound: launch floor (~1.7 us) + state fetch from L2/HBM + the ORCA
// and swept-circle chains + write-back, of which only the arithmetic is inherent.  Here the quad layout of
// env_step_quad.hip (4 lanes per human, lane k <-> candidate neighbour k) keeps the whole env state in REGISTERS
// across the T steps: humans on their quad, the robot and the clock redundantly on every lane of the env.  What a
// step of env_step_quad.hip fetches from memory -- the candidate neighbour's position / velocity / radius --
// becomes a ds_bpermute from the quad that owns that human.  Per step the only memory operations are the
// prefetched action, the discount-table entry, and (when an episode ends) the finished-episode record and the
// scenario-pool fetch of the in-kernel reset.  Arithmetic, order of operations and results are those of T
// mcn_env_step calls, bit for bit (tests/test_env_step_gpu.py::test_rollout_launch_equals_single_steps).
//
// Every lane runs exactly T iterations, there is no inter-wavefront dependency and no barrier: the grid drains.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "quad_common.hpp"
#include "env_step_params.hpp"
#include "env_common.hpp"

namespace mcn {

__device__ __forceinline__ int bperm_i(int byte_addr, int v) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }
__device__ __forceinline__ float bperm_f(int byte_addr, float v)
{
    return __builtin_bit_cast(float, bperm_i(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double bperm_d(int byte_addr, double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = bperm_i(byte_addr, (int)b), hi = bperm_i(byte_addr, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <int NT, int VIS>
__global__ __launch_bounds__(64) void env_rollout_quad_kernel(const StepParams p, const int T)
{
    if (p.debug_noop) return;
    constexpr int NC = NT - 1 + VIS;          // candidates per human, <= 4
    constexpr int LPE = 4 * NT;               // lanes per env
    constexpr int G = 64 / LPE;               // envs per wavefront
    static_assert(NC >= 0 && NC <= 4, "quad kernel handles at most 4 ORCA neighbours");
    const int lane = threadIdx.x & 63;
    const int g = lane / LPE;
    const int r = lane - g * LPE;
    const int h = r >> 2, k = r & 3;
    const long e = (long)blockIdx.x * G + g;
    const bool active = (g < G) && (e < p.E);
    const long eb = active ? e : 0;
    const long a = eb * NT + h;
    const mcn_env_cfg &c = p.cfg;
    const mcn_rollout &ro = p.roll;
    const double dt = c.time_step;
    const bool lead = active && r == 0;       // owns the per-env records
    const bool hlead = active && k == 0;      // owns the human's records
    const bool unicycle = c.robot_kinematics == MCN_KIN_UNICYCLE;
    const bool has_state = p.has_roll && ro.state != nullptr;
    const bool do_reset = p.has_roll && ro.pool_hpos != nullptr;
    const bool track = c.track_human_times && p.st.human_times != nullptr;

    // ---- state -> registers (once per launch) ----
    double2 pos = reinterpret_cast<const double2 *>(p.st.hpos)[a];
    double2 vel = reinterpret_cast<const double2 *>(p.st.hvel)[a];
    double2 goal = reinterpret_cast<const double2 *>(p.st.hgoal)[a];
    double rad = p.st.hrad[a];
    double vpref = p.st.hvpref[a];
    double htime = p.st.human_times ? p.st.human_times[a] : 0.0;
    double2 rpos = reinterpret_cast<const double2 *>(p.st.rpos)[eb];
    double2 rvel = reinterpret_cast<const double2 *>(p.st.rvel)[eb];
    double2 rgoal = reinterpret_cast<const double2 *>(p.st.rgoal)[eb];
    const double rrad = p.st.rrad[eb];
    double gtime = p.st.gtime[eb];
    double rtheta = p.st.rtheta ? p.st.rtheta[eb] : 0.0;
    mcn_roll_rec rs = {0, 0, 0, 0, 0, 0};
    if (lead && has_state) rs = ro.state[e];

    const bool cand_h = k < NT - 1;                          // candidate is another human
    const bool cand_r = VIS && (k == NT - 1);                // candidate is the robot
    const int j = cand_h ? k + (k >= h ? 1 : 0) : h;
    const int l0 = lane - r;
    const int src = ((l0 + 4 * j) & 63) << 2;                // a lane of the quad that owns human j

    double2 act_next = reinterpret_cast<const double2 *>(p.actions)[eb];
    mcn_step_rec o_last = {0, 0, 0, 0, 0, 0};
    double hax = 0, hay = 0;

    for (int t = 0; t < T; ++t) {
        const double2 act = act_next;
        if (t + 1 < T) act_next = reinterpret_cast<const double2 *>(p.actions)[(long)(t + 1) * p.E + eb];
        double ep_disc = 0;
        if (lead && has_state) ep_disc = ro.disc_table[rs.ep_steps < ro.disc_len ? rs.ep_steps : ro.disc_len - 1];

        // ---- candidate neighbour: from the quad that owns it instead of from memory ----
        double2 cpos;
        cpos.x = bperm_d(src, pos.x);
        cpos.y = bperm_d(src, pos.y);
        float cvx = bperm_f(src, (float)vel.x), cvy = bperm_f(src, (float)vel.y);
        double crd = bperm_d(src, rad);
        if (cand_r) { cpos = rpos; cvx = (float)rvel.x; cvy = (float)rvel.y; crd = rrad; }

        // ---- K1: ORCA ----
        float rx, ry;
        quad_orca_velocity(c, lane, k, cand_h || cand_r, pos, vel, goal, rad, vpref,
                           make_float4((float)cpos.x, (float)cpos.y, cvx, cvy), crd, dt, rx, ry);
        hax = (double)rx; hay = (double)ry;

        // ---- K2: swept circle per quad, one human-human pair per lane (as env_step_quad.hip) ----
        double2 eff = act;
        if (unicycle) {
            eff.x = act.x * cos(act.y + rtheta);
            eff.y = act.x * sin(act.y + rtheta);
        }
        double cd;
        {
            const double px = pos.x - rpos.x, py = pos.y - rpos.y;
            const double vx = vel.x - eff.x, vy = vel.y - eff.y;
            cd = p2s_origin(px, py, px + vx * dt, py + vy * dt) - rad - rrad;
        }
        int hh;
        {
            const double dx = pos.x - cpos.x, dy = pos.y - cpos.y;
            const bool counted = c.count_hh && cand_h && j > h;
            hh = (counted && (sqrt(dx * dx + dy * dy) - rad - crd) < 0) ? 1 : 0;
        }
        hh += __builtin_amdgcn_update_dpp(0, hh, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
        hh += __builtin_amdgcn_update_dpp(0, hh, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
        double dmin = INFINITY;
        int hh_sum = 0;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int s = (l0 + 4 * q) & 63;
            dmin = fmin(dmin, __shfl(cd, s));
            hh_sum += __shfl(hh, s);
        }

        // ---- K3: ladder, on every lane of the env ----
        double endx, endy, new_theta = rtheta, nrvx, nrvy;
        if (unicycle) {
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
        double rew; int dn, inf;
        if (gtime >= c.time_limit - 1)      { rew = 0; dn = 1; inf = MCN_INFO_TIMEOUT; }
        else if (dmin < 0)                  { rew = c.collision_penalty; dn = 1; inf = MCN_INFO_COLLISION; }
        else if (reaching)                  { rew = c.success_reward; dn = 1; inf = MCN_INFO_REACHGOAL; }
        else if (dmin < c.discomfort_dist)  { rew = (dmin - c.discomfort_dist) * c.discomfort_penalty_factor * dt; dn = 0; inf = MCN_INFO_DANGER; }
        else                                { rew = 0; dn = 0; inf = MCN_INFO_NOTHING; }
        o_last.reward = rew; o_last.dmin = dmin; o_last.done = (uint8_t)dn; o_last.info = (uint8_t)inf;
        o_last.hh_count = hh_sum;
        const double t_new = gtime + dt;

        // ---- Explorer accounting on the env's lead lane (explorer.py:88-99,124) ----
        const int case_g = __shfl(rs.next_case, l0 & 63);
        if (lead && has_state) {
            if (inf == MCN_INFO_DANGER) { rs.danger_count += 1; rs.danger_dist_sum += dmin; }
            const double ret = rs.ep_return + ep_disc * rew;
            if (dn) {
                const int kf = rs.fin_count;
                const bool keep = (ro.fin_slots == 1) || (kf < ro.fin_slots);
                const long rec = (long)(ro.fin_slots == 1 ? 0 : kf) * p.E + e;
                if (keep && ro.fin_return) ro.fin_return[rec] = ret;
                if (keep && ro.fin_time)   ro.fin_time[rec] = (inf == MCN_INFO_TIMEOUT) ? c.time_limit : t_new;
                if (keep && ro.fin_info)   ro.fin_info[rec] = (uint8_t)inf;
                rs.fin_count = kf + 1; rs.ep_return = 0; rs.ep_steps = 0;
                if (do_reset) rs.next_case = (rs.next_case + ro.case_stride) % ro.pool_size;
            } else {
                rs.ep_return = ret; rs.ep_steps += 1;
            }
        }

        // ---- integrate, or restart from the scenario pool ----
        if (do_reset && dn) {
            if (active) {
                const long pa = (long)case_g * NT + h;
                pos = reinterpret_cast<const double2 *>(ro.pool_hpos)[pa];
                goal = reinterpret_cast<const double2 *>(ro.pool_hgoal)[pa];
                rad = ro.pool_hrad[pa];
                vpref = ro.pool_hvpref[pa];
                vel = ro.pool_hvel ? reinterpret_cast<const double2 *>(ro.pool_hvel)[pa] : make_double2(0, 0);
            }
            htime = 0;
            rpos = make_double2(ro.robot_start[0], ro.robot_start[1]);
            rgoal = make_double2(ro.robot_goal[0], ro.robot_goal[1]);
            rvel = make_double2(0, 0);
            if (p.st.rtheta) rtheta = ro.robot_theta0;
            gtime = 0;
        } else {
            pos = make_double2(pos.x + hax * dt, pos.y + hay * dt);
            vel = make_double2(hax, hay);
            if (track && htime == 0 && norm2(pos.x - goal.x, pos.y - goal.y) < rad) htime = t_new;   // agent.py:137-138
            rpos = make_double2(endx, endy);
            rvel = make_double2(nrvx, nrvy);
            if (unicycle) rtheta = new_theta;
            gtime = t_new;
        }
    }

    // ---- registers -> state (once per launch) ----
    if (hlead) {
        reinterpret_cast<double2 *>(p.st.hpos)[a] = pos;
        reinterpret_cast<double2 *>(p.st.hvel)[a] = vel;
        if (do_reset) {
            reinterpret_cast<double2 *>(p.st.hgoal)[a] = goal;
            p.st.hrad[a] = rad;
            p.st.hvpref[a] = vpref;
        }
        if (p.st.human_times) p.st.human_times[a] = htime;
        if (p.out.human_act) reinterpret_cast<double2 *>(p.out.human_act)[a] = make_double2(hax, hay);
    }
    if (lead) {
        reinterpret_cast<double2 *>(p.st.rpos)[e] = rpos;
        reinterpret_cast<double2 *>(p.st.rvel)[e] = rvel;
        if (do_reset) reinterpret_cast<double2 *>(p.st.rgoal)[e] = rgoal;
        if (p.st.rtheta) p.st.rtheta[e] = rtheta;
        p.st.gtime[e] = gtime;
        p.out.rec[e] = o_last;
        if (has_state) ro.state[e] = rs;
    }
}

template <int NT, int VIS>
static void launch_rollout_one(const StepParams &p, int T, hipStream_t stream)
{
    constexpr int G = 64 / (4 * NT);
    const int blocks = (p.E + G - 1) / G;
    hipLaunchKernelGGL((env_rollout_quad_kernel<NT, VIS>), dim3(blocks), dim3(64), 0, stream, p, T);
}

// Returns true when the fused T-step kernel handles this problem (ORCA humans, <= 4 neighbours each, update).
bool launch_env_rollout_quad(const StepParams &p, int T, hipStream_t stream)
{
    if (p.cfg.human_policy != MCN_HUMANS_ORCA || p.cfg.orca_max_neighbors < 4 || !p.update) return false;
    const int vis = p.cfg.robot_visible ? 1 : 0;
    const int nc = p.N - 1 + vis;
    if (nc > 4 || p.N < 1) return false;
    switch (p.N * 2 + vis) {
        case 2:  launch_rollout_one<1, 0>(p, T, stream); break;
        case 3:  launch_rollout_one<1, 1>(p, T, stream); break;
        case 4:  launch_rollout_one<2, 0>(p, T, stream); break;
        case 5:  launch_rollout_one<2, 1>(p, T, stream); break;
        case 6:  launch_rollout_one<3, 0>(p, T, stream); break;
        case 7:  launch_rollout_one<3, 1>(p, T, stream); break;
        case 8:  launch_rollout_one<4, 0>(p, T, stream); break;
        case 9:  launch_rollout_one<4, 1>(p, T, stream); break;
        case 10: launch_rollout_one<5, 0>(p, T, stream); break;
        default: return false;
    }
    return true;
}

}  // namespace mcn
