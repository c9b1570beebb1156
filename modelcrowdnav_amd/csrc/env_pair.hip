// env_pair.hip -- the given-velocity step (ModelCrowdSim.step, model_crowd_sim.py:347-441: humans advanced by
// velocities a world model supplies; robot-human swept-circle test, reward ladder, integration, Explorer record,
// restart from the scenario pool) as a pure STREAMING kernel for large batches.  Same arithmetic, bit for bit, as
// env_step_kernel<.., MCN_HUMANS_GIVEN, 0> in env_step.hip; different memory behaviour.
//
// At a million envs that kernel is bound by what it keeps in flight, not by bytes: the per-env data (robot pose,
// goal, action, clock, the 32-byte Explorer record) is loaded and stored by ONE lane in N -- 9 load and 7 store
// instructions per wavefront that each use 12 of 64 lanes -- the discount factor is a load that depends on the
// record, and the robot data reaches the other lanes through LDS and a workgroup barrier.  Here:
//   * the per-env data is cut into 16-byte pieces (robot position, goal, action, the two halves of the record)
//     and lane h of the env loads piece h: one global_load_dwordx4 with a per-lane base pointer, plus one 8-byte
//     load for radius / clock on lanes 0 and 1;
//   * the pieces are exchanged inside the wavefront through a private LDS slab (a wavefront's DS operations
//     execute in order: no barrier), so EVERY lane of the env holds the robot state and runs the ladder and the
//     Explorer accounting redundantly (the rollout kernel's idea);
//   * the discount table (<= 128 entries) is copied to LDS by each wavefront at entry, next to its other loads,
//     so the lookup by ep_steps is an LDS read instead of a dependent global load;
//   * the results are written the same way: lane h stores 16-byte piece h of the env's outputs (new robot
//     position, velocity, step record, two record halves), lanes 0 / 1 the 8-byte ones (clock, record tail).
// Per wavefront: 6 load + 4 store instructions instead of 13 + 9, no workgroup barrier, no dependent global load.
//
// Handles: compile-time N in {5, 10}, holonomic robot, update = 1, no human-human count, no first-arrival
// bookkeeping, no human_act export; everything else stays on env_step_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "env_step_params.hpp"
#include "env_common.hpp"

namespace mcn {
void note_dispatch(const char *family);          // mcn_api.hip: mcn_last_dispatch()

__device__ __forceinline__ long long d2ll(double v) { return __builtin_bit_cast(long long, v); }
__device__ __forceinline__ double ll2d(long long v) { return __builtin_bit_cast(double, v); }

constexpr int kPairDiscMax = 128;
// where the per-step discount factor gamma^(t v_pref) of the Explorer return comes from (A/B switch):
//   0  every wavefront copies the table to its own LDS slab at entry (2 global loads + 2 LDS stores per wavefront)
//   1  one copy per workgroup + a workgroup barrier
//   2  no copy: a reward of exactly 0 adds nothing to the return (x + d * 0 == x for every finite d, and the ladder's
//      zero is +0), so only lanes whose step earned a reward fetch their ONE table entry from L2
#ifndef MCN_PAIR_DISC
#define MCN_PAIR_DISC 2
#endif

// Non-temporal loads / stores for the per-human streams (positions, velocities, given velocities, radii: 3/4 of the
// bytes, each touched exactly once per step).  Measured (round 3, MI355X, 5 humans, with Explorer record and pool
// restarts): 2^20 envs (660 MB per step, 2.6 x the 256 MB Infinity Cache) 125-145 -> 109-116 us; 2^22 envs 469-580 ->
// 460-556 us; 65 536 envs (41 MB: the state of one step is still cached when the next one starts) 9.0 -> 9.3-10.1 us,
// i.e. they only pay once the step's footprint no longer fits the memory-side cache.  Marking the per-env pieces and
// outputs too loses the gain (141 us at 2^20): neighbouring wavefronts complete each other's partial lines in cache.
#ifndef MCN_PAIR_NT_ALL
#define MCN_PAIR_NT_ALL 0      // A/B: the per-env pieces and outputs non-temporal as well (round 3 at 2^20: loses; round 4 at 2^22: see DESIGN)
#endif
typedef double pair_d2v __attribute__((ext_vector_type(2)));
template <bool NTMP> __device__ __forceinline__ double2 pair_ld(const double2 *ptr)
{
    if constexpr (NTMP) {
        const pair_d2v v = __builtin_nontemporal_load(reinterpret_cast<const pair_d2v *>(ptr));
        return make_double2(v.x, v.y);
    } else {
        return *ptr;
    }
}
template <bool NTMP> __device__ __forceinline__ double pair_ld(const double *ptr)
{
    if constexpr (NTMP) return __builtin_nontemporal_load(ptr);
    else return *ptr;
}
template <bool NTMP> __device__ __forceinline__ void pair_st(double2 *ptr, double2 v)
{
    if constexpr (NTMP) {
        pair_d2v w; w.x = v.x; w.y = v.y;
        __builtin_nontemporal_store(w, reinterpret_cast<pair_d2v *>(ptr));
    } else {
        *ptr = v;
    }
}

template <int NT, bool NTMP>
__global__ __launch_bounds__(256) void env_pair_kernel(const StepParams p)
{
    constexpr int G = 64 / NT;                 // envs per wavefront
    constexpr int WPB = 4;                     // wavefronts per workgroup
    __shared__ double2 s_piece[WPB][G][5];
    __shared__ double s_small[WPB][G][2];
    __shared__ double s_cd[WPB][64];
    __shared__ double s_disc[MCN_PAIR_DISC == 0 ? WPB : 1][MCN_PAIR_DISC == 2 ? 1 : kPairDiscMax];
#ifdef MCN_DIAG
    if (p.debug_noop) return;
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / NT;
    const int h = lane - g * NT;
    // XCD-aware chunking, as env_step_kernel: every XCD (its own L2) owns one contiguous range of envs
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned qq = nb >> 3, rr_ = nb & 7u;
    const unsigned chunk = (xcd < rr_ ? xcd * (qq + 1) : rr_ * (qq + 1) + (xcd - rr_) * qq) + idx;
    const long e = (long)chunk * (WPB * G) + wave * G + g;
    const bool active = (g < G) && (e < p.E);
    const long eb = active ? e : 0;
    const long a = eb * NT + h;
    const int gi = g < G ? g : 0;
    const mcn_env_cfg &c = p.cfg;
    const mcn_rollout &ro = p.roll;
    const double dt = c.time_step;
    const bool has_state = p.has_roll && ro.state != nullptr;
    const bool do_reset = p.has_roll && ro.pool_hpos != nullptr;

    // ---- all global loads up front ----
    const double2 pos = pair_ld<NTMP>(reinterpret_cast<const double2 *>(p.st.hpos) + a);
    const double2 vel = pair_ld<NTMP>(reinterpret_cast<const double2 *>(p.st.hvel) + a);
    const double2 gv = pair_ld<NTMP>(reinterpret_cast<const double2 *>(p.given_v) + a);
    const double rad = pair_ld<NTMP>(p.st.hrad + a);
    double2 piece = make_double2(0, 0);
    {
        // piece h of env e: 0 robot position, 1 robot goal, 2 action, 3 / 4 the halves of the Explorer record
        const double2 *src = reinterpret_cast<const double2 *>(p.st.rpos);
        src = h == 1 ? reinterpret_cast<const double2 *>(p.st.rgoal) : src;
        src = h == 2 ? reinterpret_cast<const double2 *>(p.actions) : src;
        src = h >= 3 ? reinterpret_cast<const double2 *>(ro.state) : src;
        const long pi = h >= 3 ? 2 * eb + (h - 3) : eb;
        if (h < 3 || (h < 5 && has_state)) piece = (MCN_PAIR_NT_ALL && NTMP) ? pair_ld<true>(src + pi) : src[pi];
    }
    double small = 0;
    {
        const double *src = h == 0 ? p.st.rrad : p.st.gtime;
        if (h < 2) small = (MCN_PAIR_NT_ALL && NTMP) ? pair_ld<true>(src + eb) : src[eb];
    }
    if (MCN_PAIR_DISC == 0 && has_state) {
        const int len = ro.disc_len;
        s_disc[wave][lane] = lane < len ? ro.disc_table[lane] : 0.0;
        s_disc[wave][lane + 64] = lane + 64 < len ? ro.disc_table[lane + 64] : 0.0;
    }
    if (MCN_PAIR_DISC == 1 && has_state) {
        const int t = threadIdx.x;
        if (t < kPairDiscMax) s_disc[0][t] = t < ro.disc_len ? ro.disc_table[t] : 0.0;
    }

    // ---- exchange inside the wavefront ----
    if (g < G && h < 5) s_piece[wave][gi][h] = piece;       // (idle tail lanes would alias env 0's slots)
    if (g < G && h < 2) s_small[wave][gi][h] = small;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double2 rpos = s_piece[wave][gi][0], rgoal = s_piece[wave][gi][1], act = s_piece[wave][gi][2];
    const double rrad = s_small[wave][gi][0], gtime = s_small[wave][gi][1];

    // ---- swept-circle distance to the robot (crowd_sim.py:345-365) ----
    {
        const double px = pos.x - rpos.x, py = pos.y - rpos.y;
        const double vx = vel.x - act.x, vy = vel.y - act.y;
        const double ex = px + vx * dt, ey = py + vy * dt;
        s_cd[wave][lane] = p2s_origin(px, py, ex, ey) - rad - rrad;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (MCN_PAIR_DISC == 1) __syncthreads();              // the workgroup's discount table is in place
    double dmin = INFINITY;
    {
        const int l0 = (g < G ? g : 0) * NT;
#pragma unroll
        for (int k = 0; k < NT; ++k) dmin = fmin(dmin, s_cd[wave][l0 + k]);
    }

    // ---- goal test + reward ladder, on every lane of the env (crowd_sim.py:379-403) ----
    const double endx = rpos.x + act.x * dt, endy = rpos.y + act.y * dt;
    const bool reaching = norm2(endx - rgoal.x, endy - rgoal.y) < rrad;
    double rew; int dn, inf;
    if (gtime >= c.time_limit - 1)      { rew = 0; dn = 1; inf = MCN_INFO_TIMEOUT; }
    else if (dmin < 0)                  { rew = c.collision_penalty; dn = 1; inf = MCN_INFO_COLLISION; }
    else if (reaching)                  { rew = c.success_reward; dn = 1; inf = MCN_INFO_REACHGOAL; }
    else if (dmin < c.discomfort_dist)  { rew = (dmin - c.discomfort_dist) * c.discomfort_penalty_factor * dt; dn = 0; inf = MCN_INFO_DANGER; }
    else                                { rew = 0; dn = 0; inf = MCN_INFO_NOTHING; }
    const double t_new = gtime + dt;

    // ---- Explorer accounting (explorer.py:88-99,124), on every lane of the env ----
    double2 rs_lo = make_double2(0, 0), rs_hi = make_double2(0, 0);
    int case_g = 0;
    if (has_state) {
        rs_lo = s_piece[wave][gi][3]; rs_hi = s_piece[wave][gi][4];
        double ep_return = rs_lo.x;
        int ep_steps = (int)d2ll(rs_lo.y), fin_count = (int)(d2ll(rs_lo.y) >> 32);
        int next_case = (int)d2ll(rs_hi.x), danger_count = (int)(d2ll(rs_hi.x) >> 32);
        double danger_sum = rs_hi.y;
        case_g = next_case;
        const int di = ep_steps < ro.disc_len ? ep_steps : ro.disc_len - 1;
        double ep_disc = 0.0;
        if (MCN_PAIR_DISC == 2) { if (rew != 0.0) ep_disc = ro.disc_table[di]; }
        else ep_disc = s_disc[MCN_PAIR_DISC == 0 ? wave : 0][di];
        if (inf == MCN_INFO_DANGER && (ro.danger_episodes <= 0 ||
            fin_count < ro.danger_episodes - ((ro.danger_short_from > 0 && e >= ro.danger_short_from - 1) ? 1 : 0))) {
            danger_count += 1; danger_sum += dmin;
        }
        const double ret = ep_return + ep_disc * rew;
        if (dn) {
            if (active && h == 0) {
                const bool keep = (ro.fin_slots == 1) || (fin_count < ro.fin_slots);
                const long rec = (long)(ro.fin_slots == 1 ? 0 : fin_count) * p.E + e;
                if (keep && ro.fin_return) ro.fin_return[rec] = ret;
                if (keep && ro.fin_time)   ro.fin_time[rec] = (inf == MCN_INFO_TIMEOUT) ? c.time_limit : t_new;
                if (keep && ro.fin_info)   ro.fin_info[rec] = (uint8_t)inf;
            }
            fin_count += 1; ep_return = 0; ep_steps = 0;
            if (do_reset) {
                const int nc = next_case + ro.case_stride;
                next_case = nc >= ro.pool_size ? nc - ro.pool_size : nc;
            }
        } else {
            ep_return = ret; ep_steps += 1;
        }
        rs_lo = make_double2(ep_return, ll2d(((long long)fin_count << 32) | (unsigned int)ep_steps));
        rs_hi = make_double2(ll2d(((long long)danger_count << 32) | (unsigned int)next_case), danger_sum);
    }

    // ---- humans: integrate, or restart from the scenario pool ----
    const bool restart = do_reset && dn;
    if (active) {
        if (restart) {
            const long pa = (long)case_g * NT + h;
            reinterpret_cast<double2 *>(p.st.hpos)[a]  = reinterpret_cast<const double2 *>(ro.pool_hpos)[pa];
            reinterpret_cast<double2 *>(p.st.hgoal)[a] = reinterpret_cast<const double2 *>(ro.pool_hgoal)[pa];
            p.st.hrad[a] = ro.pool_hrad[pa];
            p.st.hvpref[a] = ro.pool_hvpref[pa];
            reinterpret_cast<double2 *>(p.st.hvel)[a]  = ro.pool_hvel
                ? reinterpret_cast<const double2 *>(ro.pool_hvel)[pa] : make_double2(0, 0);
            if (p.st.human_times) p.st.human_times[a] = 0;
            if (h == 0) {
                reinterpret_cast<double2 *>(p.st.rgoal)[e] = make_double2(ro.robot_goal[0], ro.robot_goal[1]);
                if (p.st.rtheta) p.st.rtheta[e] = ro.robot_theta0;
            }
        } else {
            pair_st<NTMP>(reinterpret_cast<double2 *>(p.st.hpos) + a, make_double2(pos.x + gv.x * dt, pos.y + gv.y * dt));
            pair_st<NTMP>(reinterpret_cast<double2 *>(p.st.hvel) + a, gv);
        }
    }

    // ---- per-env outputs: lane h stores 16-byte piece h, lanes 0 / 1 the 8-byte ones ----
    {
        const double2 o_rpos = restart ? make_double2(ro.robot_start[0], ro.robot_start[1]) : make_double2(endx, endy);
        const double2 o_rvel = restart ? make_double2(0, 0) : act;
        const double o_time = restart ? 0.0 : t_new;
        const unsigned long long tail = (unsigned long long)(unsigned)(dn & 0xff) | ((unsigned long long)(unsigned)(inf & 0xff) << 8);
        double2 *dst = reinterpret_cast<double2 *>(p.st.rpos) + eb;                       // h == 0
        double2 val = o_rpos;
        if (h == 1) { dst = reinterpret_cast<double2 *>(p.st.rvel) + eb; val = o_rvel; }
        if (h == 2) { dst = reinterpret_cast<double2 *>(reinterpret_cast<double *>(p.out.rec) + 3 * eb); val = make_double2(rew, dmin); }
        if (h == 3) { dst = reinterpret_cast<double2 *>(ro.state) + 2 * eb; val = rs_lo; }
        if (h == 4) { dst = reinterpret_cast<double2 *>(ro.state) + 2 * eb + 1; val = rs_hi; }
        if (active && (h < 3 || (h < 5 && has_state))) { if (MCN_PAIR_NT_ALL && NTMP) pair_st<true>(dst, val); else *dst = val; }
        double *d8 = p.st.gtime + eb;                                                     // h == 0
        double v8 = o_time;
        if (h == 1) { d8 = reinterpret_cast<double *>(p.out.rec) + 3 * eb + 2; v8 = ll2d((long long)tail); }
        if (active && h < 2) { if (MCN_PAIR_NT_ALL && NTMP) __builtin_nontemporal_store(v8, d8); else *d8 = v8; }
    }
}

template <int NT>
static void launch_pair_one(const StepParams &p, hipStream_t stream)
{
    constexpr int G = 64 / NT;
    const long per_block = 4 * G;
    const int blocks = (int)((p.E + per_block - 1) / per_block);
    // non-temporal per-human streams once one step's footprint (~(15 + 11 N) x 8 + 70 B per env) is well past the
    // 256 MB memory-side cache (measured cross-over, 5 humans: 2^18 envs = 165 MB 27.9 vs 32 us without / with,
    // 2^19 = 330 MB equal, 2^20 = 660 MB 126-142 vs 108-109); mcn_tuning.pair_stream 2 / 3 force it on / off
    const double footprint = (double)p.E * ((15 + 11 * NT) * 8 + 70);
    const bool nt = p.pair_stream == 2 || (p.pair_stream != 3 && footprint > 400e6);
    if (nt) hipLaunchKernelGGL((env_pair_kernel<NT, true>), dim3(blocks), dim3(256), 0, stream, p);
    else    hipLaunchKernelGGL((env_pair_kernel<NT, false>), dim3(blocks), dim3(256), 0, stream, p);
}

// Returns true when the streaming kernel handles this problem.
bool launch_env_pair(const StepParams &p, hipStream_t stream)
{
    if (p.cfg.human_policy != MCN_HUMANS_GIVEN || !p.update || p.cfg.count_hh || p.cfg.robot_kinematics != MCN_KIN_HOLONOMIC)
        return false;
    if ((p.cfg.track_human_times && p.st.human_times) || p.out.human_act) return false;
    if (p.has_roll && p.roll.state && p.roll.disc_len > kPairDiscMax) return false;
    if (p.force_generic) return false;
    switch (p.N) {
        case 5:  launch_pair_one<5>(p, stream); break;
        case 10: launch_pair_one<10>(p, stream); break;
        default: return false;
    }
    note_dispatch("env_pair_kernel");
    return true;
}

}  // namespace mcn
