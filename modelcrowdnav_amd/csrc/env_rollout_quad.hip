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
void note_dispatch(const char *family);          // mcn_api.hip: mcn_last_dispatch()

// Diagnostic build only (make -C modelcrowdnav_amd/csrc stamp -> build_stamp/libmcn_hip.so, tools/fixed_cost.py):
// lane 0 of every wavefront writes the 100 MHz real-time counter at kernel entry (slot 0), after the state load
// (slot 1), at the end of each of its first 36 steps (slots 2..37) and at exit (slot 39) to a buffer nothing else
// reads.  The product build compiles none of it.
#ifdef MCN_DIAG
#define MCN_STAMP_SLOTS 40
#define MCN_STAMP_WAVES 8192
__device__ unsigned long long g_stamps[MCN_STAMP_WAVES * MCN_STAMP_SLOTS];
#define STAMP(slot)                                                                                              \
    do {                                                                                                         \
        const int w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                                      \
        if ((threadIdx.x & 63) == 0 && (slot) < MCN_STAMP_SLOTS && w_ < MCN_STAMP_WAVES)                         \
            g_stamps[w_ * MCN_STAMP_SLOTS + (slot)] = __builtin_amdgcn_s_memrealtime();                          \
    } while (0)
// slot 38: where the wavefront ran -- HW_ID (wave / simd / cu / sh / se fields) in the low word, XCC_ID in the high
#define STAMP_WHERE()                                                                                            \
    do {                                                                                                         \
        const int w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                                      \
        if ((threadIdx.x & 63) == 0 && w_ < MCN_STAMP_WAVES)                                                     \
            g_stamps[w_ * MCN_STAMP_SLOTS + 38] =                                                                \
                (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |                     \
                ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);             \
    } while (0)
int read_counts(void *dst, size_t bytes, int reset)
{
    if (bytes > sizeof(g_diag_counts)) bytes = sizeof(g_diag_counts);
    const int rc = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_diag_counts), bytes) == hipSuccess ? (int)(bytes / 4) : -1;
    if (reset) {
        void *sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_diag_counts)) == hipSuccess) (void)hipMemset(sym, 0, sizeof(g_diag_counts));
    }
    return rc;
}
int read_stamps(void *dst, size_t bytes)
{
    if (bytes > sizeof(g_stamps)) bytes = sizeof(g_stamps);
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes) == hipSuccess ? (int)(bytes / 8) : -1;
}
#else
#define STAMP(slot)
#define STAMP_WHERE()
#endif

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

// SPLIT = true: two wavefronts per workgroup work on the same G envs, as in env_step_quad.hip, but for the whole
// T-step sequence.  Wavefront 0 owns the humans (ORCA in float32, integration, scenario-pool restarts); wavefront 1
// owns the robot, the float64 swept-circle / overlap tests, the reward ladder and the Explorer record.  Per step
// they meet twice through LDS: done flag + restart case (+ robot pose when the robot is visible) one way, the new
// human positions / velocities / radii the other way.  A lone wavefront issues one instruction every ~5 cycles, so
// a step costs what its instruction count costs; splitting the step lets the two halves issue side by side and a
// step then takes max(ORCA, rest) instead of their sum.
// Pins a wave-uniform value in vector registers.  The kernel argument block is ~110 dwords; left to itself the
// compiler keeps all of it (plus every lane mask) in the ~100 scalar registers across the step loop and spills the
// excess to VGPR lanes, paying v_writelane / v_readlane / s_nop on the critical path of every step.  VGPRs are
// plentiful here (one or two wavefronts per SIMD), so loop-invariant operands live there instead.
template <class V>
__device__ __forceinline__ V in_vgpr(V v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// Operands of the RARE paths (an episode ends: finished-episode record, restart from the pool) are not kept in
// registers at all: the rare block re-reads them from the kernel-argument segment (scalar loads, constant cache).
// The empty asm makes the pointer opaque at that point, so the loads cannot be hoisted out of the step loop and
// turned back into ~30 long-lived registers (which is what spilled to scratch before).
typedef const __attribute__((address_space(4))) StepParams *KernargPtr;
__device__ __forceinline__ KernargPtr kernarg_here()
{
    KernargPtr q = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();                      // StepParams is argument 0
    asm volatile("" : "+s"(q));
    return q;
}

template <int NT, int VIS, bool UNI, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 128 : 64, SPLIT ? 3 : 1) void env_rollout_quad_kernel(const StepParams p, const int T)
{
    __shared__ double2 s_hpos[16], s_hvel[16], s_rpos[16], s_rvel[16];
    __shared__ double s_hrad[16];
    __shared__ int s_dn[16], s_case[16];
    STAMP(0);
    STAMP_WHERE();
#ifdef MCN_DIAG
    if (p.debug_noop) return;      // diagnostic build only: launch-floor measurement
#endif
    const int role = SPLIT ? (int)(threadIdx.x >> 6) : -1;      // 0: humans, 1: robot + ladder + records, -1: both
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
    const int gi = g < G ? g : 0;             // LDS slots of this lane's env / human (idle lanes alias slot 0, read-only)
    const int hi = g < G ? g * NT + h : 0;
    const mcn_env_cfg &c = p.cfg;
    const mcn_rollout &ro = p.roll;
    const double dt = in_vgpr(c.time_step);
    // The reward ladder's six constants are NOT kept in registers across the step loop (round 4): pinned in VGPRs four
    // of them were spilled to scratch at the 168-VGPR cap (9 registers, 40 B per lane: 6 MB of scratch stores per launch
    // and a scratch reload on every step's ladder); the ladder re-reads them from the kernel-argument segment instead
    // (two scalar loads from the constant cache, issued by the float64 wavefront, which has slack every step).
    const bool lead = active && r == 0 && do_pair;       // owns the per-env records
    const bool hlead = active && k == 0 && do_orca;      // owns the human's records
    constexpr bool unicycle = UNI;            // compile-time: the holonomic kernel carries no float64 sin / cos code
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
    // the Explorer record is held (redundantly, like the robot) by every lane of the env: no broadcast per step
    mcn_roll_rec rs = {0, 0, 0, 0, 0, 0};
    if (has_state) rs = ro.state[eb];

    const bool cand_h = k < NT - 1;                          // candidate is another human
    const bool cand_r = VIS && (k == NT - 1);                // candidate is the robot
    const int j = cand_h ? k + (k >= h ? 1 : 0) : h;
    const int l0 = lane - r;
    const int src = ((l0 + 4 * j) & 63) << 2;                // a lane of the quad that owns human j

    // per-lane cursors / wave-uniform operands of the rare paths, all in VGPRs
    const double2 *act_ptr = reinterpret_cast<const double2 *>(p.actions) + eb;
    const long act_stride = in_vgpr((long)p.E);
    const double *disc_table = in_vgpr(ro.disc_table);
    const int disc_last = in_vgpr(ro.disc_len - 1);
    const bool has_theta = p.st.rtheta != nullptr;
    double2 act_next = *act_ptr;
    double o_rew = 0, o_dmin = 0;             // the step record of the latest step (stored once, after the loop)
    int o_dn = 0, o_inf = 0, o_hh = 0;
    double hax = 0, hay = 0;
    // 1 / timeHorizon and 1 / timeStep once per launch, opaque (quad_common.hpp: quad_orca_velocity)
    const float inv_th = in_vgpr(1.0f / c.orca_time_horizon), inv_ts = in_vgpr(1.0f / (float)c.time_step);
    STAMP(1);

    for (int t = 0; t < T; ++t) {
        const double2 act = act_next;
        act_ptr += act_stride;
        if (do_pair && t + 1 < T) act_next = *act_ptr;
        double ep_disc = 0;
        if (do_pair && has_state) ep_disc = disc_table[rs.ep_steps < disc_last ? rs.ep_steps : disc_last];

        // ---- candidate neighbour: from the quad that owns it instead of from memory ----
        double2 cpos;
        cpos.x = bperm_d(src, pos.x);
        cpos.y = bperm_d(src, pos.y);
        double crd = bperm_d(src, rad);
        if (cand_r) { cpos = rpos; crd = rrad; }

        int dn = 0, case_g = 0;
        double t_new = 0;

        // ---- K1: ORCA ----
        if (do_orca) {
            float cvx = bperm_f(src, (float)vel.x), cvy = bperm_f(src, (float)vel.y);
            if (cand_r) { cvx = (float)rvel.x; cvy = (float)rvel.y; }
            float rx, ry;
            quad_orca_velocity(c, lane, k, cand_h || cand_r, pos, vel, goal, rad, vpref,
                               make_float4((float)cpos.x, (float)cpos.y, cvx, cvy), crd, inv_th, inv_ts, rx, ry);
            hax = (double)rx; hay = (double)ry;
        }

        if (do_pair) {
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
            int hh = 0;
            {
                // the exact test needs a float64 sqrt; it is skipped for the whole wavefront unless some counted
                // pair is within 1e-6 of touching (a conservative screen: further apart, the exact test is false)
                const double dx = pos.x - cpos.x, dy = pos.y - cpos.y;
                const bool counted = active & (c.count_hh != 0) & cand_h & (j > h);     // idle tail lanes alias env 0's humans
                const double s2 = dx * dx + dy * dy, reach = rad + crd + 1e-6;
                if (__any(counted & (s2 < reach * reach))) DIAG_COUNT(2);
                if (__any(counted & (s2 < reach * reach)))
                    hh = (counted & ((sqrt(s2) - rad - crd) < 0)) ? 1 : 0;
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
            // same screen for the goal test: the exact norm (one fma + float64 sqrt) only when some robot of the
            // wavefront ends within 1e-6 of its goal disc
            bool reaching = false;
            {
                const double gx = endx - rgoal.x, gy = endy - rgoal.y, near = rrad + 1e-6;
                if (__any(gx * gx + gy * gy < near * near)) DIAG_COUNT(3);
                if (__any(gx * gx + gy * gy < near * near)) reaching = norm2(gx, gy) < rrad;
            }
            const KernargPtr kc = kernarg_here();
            const double k_time_limit = kc->cfg.time_limit, k_timeout_at = k_time_limit - 1;
            const double k_collision = kc->cfg.collision_penalty, k_success = kc->cfg.success_reward;
            const double k_discomfort = kc->cfg.discomfort_dist, k_factor = kc->cfg.discomfort_penalty_factor;
            double rew; int inf;
            if (gtime >= k_timeout_at)          { rew = 0; dn = 1; inf = MCN_INFO_TIMEOUT; }
            else if (dmin < 0)                  { rew = k_collision; dn = 1; inf = MCN_INFO_COLLISION; }
            else if (reaching)                  { rew = k_success; dn = 1; inf = MCN_INFO_REACHGOAL; }
            else if (dmin < k_discomfort)       { rew = (dmin - k_discomfort) * k_factor * dt; dn = 0; inf = MCN_INFO_DANGER; }
            else                                { rew = 0; dn = 0; inf = MCN_INFO_NOTHING; }
            o_rew = rew; o_dmin = dmin; o_dn = dn; o_inf = inf; o_hh = hh_sum;
            t_new = gtime + dt;

            // ---- Explorer accounting (explorer.py:88-99,124): every lane of the env updates its copy of the
            //      record with selects; only the finished-episode stores are the lead lane's ----
            case_g = rs.next_case;
            if (has_state) {
                bool danger = inf == MCN_INFO_DANGER;
                if (danger) {                       // (uncommon step: the limit is re-read from the kernel arguments)
                    const KernargPtr kp = kernarg_here();
                    const int lim = kp->roll.danger_episodes, sf = kp->roll.danger_short_from;
                    danger = lim <= 0 || rs.fin_count < lim - ((sf > 0 && e >= sf - 1) ? 1 : 0);
                }
                rs.danger_count += danger ? 1 : 0;
                rs.danger_dist_sum = danger ? rs.danger_dist_sum + dmin : rs.danger_dist_sum;
                const double ret = rs.ep_return + ep_disc * rew;
                if (lead && dn) {
                    const int kf = rs.fin_count;
                    const KernargPtr kp = kernarg_here();
                    const int fin_slots = kp->roll.fin_slots;
                    double *fin_return = kp->roll.fin_return, *fin_time = kp->roll.fin_time;
                    uint8_t *fin_info = kp->roll.fin_info;
                    const bool keep = (fin_slots == 1) || (kf < fin_slots);
                    const long rec = (long)(fin_slots == 1 ? 0 : kf) * kp->E + e;
                    if (keep && fin_return) fin_return[rec] = ret;
                    if (keep && fin_time)   fin_time[rec] = (inf == MCN_INFO_TIMEOUT) ? k_time_limit : t_new;
                    if (keep && fin_info)   fin_info[rec] = (uint8_t)inf;
                }
                rs.fin_count += dn;
                rs.ep_return = dn ? 0.0 : ret;
                rs.ep_steps = dn ? 0 : rs.ep_steps + 1;
                if (do_reset) {
                    const int k_stride = kc->roll.case_stride, k_pool = kc->roll.pool_size;   // (kernel-argument reads, as the ladder's)
                    int nc = rs.next_case + k_stride;            // both < pool_size (validated on the host)
                    nc = nc >= k_pool ? nc - k_pool : nc;
                    rs.next_case = dn ? nc : rs.next_case;
                }
            }

            // ---- robot: integrate, or back to the start pose ----
            if (do_reset && dn) {
                const KernargPtr kp = kernarg_here();
                const double k_theta0 = kp->roll.robot_theta0;
                rpos = make_double2(kp->roll.robot_start[0], kp->roll.robot_start[1]);
                rgoal = make_double2(kp->roll.robot_goal[0], kp->roll.robot_goal[1]);
                rvel = make_double2(0, 0);
                if (has_theta) rtheta = k_theta0;
                gtime = 0;
            } else {
                rpos = make_double2(endx, endy);
                rvel = make_double2(nrvx, nrvy);
                if (unicycle) rtheta = new_theta;
                gtime = t_new;
            }
        }

        if (SPLIT) {
            // hand-off 1: wavefront 1 publishes the done flag, the restart case, the clock (and the robot's new pose)
            if (lead) {
                s_dn[gi] = dn; s_case[gi] = case_g;
                if (VIS) { s_rpos[gi] = rpos; s_rvel[gi] = rvel; }
            }
            __syncthreads();
            if (role == 0) {
                dn = s_dn[gi]; case_g = s_case[gi];
                if (VIS) { rpos = s_rpos[gi]; rvel = s_rvel[gi]; }
            }
        }

        // ---- humans: integrate, or restart from the scenario pool ----
        if (do_orca) {
            if (do_reset && dn) {
                DIAG_COUNT(1);
                if (active) {
                    const long pa = (long)case_g * NT + h;
                    const KernargPtr kp = kernarg_here();
                    const double2 *pool_hvel = reinterpret_cast<const double2 *>(kp->roll.pool_hvel);
                    pos = reinterpret_cast<const double2 *>(kp->roll.pool_hpos)[pa];
                    goal = reinterpret_cast<const double2 *>(kp->roll.pool_hgoal)[pa];
                    rad = kp->roll.pool_hrad[pa];
                    vpref = kp->roll.pool_hvpref[pa];
                    vel = pool_hvel ? pool_hvel[pa] : make_double2(0, 0);
                }
                htime = 0;
            } else {
                pos = make_double2(pos.x + hax * dt, pos.y + hay * dt);
                vel = make_double2(hax, hay);
                // agent.py:137-138; the clock of the step just taken (wavefront 0 keeps its own copy of it)
                if (track && htime == 0 && norm2(pos.x - goal.x, pos.y - goal.y) < rad) htime = SPLIT ? gtime + dt : t_new;
            }
        }
        if (SPLIT) {
            if (role == 0) gtime = (do_reset && dn) ? 0.0 : gtime + dt;
            // hand-off 2: wavefront 0 publishes the humans' new state
            if (hlead) { s_hpos[hi] = pos; s_hvel[hi] = vel; s_hrad[hi] = rad; }
            __syncthreads();
            if (role == 1) { pos = s_hpos[hi]; vel = s_hvel[hi]; rad = s_hrad[hi]; }
        }
        if (t < 37) STAMP(2 + t);
    }

    // ---- registers -> state (once per launch) ----
    // Addresses and pointers are re-derived here from laundered indices and a fresh read of the kernel arguments:
    // shared with the prologue's they would stay live across the step loop and spill to scratch.
    {
        long a2 = a, e2 = e;
        asm volatile("" : "+v"(a2), "+v"(e2));
        const KernargPtr kp = kernarg_here();
        if (hlead) {
            reinterpret_cast<double2 *>(kp->st.hpos)[a2] = pos;
            reinterpret_cast<double2 *>(kp->st.hvel)[a2] = vel;
            if (do_reset) {
                reinterpret_cast<double2 *>(kp->st.hgoal)[a2] = goal;
                kp->st.hrad[a2] = rad;
                kp->st.hvpref[a2] = vpref;
            }
            double *human_times = kp->st.human_times, *human_act = kp->out.human_act;
            if (human_times) human_times[a2] = htime;
            if (human_act) reinterpret_cast<double2 *>(human_act)[a2] = make_double2(hax, hay);
        }
        if (lead) {
            reinterpret_cast<double2 *>(kp->st.rpos)[e2] = rpos;
            reinterpret_cast<double2 *>(kp->st.rvel)[e2] = rvel;
            if (do_reset) reinterpret_cast<double2 *>(kp->st.rgoal)[e2] = rgoal;
            double *rth = kp->st.rtheta;
            if (rth) rth[e2] = rtheta;
            kp->st.gtime[e2] = gtime;
            store_step_rec(kp->out.rec + e2, o_rew, o_dmin, o_dn, o_inf, o_hh);
            if (has_state) kp->roll.state[e2] = rs;
        }
    }
    STAMP(39);
}

template <int NT, int VIS, bool UNI>
static void launch_rollout_kin(const StepParams &p, int T, int blocks, hipStream_t stream)
{
    if (p.quad_split)
        hipLaunchKernelGGL((env_rollout_quad_kernel<NT, VIS, UNI, true>), dim3(blocks), dim3(128), 0, stream, p, T);
    else
        hipLaunchKernelGGL((env_rollout_quad_kernel<NT, VIS, UNI, false>), dim3(blocks), dim3(64), 0, stream, p, T);
}

template <int NT, int VIS>
static void launch_rollout_one(const StepParams &p, int T, hipStream_t stream)
{
    constexpr int G = 64 / (4 * NT);
    const int blocks = (p.E + G - 1) / G;
    if (p.cfg.robot_kinematics == MCN_KIN_UNICYCLE)
        launch_rollout_kin<NT, VIS, true>(p, T, blocks, stream);
    else
        launch_rollout_kin<NT, VIS, false>(p, T, blocks, stream);
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
    note_dispatch("env_rollout_quad_kernel");
    return true;
}

}  // namespace mcn
