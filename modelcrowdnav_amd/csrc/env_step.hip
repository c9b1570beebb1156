// env_step.hip -- fused batched CrowdSim.step for gfx950 (MI355X).
//
// One launch does, for every environment of the batch:
//   K1  human velocity choice: ORCA solve per human        (crowd_sim.py:336-342, orca.py:82-132)
//   K2  robot-human swept-circle test + min distance,       (crowd_sim.py:345-365, utils.py:4-26)
//       human-human overlap count                           (crowd_sim.py:368-376)
//   K3  goal test, reward ladder, integration, look-ahead   (crowd_sim.py:379-432, agent.py:63-74,110-138)
//       + optional Explorer bookkeeping / auto-reset         (explorer.py:54-125)
//
// Mapping (wave64): lane = (env-in-wave g, human h), G = 64 / N environments per wavefront,
// so one global_load_dwordx4 per lane per field reads a contiguous run of the [E*N][2] arrays.
// Neighbour state is staged once in LDS (float4 pos/vel for the float32 ORCA solve, double2
// for the float64 overlap test); the per-env min-distance / overlap-count reductions are
// wavefront shuffles over the N lanes of a group; lane h == 0 of each group owns the robot
// and the per-env scalars.  Groups never straddle wavefronts, so no cross-wave traffic
// exists and workgroup size is only a dispatch-granularity knob (64 for small batches to
// spread over all 256 CUs, 256 for large ones).
//
// Algorithmic HBM bytes per env-step: (17 + 12 N) * 8 + 6  -> 622 B at N = 5.
// All float64 arithmetic is in the reference's operation order with contraction off;
// norm2() uses one explicit fma because numpy.linalg.norm's 2-element dot does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcn.h"
#include "orca_device.hpp"
#include "orca_static.hpp"
#include "orca_coop.hpp"
#include "env_step_params.hpp"
#include "env_common.hpp"
#include "lp3_queue.hpp"

namespace mcn {
void note_dispatch(const char *family);          // mcn_api.hip: mcn_last_dispatch()

// crowds up to this size pre-filter the human-human overlaps inside the ORCA candidate loop (squared distances, no
// sqrt); larger ones keep the separate loop: the extra live values push their unrolled solver past 168 VGPRs
constexpr int kHhPreMaxN = 6;

struct GroupCand {
    const float4 *sAg; const float *sRad;   // block-level staged humans
    float4 rob; float rob_rad;              // robot as seen by humans (if visible)
    int gbase, h, nh;                       // tid of human 0 of my env, my index, humans-as-candidates count (N-1)
    __device__ __forceinline__ void fetch(int c, float4 &pv, float &rad) const {
        if (c < nh) { const int j = c + (c >= h); pv = sAg[gbase + j]; rad = sRad[gbase + j]; }
        else { pv = rob; rad = rob_rad; }
    }
};

// NT > 0: humans per env known at compile time (register-resident ORCA, constant lane->(env,human)
// split); NT == 0: run-time N (LDS-resident lines), any N <= MCN_MAX_HUMANS.  VIS: robot visible to humans.
// MODE: MCN_HUMANS_* fixed at compile time, so the given-velocity / linear variants carry no ORCA registers,
// no neighbour staging and no goal loads (they are pure streaming kernels and want maximum occupancy).
// HH_T: 0 never count overlaps, 1 always, 2 decide at run time from cfg.count_hh.
// One env step for the workgroup's env slots: everything between the state in HBM before the step and after it.  A
// device function so that two kernels can share it: env_step_kernel (one step per launch) and env_step_loop_kernel
// (T steps per launch for the one-wavefront form, below).
template <int BLOCK, int NT, int VIS, int MODE, int HH_T>
__device__ __forceinline__ void env_step_body(const StepParams &p)
{
    constexpr bool kStageHumans = (MODE == MCN_HUMANS_ORCA) || (HH_T != 0);
#ifdef MCN_DIAG
    if (p.debug_noop) return;      // diagnostic build only: launch-floor measurement
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve (all offsets multiples of 16 B)
    float4  *sL      = reinterpret_cast<float4 *>(smem);                       // [nl_cap][BLOCK]
    float4  *sAgF    = sL + (size_t)(NT ? (MODE == MCN_HUMANS_ORCA ? NT / 2 : 0) : p.nl_cap) * BLOCK;   // static-N ORCA: the pairs' (u, dir) rows
    double2 *sPosD   = reinterpret_cast<double2 *>(sAgF + BLOCK);              // [BLOCK]
    double2 *sRobPos = sPosD + BLOCK;                                          // [BLOCK] per env slot
    double2 *sRobAct = sRobPos + BLOCK;                                        // [BLOCK]
    float4  *sRobF   = reinterpret_cast<float4 *>(sRobAct + BLOCK);            // [BLOCK]
    double  *sRadD   = reinterpret_cast<double *>(sRobF + BLOCK);              // [BLOCK]
    double  *sRobRad = sRadD + BLOCK;                                          // [BLOCK]
    float   *sRadF   = reinterpret_cast<float *>(sRobRad + BLOCK);             // [BLOCK]
    float   *sRobRadF = sRadF + BLOCK;                                         // [BLOCK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int N = NT ? NT : p.N;
    const int G = NT ? 64 / (NT ? NT : 1) : p.G;
    const int g = lane / N;
    const int h = lane - g * N;
    const int slot = wave * G + g;                          // env slot inside the block
    // XCD-aware chunking: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), each with a
    // private L2.  Give every XCD one contiguous range of envs so that the cache lines of the narrow per-env
    // outputs (u8 done/info, i32 counters), which straddle neighbouring workgroups, are completed inside ONE
    // L2 instead of being written back partially from several.  Bijective for any grid size.
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned qq = nb >> 3, rr_ = nb & 7u;
    const unsigned chunk = (xcd < rr_ ? xcd * (qq + 1) : rr_ * (qq + 1) + (xcd - rr_) * qq) + idx;
    const long e = (long)chunk * ((BLOCK / 64) * G) + slot;
    const bool active = (g < G) && (e < p.E);
    const bool leader = active && (h == 0);
    const long a = active ? e * N + h : 0;
    const int gbase = tid - h;
    const mcn_env_cfg &c = p.cfg;
    const double dt = c.time_step;

    // ---- coalesced agent-state loads (16 B per lane per field) ----
    double2 pos = make_double2(0, 0), vel = pos, goal = pos, attr = pos;
    if (active) {
        pos  = reinterpret_cast<const double2 *>(p.st.hpos)[a];
        vel  = reinterpret_cast<const double2 *>(p.st.hvel)[a];
        if constexpr (MODE != MCN_HUMANS_GIVEN) goal = reinterpret_cast<const double2 *>(p.st.hgoal)[a];
        attr.x = p.st.hrad[a];
        if constexpr (MODE != MCN_HUMANS_GIVEN) attr.y = p.st.hvpref[a];
    }
    double2 rpos = make_double2(0, 0), rvel = rpos, rgoal = rpos, rattr = rpos, act = rpos;
    double rtheta = 0, gtime = 0;
    int next_case = 0;
    double ep_disc = 0;
    mcn_roll_rec rs = {0, 0, 0, 0, 0, 0};
    if (leader) {
        rpos  = reinterpret_cast<const double2 *>(p.st.rpos)[e];
        rvel  = reinterpret_cast<const double2 *>(p.st.rvel)[e];
        rgoal = reinterpret_cast<const double2 *>(p.st.rgoal)[e];
        rattr.x = p.st.rrad[e];
        act   = reinterpret_cast<const double2 *>(p.actions)[e];
        gtime = p.st.gtime[e];
        if (c.robot_kinematics == MCN_KIN_UNICYCLE) rtheta = p.st.rtheta[e];
        // the rollout record is fetched now, so its latency hides under the ORCA solve instead of forming a
        // dependent load chain (steps -> discount) at the very end of the kernel
        if (p.has_roll && p.roll.state) {
            rs = p.roll.state[e];                       // one 32-byte record
            next_case = rs.next_case;
            ep_disc = p.roll.disc_table[rs.ep_steps < p.roll.disc_len ? rs.ep_steps : p.roll.disc_len - 1];
        }
    }
    // effective robot velocity for the swept test (crowd_sim.py:350-355)
    double2 eff = act;
    if (leader && c.robot_kinematics == MCN_KIN_UNICYCLE) {
        eff.x = act.x * cos(act.y + rtheta);
        eff.y = act.x * sin(act.y + rtheta);
    }

    // ---- stage neighbour tiles in LDS ----
    const float fpx = (float)pos.x, fpy = (float)pos.y, fvx = (float)vel.x, fvy = (float)vel.y;
    const float frad = (float)(attr.x + 0.01 + c.orca_safety_space);
    if constexpr (kStageHumans) {
        sAgF[tid] = make_float4(fpx, fpy, fvx, fvy);
        sRadF[tid] = frad;
        sPosD[tid] = pos;
        sRadD[tid] = attr.x;
    }
    if (leader) {
        sRobPos[slot] = rpos;
        sRobAct[slot] = eff;
        sRobRad[slot] = rattr.x;
        if constexpr (MODE == MCN_HUMANS_ORCA) {
            sRobF[slot] = make_float4((float)rpos.x, (float)rpos.y, (float)rvel.x, (float)rvel.y);
            sRobRadF[slot] = (float)(rattr.x + 0.01 + c.orca_safety_space);
        }
    }
    __syncthreads();

    // ---- K1: human action ----
    double hax = 0, hay = 0;
    int hh_pre = 0;                 // static-N ORCA kernels pre-filter the human-human overlaps while they hold the
    bool border_pre = false;        // squared distances
    // Dense crowds (>= 5 candidate neighbours) in a latency-bound batch (one wavefront per workgroup): the 3-D LP of the
    // few lanes that need it is solved by the whole wavefront (orca_coop.hpp) instead of running its unrolled O(n^3)
    // code for one or two active lanes.  Throughput-bound batches (BLOCK = 256) keep the unrolled form.
    constexpr int NCK = NT > 0 ? NT - 1 + VIS : 0;
    constexpr int NLK = NCK > 0 ? NCK : 1;
    constexpr bool kCoopLp3 = (MODE == MCN_HUMANS_ORCA) && NT > 0 && NCK >= 5 && BLOCK == 64;
    float4 Lk[kCoopLp3 ? NLK : 1];
    int lp_nl = 0, lp_fail = 0;
    float ox = 0, oy = 0;
    bool deferred = false;          // this human's 3-D LP is parked in the queue: env_lp3_kernel integrates it
    int qidx = 0;
    if (active) {
        if constexpr (MODE == MCN_HUMANS_ORCA) {
            if constexpr (NT > 0) {
                constexpr int NC = NT - 1 + VIS;
                // Each pair of humans builds its half-plane ONCE: the two humans' lines are (v + u/2, dir) and
                // (v' - u/2, -dir) for the same (u, dir) (orca_static.hpp: orca_u_dir).  Lane h computes the pairs
                // (h, h + s), s = 1 .. (NT-1)/2 around the env's ring of humans (for even NT the opposite pair goes to
                // the lower index), parks (u, dir) in the LDS rows the run-time-N kernel uses for its lines, and every
                // lane then reads its NT-1 half-planes back with the owner's sign.  Halves the dominant VALU block.
                constexpr int FW = NT / 2;                        // forward pairs a lane may own
                // (opaque, hoistable: the compiler would otherwise fold `apart ? 1 / a : 1 / b` in the half-plane
                //  construction into one IEEE division after the select, on every step's critical path)
                float inv_th = 1.0f / c.orca_time_horizon, inv_ts = 1.0f / (float)dt;
                asm("" : "+v"(inv_th), "+v"(inv_ts));
                if constexpr (NT >= 2) {
#pragma unroll
                    for (int s_ = 1; s_ <= FW; ++s_) {
                        int j = h + s_;
                        j = j >= NT ? j - NT : j;
                        // odd NT: distances 1 .. (NT-1)/2 cover every pair once; even NT: distance NT/2 is reached from
                        // both ends, the lower index owns it
                        const bool own = (2 * s_ < NT) || (h < j);
                        float4 ud = make_float4(0, 0, 0, 0);
                        if (own) ud = orca_u_dir(fpx, fpy, fvx, fvy, frad, sAgF[gbase + j], sRadF[gbase + j], inv_th, inv_ts);
                        sL[(s_ - 1) * BLOCK + tid] = ud;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // env groups never straddle a wavefront
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                float4 Lnat[NC > 0 ? NC : 1];
                float dd[NC > 0 ? NC : 1];
#pragma unroll
                for (int cidx = 0; cidx < NT - 1; ++cidx) {
                    const int j = cidx + (cidx >= h);
                    int sf = j - h;                               // ring distance from me to j
                    sf = sf < 0 ? sf + NT : sf;
                    const bool mine = (2 * sf < NT) || (2 * sf == NT && h < j);
                    const int sb = NT - sf;                       // ring distance from j to me
                    const int owner = mine ? tid : gbase + j;
                    const int slot = (mine ? sf : sb) - 1;
                    const float4 ud = sL[slot * BLOCK + owner];
                    const float sg = mine ? 0.5f : -0.5f;
                    Lnat[cidx] = make_float4(fvx + sg * ud.x, fvy + sg * ud.y, mine ? ud.z : -ud.z, mine ? ud.w : -ud.w);
                    const float4 q = sAgF[gbase + j];
                    const float ddx = fpx - q.x, ddy = fpy - q.y;
                    dd[cidx] = dot2(ddx, ddy, ddx, ddy);
                    if constexpr (HH_T != 0 && NT <= kHhPreMaxN) {
                        // human-human overlap pre-filter on the squared distance ORCA needs anyway (no sqrt): surely
                        // overlapping below (R - 1e-3)^2, surely apart above (R + 1e-3)^2, the exact float64 test decides
                        // in between (K2)
                        const float R = (float)attr.x + (sRadF[gbase + j] - 0.01f - (float)c.orca_safety_space);
                        const float lo = R - 1e-3f, hi = R + 1e-3f;
                        const bool sure = (j > h) & (lo > 0.0f) & (dd[cidx] < lo * lo);
                        hh_pre += sure ? 1 : 0;
                        border_pre = border_pre | ((j > h) & !sure & (dd[cidx] < hi * hi));
                    }
                }
                if constexpr (VIS) {
                    const float4 rq = sRobF[slot];
                    Lnat[NC - 1] = orca_line_select(fpx, fpy, fvx, fvy, frad, rq, sRobRadF[slot], inv_th, inv_ts);
                    const float ddx = fpx - rq.x, ddy = fpy - rq.y;
                    dd[NC - 1] = dot2(ddx, ddy, ddx, ddy);
                }
                int nl_ = 0, fail_ = 0;
                orca_sort_lp2<NC>(Lnat, dd, (float)attr.y, (float)(goal.x - pos.x), (float)(goal.y - pos.y),
                                  c.orca_neighbor_dist, c.orca_max_neighbors, ox, oy, fail_, nl_);
                bool solve_here = true;
                if (p.lp3_defer) {
                    // Deferred 3-D LP (lp3_queue.hpp): the few humans whose 2-D LP failed park their problem in the
                    // queue -- one atomic per wavefront, consecutive slots for its lanes -- and env_lp3_kernel finishes
                    // and integrates them, one per lane.  Nothing of the 3-D LP runs here unless a lane's slot falls
                    // past the bounded sub-queue: those lanes take the in-place code below (ONE call site for both
                    // uses: a second inlined copy cost 28 registers and a wavefront of occupancy).
                    const bool need = fail_ < nl_;
                    const unsigned long long m = __ballot(need);
                    bool turned_away = false;
                    if (m != 0ull) {
                        const Lp3Queue q = lp3_queue_view(p.out.lp3_queue, p.E, NT);
                        const int sub = (int)((blockIdx.x * (BLOCK / 64) + wave) & (kLp3Queues - 1));
                        const int first = __ffsll((long long)m) - 1;
                        int base = 0;
                        if (lane == first) base = atomicAdd(q.count(sub), __popcll(m));
                        base = __shfl(base, first);
                        const int slot_q = base + __popcll(m & ((1ull << lane) - 1ull));
                        if (need && slot_q < (int)q.subcap) {
                            deferred = true;
                            qidx = (int)(sub * q.subcap) + slot_q;
                            q.hdr[qidx] = make_int4((int)a, nl_ | (fail_ << 8), __float_as_int((float)attr.y), 0);
                            q.res[qidx] = make_float2(ox, oy);
#pragma unroll
                            for (int k2 = 0; k2 < NLK; ++k2) q.line[(long)k2 * q.cap + qidx] = Lnat[k2];
                        }
                        turned_away = need && !deferred;
                    }
                    solve_here = __ballot(turned_away) != 0ull;            // wave-uniform
                    if (deferred) fail_ = nl_;                             // parked: nothing left to solve in this lane
                }
                if (solve_here) {
                    if constexpr (kCoopLp3) {
                        lp_fail = fail_; lp_nl = nl_;
#pragma unroll
                        for (int k2 = 0; k2 < NLK; ++k2) Lk[k2] = Lnat[k2];    // the wavefront finishes them together below
                    } else {
                        lp3_static<NLK>(Lnat, nl_, fail_, (float)attr.y, ox, oy);
                    }
                }
            } else {
                GroupCand cand{sAgF, sRadF, make_float4(0, 0, 0, 0), 0.f, gbase, h, N - 1};
                int ncand = N - 1;
                if (c.robot_visible) { cand.rob = sRobF[slot]; cand.rob_rad = sRobRadF[slot]; ++ncand; }
                LdsLines L{sL + tid, BLOCK};
                orca_solve(cand, ncand, fpx, fpy, fvx, fvy, frad, (float)attr.y,
                           (float)(goal.x - pos.x), (float)(goal.y - pos.y),
                           c.orca_neighbor_dist, c.orca_max_neighbors, c.orca_time_horizon, (float)dt, L, ox, oy);
            }
            if constexpr (!kCoopLp3) { hax = (double)ox; hay = (double)oy; }
        } else if constexpr (MODE == MCN_HUMANS_LINEAR) {
            const double th = atan2(goal.y - pos.y, goal.x - pos.x);
            hax = cos(th) * attr.y; hay = sin(th) * attr.y;
        } else {
            const double2 gv = reinterpret_cast<const double2 *>(p.given_v)[a];
            hax = gv.x; hay = gv.y;
        }
    }

    if constexpr (kCoopLp3) {
        // (with a queue: only the lanes a full sub-queue turned away -- wave-uniform, all lanes take part)
        if (!p.lp3_defer || __ballot(lp_fail < lp_nl) != 0ull) {
            CoopLds &coop = reinterpret_cast<CoopLds *>(sRobRadF + BLOCK)[wave];
            lp3_wave_coop<NLK>(coop, Lk, lp_nl, lp_fail, (float)attr.y, ox, oy);      // every lane of the wavefront
        }
        hax = (double)ox; hay = (double)oy;
    }

    // ---- K2: swept-circle distance to the robot, overlap with later humans ----
    double cd = INFINITY;
    int hh = 0;
    if (active) {
        const double2 R = sRobPos[slot], A = sRobAct[slot];
        const double rr = sRobRad[slot];
        const double px = pos.x - R.x, py = pos.y - R.y;
        const double vx = vel.x - A.x, vy = vel.y - A.y;
        const double ex = px + vx * dt, ey = py + vy * dt;
        cd = p2s_origin(px, py, ex, ey) - attr.x - rr;
    }
    bool do_hh = false;
    if constexpr (HH_T == 1) do_hh = true;
    if constexpr (HH_T == 2) do_hh = c.count_hh != 0;
    if (do_hh) {
        // Overlap test sqrt(dx^2+dy^2) - ri - rj < 0 (crowd_sim.py:371-374).  A float32 pre-filter on the
        // staged tile decides every pair that is not within 1e-3 of touching; only if some lane of the
        // wavefront holds a borderline pair does the wave take the exact float64 path.
        bool borderline = false;
        if constexpr (MODE == MCN_HUMANS_ORCA && NT > 0 && NT <= kHhPreMaxN) {
            hh = hh_pre; borderline = border_pre;
        } else if (active) {
            for (int j = h + 1; j < N; ++j) {
                const float4 q = sAgF[gbase + j];
                const float dxf = fpx - q.x, dyf = fpy - q.y;
                const float gap = sqrtf(dxf * dxf + dyf * dyf) - ((float)attr.x + (sRadF[gbase + j] - 0.01f - (float)c.orca_safety_space));
                if (gap < -1e-3f) ++hh;
                else if (gap < 1e-3f) borderline = true;
            }
        }
        if (__any(borderline)) {
            hh = 0;
            if (active) {
                for (int j = h + 1; j < N; ++j) {
                    const double2 q = sPosD[gbase + j];
                    const double dx = pos.x - q.x, dy = pos.y - q.y;
                    const double d = sqrt(dx * dx + dy * dy) - attr.x - sRadD[gbase + j];
                    hh += (d < 0);
                }
            }
        }
    }
    // wavefront shuffle reductions over the N lanes of the group
    const int l0 = lane - h;
    double dmin = INFINITY;
    int hh_sum = 0;
    for (int k = 0; k < N; ++k) {
        const int src = (l0 + k) & 63;
        const double o = __shfl(cd, src);
        dmin = fmin(dmin, o);
        hh_sum += __shfl(hh, src);
    }

    // ---- K3: goal test + reward ladder (leader lane) ----
    double endx = 0, endy = 0, new_theta = rtheta, nrvx = 0, nrvy = 0;
    double rew = 0;
    int dn = 0, inf = MCN_INFO_NOTHING;
    if (leader) {
        if (c.robot_kinematics == MCN_KIN_UNICYCLE) {
            const double th = rtheta + act.y;                       // agent.py:115-118
            endx = rpos.x + cos(th) * act.x * dt;
            endy = rpos.y + sin(th) * act.x * dt;
            new_theta = pymod(rtheta + act.y, 2 * M_PI);            // agent.py:130-135
            nrvx = act.x * cos(new_theta); nrvy = act.x * sin(new_theta);
        } else {
            endx = rpos.x + act.x * dt; endy = rpos.y + act.y * dt;
            nrvx = act.x; nrvy = act.y;
        }
        const bool reaching = norm2(endx - rgoal.x, endy - rgoal.y) < rattr.x;
        const bool collision = dmin < 0;
        if (gtime >= c.time_limit - 1)      { rew = 0; dn = 1; inf = MCN_INFO_TIMEOUT; }
        else if (collision)                 { rew = c.collision_penalty; dn = 1; inf = MCN_INFO_COLLISION; }
        else if (reaching)                  { rew = c.success_reward; dn = 1; inf = MCN_INFO_REACHGOAL; }
        else if (dmin < c.discomfort_dist)  { rew = (dmin - c.discomfort_dist) * c.discomfort_penalty_factor * dt; dn = 0; inf = MCN_INFO_DANGER; }
        else                                { rew = 0; dn = 0; inf = MCN_INFO_NOTHING; }
        store_step_rec(p.out.rec + e, rew, dmin, dn, inf, hh_sum);   // 16 + 8 bytes
    }
    if (active && !deferred && p.out.human_act)
        reinterpret_cast<double2 *>(p.out.human_act)[a] = make_double2(hax, hay);

    // ---- integrate / look ahead ----
    // (a human whose 3-D LP is parked has no velocity yet: it only tells env_lp3_kernel what to write for it --
    //  1 integrate, 2 look-ahead observation, 0 nothing but the exported action: its env restarts from the pool)
    const double npx = pos.x + hax * dt, npy = pos.y + hay * dt;
    if (!p.update) {
        if (active) {
            if (deferred) {
                lp3_queue_view(p.out.lp3_queue, p.E, N).flag[qidx] = 2;
            } else {
                reinterpret_cast<double2 *>(p.out.nobs_pos)[a] = make_double2(npx, npy);
                reinterpret_cast<double2 *>(p.out.nobs_vel)[a] = make_double2(hax, hay);
            }
        }
        return;
    }
    const bool do_reset = p.has_roll && p.roll.pool_hpos != nullptr;
    const int dn_g = __shfl(dn, l0 & 63);
    const double t_new = __shfl(gtime, l0 & 63) + dt;
    const int case_g = __shfl(next_case, l0 & 63);
    if (active) {
        if (do_reset && dn_g) {
            const long pa = (long)case_g * N + h;
            reinterpret_cast<double2 *>(p.st.hpos)[a]  = reinterpret_cast<const double2 *>(p.roll.pool_hpos)[pa];
            reinterpret_cast<double2 *>(p.st.hgoal)[a] = reinterpret_cast<const double2 *>(p.roll.pool_hgoal)[pa];
            p.st.hrad[a] = p.roll.pool_hrad[pa];
            p.st.hvpref[a] = p.roll.pool_hvpref[pa];
            reinterpret_cast<double2 *>(p.st.hvel)[a]  = p.roll.pool_hvel
                ? reinterpret_cast<const double2 *>(p.roll.pool_hvel)[pa] : make_double2(0, 0);
            if (p.st.human_times) p.st.human_times[a] = 0;
            if (deferred) lp3_queue_view(p.out.lp3_queue, p.E, N).flag[qidx] = 0;
        } else if (deferred) {
            lp3_queue_view(p.out.lp3_queue, p.E, N).flag[qidx] = 1;
        } else {
            reinterpret_cast<double2 *>(p.st.hpos)[a] = make_double2(npx, npy);
            reinterpret_cast<double2 *>(p.st.hvel)[a] = make_double2(hax, hay);
            if (c.track_human_times && p.st.human_times) {
                // crowd_sim.py:418-421 / agent.py:137-138
                const double2 gl = reinterpret_cast<const double2 *>(p.st.hgoal)[a];
                if (p.st.human_times[a] == 0 && norm2(npx - gl.x, npy - gl.y) < attr.x)
                    p.st.human_times[a] = t_new;
            }
        }
    }
    if (leader) {
        if (p.has_roll) {
            const mcn_rollout &r = p.roll;
            if (r.state) {
                if (inf == MCN_INFO_DANGER && (r.danger_episodes <= 0 ||
            rs.fin_count < r.danger_episodes - ((r.danger_short_from > 0 && e >= r.danger_short_from - 1) ? 1 : 0))) {
                    rs.danger_count += 1; rs.danger_dist_sum += dmin;
                }
                const double ret = rs.ep_return + ep_disc * rew;
                if (dn) {
                    const int k = rs.fin_count;
                    // fin_slots == 1: keep the latest episode; otherwise keep the first fin_slots episodes
                    const bool keep = (r.fin_slots == 1) || (k < r.fin_slots);
                    const long rec = (long)(r.fin_slots == 1 ? 0 : k) * p.E + e;
                    if (keep && r.fin_return) r.fin_return[rec] = ret;
                    if (keep && r.fin_time)   r.fin_time[rec] = (inf == MCN_INFO_TIMEOUT) ? c.time_limit : t_new;
                    if (keep && r.fin_info)   r.fin_info[rec] = (uint8_t)inf;
                    rs.fin_count = k + 1; rs.ep_return = 0; rs.ep_steps = 0;
                    if (do_reset) {                          // both < pool_size (validated on the host): no division
                        const int nc = next_case + r.case_stride;
                        rs.next_case = nc >= r.pool_size ? nc - r.pool_size : nc;
                    }
                } else {
                    rs.ep_return = ret; rs.ep_steps += 1;
                }
                r.state[e] = rs;                         // one 32-byte store
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

template <int BLOCK, int NT, int VIS, int MODE, int HH_T>
__global__ __launch_bounds__(BLOCK)     // (occupancy hints spill: 5 / 6 workgroups per CU on the 5-human ORCA kernel 203 -> 320 / 485 us at 2^20 envs)
void env_step_kernel(const StepParams p)
{
    env_step_body<BLOCK, NT, VIS, MODE, HH_T>(p);
}

// T consecutive steps in ONE launch for the crowds the quad-parallel rollout kernel does not cover (6-10 humans) in
// latency-bound batches: the one-wavefront workgroup simply runs the step T times, taking step t's actions from
// actions[t][E][2].  Every value a step reads back was written by the same lane of the same wavefront one iteration
// earlier (each lane owns its human, the leader lane its robot / clock / rollout record), so program order is all the
// ordering that is needed; the state goes through L2 instead of registers, but the T - 1 launch gaps (4-6 us each
// at 4096 envs, where a step takes ~23 us) and the cold first touches disappear.  Same arithmetic, same stores, same
// order as T launches of env_step_kernel<64, ...>: bit-identical by construction.
template <int NT, int VIS>
__global__ __launch_bounds__(64) void env_step_loop_kernel(StepParams p, const int T)
{
    const double *acts = p.actions;
    for (int t = 0; t < T; ++t) {
        p.actions = acts + (size_t)t * (size_t)p.E * 2;
        env_step_body<64, NT, VIS, MCN_HUMANS_ORCA, 2>(p);
        __syncthreads();                       // the next step restages the LDS tiles this one still reads
    }
}

// The deferred 3-D LPs of one step (lp3_queue.hpp).  Finishes each parked solve exactly as the step kernel would have
// (same sorted half-planes, same running result) and then does for that human what the step kernel skipped: the
// exported action, and -- by the flag the step kernel left -- the integration (crowd_sim.py:416-421) or the look-ahead
// observation (agent.py:63-74).  Workgroup b serves sub-queue b % 256.  Two forms (A/B switch MCN_LP3_KB_COOP):
//   0  one problem per lane, the register-resident unrolled solver (lp3_static): every lane busy, but a wavefront of
//      64 different problems walks the union of all their paths through the O(n^3) code;
//   1  eight problems per wavefront, eight lanes each (lp3_wave_coop, orca_coop.hpp): a ~4x shorter dependent chain,
//      which is what bounds this launch -- it holds a few % of the step's humans and runs far below the chip's width.
// The last workgroup to finish empties the queue for the next step.
#ifndef MCN_LP3_KB_COOP
#define MCN_LP3_KB_COOP 1
#endif
template <int NL>
__global__ __launch_bounds__(256) void env_lp3_kernel(const StepParams p, const int per_queue)
{
    const Lp3Queue q = lp3_queue_view(p.out.lp3_queue, p.E, p.N);
    const int sub = blockIdx.x & (kLp3Queues - 1), rep = blockIdx.x / kLp3Queues;
    const int taken = __atomic_load_n(q.count(sub), __ATOMIC_RELAXED);
    const int count = taken < (int)q.subcap ? taken : (int)q.subcap;      // slots past subcap were solved in place
    const long qbase = (long)sub * q.subcap;
    const double dt = p.cfg.time_step;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if MCN_LP3_KB_COOP
    __shared__ CoopLds s_coop[4];
    constexpr int kPerWave = kCoopSlots;                          // problems a wavefront takes per pass
    const bool holder = lane < kPerWave;
    for (int i0 = (rep * 4 + wave) * kPerWave; i0 < count; i0 += per_queue * 4 * kPerWave) {
        const int i = i0 + lane;
        const bool mine = holder && i < count;
        const long s = qbase + (mine ? i : 0);
#else
    constexpr bool holder = true;
    for (int i0 = rep * 256; i0 < count; i0 += per_queue * 256) {
        const int i = i0 + threadIdx.x;
        const bool mine = i < count;
        const long s = qbase + (mine ? i : 0);
#endif
        int4 hd = make_int4(0, 0, 0, 0);
        float4 L[NL];
        float2 r0 = make_float2(0, 0);
        if (mine) {
            hd = q.hdr[s];
#pragma unroll
            for (int k = 0; k < NL; ++k) L[k] = q.line[(long)k * q.cap + s];
            r0 = q.res[s];
        } else {
#pragma unroll
            for (int k = 0; k < NL; ++k) L[k] = make_float4(0, 0, 1, 0);
        }
        const long a = hd.x;
        const int nl = hd.y & 255, fail = mine ? ((hd.y >> 8) & 255) : 0;
        const float ms = __int_as_float(hd.z);
        float rx = r0.x, ry = r0.y;
#if MCN_LP3_KB_COOP
        lp3_wave_coop<NL>(s_coop[wave], L, mine ? nl : 0, fail, ms, rx, ry);       // every lane of the wavefront
#else
        if (mine) lp3_static<NL>(L, nl, fail, ms, rx, ry);
#endif
        if (mine) {
            const double hax = (double)rx, hay = (double)ry;
            if (p.out.human_act) reinterpret_cast<double2 *>(p.out.human_act)[a] = make_double2(hax, hay);
            const int flag = q.flag[s];
            if (flag != 0) {
                const double2 pos = reinterpret_cast<const double2 *>(p.st.hpos)[a];
                const double npx = pos.x + hax * dt, npy = pos.y + hay * dt;
                if (flag == 2) {
                    reinterpret_cast<double2 *>(p.out.nobs_pos)[a] = make_double2(npx, npy);
                    reinterpret_cast<double2 *>(p.out.nobs_vel)[a] = make_double2(hax, hay);
                } else {
                    reinterpret_cast<double2 *>(p.st.hpos)[a] = make_double2(npx, npy);
                    reinterpret_cast<double2 *>(p.st.hvel)[a] = make_double2(hax, hay);
                    if (p.cfg.track_human_times && p.st.human_times) {
                        // crowd_sim.py:418-421; the step kernel has already advanced the env's clock to t + dt
                        const double2 gl = reinterpret_cast<const double2 *>(p.st.hgoal)[a];
                        if (p.st.human_times[a] == 0 && norm2(npx - gl.x, npy - gl.y) < p.st.hrad[a])
                            p.st.human_times[a] = p.st.gtime[a / p.N];
                    }
                }
            }
        }
    }
    (void)holder;
    // every workgroup has read its sub-queue's count before it arrives here, so the last one may clear them all
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(q.done(), 1) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (s_last) {
        static_assert(kLp3Queues == 256, "one thread per sub-queue counter");
        __atomic_store_n(q.count(threadIdx.x), 0, __ATOMIC_RELAXED);
        if (threadIdx.x == 0) __atomic_store_n(q.done(), 0, __ATOMIC_RELAXED);
    }
}

static void launch_env_lp3(const StepParams &p, hipStream_t stream)
{
    const int nc = p.N - 1 + (p.cfg.robot_visible ? 1 : 0);
    // workgroups per sub-queue: enough that the expected few % of a sub-queue's slots are taken in one or two passes
    const long subcap = lp3_subcap(p.E, p.N);
    long per_queue = subcap / 512;
    per_queue = per_queue < 1 ? 1 : (per_queue > 8 ? 8 : per_queue);
    const int blocks = (int)(kLp3Queues * per_queue);
#define MCN_LP3_CASE(NL_) case NL_: hipLaunchKernelGGL((env_lp3_kernel<NL_>), dim3(blocks), dim3(256), 0, stream, p, (int)per_queue); break;
    switch (nc) {
        MCN_LP3_CASE(1) MCN_LP3_CASE(2) MCN_LP3_CASE(3) MCN_LP3_CASE(4) MCN_LP3_CASE(5) MCN_LP3_CASE(6) MCN_LP3_CASE(7)
        MCN_LP3_CASE(8) MCN_LP3_CASE(9) MCN_LP3_CASE(10)
        default: break;
    }
#undef MCN_LP3_CASE
}

static size_t step_smem_bytes(int block, int nl_cap, bool coop = false)
{
    if (coop)   // + one CoopLds per wavefront (16-byte aligned: every array before it is a multiple of 16 B per lane x 64)
        return step_smem_bytes(block, nl_cap) + (size_t)(block / 64) * sizeof(CoopLds);
    // float4 lines + float4 sAgF + double2 sPosD + double2 sRobPos + double2 sRobAct + float4 sRobF
    // + double sRadD + double sRobRad + float sRadF + float sRobRadF
    return (size_t)block * (16u * nl_cap + 16 + 16 + 16 + 16 + 16 + 8 + 8 + 4 + 4);
}

template <int BLOCK, int NT, int VIS, int MODE, int HH_T>
static void launch_one(const StepParams &p, int blocks, hipStream_t stream)
{
    const size_t sm = step_smem_bytes(BLOCK, MODE != MCN_HUMANS_ORCA ? 0 : (NT ? NT / 2 : p.nl_cap),
                                      MODE == MCN_HUMANS_ORCA && NT > 0 && NT - 1 + VIS >= 5 && BLOCK == 64);
    note_dispatch("env_step_kernel");
    hipLaunchKernelGGL((env_step_kernel<BLOCK, NT, VIS, MODE, HH_T>), dim3(blocks), dim3(BLOCK), sm, stream, p);
}

template <int BLOCK>
static void dispatch(const StepParams &p, int blocks, hipStream_t stream)
{
    const int vis = p.cfg.robot_visible ? 1 : 0;
    if (p.cfg.human_policy == MCN_HUMANS_ORCA) {
        // register-resident ORCA specialisations for the crowd sizes the reference trains and tests on
        // (human_num 5 / 10 in the configs, 5,7,9 in test_mul_env.py:31-33, 1..5 in the 'mixed' rule)
#define MCN_CASE(NT_) case NT_: if (vis) launch_one<BLOCK, NT_, 1, MCN_HUMANS_ORCA, 2>(p, blocks, stream); \
                               else     launch_one<BLOCK, NT_, 0, MCN_HUMANS_ORCA, 2>(p, blocks, stream); return;
        switch (p.force_generic ? 0 : p.N) {
            MCN_CASE(1) MCN_CASE(2) MCN_CASE(3) MCN_CASE(4) MCN_CASE(5) MCN_CASE(6) MCN_CASE(7) MCN_CASE(8) MCN_CASE(9) MCN_CASE(10)
            default: break;
        }
#undef MCN_CASE
        launch_one<BLOCK, 0, 0, MCN_HUMANS_ORCA, 2>(p, blocks, stream);     // run-time N, LDS-resident lines
    } else if (p.cfg.human_policy == MCN_HUMANS_GIVEN) {
        if (p.cfg.count_hh) launch_one<BLOCK, 0, 0, MCN_HUMANS_GIVEN, 1>(p, blocks, stream);
        else                launch_one<BLOCK, 0, 0, MCN_HUMANS_GIVEN, 0>(p, blocks, stream);
    } else {
        if (p.cfg.count_hh) launch_one<BLOCK, 0, 0, MCN_HUMANS_LINEAR, 1>(p, blocks, stream);
        else                launch_one<BLOCK, 0, 0, MCN_HUMANS_LINEAR, 0>(p, blocks, stream);
    }
}

// Small batches run one wavefront per workgroup so that the grid covers as many CUs as possible (crowds of 9-10: the
// one-wavefront workgroups with the cooperative 3-D LP stay ahead longer -- 10 humans, 32 768 envs 63.6 vs 67.1 us,
// 65 536 envs 105 vs 110, 2^18 envs 391 vs 307; 7 humans cross at ~24 k envs).  mcn_tuning.step_block overrides.
static bool one_wave_batch(const StepParams &p, int waves_total, int nc)
{
    return p.step_block > 0 ? p.step_block == 64 : waves_total <= (nc >= 8 ? 12288 : 4096);
}

// mcn_env_rollout for 6-10 ORCA humans (and 5 with a visible robot) in a latency-bound batch: one env_step_loop_kernel launch instead of T step
// launches.  false = not applicable (the caller falls back to T launches).
bool launch_env_step_loop(const StepParams &p, int T, hipStream_t stream)
{
    // (5 humans + a visible robot = 5 neighbours: one more than the quad-parallel rollout kernel takes)
    const bool five_vis = p.N == 5 && p.cfg.robot_visible;
    if (p.cfg.human_policy != MCN_HUMANS_ORCA || p.force_generic || !p.update || ((p.N < 6 || p.N > 10) && !five_vis)) return false;
    if (p.lp3_defer > 0) return false;                                 // forced deferral: two kernels per step
    const int G = 64 / p.N;
    const int waves_total = (p.E + G - 1) / G;
    const int nc = p.N - 1 + (p.cfg.robot_visible ? 1 : 0);
    if (!one_wave_batch(p, waves_total, nc)) return false;             // launch_env_step's own rule
    // The looped kernel holds 234-282 VGPRs (one wavefront per SIMD) against the step kernel's 160 (three): it wins while
    // the batch leaves SIMDs idle anyway and loses once wavefronts queue up -- 10 humans: 4096 envs 28.9 -> 14.9 us per
    // step, 16 384 envs (2731 wavefronts) 43.0 -> 41.3, 32 768 envs (5462) 64.2 -> 80.7.
    if (p.step_block != 64 && waves_total > 3072) return false;
    StepParams q = p;
    q.lp3_defer = 0;
    q.G = G;
    const size_t sm = step_smem_bytes(64, p.N / 2, nc >= 5);
    note_dispatch("env_step_loop_kernel");
#define MCN_LOOP_CASE(NT_) case NT_: \
        if (p.cfg.robot_visible) hipLaunchKernelGGL((env_step_loop_kernel<NT_, 1>), dim3(waves_total), dim3(64), sm, stream, q, T); \
        else                     hipLaunchKernelGGL((env_step_loop_kernel<NT_, 0>), dim3(waves_total), dim3(64), sm, stream, q, T); \
        return true;
    switch (p.N) {
        case 5: hipLaunchKernelGGL((env_step_loop_kernel<5, 1>), dim3(waves_total), dim3(64), sm, stream, q, T); return true;
        MCN_LOOP_CASE(6) MCN_LOOP_CASE(7) MCN_LOOP_CASE(8) MCN_LOOP_CASE(9) MCN_LOOP_CASE(10)
        default: break;
    }
#undef MCN_LOOP_CASE
    return false;
}

bool launch_env_step_quad(const StepParams &p, hipStream_t stream);      // env_step_quad.hip
bool launch_env_pair(const StepParams &p, hipStream_t stream);           // env_pair.hip

int launch_env_step(const StepParams &p_in, hipStream_t stream)
{
    StepParams p = p_in;
    const int defer_policy = p.lp3_defer;
    p.lp3_defer = 0;                                  // what the kernels see: 0 / 1
    // <= 4 ORCA neighbours per human: quad-parallel kernel (env_step_quad.hip)
    if (p.quad_max_envs > 0 && p.E <= p.quad_max_envs && launch_env_step_quad(p, stream))
        return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
    const int waves_total = (p.E + p.G - 1) / p.G;
    // The compile-time-N ORCA kernels can park their 3-D LPs when the caller gave them a queue.  Measured on MI355X
    // (round 3, circle crossing with pool restarts; step kernel alone / + the finish kernel vs solving in place):
    // 10 humans 2^18 envs 344 -> 249 + 40 us, 32 768 envs 65 -> 65, 4096 envs 27 -> 31; 7 humans 2^18 envs 110 -> 122;
    // 5 humans 2^20 envs 207 -> 205 + 20.  Only 0.9 / 1.8 / 3.7 % of the humans need the 3-D LP at 5 / 7 / 10 humans
    // and the in-place code already skips every block no lane of the wavefront needs, so deferral pays only for the
    // largest crowds in throughput-bound batches: automatic from 8 neighbours and 16 384 wavefronts.
    const int nc = p.N - 1 + (p.cfg.robot_visible ? 1 : 0);
    if (p.out.lp3_queue && p.cfg.human_policy == MCN_HUMANS_ORCA && !p.force_generic && p.N >= 2 && p.N <= 10 &&
        nc >= 1 && nc <= kMaxLines && (defer_policy > 0 || (defer_policy < 0 && nc >= 8 && waves_total >= 16384)))
        p.lp3_defer = 1;
    // given velocities, large batch: the streaming form (env_pair.hip); mcn_tuning.pair_stream overrides
    if ((p.pair_stream > 0 || (p.pair_stream < 0 && waves_total > 4096)) && launch_env_pair(p, stream))
        return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
    if (one_wave_batch(p, waves_total, nc)) dispatch<64>(p, waves_total, stream);
    else                     dispatch<256>(p, (waves_total + 3) / 4, stream);
    if (p.lp3_defer) launch_env_lp3(p, stream);
    return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
}

}  // namespace mcn
