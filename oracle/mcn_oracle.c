/*
 * mcn_oracle.c -- ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the rollout hot path of minh86/ModelCrowdNav.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; nothing under modelcrowdnav_amd/ may import, link or
 * call it.  It is the checker, never the product.
 *
 * What it restates (reference file:line, all under /root/reference):
 *   - point_to_segment_dist            crowd_sim/envs/utils/utils.py:4-26
 *   - robot-human swept-circle test    crowd_sim/envs/crowd_sim.py:345-365
 *   - human-human overlap count        crowd_sim/envs/crowd_sim.py:368-376
 *   - goal test + reward ladder        crowd_sim/envs/crowd_sim.py:379-403
 *   - integrate / look-ahead obs       crowd_sim/envs/crowd_sim.py:405-432,
 *                                      crowd_sim/envs/utils/agent.py:63-74,110-138
 *   - ModelCrowdSim.step               crowd_sim/envs/model_crowd_sim.py:347-441
 *   - ORCA.predict parameterisation    crowd_sim/envs/policy/orca.py:82-132
 *   - Linear.predict                   crowd_sim/envs/policy/linear.py:15-22
 *   - MultiHumanRL.compute_reward      crowd_nav/policy/multi_human_rl.py:65-88
 *
 * ORCA PARITY VS rvo2 IS UNPINNED.  The velocity solve itself lives in the
 * third-party module `rvo2` (Python-RVO2 wrapping the RVO2 C++ library, UNC,
 * Apache-2.0), which is imported at crowd_sim/envs/policy/orca.py:2 but is
 * neither vendored, pinned (absent from setup.py:17-25) nor installed here.
 * The solver below restates the published algorithm (van den Berg, Guy, Lin,
 * Manocha: "Reciprocal n-body collision avoidance", 2011) as RVO2 v2.0.x is
 * documented to implement it: float32 throughout, neighbours sorted by
 * squared distance (stable, capped at maxNeighbors inside neighborDist),
 * one half-plane per neighbour (cut-off circle / left leg / right leg /
 * collision case), incremental 2-D LP with fallback to the 3-D LP that
 * minimises the maximum penetration, epsilon 1e-5.  It is pinned by
 * feasibility + optimality tests against a brute-force solver (tests/), by
 * analytic cases, and by the reference's own call-site conventions.
 *
 * Numerics notes that matter for bit-exact masks:
 *   - all env arithmetic is IEEE double in the reference's operation order;
 *     build with -ffp-contract=off.
 *   - numpy.linalg.norm of a 2-vector is sqrt(dot(x,x)); on this image's
 *     numpy/OpenBLAS that dot is sqrt(fma(x1,x1,x0*x0)) (verified against
 *     20 000 random pairs, tests/golden_tools/gen_golden.py re-checks it).  norm2() below
 *     therefore uses an explicit fma().
 *   - ORCA velocities are float32 values widened to double
 *     (rvo2 returns C floats through Cython).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define MCN_MAX_NEIGH 64
#define RVO_EPS 1e-5f

/* ------------------------------------------------------------------ */
/* float32 2-D helpers (RVO2 Vector2 semantics)                        */
/* ------------------------------------------------------------------ */
typedef struct { float x, y; } v2;
typedef struct { v2 p, d; } hline;   /* half-plane: point + direction  */

static inline v2 V(float x, float y) { v2 r = { x, y }; return r; }
static inline v2 vadd(v2 a, v2 b) { return V(a.x + b.x, a.y + b.y); }
static inline v2 vsub(v2 a, v2 b) { return V(a.x - b.x, a.y - b.y); }
static inline v2 vscale(float s, v2 a) { return V(s * a.x, s * a.y); }
static inline float vdot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline float vdet(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
static inline float vabssq(v2 a) { return vdot(a, a); }
/* RVO2's Vector2::operator/(float) multiplies by the reciprocal. */
static inline v2 vdiv(v2 a, float s) { const float inv = 1.0f / s; return V(a.x * inv, a.y * inv); }
static inline v2 vnormalize(v2 a) { return vdiv(a, sqrtf(vabssq(a))); }
static inline float sqrf(float a) { return a * a; }

/* 1-D LP along line `no`, constrained by lines [0,no) and the speed disc. */
static int lp1(const hline *L, int no, float radius, v2 opt, int dir_opt, v2 *res)
{
    const float dp = vdot(L[no].p, L[no].d);
    const float disc = sqrf(dp) + sqrf(radius) - vabssq(L[no].p);
    if (disc < 0.0f) return 0;
    const float sq = sqrtf(disc);
    float tl = -dp - sq;
    float tr = -dp + sq;
    for (int i = 0; i < no; ++i) {
        const float den = vdet(L[no].d, L[i].d);
        const float num = vdet(L[i].d, vsub(L[no].p, L[i].p));
        if (fabsf(den) <= RVO_EPS) {
            if (num < 0.0f) return 0;
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) tr = fminf(tr, t);
        else             tl = fmaxf(tl, t);
        if (tl > tr) return 0;
    }
    if (dir_opt) {
        if (vdot(opt, L[no].d) > 0.0f) *res = vadd(L[no].p, vscale(tr, L[no].d));
        else                           *res = vadd(L[no].p, vscale(tl, L[no].d));
    } else {
        const float t = vdot(L[no].d, vsub(opt, L[no].p));
        if (t < tl)      *res = vadd(L[no].p, vscale(tl, L[no].d));
        else if (t > tr) *res = vadd(L[no].p, vscale(tr, L[no].d));
        else             *res = vadd(L[no].p, vscale(t, L[no].d));
    }
    return 1;
}

/* Incremental 2-D LP.  Returns the index of the first failing line, or n. */
static int lp2(const hline *L, int n, float radius, v2 opt, int dir_opt, v2 *res)
{
    if (dir_opt)                           *res = vscale(radius, opt);   /* opt * radius */
    else if (vabssq(opt) > sqrf(radius))   *res = vscale(radius, vnormalize(opt));
    else                                   *res = opt;
    for (int i = 0; i < n; ++i) {
        if (vdet(L[i].d, vsub(L[i].p, *res)) > 0.0f) {
            const v2 keep = *res;
            if (!lp1(L, i, radius, opt, dir_opt, res)) { *res = keep; return i; }
        }
    }
    return n;
}

/* Test hook: how many agents fell through to the 3-D LP since the last reset (coverage assertions in tests/;
   not thread-safe, the oracle is single-threaded). */
static long g_lp3_entries = 0;
long mcn_oracle_lp3_entries(int reset) { const long n = g_lp3_entries; if (reset) g_lp3_entries = 0; return n; }

/* Minimise the maximum penetration once lp2 failed at line `begin`. */
static void lp3(const hline *L, int n, int begin, float radius, v2 *res)
{
    float dist = 0.0f;
    ++g_lp3_entries;
    hline P[MCN_MAX_NEIGH];
    for (int i = begin; i < n; ++i) {
        if (vdet(L[i].d, vsub(L[i].p, *res)) > dist) {
            int m = 0;
            for (int j = 0; j < i; ++j) {
                hline q;
                const float dt = vdet(L[i].d, L[j].d);
                if (fabsf(dt) <= RVO_EPS) {
                    if (vdot(L[i].d, L[j].d) > 0.0f) continue;
                    q.p = vscale(0.5f, vadd(L[i].p, L[j].p));
                } else {
                    const float s = vdet(L[j].d, vsub(L[i].p, L[j].p)) / dt;
                    q.p = vadd(L[i].p, vscale(s, L[i].d));
                }
                q.d = vnormalize(vsub(L[j].d, L[i].d));
                P[m++] = q;
            }
            const v2 keep = *res;
            if (lp2(P, m, radius, V(-L[i].d.y, L[i].d.x), 1, res) < m) *res = keep;
            dist = vdet(L[i].d, vsub(L[i].p, *res));
        }
    }
}

/*
 * Neighbour selection + half-plane construction of one agent (RVO2 Agent::computeNeighbors /
 * computeNewVelocity up to the linear programs; call sites orca.py:95-129).  ONE function: the solver
 * (mcn_oracle_orca_agent) and the test hook (mcn_oracle_orca_lines) both call it, so the independent geometry pins of
 * tests/test_orca_pins.py exercise exactly the lines the solver optimises over.
 * `o*` arrays describe the candidate neighbours in insertion order (index order, self skipped, robot last when
 * visible -- crowd_sim.py:339-341, orca.py:99-110).  Returns the number of lines written to L (solver order).
 */
static int orca_build_lines(v2 pos, v2 vel, float radius, int n_other,
                            const float *opx, const float *opy, const float *ovx, const float *ovy,
                            const float *orad, float neighbor_dist, int max_neighbors,
                            float time_horizon, float time_step, hline *L)
{
    /* neighbour selection: stable insertion by squared distance, shrinking range
       once the list is full (RVO2 Agent::insertAgentNeighbor). */
    int   idx[MCN_MAX_NEIGH];
    float dsq[MCN_MAX_NEIGH];
    int   cnt = 0;
    float range_sq = sqrf(neighbor_dist);
    if (max_neighbors > MCN_MAX_NEIGH) max_neighbors = MCN_MAX_NEIGH;
    if (max_neighbors > 0) {
        for (int j = 0; j < n_other; ++j) {
            const float d = vabssq(vsub(pos, V(opx[j], opy[j])));
            if (d < range_sq) {
                if (cnt < max_neighbors) ++cnt;
                int i = cnt - 1;
                while (i != 0 && d < dsq[i - 1]) { dsq[i] = dsq[i - 1]; idx[i] = idx[i - 1]; --i; }
                dsq[i] = d; idx[i] = j;
                if (cnt == max_neighbors) range_sq = dsq[cnt - 1];
            }
        }
    }

    const float inv_th = 1.0f / time_horizon;
    for (int k = 0; k < cnt; ++k) {
        const int j = idx[k];
        const v2 rp = vsub(V(opx[j], opy[j]), pos);
        const v2 rv = vsub(vel, V(ovx[j], ovy[j]));
        const float dist_sq = vabssq(rp);
        const float cr = radius + orad[j];
        const float cr_sq = sqrf(cr);
        hline ln; v2 u;
        if (dist_sq > cr_sq) {
            const v2 w = vsub(rv, vscale(inv_th, rp));
            const float wl_sq = vabssq(w);
            const float dp1 = vdot(w, rp);
            if (dp1 < 0.0f && sqrf(dp1) > cr_sq * wl_sq) {
                const float wl = sqrtf(wl_sq);
                const v2 uw = vdiv(w, wl);
                ln.d = V(uw.y, -uw.x);
                u = vscale(cr * inv_th - wl, uw);
            } else {
                const float leg = sqrtf(dist_sq - cr_sq);
                if (vdet(rp, w) > 0.0f) {
                    ln.d = vdiv(V(rp.x * leg - rp.y * cr, rp.x * cr + rp.y * leg), dist_sq);
                } else {
                    const v2 t = vdiv(V(rp.x * leg + rp.y * cr, -rp.x * cr + rp.y * leg), dist_sq);
                    ln.d = V(-t.x, -t.y);
                }
                const float dp2 = vdot(rv, ln.d);
                u = vsub(vscale(dp2, ln.d), rv);
            }
        } else {
            const float inv_ts = 1.0f / time_step;
            const v2 w = vsub(rv, vscale(inv_ts, rp));
            const float wl = sqrtf(vabssq(w));
            const v2 uw = vdiv(w, wl);
            ln.d = V(uw.y, -uw.x);
            u = vscale(cr * inv_ts - wl, uw);
        }
        ln.p = vadd(vel, vscale(0.5f, u));
        L[k] = ln;
    }
    return cnt;
}

/* One agent's ORCA velocity: the half-planes above, then the incremental 2-D LP and, if it fails, the 3-D LP. */
void mcn_oracle_orca_agent(float px, float py, float vx, float vy, float radius, float max_speed,
                           float pref_x, float pref_y, int n_other,
                           const float *opx, const float *opy, const float *ovx, const float *ovy,
                           const float *orad, float neighbor_dist, int max_neighbors,
                           float time_horizon, float time_step, float *out_vx, float *out_vy)
{
    hline L[MCN_MAX_NEIGH];
    const int cnt = orca_build_lines(V(px, py), V(vx, vy), radius, n_other, opx, opy, ovx, ovy, orad,
                                     neighbor_dist, max_neighbors, time_horizon, time_step, L);
    v2 res;
    const int fail = lp2(L, cnt, max_speed, V(pref_x, pref_y), 0, &res);
    if (fail < cnt) lp3(L, cnt, fail, max_speed, &res);
    *out_vx = res.x; *out_vy = res.y;
}

/* Debug/test hook: the ORCA half-planes of one agent in solver order -- the SAME lines the solver above sees
   (one shared construction, orca_build_lines). */
int mcn_oracle_orca_lines(float px, float py, float vx, float vy, float radius, int n_other,
                          const float *opx, const float *opy, const float *ovx, const float *ovy,
                          const float *orad, float neighbor_dist, int max_neighbors,
                          float time_horizon, float time_step, float *out_lines /* [n][4] */)
{
    hline L[MCN_MAX_NEIGH];
    const int cnt = orca_build_lines(V(px, py), V(vx, vy), radius, n_other, opx, opy, ovx, ovy, orad,
                                     neighbor_dist, max_neighbors, time_horizon, time_step, L);
    for (int k = 0; k < cnt; ++k) {
        out_lines[4 * k + 0] = L[k].p.x; out_lines[4 * k + 1] = L[k].p.y;
        out_lines[4 * k + 2] = L[k].d.x; out_lines[4 * k + 3] = L[k].d.y;
    }
    return cnt;
}

/* ------------------------------------------------------------------ */
/* float64 env arithmetic                                              */
/* ------------------------------------------------------------------ */
static inline double norm2(double x0, double x1) { return sqrt(fma(x1, x1, x0 * x0)); }

/* crowd_sim/envs/utils/utils.py:4-26 */
double mcn_oracle_point_to_segment_dist(double x1, double y1, double x2, double y2, double x3, double y3)
{
    const double px = x2 - x1, py = y2 - y1;
    if (px == 0 && py == 0) return norm2(x3 - x1, y3 - y1);
    double u = ((x3 - x1) * px + (y3 - y1) * py) / (px * px + py * py);
    if (u > 1) u = 1; else if (u < 0) u = 0;
    const double x = x1 + u * px, y = y1 + u * py;
    return norm2(x - x3, y - y3);
}

enum { MCN_INFO_NOTHING = 0, MCN_INFO_DANGER = 1, MCN_INFO_REACHGOAL = 2, MCN_INFO_COLLISION = 3, MCN_INFO_TIMEOUT = 4 };
enum { MCN_HUMANS_ORCA = 0, MCN_HUMANS_LINEAR = 1, MCN_HUMANS_GIVEN = 2 };

typedef struct {
    double time_step, time_limit;
    double success_reward, collision_penalty, discomfort_dist, discomfort_penalty_factor;
    int    robot_visible;        /* humans see the robot (crowd_sim.py:340) */
    int    human_policy;         /* MCN_HUMANS_* */
    int    count_hh;             /* CrowdSim counts human-human overlaps, ModelCrowdSim does not */
    int    track_human_times;    /* crowd_sim.py:418-421 */
    /* ORCA parameters, orca.py:60-66 */
    double orca_safety_space;
    float  orca_neighbor_dist; int orca_max_neighbors;
    float  orca_time_horizon;  float orca_max_speed;
    int    robot_unicycle;       /* robot action is (v, r): crowd_sim.py:353-355, agent.py:110-135 */
} mcn_oracle_cfg;

/*
 * One batched env step, scalar inside, env by env.  All arrays are row-major
 * [E] or [E*N] doubles.  With update != 0 state is advanced in place
 * (crowd_sim.py:405-427); otherwise the next observable states go to nobs_*
 * (crowd_sim.py:428-432) and state is left untouched.
 * given_v (x,y interleaved, [E*N*2]) feeds MCN_HUMANS_GIVEN (model_crowd_sim.py:347,417).
 */
void mcn_oracle_env_step(const mcn_oracle_cfg *c, int E, int N, int update,
                         double *hpx, double *hpy, double *hvx, double *hvy,
                         const double *hgx, const double *hgy, const double *hr, const double *hvpref,
                         double *rpx, double *rpy, double *rvx, double *rvy,
                         const double *rgx, const double *rgy, const double *rr, double *rtheta /* [E] or NULL */,
                         double *gtime, double *human_times /* [E*N] or NULL */,
                         const double *ax, const double *ay, const double *given_v,
                         double *reward, uint8_t *done, uint8_t *info, double *dmin_out, int32_t *hh_count,
                         double *nobs_px, double *nobs_py, double *nobs_vx, double *nobs_vy,
                         double *human_act /* [E*N*2] or NULL: the humans' chosen velocities */)
{
    const double dt = c->time_step;
    for (int e = 0; e < E; ++e) {
        const int b = e * N;
        double hax[MCN_MAX_NEIGH], hay[MCN_MAX_NEIGH];

        /* ---- human actions (crowd_sim.py:336-342) ---- */
        for (int i = 0; i < N; ++i) {
            if (c->human_policy == MCN_HUMANS_ORCA) {
                float opx[MCN_MAX_NEIGH], opy[MCN_MAX_NEIGH], ovx[MCN_MAX_NEIGH], ovy[MCN_MAX_NEIGH], orad[MCN_MAX_NEIGH];
                int m = 0;
                for (int j = 0; j < N; ++j) {
                    if (j == i) continue;
                    opx[m] = (float)hpx[b + j]; opy[m] = (float)hpy[b + j];
                    ovx[m] = (float)hvx[b + j]; ovy[m] = (float)hvy[b + j];
                    orad[m] = (float)(hr[b + j] + 0.01 + c->orca_safety_space);
                    ++m;
                }
                if (c->robot_visible) {
                    opx[m] = (float)rpx[e]; opy[m] = (float)rpy[e];
                    ovx[m] = (float)rvx[e]; ovy[m] = (float)rvy[e];
                    orad[m] = (float)(rr[e] + 0.01 + c->orca_safety_space);
                    ++m;
                }
                float nvx, nvy;
                mcn_oracle_orca_agent((float)hpx[b + i], (float)hpy[b + i], (float)hvx[b + i], (float)hvy[b + i],
                                      (float)(hr[b + i] + 0.01 + c->orca_safety_space), (float)hvpref[b + i],
                                      (float)(hgx[b + i] - hpx[b + i]), (float)(hgy[b + i] - hpy[b + i]),
                                      m, opx, opy, ovx, ovy, orad,
                                      c->orca_neighbor_dist, c->orca_max_neighbors, c->orca_time_horizon,
                                      (float)dt, &nvx, &nvy);
                hax[i] = (double)nvx; hay[i] = (double)nvy;
            } else if (c->human_policy == MCN_HUMANS_LINEAR) {
                const double th = atan2(hgy[b + i] - hpy[b + i], hgx[b + i] - hpx[b + i]);
                hax[i] = cos(th) * hvpref[b + i]; hay[i] = sin(th) * hvpref[b + i];
            } else {
                hax[i] = given_v[2 * (b + i)]; hay[i] = given_v[2 * (b + i) + 1];
            }
            if (human_act) { human_act[2 * (b + i)] = hax[i]; human_act[2 * (b + i) + 1] = hay[i]; }
        }

        /* ---- robot-human swept test (crowd_sim.py:345-365) ---- */
        double dmin = INFINITY; int collision = 0;
        double eax = ax[e], eay = ay[e];        /* robot velocity seen by the swept test */
        if (c->robot_unicycle) {
            eax = ax[e] * cos(ay[e] + rtheta[e]);
            eay = ax[e] * sin(ay[e] + rtheta[e]);
        }
        for (int i = 0; i < N; ++i) {
            const double px = hpx[b + i] - rpx[e], py = hpy[b + i] - rpy[e];
            const double vx = hvx[b + i] - eax, vy = hvy[b + i] - eay;
            const double ex = px + vx * dt, ey = py + vy * dt;
            const double cd = mcn_oracle_point_to_segment_dist(px, py, ex, ey, 0, 0) - hr[b + i] - rr[e];
            if (cd < 0) collision = 1;
            if (cd < dmin) dmin = cd;
        }

        /* ---- human-human overlaps (crowd_sim.py:368-376) ---- */
        int hh = 0;
        if (c->count_hh) {
            for (int i = 0; i < N; ++i)
                for (int j = i + 1; j < N; ++j) {
                    const double dx = hpx[b + i] - hpx[b + j], dy = hpy[b + i] - hpy[b + j];
                    /* reference: (dx**2 + dy**2)**(1/2), i.e. CPython float pow = libm pow(x, 0.5), kept literally: the
                       oracle follows the reference, not the kernels.  The kernels use the correctly rounded sqrt (a documented
                       deviation): glibc's pow differs from it by one ulp for ~0.05 % of arguments, which can change
                       the sign of `d` only for pairs within one ulp of touching
                       (tests/test_oracle_golden.py::test_pow_half_vs_sqrt_never_flips_the_overlap_test) */
                    const double d = pow(dx * dx + dy * dy, 0.5) - hr[b + i] - hr[b + j];
                    if (d < 0) ++hh;
                }
        }

        /* ---- goal test + ladder (crowd_sim.py:379-403) ---- */
        double endx, endy, nrvx = ax[e], nrvy = ay[e], nth = 0;
        if (c->robot_unicycle) {
            const double th = rtheta[e] + ay[e];                       /* agent.py:115-118 */
            endx = rpx[e] + cos(th) * ax[e] * dt; endy = rpy[e] + sin(th) * ax[e] * dt;
            nth = fmod(rtheta[e] + ay[e], 2 * M_PI);                   /* agent.py:133, Python % */
            if (nth != 0 && nth < 0) nth += 2 * M_PI;
            nrvx = ax[e] * cos(nth); nrvy = ax[e] * sin(nth);
        } else {
            endx = rpx[e] + ax[e] * dt; endy = rpy[e] + ay[e] * dt;
        }
        const int reaching = norm2(endx - rgx[e], endy - rgy[e]) < rr[e];
        double rew; uint8_t dn, inf;
        if (gtime[e] >= c->time_limit - 1)   { rew = 0; dn = 1; inf = MCN_INFO_TIMEOUT; }
        else if (collision)                  { rew = c->collision_penalty; dn = 1; inf = MCN_INFO_COLLISION; }
        else if (reaching)                   { rew = c->success_reward; dn = 1; inf = MCN_INFO_REACHGOAL; }
        else if (dmin < c->discomfort_dist)  { rew = (dmin - c->discomfort_dist) * c->discomfort_penalty_factor * dt; dn = 0; inf = MCN_INFO_DANGER; }
        else                                 { rew = 0; dn = 0; inf = MCN_INFO_NOTHING; }
        reward[e] = rew; done[e] = dn; info[e] = inf; dmin_out[e] = dmin; hh_count[e] = hh;

        /* ---- integrate or look ahead ---- */
        if (update) {
            rpx[e] = endx; rpy[e] = endy; rvx[e] = nrvx; rvy[e] = nrvy;
            if (c->robot_unicycle) rtheta[e] = nth;
            for (int i = 0; i < N; ++i) {
                hpx[b + i] = hpx[b + i] + hax[i] * dt; hpy[b + i] = hpy[b + i] + hay[i] * dt;
                hvx[b + i] = hax[i]; hvy[b + i] = hay[i];
            }
            gtime[e] += dt;
            if (c->track_human_times && human_times) {
                for (int i = 0; i < N; ++i)
                    if (human_times[b + i] == 0 &&
                        norm2(hpx[b + i] - hgx[b + i], hpy[b + i] - hgy[b + i]) < hr[b + i])
                        human_times[b + i] = gtime[e];
            }
        } else {
            for (int i = 0; i < N; ++i) {
                nobs_px[b + i] = hpx[b + i] + hax[i] * dt; nobs_py[b + i] = hpy[b + i] + hay[i] * dt;
                nobs_vx[b + i] = hax[i]; nobs_vy[b + i] = hay[i];
            }
        }
    }
}

/*
 * MultiHumanRL.compute_reward (multi_human_rl.py:65-88): end-position distance test
 * with the reference's hard-coded -0.25 / 1 / 0.2 / 0.5 constants.  nav = robot after
 * the candidate action, humans = constant-velocity propagated.  [E] x [A] candidates.
 */
void mcn_oracle_lookahead_reward(int E, int N, int A, double dt,
                                 const double *rpx, const double *rpy, const double *rgx, const double *rgy, const double *rr,
                                 const double *hpx, const double *hpy, const double *hvx, const double *hvy, const double *hr,
                                 const double *act /* [A*2] */, double *out /* [E*A] */)
{
    for (int e = 0; e < E; ++e)
        for (int a = 0; a < A; ++a) {
            const double nx = rpx[e] + act[2 * a] * dt, ny = rpy[e] + act[2 * a + 1] * dt;
            double dmin = INFINITY; int coll = 0;
            for (int i = 0; i < N; ++i) {
                const int k = e * N + i;
                const double qx = hpx[k] + hvx[k] * dt, qy = hpy[k] + hvy[k] * dt;
                const double d = norm2(nx - qx, ny - qy) - rr[e] - hr[k];
                if (d < 0) { coll = 1; break; }
                if (d < dmin) dmin = d;
            }
            const int reach = norm2(nx - rgx[e], ny - rgy[e]) < rr[e];
            double r;
            if (coll) r = -0.25; else if (reach) r = 1; else if (dmin < 0.2) r = (dmin - 0.2) * 0.5 * dt; else r = 0;
            out[e * A + a] = r;
        }
}
