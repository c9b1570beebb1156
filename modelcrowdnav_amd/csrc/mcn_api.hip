// mcn_api.hip -- extern "C" entry points of libmcn_hip.so (see include/mcn.h).
// Argument validation happens here, on the host, before any kernel is launched: a kernel
// is never started on shapes its indexing does not assume.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/mcn.h"

#include "env_step_params.hpp"
#include "lp3_queue.hpp"

namespace mcn {
int launch_env_step(const StepParams &p, hipStream_t stream);
bool launch_env_rollout_quad(const StepParams &p, int T, hipStream_t stream);
bool launch_env_step_loop(const StepParams &p, int T, hipStream_t stream);
int launch_scenario_pool(const mcn_scenario_cfg &c, uint64_t seed, int64_t first_case, int P, int N, double *hpos,
                         double *hgoal, double *hrad, double *hvpref, hipStream_t stream);
struct SarlParams;
long sarl_workspace_float4s(int E, int N, int A);
int launch_mlp_world(const mcn_mlp_world_net *net, const double *hpos, const double *hvel, double *out_vel, int E, int N,
                     hipStream_t stream);
int launch_attn_world(const mcn_attn_world_net *net, const double *hpos, const double *hvel, const int32_t *hcount,
                      void *workspace, double *out_vel, int E, int N, hipStream_t stream);
long attn_world_workspace_float4s(int E, int N);
int launch_sarl_c(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int A, double dt,
                  double gamma_pow, int kinematics, void *workspace, double *values, int32_t *best, double *best_val,
                  float *attention, const double *next_hpos, const double *next_hvel, const double *reward_in,
                  double *action_out, double epsilon, unsigned long long seed, int E, int N, hipStream_t stream);
#ifdef MCN_DIAG
int read_pool_clock(void *dst, size_t bytes);
int read_sarl_phases(void *dst, size_t bytes, int reset);
#endif
int launch_sgan(const mcn_sgan_net *net, double *hist, int push_slot, int oldest, const double *cur_pos,
                const float *noise, const int32_t *hcount, void *workspace, double *out_vel, float *out_rel,
                double time_step, int E, int N, hipStream_t stream);
int launch_orca_batch(const float *self, const float *others, const int32_t *n_other, float *out,
                      int B, int M, float neighbor_dist, int max_neighbors, float time_horizon, float time_step,
                      hipStream_t stream);
#ifdef MCN_DIAG
int read_stamps(void *dst, size_t bytes);
int read_counts(void *dst, size_t bytes, int reset);
#endif
}  // namespace mcn

// Dispatch overrides (include/mcn.h: mcn_tuning).  The MCN_* environment variables are read ONCE, the first time a
// launch needs them, as the initial values; after that only mcn_set_tuning changes them.  No getenv on the launch path.
static mcn_tuning g_tuning;
static bool g_tuning_init = false;
static std::mutex g_tuning_mu;          // launches copy the settings under it: a concurrent mcn_set_tuning never tears them
static int env_or(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
static mcn_tuning tuning_from_env()
{
    mcn_tuning t;
    memset(&t, 0, sizeof(t));
    t.force_generic = env_or("MCN_FORCE_GENERIC", 0);
    t.quad_max_envs = env_or("MCN_QUAD_MAX_ENVS", -1);
    t.quad_split = env_or("MCN_QUAD_SPLIT", -1);
    t.rollout_fused = env_or("MCN_ROLLOUT_FUSED", -1);
    t.rollout_split = env_or("MCN_ROLLOUT_SPLIT", -1);
    t.step_block = env_or("MCN_STEP_BLOCK", -1);
    t.pair_stream = env_or("MCN_PAIR_STREAM", -1);
    t.lp3_defer = env_or("MCN_LP3_DEFER", -1);
    t.sarl_x3 = env_or("MCN_SARL_X3", -1);
    return t;
}
static mcn_tuning tuning()
{
    std::lock_guard<std::mutex> lock(g_tuning_mu);
    if (!g_tuning_init) { g_tuning = tuning_from_env(); g_tuning_init = true; }
    return g_tuning;
}

namespace mcn { bool tuning_sarl_x3() { return tuning().sarl_x3 != 0; } }

// bf16 round-to-nearest-even of a float32 (NaN kept quiet), as the device's v_cvt_pk_bf16_f32
static uint16_t bf16_rne(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int64_t mcn_pack_x3_bytes(int32_t NT, int32_t KT)
{
    if (NT <= 0 || KT <= 0) return 0;
    return (int64_t)NT * ((KT + 1) / 2) * 3 * 64 * 8 * 2;
}

int mcn_pack_x3(const float *wfrag, int32_t NT, int32_t KT, void *x3_out)
{
    if (!wfrag || !x3_out || NT <= 0 || KT <= 0) return MCN_EINVAL;
    const int KB = (KT + 1) / 2;
    uint16_t *out = reinterpret_cast<uint16_t *>(x3_out);
    for (int n = 0; n < NT; ++n)
        for (int m = 0; m < KB; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 8; ++s) {
                    // k-slot 8 q + s of input block m = slot 4 q + (s & 3) of k-tile 2 m + (s >> 2)
                    const int t = 2 * m + (s >> 2);
                    const float w = t < KT ? wfrag[(((size_t)n * KT + t) * 64 + lane) * 4 + (s & 3)] : 0.0f;
                    const uint16_t hi = bf16_rne(w);
                    const float r1 = w - bf16_to_f32(hi);
                    const uint16_t mid = bf16_rne(r1);
                    const float r2 = r1 - bf16_to_f32(mid);
                    const size_t at = (((size_t)n * KB + m) * 3 * 64 + lane) * 8 + s;
                    out[at] = hi; out[at + 64 * 8] = mid; out[at + 2 * 64 * 8] = bf16_rne(r2);
                }
    return MCN_OK;
}

// mcn_last_dispatch(): the launch functions of the env kernels note which family they picked (per host thread).
static thread_local const char *g_last_dispatch = "";
namespace mcn { void note_dispatch(const char *family) { g_last_dispatch = family; } }
const char *mcn_last_dispatch(void) { return g_last_dispatch; }

int32_t mcn_abi_version(void) { return MCN_ABI_VERSION; }
int64_t mcn_sizeof(int32_t which)
{
    switch (which) {
        case MCN_SIZEOF_ENV_CFG: return sizeof(mcn_env_cfg);
        case MCN_SIZEOF_ENV_STATE: return sizeof(mcn_env_state);
        case MCN_SIZEOF_ENV_OUT: return sizeof(mcn_env_out);
        case MCN_SIZEOF_ROLLOUT: return sizeof(mcn_rollout);
        case MCN_SIZEOF_TUNING: return sizeof(mcn_tuning);
        case MCN_SIZEOF_STEP_REC: return sizeof(mcn_step_rec);
        case MCN_SIZEOF_ROLL_REC: return sizeof(mcn_roll_rec);
        case MCN_SIZEOF_SARL_NET: return sizeof(mcn_sarl_net);
        case MCN_SIZEOF_SGAN_NET: return sizeof(mcn_sgan_net);
        case MCN_SIZEOF_SCENARIO_CFG: return sizeof(mcn_scenario_cfg);
        case MCN_SIZEOF_MLP_WORLD_NET: return sizeof(mcn_mlp_world_net);
        case MCN_SIZEOF_ATTN_WORLD_NET: return sizeof(mcn_attn_world_net);
        case MCN_SIZEOF_SARL_X3: return sizeof(mcn_sarl_x3);
        default: return -1;
    }
}

// Validates one env-step problem and fills the kernel argument block.  Shared by mcn_env_step / mcn_env_rollout.
static int fill_step_params(mcn::StepParams &p, const mcn_env_cfg *cfg, const mcn_env_state *st, const double *actions,
                            const double *given_v, const mcn_env_out *out, const mcn_rollout *roll,
                            int32_t E, int32_t N, int32_t update)
{
    if (!cfg || !st || !out || !actions) return MCN_EINVAL;
    if (E <= 0 || N <= 0 || N > MCN_MAX_HUMANS) return MCN_EINVAL;
    if (!st->hpos || !st->hvel || !st->hgoal || !st->hrad || !st->hvpref || !st->rpos || !st->rvel || !st->rgoal ||
        !st->rrad || !st->gtime) return MCN_EINVAL;
    if (cfg->robot_kinematics == MCN_KIN_UNICYCLE && !st->rtheta) return MCN_EINVAL;
    if (!out->rec) return MCN_EINVAL;
    if (!update && (!out->nobs_pos || !out->nobs_vel)) return MCN_EINVAL;
    if (cfg->human_policy == MCN_HUMANS_GIVEN && !given_v) return MCN_EINVAL;
    if (cfg->human_policy < MCN_HUMANS_ORCA || cfg->human_policy > MCN_HUMANS_GIVEN) return MCN_EINVAL;
    if (cfg->orca_max_neighbors < 0 || cfg->orca_max_neighbors > MCN_MAX_LINES) return MCN_EINVAL;
    if (!(cfg->time_step > 0)) return MCN_EINVAL;
    if (roll) {
        if (roll->state && (!roll->disc_table || roll->disc_len <= 0 || roll->fin_slots < 1)) return MCN_EINVAL;
        if (roll->pool_hpos && !roll->state) return MCN_EINVAL;
        if (roll->pool_hpos && (!roll->pool_hgoal || !roll->pool_hrad || !roll->pool_hvpref || roll->pool_size <= 0)) return MCN_EINVAL;
        if (roll->pool_hpos && (roll->case_stride < 0 || roll->case_stride >= roll->pool_size)) return MCN_EINVAL;
    }
    memset(&p, 0, sizeof(p));
    p.cfg = *cfg; p.st = *st; p.out = *out;
    if (roll) { p.roll = *roll; p.has_roll = 1; }
    p.actions = actions; p.given_v = given_v;
    p.E = E; p.N = N; p.G = 64 / N; p.update = update ? 1 : 0;
    int ncand = N - 1 + (cfg->robot_visible ? 1 : 0);
    int nl = ncand < cfg->orca_max_neighbors ? ncand : cfg->orca_max_neighbors;
    if (cfg->human_policy != MCN_HUMANS_ORCA) nl = 0;
    p.nl_cap = nl;
    // Small batches are latency-bound: use the quad-parallel kernel (env_step_quad.hip) while its 4x wider
    // grid still fits the chip about twice over (measured cross-over on MI355X: ~2800 wavefronts, i.e.
    // E <= 8192 at 5 humans); above that the lane-per-human kernel wins on throughput.  Inside the quad
    // kernel, ORCA and the float64 pairwise work go to two cooperating wavefronts only while BOTH still get a
    // SIMD of their own (grid <= 512 workgroups).  mcn_set_tuning overrides (tests, tuning).
    const mcn_tuning tu = tuning();
    p.force_generic = tu.force_generic > 0 ? 1 : 0;
    p.pair_stream = tu.pair_stream;
    p.step_block = tu.step_block;
    p.lp3_defer = tu.lp3_defer;
#ifdef MCN_DIAG
    p.debug_noop = tu.diag_noop;
#endif
    const int envs_per_wave = 64 / (4 * N);
    const long quad_waves = envs_per_wave > 0 ? ((long)E + envs_per_wave - 1) / envs_per_wave : (1L << 40);
    p.quad_max_envs = tu.quad_max_envs >= 0 ? tu.quad_max_envs : (quad_waves <= 2800 ? E : 0);
    // two cooperating wavefronts per env group while the doubled grid still finds idle issue slots; re-measured in round 4
    // (5 humans, us per step without / with: 1024 envs 5.87 / 5.37, 2048 6.28 / 6.10, 4096 = 1366 wavefronts 7.20 / 6.95,
    //  5120 7.60 / 8.01, 6144 7.71 / 8.68): up to ~1400 wavefronts (round 3's rule stopped at 512)
    p.quad_split = tu.quad_split >= 0 ? tu.quad_split : (quad_waves <= 1400 ? 1 : 0);
    return MCN_OK;
}

extern "C" {

#ifdef MCN_DIAG
const char *mcn_version(void) { return "modelcrowdnav_amd 0.4 (gfx950) DIAGNOSTIC BUILD (time stamps)"; }
#else
const char *mcn_version(void) { return "modelcrowdnav_amd 0.4 (gfx950)"; }
#endif

int mcn_set_tuning(const mcn_tuning *t)
{
    std::lock_guard<std::mutex> lock(g_tuning_mu);
    if (!t) { g_tuning = tuning_from_env(); g_tuning_init = true; return MCN_OK; }
    if (t->quad_split > 1 || t->rollout_fused > 1 || t->rollout_split > 1 || t->pair_stream > 3 || t->force_generic < 0 || t->force_generic > 1) return MCN_EINVAL;
    if (t->quad_max_envs < -1 || t->quad_split < -1 || t->rollout_fused < -1 || t->rollout_split < -1 || t->pair_stream < -1) return MCN_EINVAL;
    if (t->step_block != -1 && t->step_block != 64 && t->step_block != 256) return MCN_EINVAL;
    if (t->lp3_defer < -1 || t->lp3_defer > 1) return MCN_EINVAL;
    if (t->sarl_x3 < -1 || t->sarl_x3 > 1) return MCN_EINVAL;
#ifndef MCN_DIAG
    if (t->diag_noop) return MCN_EINVAL;          // kernels that do nothing exist in the diagnostic build only
#endif
    g_tuning = *t;
    g_tuning_init = true;
    return MCN_OK;
}

int mcn_get_tuning(mcn_tuning *t)
{
    if (!t) return MCN_EINVAL;
    *t = tuning();
    return MCN_OK;
}

int64_t mcn_env_lp3_queue_bytes(int32_t E, int32_t N)
{
    if (E <= 0 || N < 2 || N > 10) return 0;          // the compile-time-N ORCA kernels (env_step.hip)
    return mcn::lp3_queue_bytes(E, N, N < MCN_MAX_LINES ? N : MCN_MAX_LINES);
}

int mcn_env_step(const mcn_env_cfg *cfg, const mcn_env_state *st, const double *actions,
                 const double *given_v, const mcn_env_out *out, const mcn_rollout *roll,
                 int32_t E, int32_t N, int32_t update, void *stream)
{
    mcn::StepParams p;
    const int rc = fill_step_params(p, cfg, st, actions, given_v, out, roll, E, N, update);
    if (rc != MCN_OK) return rc;
    return mcn::launch_env_step(p, (hipStream_t)stream);
}

int mcn_env_rollout(const mcn_env_cfg *cfg, const mcn_env_state *st, const double *actions, int32_t T,
                    const mcn_env_out *out, const mcn_rollout *roll, int32_t E, int32_t N, void *stream)
{
    if (T <= 0) return MCN_EINVAL;
    mcn::StepParams p;
    const int rc = fill_step_params(p, cfg, st, actions, nullptr, out, roll, E, N, 1);
    if (rc != MCN_OK) return rc;
    // One launch with the env state held in registers for all T steps where the quad layout applies and beats T
    // launches of the throughput kernel: measured cross-over on MI355X at ~45 k envs of 5 humans (19.2 vs 18.6 us per
    // step at 49 152 envs, 13.9 vs 15.3 at 32 768), i.e. ~14 k env groups; below that the fused launch wins by up to
    // 2.7x.  Two cooperating wavefronts per env group while the doubled grid still finds idle issue slots (measured:
    // wins up to 1536 groups = 4608 envs, loses from 1707).
    // mcn_set_tuning (rollout_fused / rollout_split) overrides (tests, tuning).
    const mcn_tuning tu = tuning();
    const int envs_per_wave = 64 / (4 * N) > 0 ? 64 / (4 * N) : 1;
    const long waves = ((long)E + envs_per_wave - 1) / envs_per_wave;
    const bool fused = tu.rollout_fused >= 0 ? tu.rollout_fused != 0 : waves <= 14000;
    const int step_split = p.quad_split;              // the single-step kernel's own choice, for the T-launch path
    p.quad_split = tu.rollout_split >= 0 ? tu.rollout_split : (waves <= 1536 ? 1 : 0);
    if (fused && !p.force_generic && mcn::launch_env_rollout_quad(p, T, (hipStream_t)stream))
        return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
    p.quad_split = step_split;
    // 6-10 ORCA humans, latency-bound batch: the one-wavefront step kernel run T times inside one launch
    // (env_step.hip: env_step_loop_kernel); rollout_fused = 0 keeps the T launches
    if (tu.rollout_fused != 0 && T > 1 && mcn::launch_env_step_loop(p, T, (hipStream_t)stream))
        return hipGetLastError() == hipSuccess ? MCN_OK : MCN_ELAUNCH;
    for (int32_t t = 0; t < T; ++t) {
        p.actions = actions + (size_t)t * E * 2;
        const int r = mcn::launch_env_step(p, (hipStream_t)stream);
        if (r != MCN_OK) return r;
    }
    return MCN_OK;
}

int mcn_scenario_pool(const mcn_scenario_cfg *cfg, uint64_t seed, int64_t first_case, int32_t P, int32_t N,
                      double *hpos, double *hgoal, double *hrad, double *hvpref, void *stream)
{
    if (!cfg || !hpos || !hgoal || !hrad || !hvpref) return MCN_EINVAL;
    if (P <= 0 || N <= 0 || N > MCN_MAX_HUMANS) return MCN_EINVAL;
    if (cfg->rule != MCN_RULE_CIRCLE && cfg->rule != MCN_RULE_SQUARE) return MCN_EINVAL;
    if (!(cfg->circle_radius > 0) || !(cfg->square_width > 0) || !(cfg->human_radius > 0)) return MCN_EINVAL;
    return mcn::launch_scenario_pool(*cfg, seed, first_case, P, N, hpos, hgoal, hrad, hvpref, (hipStream_t)stream);
}

int mcn_orca_batch(const float *self, const float *others, const int32_t *n_other, float *out,
                   int32_t B, int32_t M, float neighbor_dist, int32_t max_neighbors,
                   float time_horizon, float time_step, void *stream)
{
    if (!self || !n_other || !out || B <= 0 || M < 0 || M > MCN_MAX_HUMANS) return MCN_EINVAL;
    if (M > 0 && !others) return MCN_EINVAL;
    if (max_neighbors < 0 || max_neighbors > MCN_MAX_LINES) return MCN_EINVAL;
    if (!(time_horizon > 0) || !(time_step > 0)) return MCN_EINVAL;
    return mcn::launch_orca_batch(self, others, n_other, out, B, M, neighbor_dist, max_neighbors,
                                  time_horizon, time_step, (hipStream_t)stream);
}

int mcn_pack_linear(const float *weight, const float *bias, int32_t nout, int32_t kin,
                    const int32_t *kmap, int32_t KT, const int32_t *omap, int32_t NT,
                    float *wfrag_out, float *bfrag_out)
{
    if (!weight || !kmap || !wfrag_out || nout <= 0 || kin <= 0 || KT <= 0 || NT <= 0) return MCN_EINVAL;
    if (!omap && NT != (nout + 15) / 16) return MCN_EINVAL;
    auto orow = [&](int n, int slot) { return omap ? omap[n * 16 + slot] : 16 * n + slot; };
    for (int n = 0; n < NT; ++n)
        for (int slot = 0; slot < 16; ++slot)
            if (orow(n, slot) >= nout) { if (omap) return MCN_EINVAL; }
    for (int n = 0; n < NT; ++n)
        for (int t = 0; t < KT; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int row = orow(n, lane & 15);
                    const int col = kmap[t * 16 + 4 * (lane >> 4) + r];
                    float v = 0.0f;
                    if (row >= 0 && row < nout && col >= 0) {
                        if (col >= kin) return MCN_EINVAL;
                        v = weight[(size_t)row * kin + col];
                    }
                    wfrag_out[(((size_t)n * KT + t) * 64 + lane) * 4 + r] = v;
                }
    if (bfrag_out)
        for (int n = 0; n < NT; ++n)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int row = orow(n, 4 * (lane >> 4) + r);
                    bfrag_out[((size_t)n * 64 + lane) * 4 + r] = (bias && row >= 0 && row < nout) ? bias[row] : 0.0f;
                }
    return MCN_OK;
}

int64_t mcn_sarl_workspace_bytes(int32_t E, int32_t N, int32_t A)
{
    if (E <= 0 || N <= 0 || A <= 0) return 0;
    return (int64_t)mcn::sarl_workspace_float4s(E, N, A) * 16;
}

static int sarl_lookahead_impl(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int32_t A,
                               double time_step, double gamma_pow, int32_t kinematics, void *workspace,
                               double *values, int32_t *best, double *best_val, float *attention,
                               const double *next_hpos, const double *next_hvel, const double *rewards,
                               double *action_out, double epsilon, uint64_t seed, int32_t E, int32_t N, void *stream)
{
    if (!net || !st || !actions || !workspace || !values) return MCN_EINVAL;
    if (action_out && !best) return MCN_EINVAL;
    if (!(epsilon >= 0.0 && epsilon <= 1.0)) return MCN_EINVAL;
    if (E <= 0 || N <= 0 || N > MCN_MAX_HUMANS || A <= 0) return MCN_EINVAL;
    if (best && !best_val) return MCN_EINVAL;
    if (!st->hpos || !st->hvel || !st->hrad || !st->rpos || !st->rgoal || !st->rrad || !st->rvpref) return MCN_EINVAL;
    if (kinematics == MCN_KIN_UNICYCLE && !st->rtheta) return MCN_EINVAL;
    if ((next_hpos == nullptr) != (next_hvel == nullptr)) return MCN_EINVAL;
    const float *const *fp = reinterpret_cast<const float *const *>(net);
    for (size_t k = 0; k < sizeof(mcn_sarl_net) / sizeof(float *); ++k)
        if (!fp[k]) return MCN_EINVAL;
    if (!(time_step > 0)) return MCN_EINVAL;
    return mcn::launch_sarl_c(net, st, actions, A, time_step, gamma_pow, kinematics, workspace, values, best,
                              best_val, attention, next_hpos, next_hvel, rewards, action_out, epsilon,
                              (unsigned long long)seed, E, N, (hipStream_t)stream);
}

int mcn_sarl_lookahead(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int32_t A,
                       double time_step, double gamma_pow, int32_t kinematics, void *workspace,
                       double *values, int32_t *best, double *best_val, float *attention,
                       int32_t E, int32_t N, void *stream)
{
    return sarl_lookahead_impl(net, st, actions, A, time_step, gamma_pow, kinematics, workspace, values, best, best_val,
                               attention, nullptr, nullptr, nullptr, nullptr, 0.0, 0, E, N, stream);
}

int mcn_sarl_lookahead_env(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int32_t A,
                           double time_step, double gamma_pow, int32_t kinematics, void *workspace,
                           double *values, int32_t *best, double *best_val, float *attention,
                           const double *next_hpos, const double *next_hvel, const double *rewards,
                           int32_t E, int32_t N, void *stream)
{
    if (!next_hpos || !next_hvel || !rewards) return MCN_EINVAL;
    return sarl_lookahead_impl(net, st, actions, A, time_step, gamma_pow, kinematics, workspace, values, best, best_val,
                               attention, next_hpos, next_hvel, rewards, nullptr, 0.0, 0, E, N, stream);
}

int mcn_sarl_predict(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int32_t A,
                     double time_step, double gamma_pow, int32_t kinematics, void *workspace,
                     double *values, int32_t *best, double *best_val, float *attention,
                     const double *next_hpos, const double *next_hvel, const double *rewards,
                     double *action_out, double epsilon, uint64_t seed, int32_t E, int32_t N, void *stream)
{
    if (!best || !action_out) return MCN_EINVAL;
    if ((next_hpos == nullptr) != (rewards == nullptr)) return MCN_EINVAL;
    return sarl_lookahead_impl(net, st, actions, A, time_step, gamma_pow, kinematics, workspace, values, best, best_val,
                               attention, next_hpos, next_hvel, rewards, action_out, epsilon, seed, E, N, stream);
}

int mcn_mlp_world_step(const mcn_mlp_world_net *net, const double *hpos, const double *hvel, double *out_vel,
                       int32_t E, int32_t N, void *stream)
{
    if (!net || !hpos || !hvel || !out_vel || E <= 0 || N <= 0 || N > 10) return MCN_EINVAL;
    const float *const *fp = reinterpret_cast<const float *const *>(net);
    for (int k = 0; k < 8; ++k)
        if (!fp[k]) return MCN_EINVAL;
    return mcn::launch_mlp_world(net, hpos, hvel, out_vel, E, N, (hipStream_t)stream);
}

int64_t mcn_attn_world_workspace_bytes(int32_t E, int32_t N)
{
    if (E <= 0 || N <= 0) return 0;
    return (int64_t)mcn::attn_world_workspace_float4s(E, N) * 16;
}

int mcn_attn_world_step(const mcn_attn_world_net *net, const double *hpos, const double *hvel, const int32_t *hcount,
                        void *workspace, double *out_vel, int32_t E, int32_t N, void *stream)
{
    if (!net || !hpos || !hvel || !workspace || !out_vel || E <= 0 || N <= 0 || N > MCN_MAX_HUMANS) return MCN_EINVAL;
    const float *const *fp = reinterpret_cast<const float *const *>(net);
    for (size_t k = 0; k < sizeof(mcn_attn_world_net) / sizeof(float *); ++k)
        if (!fp[k]) return MCN_EINVAL;
    return mcn::launch_attn_world(net, hpos, hvel, hcount, workspace, out_vel, E, N, (hipStream_t)stream);
}

int64_t mcn_sgan_workspace_bytes(int32_t E, int32_t N)
{
    if (E <= 0 || N <= 0) return 0;
    return (int64_t)E * N * (32 + 4 + 8) * 4;     // encoder state, last position / displacement, pooled features
}

int mcn_sgan_step(const mcn_sgan_net *net, double *hist, int32_t push_slot, int32_t oldest, const double *cur_pos,
                  const float *noise, const int32_t *hcount, void *workspace, double *out_vel, float *out_rel,
                  double time_step, int32_t E, int32_t N, void *stream)
{
    if (!net || !hist || !noise || !workspace || !out_vel) return MCN_EINVAL;
    if (E <= 0 || N <= 0 || N > MCN_MAX_HUMANS) return MCN_EINVAL;
    if (push_slot < 0 || push_slot > 7 || oldest < 0 || oldest > 7 || !(time_step > 0)) return MCN_EINVAL;
    const float *const *fp = reinterpret_cast<const float *const *>(net);
    for (int k = 0; k < 14; ++k) {
        const bool pool_only = (k >= 2 && k < 6);
        if (!fp[k] && !(pool_only && !net->pooling)) return MCN_EINVAL;
    }
    return mcn::launch_sgan(net, hist, push_slot, oldest, cur_pos, noise, hcount, workspace, out_vel, out_rel,
                            time_step, E, N, (hipStream_t)stream);
}

#ifdef MCN_DIAG
// diagnostic build only: [workgroup][4] = s_memtime, s_memrealtime before / after the SGAN pool kernel's unit loop
int mcn_debug_pool_clock(void *dst_host, int64_t bytes) { return mcn::read_pool_clock(dst_host, (size_t)bytes); }
// diagnostic build only: [resident wavefront][16] shader cycles per phase of sarl_value_kernel (tools/sarl_phases.py)
int mcn_debug_sarl_phases(void *dst_host, int64_t bytes, int32_t reset) { return mcn::read_sarl_phases(dst_host, (size_t)bytes, reset); }
// diagnostic build only: copies the rollout kernel's time stamps to host memory, returns the number of 8-byte words
int mcn_debug_stamps(void *dst_host, int64_t bytes) { return mcn::read_stamps(dst_host, (size_t)bytes); }
// per-wavefront counts of the data-dependent paths taken: [wave][4] = 3-D LP, restarts, overlap sqrt, goal sqrt
int mcn_debug_counts(void *dst_host, int64_t bytes, int reset) { return mcn::read_counts(dst_host, (size_t)bytes, reset); }
#endif

}  // extern "C"
