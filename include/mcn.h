/*
 * mcn.h -- C ABI of libmcn_hip.so, the MI355X (gfx950) implementation of the
 * ModelCrowdNav data-parallel rollout hot path.
 *
 * The reference has no FFI of its own for this path: its boundary is three duck-typed
 * Python protocols (gym.Env, Policy, world-model callable) plus one native module, rvo2
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it
 * replaces.  All pointers are DEVICE pointers into caller-owned HBM buffers unless
 * marked host; every call is stream-ordered on `stream` (a hipStream_t passed as
 * void*), performs no allocation and no synchronisation, and returns 0 on success or a
 * negative MCN_E* code.  No torch types appear in any signature.
 *
 * Buffer conventions (E environments, N humans, row-major, env-major):
 *   double2-packed arrays are [E*N][2] or [E][2] doubles, 16-byte aligned.
 *   hpos  (px,py)   hvel (vx,vy)   hgoal (gx,gy)                              [E*N][2]
 *   hrad, hvpref (radius, v_pref: separate so kernels that need only the radius stream only it)  [E*N]
 *   rpos  (px,py)   rvel (vx,vy)   rgoal (gx,gy)                              [E][2]
 *   rrad, rvpref                                                              [E]
 *   rtheta, gtime                                                             [E]
 * The observation the reference returns from step()/reset() -- a list of
 * ObservableState(px,py,vx,vy,radius), crowd_sim/envs/utils/state.py:27-43 -- is
 * (hpos, hvel, hrad) and is therefore never copied.
 */
#ifndef MCN_H_
#define MCN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCN_OK            0
#define MCN_EINVAL      (-1)   /* bad shape / null pointer / unsupported N */
#define MCN_ELAUNCH     (-2)   /* hipLaunchKernel reported an error        */

#define MCN_MAX_HUMANS   32    /* humans per environment (one wavefront holds >= 2 envs) */
#define MCN_MAX_LINES    10    /* ORCA maxNeighbors ceiling handled on device            */

/* info codes: crowd_sim/envs/utils/info.py:1-38 */
enum { MCN_INFO_NOTHING = 0, MCN_INFO_DANGER = 1, MCN_INFO_REACHGOAL = 2, MCN_INFO_COLLISION = 3, MCN_INFO_TIMEOUT = 4 };
/* how humans choose their velocity */
enum { MCN_HUMANS_ORCA = 0,     /* crowd_sim/envs/policy/orca.py:82-132            */
       MCN_HUMANS_LINEAR = 1,   /* crowd_sim/envs/policy/linear.py:15-22           */
       MCN_HUMANS_GIVEN = 2 };  /* model_crowd_sim.py:347 step(new_v=...) / world model output */
enum { MCN_KIN_HOLONOMIC = 0, MCN_KIN_UNICYCLE = 1 };

/* Scalar configuration: env.config [env]/[reward] + orca.py:60-66 + robot flags. */
typedef struct mcn_env_cfg {
    double time_step;                  /* env.config:3   */
    double time_limit;                 /* env.config:2   */
    double success_reward;             /* env.config:10  */
    double collision_penalty;          /* env.config:11  */
    double discomfort_dist;            /* env.config:12  */
    double discomfort_penalty_factor;  /* env.config:13  */
    double orca_safety_space;          /* orca.py:59     */
    float  orca_neighbor_dist;         /* orca.py:60     */
    float  orca_time_horizon;          /* orca.py:62     */
    int32_t orca_max_neighbors;        /* orca.py:61 (<= MCN_MAX_LINES) */
    int32_t robot_visible;             /* humans see the robot, crowd_sim.py:340 */
    int32_t human_policy;              /* MCN_HUMANS_*   */
    int32_t robot_kinematics;          /* MCN_KIN_*      */
    int32_t count_hh;                  /* crowd_sim.py:368-376 (CrowdSim yes, ModelCrowdSim no) */
    int32_t track_human_times;         /* crowd_sim.py:418-421 */
} mcn_env_cfg;

/* Device state of E environments (all device pointers, caller-owned). */
typedef struct mcn_env_state {
    double *hpos, *hvel, *hgoal;               /* [E*N][2] */
    double *hrad, *hvpref;                     /* [E*N]    */
    double *rpos, *rvel, *rgoal;               /* [E][2]   */
    double *rrad, *rvpref;                     /* [E]      */
    double *rtheta;                            /* [E]      */
    double *gtime;                             /* [E]  CrowdSim.global_time */
    double *human_times;                       /* [E*N] or NULL */
    const int32_t *hcount;                     /* [E] or NULL: pedestrians the ROBOT'S POLICY sees, 1 <= hcount[e] <= N.
                                                * mcn_sarl_lookahead ignores slots i >= hcount[e] (attention sum, mean,
                                                * distance test); the env kernels step all N slots regardless */
} mcn_env_state;

/* What one env reports per step, as ONE 24-byte record: the kernel issues one store per env instead of five
 * scattered 1..8-byte ones (at a million envs the narrow per-env streams, not the bytes, limit the kernel). */
/* Alignment: 8 bytes (24-byte stride); the first 16 bytes are written with one 16-byte store, which gfx950 global
 * memory accepts at 8-byte alignment. */
typedef struct mcn_step_rec {
    double  reward;
    double  dmin;         /* min boundary distance robot-human over the step */
    uint8_t done;
    uint8_t info;         /* MCN_INFO_* */
    uint16_t reserved;
    int32_t hh_count;     /* human-human overlaps (the reference only logs them) */
} mcn_step_rec;

/* Per-step outputs (device pointers). */
typedef struct mcn_env_out {
    mcn_step_rec *rec;    /* [E] */
    double  *human_act;   /* [E*N][2] velocity each human chose, or NULL */
    double  *nobs_pos;    /* [E*N][2] next observable positions  (update == 0 only) */
    double  *nobs_vel;    /* [E*N][2] next observable velocities (update == 0 only) */
    void    *lp3_queue;   /* optional: mcn_env_lp3_queue_bytes(E, N) bytes of device memory, zero-filled ONCE by the caller
                           * and then owned by the env (the kernels keep its header consistent), or NULL.  With a queue,
                           * large ORCA batches of <= 10 humans solve RVO2's rare linearProgram3 in a second, dense launch
                           * (one parked problem per lane) instead of inside the step kernel, where one or two lanes of a
                           * wavefront would run it while the others wait.  Same results bit for bit.  The queue holds
                           * half of the worst case (0.9-3.7 % of the humans use it in a circle crossing, in bursts):
                           * humans beyond that are solved inside the step kernel as without a queue. */
} mcn_env_out;

/* Bytes of mcn_env_out.lp3_queue for E envs of N humans (0 when N is outside the deferred path's range):
 * 65 KiB of counters + 256 sub-queues x max(64, worst case / 2) entries x (28 + 16 N) bytes -- 264 MB for 2^18 x 10. */
int64_t mcn_env_lp3_queue_bytes(int32_t E, int32_t N);

/*
 * Optional fused bookkeeping of Explorer.run_k_episodes (crowd_nav/utils/explorer.py:54-125)
 * and vector-env style auto-reset.  Any pointer may be NULL to disable that part.
 */
/* Rollout state of one env, ONE 32-byte record (one load + one store per step). */
typedef struct mcn_roll_rec {
    double  ep_return;         /* running discounted sum */
    int32_t ep_steps;          /* steps taken in the running episode */
    int32_t fin_count;         /* episodes finished so far */
    int32_t next_case;         /* pool index used at the next reset, in [0, pool_size); advanced by case_stride mod pool_size */
    int32_t danger_count;      /* steps whose info was Danger ("too close", explorer.py:88-90) */
    double  danger_dist_sum;   /* sum of their min_dist */
} mcn_roll_rec;

typedef struct mcn_rollout {
    /* discounted return: sum_t disc_table[t] * r_t, disc_table[t] = pow(gamma, t*dt*v_pref) (explorer.py:124) */
    const double *disc_table;  int32_t disc_len;
    /* the "too close" statistics (danger_count / danger_dist_sum, explorer.py:88-90,138-141) only count steps of an env's
     * first `danger_episodes` episodes; 0 = every step (see danger_short_from below) */
    int32_t danger_episodes;
    mcn_roll_rec *state;       /* [E], or NULL: no return accounting */
    /* records of finished episodes: with fin_slots == 1 slot 0 holds the latest episode of env e; with
     * fin_slots > 1 episode number k < fin_slots of env e lands in slot k and later ones are not recorded */
    double  *fin_return;       /* [fin_slots][E] */
    double  *fin_time;         /* [fin_slots][E] env.global_time at the end (explorer.py:95,99) */
    uint8_t *fin_info;         /* [fin_slots][E] */
    int32_t  fin_slots;        /* >= 1 */
    /* k episodes over E envs leave the last round partial: when danger_short_from > 0, envs with index >=
     * danger_short_from - 1 count one episode fewer than danger_episodes; 0 = all envs count danger_episodes */
    int32_t  danger_short_from;
    /* auto-reset from a pool of host-generated scenarios (bit-exact CrowdSim.reset output) */
    const double *pool_hpos, *pool_hgoal;                /* [P*N][2] */
    const double *pool_hrad, *pool_hvpref;               /* [P*N]    */
    const double *pool_hvel;                             /* [P*N][2] or NULL (zeros) */
    int32_t pool_size;
    int32_t case_stride;       /* in [0, pool_size); pool resets need `state` (its next_case field) */
    double  robot_start[2], robot_goal[2], robot_theta0;  /* crowd_sim.py:284 */
} mcn_rollout;

/*
 * mcn_env_step -- one batched CrowdSim.step / ModelCrowdSim.step.
 * Replaces: crowd_sim/envs/crowd_sim.py:331-434 (step, update=True/False),
 *           crowd_sim/envs/crowd_sim.py:325-329 (onestep_lookahead),
 *           crowd_sim/envs/model_crowd_sim.py:347-441,
 *           and the per-human rvo2 round trip of crowd_sim/envs/policy/orca.py:82-132.
 * actions: [E][2] (vx,vy) holonomic or (v,r) unicycle.  given_v: [E*N][2] or NULL.
 * update != 0 advances state in place; update == 0 leaves state untouched and fills nobs_*.
 * roll may be NULL.
 */
int mcn_env_step(const mcn_env_cfg *cfg, const mcn_env_state *st, const double *actions,
                 const double *given_v, const mcn_env_out *out, const mcn_rollout *roll,
                 int32_t E, int32_t N, int32_t update, void *stream);

/*
 * mcn_env_rollout -- T consecutive mcn_env_step(update = 1) calls with the action sequence actions[T][E][2],
 * same results bit for bit.  Replaces the step loop of Explorer.run_k_episodes (crowd_nav/utils/explorer.py:69-99)
 * for robots whose actions do not depend on the observation (random / scripted sequences).  Where the batch is
 * small enough to be latency-bound the T steps run in ONE launch: with <= 4 ORCA neighbours per human the env state
 * stays in registers (env_rollout_quad.hip), with 6-10 ORCA humans the one-wavefront step kernel runs T times and the
 * state goes through L2 (env_step.hip: env_step_loop_kernel); otherwise this is T launches.  On return `st` is the state after step T,
 * `out->rec` / `out->human_act` are those of step T, `roll` has accounted for all T steps (episodes that end
 * inside the sequence are recorded and, with a pool, restarted in-kernel).  Humans: ORCA or linear (no given_v).
 */
int mcn_env_rollout(const mcn_env_cfg *cfg, const mcn_env_state *st, const double *actions, int32_t T,
                    const mcn_env_out *out, const mcn_rollout *roll, int32_t E, int32_t N, void *stream);

/* Scenario rules of CrowdSim.reset (crowd_sim.py:120-163). */
enum { MCN_RULE_CIRCLE = 0, MCN_RULE_SQUARE = 1 };

typedef struct mcn_scenario_cfg {
    double circle_radius, square_width;      /* env.config [sim] */
    double discomfort_dist;                  /* env.config [reward] */
    double human_radius, human_v_pref;       /* env.config [humans] (used when randomize_attributes == 0) */
    double robot_radius;
    double robot_start[2], robot_goal[2];    /* crowd_sim.py:284 */
    int32_t rule;                            /* MCN_RULE_* */
    int32_t randomize_attributes;            /* agent.py:39-45 */
} mcn_scenario_cfg;

/*
 * mcn_scenario_pool -- P crowd scenarios of N humans generated ON the device into the pool arrays that
 * mcn_rollout's in-kernel restart reads (pool_hpos / pool_hgoal [P*N][2], pool_hrad / pool_hvpref [P*N]).
 * Replaces, for rollouts that do not need bit-parity with the reference's random stream,
 * generate_random_human_position / generate_circle_crossing_human / generate_square_crossing_human
 * (crowd_sim/envs/crowd_sim.py:94-215): same placement rules and rejection tests, counter-based random numbers
 * keyed by (seed, first_case + i), so case i is the same whatever P or the partition.  The bit-exact generator
 * (numpy MT19937, host) stays in modelcrowdnav_amd/envs/scenarios.py.
 */
int mcn_scenario_pool(const mcn_scenario_cfg *cfg, uint64_t seed, int64_t first_case, int32_t P, int32_t N,
                      double *hpos, double *hgoal, double *hrad, double *hvpref, void *stream);

/*
 * mcn_orca_batch -- ORCA velocity for B independent agents, each with up to M candidate
 * neighbours (float32, RVO2 semantics).  Replaces the rvo2.PyRVOSimulator
 * addAgent / setAgent... / doStep / getAgentVelocity(0) sequence of orca.py:95-129 for arbitrary agents
 * (used for an ORCA-driven robot, test.py:52 --policy orca).
 * self:   [B][8] float  (px,py,vx,vy,radius,max_speed,pref_x,pref_y)
 * others: [B][M][5] float (px,py,vx,vy,radius); n_other: [B] int32 valid counts
 * out:    [B][2] float new velocity
 */
int mcn_orca_batch(const float *self, const float *others, const int32_t *n_other, float *out,
                   int32_t B, int32_t M, float neighbor_dist, int32_t max_neighbors,
                   float time_horizon, float time_step, void *stream);


/* ------------------------------------------------------------------------------------------------
 * SARL attention value network: 81-action one-step look-ahead (crowd_nav/policy/sarl.py:28-65,
 * multi_human_rl.py:11-63, cadrl.py:104-129,217-252).
 * ---------------------------------------------------------------------------------------------- */

/*
 * mcn_pack_linear -- HOST helper: permute one nn.Linear (weight [nout][kin] row-major, bias [nout])
 * into the MFMA operand order the network kernels stream (see mfma_chain.hpp; used for SARL and SGAN).  kmap[t*16 + s] is the column of
 * `weight` that input slot s of input tile t carries, or -1 for padding; KT input tiles.  omap[n*16 + s] is the
 * row of `weight` produced in output slot s of output tile n, or -1 (NULL = identity: row 16n + s); NT output
 * tiles.  Slot s = 4q + r lives in accumulator register r of lane group q, so a ragged last tile packed
 * "q first" (feature j at slot 4(j%4) + j/4) needs only ceil(w/4) k-steps in the layer that consumes it.
 * wfrag_out: [NT][KT][64][4] floats, bfrag_out: [NT][64][4] floats (may be NULL).
 */
int mcn_pack_linear(const float *weight, const float *bias, int32_t nout, int32_t kin,
                    const int32_t *kmap, int32_t KT, const int32_t *omap, int32_t NT,
                    float *wfrag_out, float *bfrag_out);

/*
 * mcn_pack_x3 -- HOST helper: regroup float32 weight fragments ([NT][KT][64][4] floats, mcn_pack_linear's wfrag_out)
 * for the bf16 matrix pipe: every weight as three bfloat16 pieces hi + mid + lo (round-to-nearest-even, exact to
 * 2^-24), per (output tile, 32-feature input block) three 16-byte rows per lane:
 * x3_out [NT][ceil(KT / 2)][3][64][8] bfloat16 (mcn_pack_x3_bytes bytes).  With these the look-ahead kernels run
 * v_mfma_f32_16x16x32_bf16 on six of the nine piece products instead of v_mfma_f32_16x16x4_f32: float32-accurate
 * (error against float64 as the float32 chain's), ~2.7 x fewer matrix-pipe cycles.
 */
int64_t mcn_pack_x3_bytes(int32_t NT, int32_t KT);
int mcn_pack_x3(const float *wfrag, int32_t NT, int32_t KT, void *x3_out);

/* Device pointers to the bf16x3 fragments of one ValueNetwork (biases stay the float32 fragments of mcn_sarl_net). */
typedef struct mcn_sarl_x3 {
    const void *w_m1a, *w_m1b, *w_m2a, *w_m2b, *w_ata, *w_atg, *w_atb, *w_atc, *w_m3a, *w_m3b, *w_m3c, *w_m3d;
} mcn_sarl_x3;

/* Device pointers to the packed fragments of one ValueNetwork (state_dict keys in comments). */
typedef struct mcn_sarl_net {
    const float *w_m1a, *b_m1a;   /* mlp1.0        13 -> 150 */
    const float *w_m1b, *b_m1b;   /* mlp1.2       150 -> 100 */
    const float *w_m2a, *b_m2a;   /* mlp2.0       100 -> 100 */
    const float *w_m2b, *b_m2b;   /* mlp2.2       100 -> 50  */
    const float *w_ata, *b_ata;   /* attention.0  columns   0..99  (per-human half) + bias */
    const float *w_atg;           /* attention.0  columns 100..199 (global-state half)     */
    const float *w_atb, *b_atb;   /* attention.2  100 -> 100 */
    const float *w_atc, *b_atc;   /* attention.4  100 -> 1   */
    const float *w_m3a, *b_m3a;   /* mlp3.0        56 -> 150 (input tiles: pooled x4, self x1) */
    const float *w_m3b, *b_m3b;   /* mlp3.2       150 -> 100 */
    const float *w_m3c, *b_m3c;   /* mlp3.4       100 -> 100 */
    const float *w_m3d, *b_m3d;   /* mlp3.6       100 -> 1   */
    const mcn_sarl_x3 *x3;        /* HOST pointer or NULL: when set, the layers run on the bf16 pipe from these fragments
                                   * (ABI 5; the float32 fragments above are still required: biases, and the fallback) */
} mcn_sarl_net;

/* Bytes of device workspace mcn_sarl_lookahead needs for (E, N, A): one slot per RESIDENT wavefront of the persistent
 * grid (at most 2 048), N x 12 KiB each (the bf16x3 path parks mlp1's output already split; the float32 path uses
 * 7 KiB of it) -- 123 MB at N = 5, whatever E. */
int64_t mcn_sarl_workspace_bytes(int32_t E, int32_t N, int32_t A);

/*
 * mcn_sarl_lookahead -- for every env and every candidate action: propagate, reward, rotate, value
 * network, value = reward + gamma_pow * V; then the first-max-wins argmax.
 * Replaces the `for action in self.action_space` loop of MultiHumanRL.predict (multi_human_rl.py:35-55).
 * actions: [A][2] device; values: [E][A] device out; best: [E] int32 out (-1 = robot already at its goal,
 * multi_human_rl.py:22; may be NULL); best_val: [E]; attention: [E][A][N] float out or NULL.
 * gamma_pow = pow(gamma, time_step * v_pref) computed by the caller (multi_human_rl.py:52).
 */
int mcn_sarl_lookahead(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int32_t A,
                       double time_step, double gamma_pow, int32_t kinematics, void *workspace,
                       double *values, int32_t *best, double *best_val, float *attention,
                       int32_t E, int32_t N, void *stream);

/*
 * mcn_sarl_lookahead_env -- the same look-ahead with `query_env = true` (multi_human_rl.py:37-38, cadrl.py:158-159):
 * the humans' next states and the per-action rewards are what the env's own one-step look-ahead returned
 * (CrowdSim.onestep_lookahead, crowd_sim.py:325-329) instead of constant-velocity propagation + compute_reward.
 * next_hpos / next_hvel: [E*N][2] -- out->nobs_pos / nobs_vel of one mcn_env_step(update = 0) (the humans' reaction
 * does not depend on the candidate action: they see the robot's current state, crowd_sim.py:336-342);
 * rewards: [E][A] -- reward of mcn_env_step for every (env, action).  Everything else as mcn_sarl_lookahead.
 */
int mcn_sarl_lookahead_env(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int32_t A,
                           double time_step, double gamma_pow, int32_t kinematics, void *workspace,
                           double *values, int32_t *best, double *best_val, float *attention,
                           const double *next_hpos, const double *next_hvel, const double *rewards,
                           int32_t E, int32_t N, void *stream);

/*
 * mcn_sarl_predict -- the look-ahead AND the action MultiHumanRL.predict returns (multi_human_rl.py:22-23,53-63), for
 * every env, with no host round trip: action_out[e] = actions[best[e]], or (0, 0) where best[e] == -1 (robot on its
 * goal) or every value is NaN (best[e] < 0 with best_val = -inf: "Value network is not well trained", the caller's
 * error).  next_hpos / next_hvel / rewards: all NULL (mcn_sarl_lookahead's propagate + compute_reward) or all set
 * (mcn_sarl_lookahead_env's `query_env` form).  action_out: [E][2] device out.  Everything else as above.
 * epsilon > 0 (phase 'train', multi_human_rl.py:27-29): each env not on its goal independently takes, with
 * probability epsilon, a uniformly drawn row of `actions` instead of the best one (best[e] = -2); the draws are a
 * counter-based function of (seed, env), so pass a fresh seed per call.  epsilon = 0: greedy, seed ignored.
 */
int mcn_sarl_predict(const mcn_sarl_net *net, const mcn_env_state *st, const double *actions, int32_t A,
                     double time_step, double gamma_pow, int32_t kinematics, void *workspace,
                     double *values, int32_t *best, double *best_val, float *attention,
                     const double *next_hpos, const double *next_hvel, const double *rewards,
                     double *action_out, double epsilon, uint64_t seed, int32_t E, int32_t N, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Social-GAN one-step world model (crowd_nav/policy/world_model.py:134-268, sgan/models.py:501-553).
 * ---------------------------------------------------------------------------------------------- */

/* Device pointers to packed fragments of one TrajectoryGenerator (shipped architecture: embedding 16,
 * encoder/decoder hidden 32, mlp 64, bottleneck 8, noise 8 'global', pool hidden 512, no batch norm).
 * The three spatial_embedding layers (Linear(2, 16), models.py:47,120,186) feed a Linear / LSTM input directly, so the
 * packer folds them into it (in float64, rounded once): a layer "W [emb | h] , b" arrives here as
 * "[W_emb W_se | W_h] , b + W_emb b_se" with 2 + 32 input columns.  Input tile 0 carries the displacement: x in slot
 * 0, y in slot 4 (both in MFMA k-step 0), tiles 1-2 the 32 hidden features. */
typedef struct mcn_sgan_net {
    const float *w_elstm, *b_elstm;   /* encoder: [weight_ih_l0 W_se | weight_hh_l0], bias_ih + bias_hh + weight_ih b_se */
    const float *w_p1, *b_p1;         /* pool_net.mlp_pre_pool.0 folded with pool_net.spatial_embedding (pooling only) */
    const float *w_p2, *b_p2;         /* pool_net.mlp_pre_pool.2                                         (pooling only) */
    const float *w_c1, *b_c1;         /* mlp_decoder_context.0 */
    const float *w_c2, *b_c2;         /* mlp_decoder_context.2 */
    const float *w_dlstm, *b_dlstm;   /* decoder: folded like the encoder's */
    const float *w_h2p, *b_h2p;       /* decoder.hidden2pos */
    int32_t pooling;                  /* 1: pooling_type == 'pool_net', 0: none */
} mcn_sgan_net;

int64_t mcn_sgan_workspace_bytes(int32_t E, int32_t N);

/*
 * mcn_sgan_step -- one SGANWorld.forward for E scenes of N pedestrians.
 * hist: [E][8][N][2] float64 ring of positions already rounded to 1e-4.  If cur_pos ([E*N][2]) is not NULL it
 * is rounded and written to ring slot `push_slot` first (the frame the reference appends to its cache file,
 * world_model.py:238-240).  `oldest` is the ring slot of the oldest of the 8 frames AFTER that push.
 * noise: [E][8] float (user_noise, sgan/models.py:475); out_vel: [E*N][2] float64 velocities
 * (world_model.py:266-268); out_rel: [E*N][2] float predicted displacement or NULL.
 * hcount: [E] or NULL -- pedestrians present in scene e (1 <= hcount[e] <= N): the pooling module looks only at
 * partners k < hcount[e] (a scene of hcount[e] pedestrians in the reference); outputs of slots >= hcount[e] are
 * meaningless.
 */
int mcn_sgan_step(const mcn_sgan_net *net, double *hist, int32_t push_slot, int32_t oldest, const double *cur_pos,
                  const float *noise, const int32_t *hcount, void *workspace, double *out_vel, float *out_rel,
                  double time_step, int32_t E, int32_t N, void *stream);

/* ------------------------------------------------------------------------------------------------
 * MlpWorld one-step world model (crowd_nav/policy/world_model.py:22-42).
 * ---------------------------------------------------------------------------------------------- */

/* Device pointers to the four Linear layers of MlpWorld.mlp (state_dict keys mlp.0 / mlp.3 / mlp.6 / mlp.8), packed by
 * mcn_pack_linear: mlp.0 with KT = ceil(4N / 16) natural input tiles and 8 output tiles; mlp.3 8 -> 4 tiles; mlp.6
 * 4 -> 1 tile with its 12 outputs packed "q first" (omap: feature j at slot 4 (j % 4) + j / 4); mlp.8 with the matching
 * kmap, 1 input tile, ceil(2N / 16) natural output tiles. */
typedef struct mcn_mlp_world_net {
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
} mcn_mlp_world_net;

/*
 * mcn_mlp_world_step -- MlpWorld.forward (eval mode: dropout is the identity) for E scenes of N <= 10 pedestrians:
 * input row = the scene's [px, py, vx, vy] x N as float32 (model_crowd_sim.py:401-405), output = the N predicted
 * velocities (tanh of mlp.8's output, world_model.py:33-35) as float64 [E*N][2] -- what ModelCrowdSim.step hands to
 * humans[i].step(ActionXY(new_v[i])) (model_crowd_sim.py:406-417), i.e. mcn_env_step's given_v.
 */
int mcn_mlp_world_step(const mcn_mlp_world_net *net, const double *hpos, const double *hvel, double *out_vel,
                       int32_t E, int32_t N, void *stream);

/* ------------------------------------------------------------------------------------------------
 * AttentionWorld one-step world model (crowd_nav/policy/world_model.py:54-106).
 * ---------------------------------------------------------------------------------------------- */

/* Device pointers to the packed fragments of one AttentionWorld (state_dict keys in comments; ragged tiles packed
 * "q first" as for mcn_sarl_net).  The 4-wide pedestrian state [px, py, vx, vy] is one input tile with feature c in
 * slot 4 c (one MFMA k-step). */
typedef struct mcn_attn_world_net {
    const float *w_m1a, *b_m1a;   /* mlp1.0         4 -> 150 */
    const float *w_m1b, *b_m1b;   /* mlp1.2       150 -> 100 */
    const float *w_m2a, *b_m2a;   /* mlp2.0       100 -> 100 */
    const float *w_m2b, *b_m2b;   /* mlp2.2       100 -> 50  */
    const float *w_ata, *b_ata;   /* attention.0  columns   0..99  (per-pedestrian half) + bias */
    const float *w_atg;           /* attention.0  columns 100..199 (global-state half)          */
    const float *w_atb, *b_atb;   /* attention.2  100 -> 100 */
    const float *w_atc, *b_atc;   /* attention.4  100 -> 1   */
    const float *w_m3p, *b_m3p;   /* mlp3.0       columns 4..53 (pooled feature) + bias: 50 -> 150 */
    const float *w_m3s;           /* mlp3.0       columns 0..3  (the pedestrian's own state): 4 -> 150 */
    const float *w_m3b, *b_m3b;   /* mlp3.2       150 -> 100 */
    const float *w_m3c, *b_m3c;   /* mlp3.4       100 -> 100 */
    const float *w_m3d, *b_m3d;   /* mlp3.6       100 -> 2 (natural row order) */
} mcn_attn_world_net;

int64_t mcn_attn_world_workspace_bytes(int32_t E, int32_t N);

/*
 * mcn_attn_world_step -- AttentionWorld.forward for E scenes of N pedestrians: input = each pedestrian's
 * [px, py, vx, vy] as float32 (model_crowd_sim.py:401-405), output = its predicted velocity (mlp3's two outputs, no
 * output non-linearity, world_model.py:104-105) as float64 [E*N][2].  hcount ([E] int32 or NULL): scene e has only
 * its first hcount[e] pedestrians (the rest of the N slots is ignored and not written).
 */
int mcn_attn_world_step(const mcn_attn_world_net *net, const double *hpos, const double *hvel, const int32_t *hcount,
                        void *workspace, double *out_vel, int32_t E, int32_t N, void *stream);

/*
 * Dispatch overrides (host, process-wide, not stream-ordered; for tests and tuning).  Every env-step arithmetic
 * exists in several kernel decompositions with bit-identical results; by default the entry points pick one from
 * the batch shape.  -1 = automatic.  The MCN_FORCE_GENERIC / MCN_QUAD_MAX_ENVS / MCN_QUAD_SPLIT /
 * MCN_ROLLOUT_FUSED / MCN_ROLLOUT_SPLIT / MCN_PAIR_STREAM / MCN_STEP_BLOCK / MCN_LP3_DEFER / MCN_SARL_X3 environment variables give the initial values and are read once, at
 * the first launch; no entry point calls getenv after that.
 */
typedef struct mcn_tuning {
    int32_t force_generic;   /* 1: run-time-N lane-per-human kernel even where a compile-time-N form exists */
    int32_t quad_max_envs;   /* mcn_env_step: lanes-per-neighbour ("quad") kernel up to this batch size (0 = never) */
    int32_t quad_split;      /* quad kernel: ORCA and float64 pairwise work on two cooperating wavefronts (0/1) */
    int32_t rollout_fused;   /* mcn_env_rollout: one T-step launch (1) or T single-step launches (0) */
    int32_t rollout_split;   /* fused rollout: two cooperating wavefronts per env group (0/1) */
    int32_t step_block;      /* lane-per-human step kernels: workgroup of 64 or 256 lanes (-1: 64 up to 4096 wavefronts, 256 above) */
    int32_t diag_noop;       /* DIAGNOSTIC build only (make stamp): env kernels return at entry; MCN_EINVAL otherwise */
    int32_t pair_stream;     /* given-velocity step: streaming kernel (env_pair.hip) 1 wherever it applies / 0 never;
                              * 2 / 3: wherever it applies, with non-temporal per-human streams forced on / off */
    int32_t lp3_defer;       /* lane-per-human ORCA kernels with mcn_env_out.lp3_queue set: park the 3-D LPs for a second,
                              * dense launch (1) or solve them in the step kernel (0); -1: defer from 8 ORCA neighbours and 16 384 wavefronts */
    int32_t sarl_x3;         /* SARL look-ahead with mcn_sarl_net.x3 set: bf16x3 layers (1 / -1) or the float32 MFMA layers (0) */
} mcn_tuning;

/* NULL restores the initial values.  Returns MCN_EINVAL for out-of-range fields.  The settings are one process-wide
 * struct; every launch takes a consistent snapshot of it under a lock, so mcn_set_tuning may be called while other
 * threads launch (they see the old or the new settings, never a mixture). */
int mcn_set_tuning(const mcn_tuning *t);
int mcn_get_tuning(mcn_tuning *t);

/* Library self-description (host). */
const char *mcn_version(void);

/*
 * ABI guard.  MCN_ABI_VERSION changes whenever a struct of this header changes size or layout or an entry point
 * changes its signature (0.3 grew mcn_tuning by lp3_defer and mcn_env_out by lp3_queue: ABI 4; 0.4 grew mcn_sarl_net by
 * x3: ABI 5).  A caller built against
 * another header must not pass structs to this library: compare mcn_abi_version() with the MCN_ABI_VERSION it was
 * compiled with, and (bindings without the header: ctypes, cgo) mcn_sizeof() with the size of its own struct mirrors.
 */
#define MCN_ABI_VERSION 5
int32_t mcn_abi_version(void);
enum { MCN_SIZEOF_ENV_CFG = 0, MCN_SIZEOF_ENV_STATE = 1, MCN_SIZEOF_ENV_OUT = 2, MCN_SIZEOF_ROLLOUT = 3,
       MCN_SIZEOF_TUNING = 4, MCN_SIZEOF_STEP_REC = 5, MCN_SIZEOF_ROLL_REC = 6, MCN_SIZEOF_SARL_NET = 7,
       MCN_SIZEOF_SGAN_NET = 8, MCN_SIZEOF_SCENARIO_CFG = 9, MCN_SIZEOF_MLP_WORLD_NET = 10, MCN_SIZEOF_ATTN_WORLD_NET = 11,
       MCN_SIZEOF_SARL_X3 = 12 };
int64_t mcn_sizeof(int32_t which);                 /* sizeof the struct named by MCN_SIZEOF_*; -1 for an unknown id */

/*
 * Diagnostic: the kernel family ("env_step_quad_kernel", "env_rollout_quad_kernel", "env_step_loop_kernel",
 * "env_pair_kernel", "env_step_kernel") that the calling thread's latest mcn_env_step / mcn_env_rollout call
 * dispatched to; "" before the first call.  bench.py attributes profiles to the kernel that really ran with it.
 */
const char *mcn_last_dispatch(void);

#ifdef __cplusplus
}
#endif
#endif /* MCN_H_ */
