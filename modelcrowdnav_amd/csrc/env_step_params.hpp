// env_step_params.hpp -- kernel argument block of env_step_kernel (passed by value).
#pragma once
#include <stdint.h>
#include "../../include/mcn.h"

namespace mcn {

struct StepParams {
    mcn_env_cfg cfg;
    mcn_env_state st;
    mcn_env_out out;
    mcn_rollout roll;
    const double *actions;
    const double *given_v;
    int E, N, G, update, has_roll;
    int nl_cap;   // LDS line slots per lane
    int quad_max_envs;   // use the quad-parallel kernel up to this batch size (0: never)
    int force_generic;   // mcn_tuning.force_generic: run-time-N kernel even where a specialisation exists (tests, tuning)
    int debug_noop;      // mcn_tuning.diag_noop, DIAGNOSTIC BUILD ONLY: kernels return at entry (launch-floor measurement)
    int quad_split;      // quad kernel: ORCA and pairwise work on two cooperating wavefronts
    int step_block;      // mcn_tuning.step_block: 64 / 256 lanes per workgroup in the lane-per-human kernels, -1 automatic
    int lp3_defer;       // mcn_tuning.lp3_defer (-1 automatic); launch_env_step turns it into 0 / 1 for the kernel
    int pair_stream;     // given-velocity step: streaming kernel of env_pair.hip (-1 automatic, 0 never, 1 where it applies)
};

}  // namespace mcn
