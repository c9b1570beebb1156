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
};

}  // namespace mcn
