"""Policy half of __graft_entry__.smoke(): one tiny SARL look-ahead + SGAN step on cuda:0 vs the oracle."""
import os

import numpy as np
import torch


def run():
    from oracle import pyref
    from tests import helpers as H
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.policy.sarl import SARL
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pol = SARL(); pol.configure(configs.policy_config()); pol.kinematics = "holonomic"
    pol.set_device(dev); pol.set_phase("test"); pol.time_step = 0.25
    E, N = 8, 5
    env = H.make_vec_env(E, N)
    env.reset("test", test_cases=list(range(E)))
    st = H.download(env)
    actions, best, values = pol.predict_batch(env, want_values=True)
    w = {k: v.detach().cpu() for k, v in pol.model.state_dict().items()}
    for e in (0, E - 1):
        row = [st.rpx[e], st.rpy[e], st.rvx[e], st.rvy[e], st.rr[e], st.rgx[e], st.rgy[e], 1.0, 0.0]
        hum = np.stack([st.hpx[e], st.hpy[e], st.hvx[e], st.hvy[e], st.hr[e]], 1)
        ref, idx = pyref.sarl_predict(w, row, hum, pol._action_table)
        assert np.abs(values[e].cpu().numpy() - ref).max() < 1e-5
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "g6_sgan.npz")
    if os.path.exists(gold):
        from modelcrowdnav_amd.policy.world_model import generator_from_arrays
        g = np.load(gold)
        gen = generator_from_arrays(g, "p", dev)
        key = "p__S6_N5__"
        sse = torch.tensor([[i * 5, (i + 1) * 5] for i in range(6)])
        pr = gen(torch.from_numpy(g[key + "obs_traj"]).to(dev), torch.from_numpy(g[key + "obs_rel"]).to(dev), sse,
                 user_noise=torch.from_numpy(g[key + "noise"]).to(dev))
        assert np.abs(pr.cpu().numpy() - g[key + "pred_rel"]).max() < 1e-5
    print("policy smoke ok: SARL look-ahead and SGAN step match the oracle / reference fixtures")
