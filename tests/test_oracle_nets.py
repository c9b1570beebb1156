"""CPU: pin the torch-fp32 restatements (oracle/pyref.py) against fixtures recorded from the real
reference's SARL and SGAN code (tests/golden_tools/gen_golden_nets.py)."""
import os

import numpy as np
import torch

from oracle import pyref


def _weights(g, prefix):
    return {k[len(prefix):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def test_rotate(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_sarl.npz"))
    for kin in ("holonomic", "unicycle"):
        y = pyref.rotate(torch.from_numpy(g["rotate_in_" + kin]), kin).numpy()
        np.testing.assert_allclose(y, g["rotate_out_" + kin], rtol=0, atol=2e-6)


def test_value_network(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_sarl.npz"))
    for seed in (0, 1):
        w = _weights(g, "w%d__" % seed)
        assert sum(v.numel() for v in w.values()) == 96502
        for N in (5, 10, 1):
            v, a = pyref.sarl_forward(w, torch.from_numpy(g["vn%d_in_N%d" % (seed, N)]))
            np.testing.assert_allclose(v.numpy(), g["vn%d_out_N%d" % (seed, N)][:, 0], rtol=0, atol=2e-6)
            np.testing.assert_allclose(a.numpy(), g["vn%d_att_N%d" % (seed, N)], rtol=0, atol=2e-6)


def test_predict_values_and_argmax(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_sarl.npz"))
    for seed in (0, 1):
        w = _weights(g, "w%d__" % seed)
        for N in (5, 10):
            key = "pred%d_N%d_" % (seed, N)
            table = g[key + "table"]
            for s in range(0, g[key + "self"].shape[0], 3):
                want = g[key + "values"][s]
                if np.isnan(want[0]):
                    continue          # reach_destination short-circuit (multi_human_rl.py:22)
                vals, idx = pyref.sarl_predict(w, g[key + "self"][s], g[key + "humans"][s], table)
                np.testing.assert_allclose(vals, want, rtol=0, atol=5e-6)
                assert tuple(table[idx]) == tuple(g[key + "action"][s])


def test_predict_unicycle_values_and_argmax(golden_dir):
    """g17_sarl_unicycle.npz = the reference's MultiHumanRL.predict with kinematics 'unicycle': (v, r) table
    (cadrl.py:97-99), heading-dependent propagate (:118-124), theta kept as a feature (:236-237)."""
    g = np.load(os.path.join(golden_dir, "g17_sarl_unicycle.npz"))
    w = _weights(g, "w__")
    table = g["table"]
    assert table.shape == (81, 2) and table[1, 1] == -np.pi / 4
    seen = 0
    for N in (5, 10, 2):
        for s in range(g["N%d_self" % N].shape[0]):
            want = g["N%d_values" % N][s]
            if np.isnan(want[0]):
                assert tuple(g["N%d_action" % N][s]) == (0, 0)        # reach_destination short-circuit: ActionRot(0, 0)
                continue
            vals, idx = pyref.sarl_predict(w, g["N%d_self" % N][s], g["N%d_humans" % N][s], table, kinematics="unicycle")
            np.testing.assert_allclose(vals, want, rtol=0, atol=5e-6)
            top2 = np.sort(want)[-2:]
            if top2[1] - top2[0] > 1e-5:
                assert tuple(table[idx]) == tuple(g["N%d_action" % N][s]), (N, s)
            seen += 1
    assert seen > 90


def test_sgan_generator(golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_sgan.npz"))
    for tag in ("np", "p"):
        w = _weights(g, tag + "__w__")
        pooling = str(g[tag + "__pooling_type"]) == "pool_net"
        assert sum(v.numel() for v in w.values()) == (46386 if pooling else 16634)
        for S, N in ((6, 5), (3, 10), (4, 1)):
            key = "%s__S%d_N%d__" % (tag, S, N)
            pr = pyref.sgan_generator(w, torch.from_numpy(g[key + "obs_traj"]), torch.from_numpy(g[key + "obs_rel"]),
                                      N, torch.from_numpy(g[key + "noise"]), pooling)
            np.testing.assert_allclose(pr.numpy(), g[key + "pred_rel"][0], rtol=0, atol=2e-6)
