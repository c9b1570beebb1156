"""ORACLE -- test infrastructure only.  numpy / torch-fp32 CPU restatements of the Python-side
arithmetic of the rollout path.  Pinned against fixtures recorded from the real reference
(tests/golden/g5_sarl.npz, g6_sgan.npz, g7_episode.npz; tests/test_oracle_nets.py).

  rotate()            crowd_nav/policy/cadrl.py:217-252
  sarl_forward()      crowd_nav/policy/sarl.py:28-65 (+ mlp, cadrl.py:11-19)
  sarl_predict()      crowd_nav/policy/multi_human_rl.py:11-63 (query_env = false path)
  sgan_generator()    sgan/models.py:501-553 with Encoder :28-71, PoolHiddenNet :167-232,
                      add_noise :454-490, Decoder.forward :127-164 (pred_len 1)
  sgan_velocities()   crowd_nav/policy/world_model.py:252-268
"""
import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------- SARL
def rotate(x, kinematics="holonomic"):
    """[B,14] float32 joint rows -> [B,13] agent-centric features."""
    px, py, vx, vy, r, gx, gy, vpref, theta = [x[:, k] for k in range(9)]
    hx, hy, hvx, hvy, hr = [x[:, k] for k in range(9, 14)]
    dx, dy = gx - px, gy - py
    rot = torch.atan2(dy, dx)
    c, s = torch.cos(rot), torch.sin(rot)
    dg = torch.norm(torch.stack([dx, dy], 1), 2, dim=1)
    th = (theta - rot) if kinematics == "unicycle" else torch.zeros_like(vpref)
    da = torch.norm(torch.stack([px - hx, py - hy], 1), 2, dim=1)
    cols = [dg, vpref, th, r, vx * c + vy * s, vy * c - vx * s,
            (hx - px) * c + (hy - py) * s, (hy - py) * c - (hx - px) * s,
            hvx * c + hvy * s, hvy * c - hvx * s, hr, da, r + hr]
    return torch.stack(cols, 1)


def _mlp(x, w, prefix, idx, last_relu):
    for n, k in enumerate(idx):
        x = F.linear(x, w["%s.%d.weight" % (prefix, k)], w["%s.%d.bias" % (prefix, k)])
        if n != len(idx) - 1 or last_relu:
            x = torch.relu(x)
    return x


def sarl_forward(w, x):
    """w: state_dict-style {name: float32 tensor}; x: [B,N,13] float32 -> (values [B], attention [B,N])."""
    B, N, _ = x.shape
    flat = x.reshape(B * N, -1)
    h = _mlp(flat, w, "mlp1", (0, 2), True)
    m = _mlp(h, w, "mlp2", (0, 2), False)
    g = h.view(B, N, -1).mean(1, keepdim=True).expand(B, N, h.shape[1]).reshape(B * N, -1)
    s = _mlp(torch.cat([h, g], 1), w, "attention", (0, 2, 4), False).view(B, N)
    e = torch.exp(s) * (s != 0).float()
    a = e / e.sum(1, keepdim=True)
    pooled = (a.unsqueeze(2) * m.view(B, N, -1)).sum(1)
    joint = torch.cat([x[:, 0, :6], pooled], 1)
    v = _mlp(joint, w, "mlp3", (0, 2, 4, 6), False)
    return v[:, 0], a


def lookahead_reward(nav_px, nav_py, nav_r, gx, gy, humans_next, dt):
    """multi_human_rl.py:65-88 for one candidate (float64)."""
    dmin, coll = float("inf"), False
    for hx, hy, hr in humans_next:
        d = float(np.linalg.norm((nav_px - hx, nav_py - hy))) - nav_r - hr
        if d < 0:
            coll = True
            break
        dmin = min(dmin, d)
    reach = float(np.linalg.norm((nav_px - gx, nav_py - gy))) < nav_r
    if coll:
        return -0.25
    if reach:
        return 1
    if dmin < 0.2:
        return (dmin - 0.2) * 0.5 * dt
    return 0


def sarl_predict(w, self_row, humans, table, dt=0.25, gamma=0.9, kinematics="holonomic"):
    """self_row: 9 float64 (px,py,vx,vy,r,gx,gy,v_pref,theta); humans [N,5] float64.
    Returns (values [A] float64, argmax index with first-max-wins)."""
    px, py, vx, vy, r, gx, gy, vpref, theta = [float(v) for v in self_row]
    nxt = [(h[0] + h[2] * dt, h[1] + h[3] * dt, h[2], h[3], h[4]) for h in humans]
    vals = []
    for a in table:
        if kinematics == "holonomic":
            npx, npy, nvx, nvy, nth = px + a[0] * dt, py + a[1] * dt, a[0], a[1], theta
        else:
            nth = theta + a[1]
            nvx, nvy = a[0] * np.cos(nth), a[0] * np.sin(nth)
            npx, npy = px + nvx * dt, py + nvy * dt
        rew = lookahead_reward(npx, npy, r, gx, gy, [(h[0], h[1], h[4]) for h in nxt], dt)
        rows = torch.tensor([[npx, npy, nvx, nvy, r, gx, gy, vpref, nth, h[0], h[1], h[2], h[3], h[4]] for h in nxt],
                            dtype=torch.float64).float()
        v, _ = sarl_forward(w, rotate(rows, kinematics).unsqueeze(0))
        vals.append(rew + pow(gamma, dt * vpref) * float(v[0]))
    vals = np.array(vals)
    best, idx = float("-inf"), -1
    for k, v in enumerate(vals):
        if v > best:
            best, idx = v, k
    return vals, idx


# ------------------------------------------------------------------------------- SGAN
def _lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    g = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, gg, o = g.chunk(4, 1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c), c


def sgan_generator(w, obs_traj, obs_rel, n_per_scene, noise, pooling):
    """One-step generator.  obs_*: [8, S*N, 2] float32, noise [S, noise_dim] -> pred_rel [S*N, 2]."""
    T, B, _ = obs_rel.shape
    emb = F.linear(obs_rel.reshape(-1, 2), w["encoder.spatial_embedding.weight"], w["encoder.spatial_embedding.bias"])
    emb = emb.view(T, B, -1)
    H = w["encoder.encoder.weight_hh_l0"].shape[1]
    h = torch.zeros(B, H); c = torch.zeros(B, H)
    for t in range(T):
        h, c = _lstm_cell(emb[t], h, c, w["encoder.encoder.weight_ih_l0"], w["encoder.encoder.weight_hh_l0"],
                          w["encoder.encoder.bias_ih_l0"], w["encoder.encoder.bias_hh_l0"])
    ctx_in = h
    if pooling:
        end = obs_traj[-1]
        pools = []
        for s in range(B // n_per_scene):
            sl = slice(s * n_per_scene, (s + 1) * n_per_scene)
            hs, ps = h[sl], end[sl]
            n = n_per_scene
            rel = ps.repeat(n, 1) - ps.unsqueeze(1).repeat(1, n, 1).view(-1, 2)       # P_j - P_i, i-major
            re = F.linear(rel, w["pool_net.spatial_embedding.weight"], w["pool_net.spatial_embedding.bias"])
            x = torch.cat([re, hs.repeat(n, 1)], 1)
            x = torch.relu(F.linear(x, w["pool_net.mlp_pre_pool.0.weight"], w["pool_net.mlp_pre_pool.0.bias"]))
            x = torch.relu(F.linear(x, w["pool_net.mlp_pre_pool.2.weight"], w["pool_net.mlp_pre_pool.2.bias"]))
            pools.append(x.view(n, n, -1).max(1)[0])
        ctx_in = torch.cat([h, torch.cat(pools, 0)], 1)
    x = torch.relu(F.linear(ctx_in, w["mlp_decoder_context.0.weight"], w["mlp_decoder_context.0.bias"]))
    x = torch.relu(F.linear(x, w["mlp_decoder_context.2.weight"], w["mlp_decoder_context.2.bias"]))
    z = noise.repeat_interleave(n_per_scene, 0)
    dh = torch.cat([x, z], 1)
    dc = torch.zeros_like(dh)
    din = F.linear(obs_rel[-1], w["decoder.spatial_embedding.weight"], w["decoder.spatial_embedding.bias"])
    dh, dc = _lstm_cell(din, dh, dc, w["decoder.decoder.weight_ih_l0"], w["decoder.decoder.weight_hh_l0"],
                        w["decoder.decoder.bias_ih_l0"], w["decoder.decoder.bias_hh_l0"])
    return F.linear(dh, w["decoder.hidden2pos.weight"], w["decoder.hidden2pos.bias"])


def sgan_velocities(pred_rel, last_pos, time_step):
    """world_model.py:258-268: pred_abs = last + rel (float32), then (pred - last) / time_step in float64."""
    pred_abs = (pred_rel + last_pos).double().numpy()
    return (pred_abs - last_pos.double().numpy()) / time_step
