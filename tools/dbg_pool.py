import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from modelcrowdnav_amd.policy.world_model import generator_from_arrays
from modelcrowdnav_amd.sgan import models as M
dev = torch.device("cuda", 0)
g = np.load(os.path.join(ROOT, "tests/golden/g6_sgan.npz"))
gen = generator_from_arrays(g, "p", dev)
E, N = 3, 10
gg = torch.Generator().manual_seed(0)
hist = (torch.rand(E, 8, N, 2, dtype=torch.float64, generator=gg) * 4).to(dev)
noise = torch.zeros(E, 8, device=dev)
vel, rel = M.sgan_step(gen, hist, 0, 0, None, noise, 1.0, want_rel=True)
torch.cuda.synchronize()
ws = list(M._WS.values())[0].cpu().numpy()
nped = E * N
henc = ws[:nped * 32].reshape(nped, 32); last = ws[nped * 32:nped * 36].reshape(nped, 4); pool = ws[nped * 36:nped * 44].reshape(nped, 8)
sd = {k: v.detach().cpu().double().numpy() for k, v in gen.state_dict().items()}
W1, b1 = sd["pool_net.mlp_pre_pool.0.weight"], sd["pool_net.mlp_pre_pool.0.bias"]
W2, b2 = sd["pool_net.mlp_pre_pool.2.weight"], sd["pool_net.mlp_pre_pool.2.bias"]
We, be = sd["pool_net.spatial_embedding.weight"], sd["pool_net.spatial_embedding.bias"]
want = np.zeros((nped, 8))
for e in range(E):
    for i in range(N):
        outs = []
        for k in range(N):
            d = last[e * N + k, :2].astype(np.float64) - last[e * N + i, :2]
            x = np.concatenate([We @ d + be, henc[e * N + k]])
            hid = np.maximum(W1 @ x + b1, 0)
            outs.append(np.maximum(W2 @ hid + b2, 0))
        want[e * N + i] = np.max(outs, 0)
print("max |pool - want|", np.abs(pool - want).max())
print(pool[:3]); print(want[:3])
bad = np.abs(pool - want).max(1)
print("per ped err", np.round(bad, 4))
