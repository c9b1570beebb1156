"""World models that advance the humans inside ModelCrowdSim (reference: crowd_nav/policy/world_model.py).

  get_generator   :108-131  build a TrajectoryGenerator from a checkpoint dict
  SGANWorld       :134-268  E = 1 callable with the reference's constructor and return type
  VecSGANWorld              the same step for E scenes, history kept as a ring in HBM

  MlpWorld        :22-51    torch module (trained by Trainer_Sim); VecMlpWorld runs it for E scenes in one HIP launch
  AttentionWorld  :54-106   torch module; VecTorchWorld runs any [B,4N] -> [B,2N] module on a VecModelCrowdSim
"""
import ctypes as C
import logging
import os

import numpy as np
import torch
from torch import nn

from ..sgan.models import TrajectoryGenerator, sgan_step


def get_generator(checkpoint, device):
    a = checkpoint["args"]
    get = a.get if isinstance(a, dict) else (lambda k: getattr(a, k))
    gen = TrajectoryGenerator(
        obs_len=get("obs_len"), pred_len=get("pred_len"), embedding_dim=get("embedding_dim"),
        encoder_h_dim=get("encoder_h_dim_g"), decoder_h_dim=get("decoder_h_dim_g"), mlp_dim=get("mlp_dim"),
        num_layers=get("num_layers"), noise_dim=get("noise_dim"), noise_type=get("noise_type"),
        noise_mix_type=get("noise_mix_type"), pooling_type=get("pooling_type"),
        pool_every_timestep=get("pool_every_timestep"), dropout=get("dropout"), bottleneck_dim=get("bottleneck_dim"),
        neighborhood_size=get("neighborhood_size"), grid_size=get("grid_size"), batch_norm=get("batch_norm"),
        device=device)
    gen.load_state_dict(checkpoint["g_state"])
    gen.train()
    return gen


def round4(x):
    """np.around(x, 4) on float64 tensors (round-half-even), world_model.py:169,192."""
    return torch.round(x * 10000.0) / 10000.0


class VecSGANWorld(object):
    """Batched SGANWorld: hist [E,8,N,2] float64 ring in HBM, one mcn_sgan_step per call."""

    def __init__(self, generator, num_envs, num_humans, device, time_step=0.25, seed=None):
        self.generator = generator
        self.E, self.N = int(num_envs), int(num_humans)
        self.device = torch.device(device)
        self.time_step = float(time_step)
        self.hist = torch.zeros(self.E, 8, self.N, 2, dtype=torch.float64, device=self.device)
        self.oldest = 0
        self.out_vel = torch.zeros(self.E, self.N, 2, dtype=torch.float64, device=self.device)
        self._gen = torch.Generator(device="cpu")
        if seed is not None:
            self._gen.manual_seed(int(seed))
        self.fixed_noise = None      # [E,8] float32 device tensor: used instead of fresh draws (reproducible runs)
        self._noise_buf, self._noise_at = None, 0

    def reset_history(self, hist):
        """hist: [E,8,N,2] positions, oldest frame first (datagen.py:423-430 writes the last obs_len frames)."""
        self.hist.copy_(round4(hist.to(self.device, torch.float64)))
        self.oldest = 0

    def init_constant_velocity(self, pos, vel):
        """8 frames ending at `pos`, walking backwards at `vel`."""
        k = torch.arange(7, -1, -1, dtype=torch.float64, device=self.device).view(1, 8, 1, 1)
        self.reset_history(pos.unsqueeze(1) - vel.unsqueeze(1) * (k * self.time_step))

    _NOISE_BLOCK = 32

    def draw_noise(self):
        """One [E, 8] standard-normal sample per call (sgan/models.py:475-480 draws it on the host), taken from blocks
        of `_NOISE_BLOCK` calls drawn and uploaded together: one host draw + copy per 32 world-model steps."""
        if self._noise_buf is None or self._noise_at == self._NOISE_BLOCK:
            host = torch.randn(self._NOISE_BLOCK, self.E, 8, generator=self._gen)
            self._noise_buf = host.to(self.device)
            self._noise_at = 0
        self._noise_at += 1
        return self._noise_buf[self._noise_at - 1]

    def __call__(self, cur_pos, noise=None, hcount=None):
        """cur_pos [E,N,2] float64 -> velocities [E,N,2] float64 (a view reused by the next call).
        hcount ([E] int32, optional): scene e has only its first hcount[e] pedestrians (the rest of the N slots is
        ignored by the pooling module and their outputs are meaningless)."""
        if noise is None:
            noise = self.fixed_noise if self.fixed_noise is not None else self.draw_noise()
        push = self.oldest
        self.oldest = (self.oldest + 1) & 7
        cur = cur_pos if (cur_pos.dtype == torch.float64 and cur_pos.is_contiguous()) else \
            cur_pos.to(self.device, torch.float64).contiguous()
        sgan_step(self.generator, self.hist, push, self.oldest, cur, noise.float().contiguous(), self.time_step,
                  out_vel=self.out_vel, hcount=hcount)
        return self.out_vel


class SGANWorld(nn.Module):
    """E = 1 drop-in (world_model.py:134-268): `sim_world(list[[px,py,vx,vy]]) -> ndarray [N,2]`.

    `dataFile` is read once if it exists (the `frame<TAB>ped<TAB>x<TAB>y` cache the reference rewrites every
    step); afterwards the history lives on the device and no file I/O happens per step."""

    def __init__(self, dataFile, device, obs_len=8, pred_len=1, skip=1, delim="tab", time_step=0.4, pretrainPath=""):
        super().__init__()
        if obs_len != 8 or pred_len != 1:
            raise NotImplementedError("the shipped generators use obs_len 8 and one-step prediction")
        self.dataFile, self.device = dataFile, torch.device(device)
        self.obs_len, self.pred_len, self.skip, self.delim = obs_len, pred_len, skip, delim
        self.frameid = obs_len + 10
        self.time_step = time_step
        self.generator = None
        self._vec = None
        if pretrainPath != "":
            logging.info("Loading SGAN pretrain generator: %s", pretrainPath)
            checkpoint = torch.load(pretrainPath, map_location="cpu", weights_only=True)
            self.generator = get_generator(checkpoint, self.device)

    def _read_cache(self):
        rows = []
        sep = {"tab": "\t", "space": " "}.get(self.delim, self.delim)
        with open(self.dataFile) as fh:
            for line in fh:
                parts = line.strip().split(sep)
                if len(parts) == 4:
                    rows.append([float(v) for v in parts])
        return np.asarray(rows, np.float64)

    def load_history(self, n_humans):
        """(Re)build the device ring from the cache file (call after the driver rewrites it)."""
        dev = self.device if self.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        self._vec = VecSGANWorld(self.generator, 1, n_humans, dev, self.time_step)
        if self.dataFile and os.path.exists(self.dataFile):
            data = self._read_cache()
            frames = np.unique(data[:, 0])[-8:] if len(data) else []
            if len(frames) == 8:
                hist = np.full((1, 8, n_humans, 2), np.nan)
                for t, f in enumerate(frames):
                    sel = data[data[:, 0] == f]
                    for row in sel:
                        if int(row[1]) < n_humans:
                            hist[0, t, int(row[1])] = row[2:4]
                # a pedestrian that enters after the first frame / leaves before the last one stands on its first /
                # last recorded position for the missing frames (world_model.py:166-180)
                for p_ in range(n_humans):
                    seen = np.nonzero(~np.isnan(hist[0, :, p_, 0]))[0]
                    if len(seen) == 0:
                        hist[0, :, p_] = 0.0
                        continue
                    hist[0, :seen[0], p_] = hist[0, seen[0], p_]
                    hist[0, seen[-1] + 1:, p_] = hist[0, seen[-1], p_]
                if np.isnan(hist).any():
                    raise NotImplementedError("a pedestrian with a gap inside the cached window")
                self._vec.reset_history(torch.from_numpy(hist))
                return
        self._vec = None

    def forward(self, in_state):
        self.frameid += 1
        cur = torch.tensor([[s[0], s[1]] for s in in_state], dtype=torch.float64).unsqueeze(0)
        n = cur.shape[1]
        if self._vec is None or self._vec.N != n:
            self.load_history(n)
            if self._vec is None:          # no usable cache: start from a constant-velocity history
                dev = self.device if self.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
                self._vec = VecSGANWorld(self.generator, 1, n, dev, self.time_step)
                vel = torch.tensor([[s[2], s[3]] for s in in_state], dtype=torch.float64).unsqueeze(0)
                self._vec.init_constant_velocity(cur.to(dev), vel.to(dev))
        noise = torch.randn(1, 8)             # same global-stream draw as get_noise (sgan/models.py:20-25)
        vel = self._vec(cur.to(self._vec.device), noise.to(self._vec.device))
        return vel[0].cpu().numpy()


def generator_from_arrays(arrays, prefix, device=None):
    """Build a TrajectoryGenerator from plain arrays (e.g. the re-serialised zara1_8 weights in
    tests/golden/g6_sgan.npz: keys '<prefix>__w__<state_dict key with . -> __>')."""
    tag = prefix + "__w__"
    sd = {k[len(tag):].replace("__", "."): torch.from_numpy(np.asarray(arrays[k])) for k in arrays.files
          if k.startswith(tag)}
    pooling = "pool_net" if any(k.startswith("pool_net.") for k in sd) else None
    gen = TrajectoryGenerator(pooling_type=pooling, device=device)
    gen.load_state_dict(sd)
    return gen


# ---------------------------------------------------------------------------------------------
# Learned one-step world models other than SGAN (reference: world_model.py:17-106).  These are the next tier
# (SURVEY.md 8f, f3): kept as torch modules with the reference's parameter names so their checkpoints load and
# ModelCrowdSim can call them (`sim_world(Tensor[B,4N]) -> Tensor[B,2N]`, model_crowd_sim.py:404-407); they run
# through torch's ROCm GEMMs, not through hand-written kernels.
def init_weight(m):
    if type(m) == nn.Linear:
        nn.init.xavier_uniform_(m.weight)


class MlpWorld(nn.Module):
    """world_model.py:22-51: 4N -> 128 -> 64 -> 12 -> 2N with dropout and a final tanh."""

    def __init__(self, num_human, drop_rate=0.5, multihuman=True):
        super().__init__()
        if not multihuman:
            num_human = 1
        self.mlp = nn.Sequential(nn.Linear(num_human * 4, 128), nn.ReLU(True), nn.Dropout(drop_rate),
                                 nn.Linear(128, 64), nn.ReLU(True), nn.Dropout(drop_rate),
                                 nn.Linear(64, 12), nn.ReLU(True), nn.Linear(12, num_human * 2), nn.Tanh())
        self.mse = 0
        self.device = None

    def forward(self, x):
        return self.mlp(x)

    def noise_pre(self, x):
        import math
        x = self.forward(x)
        return x + (torch.randn(x.shape) * math.sqrt(self.mse)).to(self.device)


class AttentionWorld(nn.Module):
    """world_model.py:54-106: SARL-style attention over humans, 2 outputs per human (no final tanh)."""

    def __init__(self, input_dim=4, with_global_state=True):
        super().__init__()
        from .cadrl import mlp
        self.input_dim = input_dim
        self.with_global_state = with_global_state
        self.global_state_dim = 100
        self.mlp1 = mlp(input_dim, [150, 100], last_relu=True)
        self.mlp2 = mlp(100, [100, 50])
        self.attention = mlp(200 if with_global_state else 100, [100, 100, 1])
        self.mlp3_input_dim = 50 + input_dim
        self.mlp3 = mlp(self.mlp3_input_dim, [150, 100, 100, 2])
        self.attention_weights = None
        self.output_func = nn.Tanh()

    def forward(self, in_state):
        state = in_state.view(in_state.shape[0], -1, self.input_dim)
        B, N, _ = state.shape
        h = self.mlp1(state.reshape(B * N, -1))
        feat = self.mlp2(h)
        if self.with_global_state:
            g = h.view(B, N, -1).mean(1, keepdim=True).expand(B, N, self.global_state_dim)
            att_in = torch.cat([h, g.reshape(B * N, -1)], dim=1)
        else:
            att_in = h
        scores = self.attention(att_in).view(B, N)
        e = torch.exp(scores) * (scores != 0).float()
        w = (e / e.sum(dim=1, keepdim=True)).unsqueeze(2)
        self.attention_weights = w[0, :, 0].data.cpu().numpy()
        pooled = (w * feat.view(B, N, -1)).sum(dim=1, keepdim=True).expand(B, N, feat.shape[1])
        joint = torch.cat([state, pooled], dim=2)
        return self.mlp3(joint.reshape(B * N, self.mlp3_input_dim)).view(B, -1)


class _MlpWorldNet(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4")]


def _weights_stamp(module):
    """Changes whenever a parameter of `module` is written in place (optimizer steps, load_state_dict, .copy_) or
    replaced (another tensor object / device move): what the packed HIP fragments were made from."""
    return tuple((id(p), p.data_ptr(), p._version) for p in module.parameters())


def _dropout_active(module):
    """A module left in train() mode with a Dropout of p > 0 draws masks; the HIP kernels are eval-mode forwards."""
    return module.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in module.modules())


def pack_mlp_world(module, num_human, dev):
    """MlpWorld.state_dict -> (ctypes mcn_mlp_world_net, [device tensors kept alive]); layouts in include/mcn.h."""
    from .. import _hip
    from .sarl import _ident, _natural
    sd = {k: v.detach().to("cpu", torch.float32).contiguous().numpy() for k, v in module.state_dict().items()}
    N = int(num_human)
    if tuple(sd["mlp.0.weight"].shape) != (128, 4 * N) or tuple(sd["mlp.8.weight"].shape) != (2 * N, 12):
        raise ValueError("MlpWorld built for %d pedestrians, asked to run %d" % (sd["mlp.0.weight"].shape[1] // 4, N))
    kt1, nt4 = (4 * N + 15) // 16, (2 * N + 15) // 16
    plan = [("1", "mlp.0", _natural(4 * N, kt1), kt1, None, 8),
            ("2", "mlp.3", _natural(128, 8), 8, None, 4),
            ("3", "mlp.6", _natural(64, 4), 4, _ident(12, 1), 1),
            ("4", "mlp.8", _ident(12, 1), 1, None, nt4)]
    net, keep = _MlpWorldNet(), []
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    for tag, key, kmap, KT, omap, NT in plan:
        W, b = sd[key + ".weight"], sd[key + ".bias"]
        nout, kin = W.shape
        wf = np.zeros((NT, KT, 64, 4), np.float32)
        bf = np.zeros((NT, 64, 4), np.float32)
        om = omap if omap is not None else _natural(nout, NT)
        rc = _hip.lib.mcn_pack_linear(W.ctypes.data_as(fp), b.ctypes.data_as(fp), nout, kin, kmap.ctypes.data_as(ip), KT,
                                      om.ctypes.data_as(ip), NT, wf.ctypes.data_as(fp), bf.ctypes.data_as(fp))
        _hip.check(rc, "mcn_pack_linear(mlp_world %s)" % key)
        dw, db = torch.from_numpy(wf).to(dev), torch.from_numpy(bf).to(dev)
        keep += [dw, db]
        setattr(net, "w" + tag, dw.data_ptr())
        setattr(net, "b" + tag, db.data_ptr())
    return net, keep


class VecMlpWorld(object):
    """MlpWorld as a VecModelCrowdSim `sim_world`: one mcn_mlp_world_step launch for all E scenes (eval mode: the
    reference calls the model under `model_sim.eval()` when it imagines, train_model_based_sgan.py).  The packed weight
    fragments follow the module: they are re-packed whenever a parameter changed since they were made (training rounds
    alternate with imagination on the same module, train_model_based.py) -- `refresh()` is never needed.  A module left
    in train() mode with active Dropout is NOT an eval-mode forward: that call goes through the torch module instead
    (same masks / random stream as the E = 1 path)."""

    def __init__(self, module, env):
        self.module, self.env = module, env
        self._net = None
        self.out_vel = None
        self._torch = None

    def refresh(self):
        self._net = None

    def __call__(self, hpos, noise=None):
        from .. import _hip
        env = self.env
        E, N, dev = env.num_envs, env._alloc_N, env.device
        if _dropout_active(self.module):
            if self._torch is None:
                self._torch = VecTorchWorld(self.module, env)
            return self._torch(hpos, noise)
        stamp = _weights_stamp(self.module)
        if self._net is None or self._net[2] != (E, N) or self._net[3] != stamp:
            net, keep = pack_mlp_world(self.module, N, dev)
            self._net = (net, keep, (E, N), stamp)
            self.out_vel = torch.empty(E, N, 2, dtype=torch.float64, device=dev)
        rc = _hip.lib.mcn_mlp_world_step(C.byref(self._net[0]), _hip.ptr(env.hpos), _hip.ptr(env.hvel),
                                         _hip.ptr(self.out_vel), E, N, _hip.stream_ptr(dev))
        _hip.check(rc, "mcn_mlp_world_step")
        return self.out_vel


class _AttnWorldNet(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "w_m1a", "b_m1a", "w_m1b", "b_m1b", "w_m2a", "b_m2a", "w_m2b", "b_m2b", "w_ata", "b_ata", "w_atg", "w_atb", "b_atb",
        "w_atc", "b_atc", "w_m3p", "b_m3p", "w_m3s", "w_m3b", "b_m3b", "w_m3c", "b_m3c", "w_m3d", "b_m3d")]


def pack_attn_world(module, dev):
    """AttentionWorld.state_dict -> (ctypes mcn_attn_world_net, [device tensors kept alive]); layouts in include/mcn.h."""
    from .. import _hip
    from .sarl import _ident, _natural
    if not module.with_global_state or module.input_dim != 4:
        raise NotImplementedError("world_attn.hip is built for input_dim 4 with the global state (the reference's defaults)")
    sd = {k: v.detach().to("cpu", torch.float32).contiguous().numpy() for k, v in module.state_dict().items()}
    #        name   state_dict key  kmap                          KT  bias  omap (None = ragged "q first" like its consumers)
    plan = [("m1a", "mlp1.0", _ident(4, 1), 1, True, None),
            ("m1b", "mlp1.2", _ident(150, 10), 10, True, None),
            ("m2a", "mlp2.0", _ident(100, 7), 7, True, None),
            ("m2b", "mlp2.2", _ident(100, 7), 7, True, None),
            ("ata", "attention.0", _ident(100, 7), 7, True, None),
            ("atg", "attention.0", _ident(100, 7, offset=100), 7, False, None),
            ("atb", "attention.2", _ident(100, 7), 7, True, None),
            ("atc", "attention.4", _ident(100, 7), 7, True, None),
            ("m3p", "mlp3.0", _ident(50, 4, offset=4), 4, True, None),
            ("m3s", "mlp3.0", _ident(4, 1), 1, False, None),
            ("m3b", "mlp3.2", _ident(150, 10), 10, True, None),
            ("m3c", "mlp3.4", _ident(100, 7), 7, True, None),
            ("m3d", "mlp3.6", _ident(100, 7), 7, True, _natural(2, 1))]
    net, keep = _AttnWorldNet(), []
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    for name, key, kmap, KT, with_bias, omap in plan:
        W, b = sd[key + ".weight"], sd[key + ".bias"]
        nout, kin = W.shape
        NT = (nout + 15) // 16
        wf = np.zeros((NT, KT, 64, 4), np.float32)
        bf = np.zeros((NT, 64, 4), np.float32)
        om = omap if omap is not None else _ident(nout, NT)
        rc = _hip.lib.mcn_pack_linear(W.ctypes.data_as(fp), b.ctypes.data_as(fp), nout, kin, kmap.ctypes.data_as(ip), KT,
                                      om.ctypes.data_as(ip), NT, wf.ctypes.data_as(fp),
                                      bf.ctypes.data_as(fp) if with_bias else None)
        _hip.check(rc, "mcn_pack_linear(attn_world %s)" % name)
        dw = torch.from_numpy(wf).to(dev)
        keep.append(dw)
        setattr(net, "w_" + name, dw.data_ptr())
        if with_bias:
            db = torch.from_numpy(bf).to(dev)
            keep.append(db)
            setattr(net, "b_" + name, db.data_ptr())
    return net, keep


class VecAttnWorld(object):
    """AttentionWorld as a VecModelCrowdSim `sim_world`: one mcn_attn_world_step launch for all E scenes.  The packed
    weight fragments follow the module (re-packed whenever a parameter changed since they were made; `refresh()` is never
    needed).  `hcount` ([E] int32 device tensor): scene e has only its first hcount[e] pedestrians."""

    def __init__(self, module, env):
        self.module, self.env = module, env
        self._net = None
        self.out_vel = None

    def refresh(self):
        self._net = None

    def __call__(self, hpos, noise=None, hcount=None):
        from .. import _hip
        env = self.env
        E, N, dev = env.num_envs, env._alloc_N, env.device
        if _dropout_active(self.module):
            raise RuntimeError("AttentionWorld in train() mode with active Dropout: the HIP kernel is an eval-mode forward")
        stamp = _weights_stamp(self.module)
        if self._net is None or self._net[2] != (E, N) or self._net[4] != stamp:
            net, keep = pack_attn_world(self.module, dev)
            ws = torch.empty(_hip.lib.mcn_attn_world_workspace_bytes(E, N) // 4, dtype=torch.float32, device=dev)
            self._net = (net, keep, (E, N), ws, stamp)
            self.out_vel = torch.zeros(E, N, 2, dtype=torch.float64, device=dev)
        rc = _hip.lib.mcn_attn_world_step(C.byref(self._net[0]), _hip.ptr(env.hpos), _hip.ptr(env.hvel), _hip.ptr(hcount),
                                          _hip.ptr(self._net[3]), _hip.ptr(self.out_vel), E, N, _hip.stream_ptr(dev))
        _hip.check(rc, "mcn_attn_world_step")
        return self.out_vel


def vec_world(module, env):
    """The fastest `sim_world` adapter for a world-model module on a VecModelCrowdSim: the HIP kernels for MlpWorld and
    (default-shaped) AttentionWorld, torch for anything else."""
    if isinstance(module, MlpWorld) and env.human_num <= 10:
        return VecMlpWorld(module, env)
    if isinstance(module, AttentionWorld) and module.with_global_state and module.input_dim == 4:
        return VecAttnWorld(module, env)
    return VecTorchWorld(module, env)


class VecTorchWorld(object):
    """Adapter: a [B,4N] -> [B,2N] module (MlpWorld / AttentionWorld) as a VecModelCrowdSim `sim_world`."""

    def __init__(self, module, env):
        self.module, self.env = module, env

    def __call__(self, hpos, noise=None):
        env = self.env
        x = torch.cat([env.hpos, env.hvel], dim=2).reshape(env.num_envs, -1).float()
        prm = next(self.module.parameters(), None)
        if prm is not None and prm.device != x.device:          # a module left on the host (the E = 1 drivers do that)
            x = x.to(prm.device)
        with torch.no_grad():
            v = self.module(x)
        return v.view(env.num_envs, -1, 2).double().to(env.device).contiguous()
