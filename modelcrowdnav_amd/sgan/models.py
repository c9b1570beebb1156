"""TrajectoryGenerator of the shipped Social-GAN checkpoints (reference: sgan/models.py).

The module tree reproduces the reference's parameter names so `checkpoint['g_state']` of
sgan/models/sgan-models/*.pt and sgan-p-models/*.pt loads with load_state_dict
(encoder.encoder.*, encoder.spatial_embedding.*, decoder.decoder.*, decoder.spatial_embedding.*,
decoder.hidden2pos.*, pool_net.spatial_embedding.*, pool_net.mlp_pre_pool.{0,2}.*,
mlp_decoder_context.{0,2}.*).  Inference does not run these modules: `pack()` permutes the
weights into MFMA operand order and sgan_step.hip evaluates the network (one-step prediction,
models.py:501-553).  SocialPooling, the discriminator, batch-norm variants and
pool_every_timestep are not used by any shipped checkpoint and are not built.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _hip


def make_mlp(dim_list):
    """models.py:5-17 with batch_norm=False, dropout=0 (the shipped configuration): Linear+ReLU pairs."""
    layers = []
    for a, b in zip(dim_list[:-1], dim_list[1:]):
        layers += [nn.Linear(a, b), nn.ReLU()]
    return nn.Sequential(*layers)


class _Encoder(nn.Module):
    def __init__(self, emb, h):
        super().__init__()
        self.encoder = nn.LSTM(emb, h, 1)
        self.spatial_embedding = nn.Linear(2, emb)


class _Decoder(nn.Module):
    def __init__(self, emb, h):
        super().__init__()
        self.seq_len = 1
        self.decoder = nn.LSTM(emb, h, 1)
        self.spatial_embedding = nn.Linear(2, emb)
        self.hidden2pos = nn.Linear(h, 2)


class _PoolHiddenNet(nn.Module):
    def __init__(self, emb, h, bottleneck):
        super().__init__()
        self.spatial_embedding = nn.Linear(2, emb)
        self.mlp_pre_pool = make_mlp([emb + h, 512, bottleneck])


class _SganNet(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_elstm", "b_elstm", "w_p1", "b_p1", "w_p2", "b_p2", "w_c1", "b_c1",
                                          "w_c2", "b_c2", "w_dlstm", "b_dlstm", "w_h2p", "b_h2p")] + [("pooling", C.c_int32)]


def _fold_embedding(W, b, W_se, b_se):
    """A layer W [nout, 16 + h] applied to [spatial_embedding(d), h] equals [W_e W_se | W_h] applied to [d, h] with
    bias b + W_e b_se (the embedding is a plain Linear(2, 16): sgan/models.py:47,66 / 120,142 / 186,221-223).
    Composed in float64 and rounded once."""
    W, W_se = W.astype(np.float64), W_se.astype(np.float64)
    Wf = np.concatenate([W[:, :16] @ W_se, W[:, 16:]], 1)
    bf = b.astype(np.float64) + W[:, :16] @ b_se.astype(np.float64)
    return Wf.astype(np.float32), bf.astype(np.float32)


def _kmap_rel_h():
    """Input tiles of a folded layer: tile 0 = displacement (x in slot 0, y in slot 4: one MFMA k-step), tiles 1-2 =
    the 32 hidden features."""
    m = np.full(48, -1, np.int32)
    m[0], m[4] = 0, 1
    m[16:48] = 2 + np.arange(32)
    return m


def _ident(kin, tiles, offset=0):
    m = np.full(tiles * 16, -1, np.int32)
    m[:kin] = np.arange(kin) + offset
    return m


class TrajectoryGenerator(nn.Module):
    def __init__(self, obs_len=8, pred_len=1, embedding_dim=16, encoder_h_dim=32, decoder_h_dim=32, mlp_dim=64,
                 num_layers=1, noise_dim=(8,), noise_type="gaussian", noise_mix_type="global", pooling_type=None,
                 pool_every_timestep=False, dropout=0.0, bottleneck_dim=8, activation="relu", batch_norm=False,
                 neighborhood_size=2.0, grid_size=8, device=None):
        super().__init__()
        if pooling_type and str(pooling_type).lower() == "none":
            pooling_type = None
        supported = (embedding_dim == 16 and encoder_h_dim == 32 and decoder_h_dim == 32 and mlp_dim == 64 and
                     num_layers == 1 and tuple(noise_dim) == (8,) and noise_mix_type == "global" and
                     bottleneck_dim == 8 and not batch_norm and not pool_every_timestep and
                     pooling_type in (None, "pool_net"))
        if not supported:
            raise NotImplementedError("sgan_step.hip is built for the architecture of the shipped checkpoints "
                                      "(emb 16, h 32, mlp 64, bottleneck 8, noise (8,) global, no batch norm)")
        self.obs_len, self.pred_len = obs_len, pred_len
        self.noise_type, self.noise_dim, self.pooling_type = noise_type, tuple(noise_dim), pooling_type
        self.device = device
        self.encoder = _Encoder(embedding_dim, encoder_h_dim)
        self.decoder = _Decoder(embedding_dim, decoder_h_dim)
        if pooling_type == "pool_net":
            self.pool_net = _PoolHiddenNet(embedding_dim, encoder_h_dim, bottleneck_dim)
        ctx_in = encoder_h_dim + (bottleneck_dim if pooling_type else 0)
        self.mlp_decoder_context = make_mlp([ctx_in, mlp_dim, decoder_h_dim - noise_dim[0]])
        self._packed = None

    # ------------------------------------------------------------------ packing
    def pack(self, dev):
        version = tuple(p._version for p in self.parameters()) + (str(dev),)
        if self._packed is not None and self._packed[0] == version:
            return self._packed[1]
        sd = {k: v.detach().to("cpu", torch.float32).contiguous().numpy() for k, v in self.state_dict().items()}
        net, keep = _SganNet(), []
        fp = C.POINTER(C.c_float)

        def put(name, W, b, kmap, KT):
            W, b = np.ascontiguousarray(W, np.float32), np.ascontiguousarray(b, np.float32)
            nout, kin = W.shape
            NT = (nout + 15) // 16
            wf, bf = np.zeros((NT, KT, 64, 4), np.float32), np.zeros((NT, 64, 4), np.float32)
            _hip.check(_hip.lib.mcn_pack_linear(W.ctypes.data_as(fp), b.ctypes.data_as(fp), nout, kin,
                                                kmap.ctypes.data_as(C.POINTER(C.c_int32)), KT, None, NT,
                                                wf.ctypes.data_as(fp), bf.ctypes.data_as(fp)), "mcn_pack_linear")
            for pre, arr in (("w_", wf), ("b_", bf)):
                t = torch.from_numpy(arr).to(dev)
                keep.append(t)
                setattr(net, pre + name, t.data_ptr())

        def lstm(prefix):
            W = np.concatenate([sd[prefix + ".weight_ih_l0"], sd[prefix + ".weight_hh_l0"]], 1)     # [128, 16+32]
            return W, sd[prefix + ".bias_ih_l0"] + sd[prefix + ".bias_hh_l0"]

        put("elstm", *_fold_embedding(*lstm("encoder.encoder"), sd["encoder.spatial_embedding.weight"],
                                      sd["encoder.spatial_embedding.bias"]), _kmap_rel_h(), 3)
        if self.pooling_type:
            put("p1", *_fold_embedding(sd["pool_net.mlp_pre_pool.0.weight"], sd["pool_net.mlp_pre_pool.0.bias"],
                                       sd["pool_net.spatial_embedding.weight"], sd["pool_net.spatial_embedding.bias"]),
                _kmap_rel_h(), 3)
            put("p2", sd["pool_net.mlp_pre_pool.2.weight"], sd["pool_net.mlp_pre_pool.2.bias"], _ident(512, 32), 32)
            put("c1", sd["mlp_decoder_context.0.weight"], sd["mlp_decoder_context.0.bias"], _ident(40, 3), 3)
        else:
            put("c1", sd["mlp_decoder_context.0.weight"], sd["mlp_decoder_context.0.bias"], _ident(32, 2), 2)
        put("c2", sd["mlp_decoder_context.2.weight"], sd["mlp_decoder_context.2.bias"], _ident(64, 4), 4)
        put("dlstm", *_fold_embedding(*lstm("decoder.decoder"), sd["decoder.spatial_embedding.weight"],
                                      sd["decoder.spatial_embedding.bias"]), _kmap_rel_h(), 3)
        put("h2p", sd["decoder.hidden2pos.weight"], sd["decoder.hidden2pos.bias"], _ident(32, 2), 2)
        net.pooling = 1 if self.pooling_type else 0
        self._packed = (version, (net, keep))
        return self._packed[1]

    # ------------------------------------------------------------------ reference call signature
    def forward(self, obs_traj, obs_traj_rel, seq_start_end, user_noise=None):
        """models.py:501-553 for pred_len 1 and equally sized scenes.  obs_traj [8,B,2]; returns [1,B,2]
        float32 on obs_traj's device.  obs_traj_rel is recomputed from obs_traj (identical values)."""
        if self.decoder.seq_len != 1:
            raise NotImplementedError("only one-step prediction (decoder.seq_len = 1, world_model.py:252) is built")
        dev = obs_traj.device
        if dev.type != "cuda":
            raise RuntimeError("TrajectoryGenerator inference only exists as HIP kernels; move inputs to the GPU")
        sizes = (seq_start_end[:, 1] - seq_start_end[:, 0]).tolist()
        if len(set(sizes)) != 1:
            raise NotImplementedError("scenes of different sizes in one batch are not supported")
        N, S = int(sizes[0]), len(sizes)
        noise = user_noise if user_noise is not None else torch.randn(S, self.noise_dim[0]).to(dev)
        hist = obs_traj.double().view(8, S, N, 2).permute(1, 0, 2, 3).contiguous()
        vel, rel = sgan_step(self, hist, 0, 0, None, noise.float().contiguous(), 1.0, want_rel=True)
        return rel.view(1, S * N, 2)


_WS = {}


def sgan_step(gen, hist, push_slot, oldest, cur_pos, noise, time_step, want_rel=False, out_vel=None, hcount=None):
    """Thin wrapper over mcn_sgan_step.  hist [E,8,N,2] f64 (modified in place when cur_pos is given)."""
    E, T, N, _ = hist.shape
    dev = hist.device
    net, _keep = gen.pack(dev)
    key = (E, N, str(dev))
    if key not in _WS:
        _WS.clear()
        _WS[key] = torch.empty(_hip.lib.mcn_sgan_workspace_bytes(E, N) // 4, dtype=torch.float32, device=dev)
    if out_vel is None:
        out_vel = torch.empty(E, N, 2, dtype=torch.float64, device=dev)
    rel = torch.empty(E * N, 2, dtype=torch.float32, device=dev) if want_rel else None
    rc = _hip.lib.mcn_sgan_step(C.byref(net), _hip.ptr(hist), int(push_slot), int(oldest), _hip.ptr(cur_pos),
                                _hip.ptr(noise), _hip.ptr(hcount), _hip.ptr(_WS[key]), _hip.ptr(out_vel), _hip.ptr(rel),
                                float(time_step), E, N, _hip.stream_ptr(dev))
    _hip.check(rc, "mcn_sgan_step")
    return out_vel, rel
