"""SARL: socially attentive value network (reference: crowd_nav/policy/sarl.py:9-89).

ValueNetwork keeps the reference's module tree, so `state_dict()` keys are
mlp1.{0,2} / mlp2.{0,2} / attention.{0,2,4} / mlp3.{0,2,4,6} .weight/.bias and `rl_model.pth`
files load unchanged (train.py:57-58,147-148).  Its torch forward exists for training code that
back-propagates through it; every inference on the rollout path (`SARL.predict`,
`SARL.predict_batch`) runs sarl_value.hip on weights re-packed into MFMA operand order.
"""
import ctypes as C
import logging

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from .cadrl import mlp
from .multi_human_rl import MultiHumanRL


class ValueNetwork(nn.Module):
    def __init__(self, input_dim, self_state_dim, mlp1_dims, mlp2_dims, mlp3_dims, attention_dims, with_global_state,
                 cell_size, cell_num):
        super().__init__()
        self.self_state_dim = self_state_dim
        self.global_state_dim = mlp1_dims[-1]
        self.with_global_state = with_global_state
        self.cell_size, self.cell_num = cell_size, cell_num
        self.mlp1 = mlp(input_dim, mlp1_dims, last_relu=True)
        self.mlp2 = mlp(mlp1_dims[-1], mlp2_dims)
        self.attention = mlp(mlp1_dims[-1] * (2 if with_global_state else 1), attention_dims)
        self.mlp3 = mlp(mlp2_dims[-1] + self_state_dim, mlp3_dims)
        self.attention_weights = None

    def forward(self, state):
        """sarl.py:28-65 in torch ops (autograd-capable; used by training, not by the rollout)."""
        B, N, D = state.shape
        h = self.mlp1(state.reshape(B * N, D))
        feat = self.mlp2(h)
        if self.with_global_state:
            g = h.view(B, N, -1).mean(1, keepdim=True).expand(B, N, self.global_state_dim)
            att_in = torch.cat([h, g.reshape(B * N, -1)], dim=1)
        else:
            att_in = h
        scores = self.attention(att_in).view(B, N)
        e = torch.exp(scores) * (scores != 0).float()          # un-stabilised masked softmax, literally
        w = (e / e.sum(dim=1, keepdim=True)).unsqueeze(2)
        self.attention_weights = w[0, :, 0].data.cpu().numpy()
        pooled = (w * feat.view(B, N, -1)).sum(dim=1)
        return self.mlp3(torch.cat([state[:, 0, :self.self_state_dim], pooled], dim=1))


# input-slot -> weight-column maps of each packed layer (see mcn_pack_linear in include/mcn.h)
def _ident(kin, tiles, offset=0):
    """Slot -> feature map of a `kin`-wide activation held in `tiles` tiles of 16.  Full tiles are in natural
    order; the ragged last tile is packed "q first" (feature j at slot 4(j%4) + j/4) so that its consumers need
    only ceil(w/4) k-steps (include/mcn.h, mcn_pack_linear)."""
    m = np.full(tiles * 16, -1, np.int32)
    full = (kin // 16) * 16 if kin % 16 else kin
    m[:full] = np.arange(full)
    for j in range(kin - full):
        m[full + 4 * (j % 4) + j // 4] = full + j
    m[m >= 0] += offset
    return m


def _natural(kin, tiles, offset=0):
    m = np.full(tiles * 16, -1, np.int32)
    m[:kin] = np.arange(kin) + offset
    return m


def _pack_plan():
    m3a = np.full(5 * 16, -1, np.int32)
    m3a[:64] = _ident(50, 4, offset=6)    # pooled features occupy tiles 0..3 -> mlp3.0 columns 6..55
    m3a[64:80] = _ident(6, 1)             # self features occupy tile 4      -> mlp3.0 columns 0..5
    #        name   state_dict key   kmap                KT  bias?
    return [("m1a", "mlp1.0", _natural(13, 1), 1, True),
            ("m1b", "mlp1.2", _ident(150, 10), 10, True),
            ("m2a", "mlp2.0", _ident(100, 7), 7, True),
            ("m2b", "mlp2.2", _ident(100, 7), 7, True),
            ("ata", "attention.0", _ident(100, 7), 7, True),
            ("atg", "attention.0", _ident(100, 7, offset=100), 7, False),
            ("atb", "attention.2", _ident(100, 7), 7, True),
            ("atc", "attention.4", _ident(100, 7), 7, True),
            ("m3a", "mlp3.0", m3a, 5, True),
            ("m3b", "mlp3.2", _ident(150, 10), 10, True),
            ("m3c", "mlp3.4", _ident(100, 7), 7, True),
            ("m3d", "mlp3.6", _ident(100, 7), 7, True)]


class _SarlX3(C.Structure):
    """mcn_sarl_x3: device pointers to the bf16x3 weight fragments (mcn_pack_x3)."""
    _fields_ = [(n, C.c_void_p) for n in ("w_m1a", "w_m1b", "w_m2a", "w_m2b", "w_ata", "w_atg", "w_atb", "w_atc",
                                          "w_m3a", "w_m3b", "w_m3c", "w_m3d")]


class _SarlNet(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_m1a", "b_m1a", "w_m1b", "b_m1b", "w_m2a", "b_m2a", "w_m2b", "b_m2b",
                                          "w_ata", "b_ata", "w_atg", "w_atb", "b_atb", "w_atc", "b_atc",
                                          "w_m3a", "b_m3a", "w_m3b", "b_m3b", "w_m3c", "b_m3c", "w_m3d", "b_m3d",
                                          "x3")]


def pack_value_network(model, dev):
    """state_dict -> (ctypes mcn_sarl_net, [device tensors kept alive])."""
    sd = {k: v.detach().to("cpu", torch.float32).contiguous().numpy() for k, v in model.state_dict().items()}
    expect = {"mlp1.0": (150, 13), "mlp1.2": (100, 150), "mlp2.0": (100, 100), "mlp2.2": (50, 100),
              "attention.0": (100, 200), "attention.2": (100, 100), "attention.4": (1, 100),
              "mlp3.0": (150, 56), "mlp3.2": (100, 150), "mlp3.4": (100, 100), "mlp3.6": (1, 100)}
    for k, shp in expect.items():
        if tuple(sd[k + ".weight"].shape) != shp:
            raise ValueError("sarl_value.hip is built for the shipped SARL dimensions (policy.config:44-50); "
                             "%s.weight is %s, expected %s" % (k, sd[k + ".weight"].shape, shp))
    net, keep = _SarlNet(), []
    x3 = _SarlX3()
    fp = C.POINTER(C.c_float)
    for name, key, kmap, KT, with_bias in _pack_plan():
        W, b = sd[key + ".weight"], sd[key + ".bias"]
        nout, kin = W.shape
        NT = (nout + 15) // 16
        wf = np.zeros((NT, KT, 64, 4), np.float32)
        bf = np.zeros((NT, 64, 4), np.float32)
        omap = _ident(nout, NT)             # outputs use the same ragged-tile packing their consumers assume
        ip = C.POINTER(C.c_int32)
        rc = _hip.lib.mcn_pack_linear(W.ctypes.data_as(fp), b.ctypes.data_as(fp), nout, kin,
                                      kmap.ctypes.data_as(ip), KT, omap.ctypes.data_as(ip), NT,
                                      wf.ctypes.data_as(fp), bf.ctypes.data_as(fp) if with_bias else None)
        _hip.check(rc, "mcn_pack_linear(%s)" % name)
        dw = torch.from_numpy(wf).to(dev)
        keep.append(dw)
        setattr(net, "w_" + name, dw.data_ptr())
        if with_bias:
            db = torch.from_numpy(bf).to(dev)
            keep.append(db)
            setattr(net, "b_" + name, db.data_ptr())
        # the same fragments as three bfloat16 pieces per weight, for the bf16 matrix pipe (include/mcn.h: mcn_pack_x3)
        xb = np.zeros(int(_hip.lib.mcn_pack_x3_bytes(NT, KT)), np.uint8)
        _hip.check(_hip.lib.mcn_pack_x3(wf.ctypes.data_as(fp), NT, KT, xb.ctypes.data), "mcn_pack_x3(%s)" % name)
        dx = torch.from_numpy(xb).to(dev)
        keep.append(dx)
        setattr(x3, "w_" + name, dx.data_ptr())
    keep.append(x3)                                     # the host struct mcn_sarl_net.x3 points at
    net.x3 = C.addressof(x3)
    return net, keep


class SARL(MultiHumanRL):
    def __init__(self):
        super().__init__()
        self.name = "SARL"
        self._last_attention = None
        self.chosen_attention_weights = None

    def configure(self, config):
        self.set_common_parameters(config)
        dims = lambda key: [int(x) for x in config.get("sarl", key).split(", ")]
        self.with_om = config.getboolean("sarl", "with_om")
        with_global_state = config.getboolean("sarl", "with_global_state")
        if not with_global_state:
            raise NotImplementedError("with_global_state=false is outside this build's scope (shipped config: true)")
        self.model = ValueNetwork(self.input_dim(), self.self_state_dim, dims("mlp1_dims"), dims("mlp2_dims"),
                                  dims("mlp3_dims"), dims("attention_dims"), with_global_state, self.cell_size,
                                  self.cell_num)
        self.multiagent_training = config.getboolean("sarl", "multiagent_training")
        if self.with_om:
            self.name = "OM-SARL"
        logging.info("Policy: %s %s global state", self.name, "w/" if with_global_state else "w/o")

    def _packed(self, dev):
        version = tuple(p._version for p in self.model.parameters()) + (str(dev),)
        if self._frags is None or self._frags[0] != version:
            net, keep = pack_value_network(self.model, dev)
            self._frags = (version, net, keep)
        return self._frags[1]

    def get_attention_weights(self):
        """sarl.py:88-89: the attention weights of the model's last forward -- after predict() those of the LAST
        candidate action of the table, as in the reference (stale after a call that returned without a look-ahead).
        `chosen_attention_weights` holds the weights of the action predict() picked."""
        return self._last_attention
