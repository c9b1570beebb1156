#!/usr/bin/env python3
"""GPU box, DIAGNOSTIC build: where a wavefront of mcn::sarl_value_kernel spends its cycles.

    make -C modelcrowdnav_amd/csrc diag
    MCN_HIP_LIB=modelcrowdnav_amd/csrc/build_diag/libmcn_hip.so python tools/sarl_phases.py [--humans 5]

The diagnostic kernel sums, per resident wavefront, the shader cycles between phase boundaries over all the tiles it
walks (mcn_debug_sarl_phases).  Printed: cycles per tile per phase (median over wavefronts), the MFMAs of the phase x
32 cycles (what the matrix pipe needs for ONE wavefront; two wavefronts share a SIMD, so a phase that keeps the pipe
busy shows about twice that), and their ratio.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from modelcrowdnav_amd import _hip  # noqa: E402

PHASES = ["tile set-up", "per human: loads, float64 distance, features", "mlp1.0", "mlp1.2", "workspace store + global sum",
          "reward, mean, attention.0 global half", "workspace load", "attention.0", "attention.2", "attention.4",
          "exp / attention output", "mlp2.0", "weighted accumulation", "normalise, mlp2.2, self tile", "mlp3 + store"]


def mfma_per_phase(N):
    per_h = {2: 40, 3: 266, 7: 175, 8: 175, 9: 25, 11: 175}
    per_tile = {5: 175, 13: 100, 14: 150 + 266 + 175 + 25}
    out = [0] * 15
    for k, v in per_h.items():
        out[k] = v * N
    for k, v in per_tile.items():
        out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--humans", type=int, default=5)
    ap.add_argument("--envs", type=int, default=4096)
    a = ap.parse_args()
    if "DIAGNOSTIC" not in _hip.version():
        raise SystemExit("needs the diagnostic build: MCN_HIP_LIB=.../build_diag/libmcn_hip.so")
    dev = torch.device("cuda", 0)
    E, N = a.envs, a.humans
    env, _ = bench.build_env(E, N, 0, dev)
    pol = bench._sarl_policy(dev, env.time_step)
    for _ in range(3):
        pol.predict_batch(env)
    torch.cuda.synchronize()
    buf = np.zeros((2048, 16), np.uint64)
    f = _hip.lib.mcn_debug_sarl_phases
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    assert f(buf.ctypes.data, buf.nbytes, 1) > 0          # read + reset
    iters = 5
    ms = bench._timed(lambda: pol.predict_batch(env), iters)          # _timed runs fn iters + 2 times
    torch.cuda.synchronize()
    assert f(buf.ctypes.data, buf.nbytes, 0) > 0
    tiles = (E * 81 + 15) // 16
    launches = iters + 2
    waves = min(2048, (tiles + 3) // 4 * 4)
    per_wave_tiles = tiles * launches / waves
    cyc = buf[:waves, :15].astype(np.float64) / per_wave_tiles          # cycles per tile, per wavefront
    med = np.median(cyc, 0)
    need = np.array(mfma_per_phase(N), np.float64) * 32
    print("SARL look-ahead %d x %d: %.3f ms per launch (diagnostic build), %d tiles, %.2f tiles per resident wavefront "
          "and launch" % (E, N, ms, tiles, tiles / waves))
    print("%-48s %12s %14s %8s" % ("phase", "cycles/tile", "MFMA x 32", "ratio"))
    for k in range(15):
        print("%-48s %12.0f %14.0f %8s" % (PHASES[k], med[k], need[k], ("%.2f" % (med[k] / need[k])) if need[k] else "-"))
    print("%-48s %12.0f %14.0f %8.2f" % ("total", med.sum(), need.sum(), med.sum() / need.sum()))
    print("no-MFMA phases together: %.0f cycles per tile = %.1f %% of the tile" % (
        med[need == 0].sum(), 100 * med[need == 0].sum() / med.sum()))


if __name__ == "__main__":
    main()
