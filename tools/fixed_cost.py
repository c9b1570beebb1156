#!/usr/bin/env python3
"""Where does the per-launch fixed cost of mcn_env_rollout go?  (GPU box)

  python tools/fixed_cost.py sweep            # product library: time per launch for T = 1..256, fit a + b*T
  MCN_HIP_LIB=modelcrowdnav_amd/csrc/build_diag/libmcn_hip.so python tools/fixed_cost.py stamps
                                              # diagnostic library: in-kernel time stamps per wavefront and step

`sweep` times R back-to-back launches replayed from one hipGraph (HIP events on the launch stream).
`stamps` reads the 100 MHz real-time counter values the diagnostic build's rollout kernel stores (kernel entry,
state loaded, end of each of the first 36 steps, exit) and prints, over all wavefronts: the spread of entry times,
the state-load time, the duration of each step index, and the exit spread.
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from modelcrowdnav_amd import _hip  # noqa: E402


def time_launches(env, acts, T, R, reps=5):
    """ms per launch of one T-step mcn_env_rollout, from a graph of R launches."""
    env.rollout(acts[:T])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(R):
            env.rollout(acts[:T])
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    per = []
    for _ in range(reps):
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        per.append(s.elapsed_time(e) / R)
    return float(np.median(per)), float(np.min(per))


def sweep(E=4096, N=5):
    dev = torch.device("cuda", 0)
    print(_hip.version())
    acts = bench.make_actions(256, E, E, 0, dev)
    for split in (1, 0):
        _hip.set_tuning(rollout_fused=1, rollout_split=split)
        env, _ = bench.build_env(E, N, 0, dev)
        env.rollout(acts[:200])             # spread the episodes' phases as in a running rollout
        xs, ys = [], []
        for T in (1, 2, 4, 8, 16, 20, 32, 64, 128, 256):
            med, best = time_launches(env, acts, T, R=max(4, 256 // T))
            xs.append(T); ys.append(med)
            print("split=%d T=%4d  %9.2f us/launch (best %9.2f)  %7.3f us/step" % (split, T, med * 1e3, best * 1e3, med * 1e3 / T))
        b, a = np.polyfit(xs[3:], ys[3:], 1)
        print("split=%d fit over T>=8: fixed %.2f us + %.3f us/step" % (split, a * 1e3, b * 1e3))
        del env
    _hip.set_tuning()


def stamps(E=4096, N=5, T=20):
    dev = torch.device("cuda", 0)
    print(_hip.version())
    if "DIAGNOSTIC" not in _hip.version():
        raise SystemExit("needs the diagnostic build: MCN_HIP_LIB=.../build_diag/libmcn_hip.so")
    fn = _hip.lib.mcn_debug_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    fn.restype = ctypes.c_int
    fc = _hip.lib.mcn_debug_counts
    fc.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    fc.restype = ctypes.c_int
    cnt = np.zeros(8192 * 4, dtype=np.uint32)
    acts = bench.make_actions(256, E, E, 0, dev)
    for split in (1, 0):
        _hip.set_tuning(rollout_fused=1, rollout_split=split)
        env, _ = bench.build_env(E, N, 0, dev)
        env.rollout(acts[:200])
        torch.cuda.synchronize()
        for trial in range(3):
            fc(cnt.ctypes.data, cnt.nbytes, 1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); env.rollout(acts[:T]); e.record(); torch.cuda.synchronize()
            buf = np.zeros(8192 * 40, dtype=np.uint64)
            n = fn(buf.ctypes.data, buf.nbytes)
            assert n > 0
            G = 64 // (4 * N)
            waves = ((E + G - 1) // G) * (2 if split else 1)
            st = buf.reshape(8192, 40)[:waves].astype(np.int64)
            t0 = st[:, 0].min()
            us = lambda x: x * 0.01                 # 100 MHz ticks -> us
            ent = us(st[:, 0] - t0)
            load = us(st[:, 1] - st[:, 0])
            ends = us(st[:, 39] - t0)
            print("split=%d trial %d: event time %.2f us; %d wavefronts" % (split, trial, s.elapsed_time(e) * 1e3, waves))
            print("   entry after first entry:  median %.2f  p99 %.2f  max %.2f us" % (np.median(ent), np.percentile(ent, 99), ent.max()))
            print("   state load:               median %.2f  p99 %.2f  max %.2f us" % (np.median(load), np.percentile(load, 99), load.max()))
            prev = st[:, 1]
            line = []
            for t in range(min(T, 36)):
                d = us(st[:, 2 + t] - prev)
                prev = st[:, 2 + t]
                line.append("%d:%.2f/%.2f" % (t, np.median(d), d.max()))
            print("   step durations median/max us: " + " ".join(line))
            print("   exit after first entry:   median %.2f  min %.2f  max %.2f us; last step end -> exit median %.2f us" % (
                np.median(ends), ends.min(), ends.max(), np.median(us(st[:, 39] - st[:, 2 + min(T, 36) - 1]))))
            # census: HW_ID bits wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID[3:0]
            hw, xcc = st[:, 38] & 0xffffffff, (st[:, 38] >> 32) & 0xf
            simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
            cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
            simd_key = cu_key * 4 + simd
            nsteps = min(T, 36)
            mean_step = us(st[:, 2 + nsteps - 1] - st[:, 1]) / nsteps
            uc, cc = np.unique(cu_key, return_counts=True)
            usd, inv, sc = np.unique(simd_key, return_inverse=True, return_counts=True)
            print("   census: %d CUs used, waves per CU min %d median %d max %d; %d SIMDs used, waves per SIMD histogram %s" % (
                len(uc), cc.min(), np.median(cc), cc.max(), len(usd), {int(a): int(b) for a, b in zip(*np.unique(sc, return_counts=True))}))
            pop = sc[inv]
            for k in np.unique(pop):
                m = pop == k
                print("   waves on a SIMD hosting %d: n=%d mean step %.3f us (min %.3f max %.3f), exit median %.1f us" % (
                    k, m.sum(), mean_step[m].mean(), mean_step[m].min(), mean_step[m].max(), np.median(ends[m])))
            cpop = cc[np.searchsorted(uc, cu_key)]
            for k in np.unique(cpop):
                m = cpop == k
                print("   waves on a CU hosting %d: n=%d mean step %.3f us, exit median %.1f max %.1f us" % (
                    k, m.sum(), mean_step[m].mean(), np.median(ends[m]), ends[m].max()))
            fc(cnt.ctypes.data, cnt.nbytes, 0)
            c4 = cnt.reshape(8192, 4)[:waves].astype(np.int64)
            if split:                       # a workgroup's two wavefronts wait for each other: add their counts
                c4 = c4[0::2] + c4[1::2]
                ms = np.maximum(mean_step[0::2], mean_step[1::2])
            else:
                ms = mean_step
            print("   per env group over %d steps: 3-D LP entries mean %.1f max %d; restarts mean %.2f; overlap sqrt %.2f; goal sqrt %.2f" % (
                nsteps, c4[:, 0].mean(), c4[:, 0].max(), c4[:, 1].mean(), c4[:, 2].mean(), c4[:, 3].mean()))
            A = np.column_stack([np.ones(len(ms)), c4[:, 0], c4[:, 1], c4[:, 2], c4[:, 3]]) 
            coef, *_ = np.linalg.lstsq(A, ms * nsteps, rcond=None)
            print("   least squares, us per launch = %.2f + %.3f * lp3 + %.3f * restarts + %.3f * overlap + %.3f * goal;  residual std %.2f us" % (
                *coef, np.std(ms * nsteps - A @ coef)))
            order = np.argsort(ms)
            for name, sel in (("fastest 5%", order[:len(order) // 20]), ("slowest 5%", order[-(len(order) // 20):])):
                print("   %s: mean step %.3f us, 3-D LP entries %.1f, restarts %.2f" % (name, ms[sel].mean(), c4[sel, 0].mean(), c4[sel, 1].mean()))
            if split:
                for role in (0, 1):
                    sel = st[role::2]
                    d = us(sel[:, 2 + min(T, 36) - 1] - sel[:, 1]) / min(T, 36)
                    print("   role %d: mean step %.3f us (median over wavefronts)" % (role, np.median(d)))
        del env
    _hip.set_tuning()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "sweep"
    {"sweep": sweep, "stamps": stamps}[mode]()
