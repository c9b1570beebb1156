"""ORACLE -- test infrastructure only.  One CPU replica of the benchmark workload for bench.py's cpu_baseline leg:
the C oracle stepping 4096 envs x N humans with auto-reset, started as a child process (numpy + ctypes only, no
torch, no GPU) so that replicas on several host cores really run side by side.

    python -m oracle.cpu_replica <workload.npz> <seconds> <seed> <start_at_unix_time>

workload.npz: sc [E,N,9] scenarios, tab [81,2] action table.  Prints "env_steps elapsed_s steps".
"""
import sys
import time

import numpy as np

from oracle import cport


def setup(sc, tab, seed):
    E, N = sc.shape[0], sc.shape[1]
    st = cport.EnvState(E, N)
    st.hpx[:], st.hpy[:], st.hgx[:], st.hgy[:] = sc[..., 0], sc[..., 1], sc[..., 2], sc[..., 3]
    st.hr[:], st.hvpref[:] = sc[..., 7], sc[..., 8]
    st.rpy[:], st.rgy[:], st.rr[:] = -4.0, 4.0, 0.3
    cfg = cport.default_cfg()
    cport.env_step(cfg, st, np.zeros(E), np.zeros(E))      # warm
    return dict(E=E, st=st, fresh=st.copy(), cfg=cfg, tab=tab, rng=np.random.RandomState(seed))


def run(w, seconds):
    E, st, fresh = w["E"], w["st"], w["fresh"]
    steps, t0 = 0, time.perf_counter()
    while True:
        a = w["tab"][w["rng"].randint(0, 81, E)]
        o = cport.env_step(w["cfg"], st, np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1]))
        d = o["done"].astype(bool)
        if d.any():        # auto-reset like the GPU run
            for k in cport.EnvState.FIELDS_H + cport.EnvState.FIELDS_R + ("gtime", "human_times"):
                getattr(st, k)[d] = getattr(fresh, k)[d]
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds or steps >= 20000:
            break
    return E * steps, el, steps


if __name__ == "__main__":
    z = np.load(sys.argv[1])
    w = setup(z["sc"], z["tab"], int(sys.argv[3]))
    while time.time() < float(sys.argv[4]):
        time.sleep(0.005)
    print("%d %.6f %d" % run(w, float(sys.argv[2])))
