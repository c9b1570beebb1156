"""GPU box, diagnostic library only: the clock the chip holds inside mcn::sgan_pool_kernel.

    MCN_HIP_LIB=modelcrowdnav_amd/csrc/build_diag/libmcn_hip.so python tools/pool_clock.py

Runs the SGAN step back to back for a few seconds (4096 scenes x 10 pedestrians, shipped pool-net weights), then reads the
(s_memtime, s_memrealtime) pairs each workgroup stored around its unit loop: in-kernel clock = d(shader cycles) /
d(100 MHz ticks) x 100 MHz (MI355X_MICROARCH.md, 'DVFS give-back' item 6)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from modelcrowdnav_amd import _hip  # noqa: E402
from modelcrowdnav_amd.policy.world_model import VecSGANWorld, generator_from_arrays  # noqa: E402


def main():
    fn = getattr(_hip.lib, "mcn_debug_pool_clock", None)
    if fn is None:
        sys.exit("needs the diagnostic library: make -C modelcrowdnav_amd/csrc diag; MCN_HIP_LIB=.../build_diag/libmcn_hip.so")
    fn.argtypes = [C.c_void_p, C.c_int64]
    fn.restype = C.c_int
    dev = torch.device("cuda", 0)
    E, N = 4096, 10
    gen = generator_from_arrays(np.load(os.path.join(ROOT, "tests", "golden", "g6_sgan.npz")), "p", dev)
    world = VecSGANWorld(gen, E, N, dev, time_step=0.25, seed=0)
    world.fixed_noise = torch.randn(E, 8, device=dev)
    g = torch.Generator().manual_seed(0)
    pos = (torch.rand(E, N, 2, dtype=torch.float64, generator=g) * 8 - 4).to(dev)
    world.init_constant_velocity(pos, (torch.rand(E, N, 2, dtype=torch.float64, generator=g) - 0.5).to(dev))
    t0 = time.time()
    while time.time() - t0 < 3.0:
        for _ in range(200):
            world(pos)
        torch.cuda.synchronize()
    W = int(os.environ.get("MCN_POOL_WAVES", "16"))          # must match the library's build
    buf = np.zeros((256 * W, 8), np.uint64)
    n = fn(buf.ctypes.data, buf.nbytes)
    assert n > 0
    buf = buf.astype(np.int64).reshape(256, W, 8)
    t0 = buf[:, :, 1].min()
    cyc, ticks = buf[:, :, 6] - buf[:, :, 0], buf[:, :, 7] - buf[:, :, 1]
    ghz = cyc / ticks * 0.1
    print("in-kernel clock %.3f GHz (min %.3f max %.3f)" % (float(np.median(ghz)), ghz.min(), ghz.max()))
    us = lambda a: (a - t0) / 100.0
    ent, fill, end = us(buf[:, :, 1]), us(buf[:, :, 2]), us(buf[:, :, 7])
    print("entry  %6.1f .. %6.1f us   LDS fill done %6.1f .. %6.1f (median %.1f)   exit %6.1f .. %6.1f (median %.1f)" % (
        ent.min(), ent.max(), fill.min(), fill.max(), np.median(fill), end.min(), end.max(), np.median(end)))
    u1 = (buf[:, :, 3] - buf[:, :, 2]) / 100.0
    pro, loop, epi = (buf[:, :, 4] - buf[:, :, 2]) / 100.0, (buf[:, :, 5] - buf[:, :, 4]) / 100.0, (buf[:, :, 3] - buf[:, :, 5]) / 100.0
    print("first unit %.1f us median (%.1f .. %.1f) = inputs + embeddings %.1f, hidden-tile loop %.1f, scan + atomics %.1f" % (
        np.median(u1), u1.min(), u1.max(), np.median(pro), np.median(loop), np.median(epi)))
    wg_end = end.max(axis=1)
    print("workgroup finish: median %.1f, 90 %% %.1f, max %.1f us" % (np.median(wg_end), np.percentile(wg_end, 90), wg_end.max()))

if __name__ == "__main__":
    main()
