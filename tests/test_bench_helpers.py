"""CPU: bench.py's roofline bookkeeping against the committed PMC summaries (profiles/rNN_pmc_*.json) -- the fields
the driver's JSON line must carry (`roofline.traffic` for whatever launch length ran, `roofline_valu`), and the rule
that a traffic figure is only ever attached to the kernel it was measured on, from the latest round's files."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest(pattern):
    files = glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern))
    last = max(re.match(r"r(\d\d)_", os.path.basename(f)).group(1) for f in files)
    return [f for f in files if os.path.basename(f).startswith("r%s_" % last)], "r%s" % last


def test_algorithmic_bytes_follow_survey_8d():
    import bench
    assert bench.algorithmic_bytes_per_env_step(5) == 622 and bench.algorithmic_bytes_per_env_step(10) == 1102
    assert bench.pairwise_bytes_per_env_step(5) == 566


def test_kernel_classification_reads_template_arguments():
    import bench
    c = bench.classify_kernel
    assert c("mcn::env_pair_kernel<5>") == ("env_pair_kernel", 5, "given", 1)
    assert c("mcn::env_pair_kernel<10>") == ("env_pair_kernel", 10, "given", 1)
    assert c("mcn::env_step_kernel<256, 5, 0, 0, 2>") == ("env_step_kernel", 5, "orca", 1)       # HH = 2 is not MODE
    assert c("mcn::env_step_kernel<64, 0, 0, 2, 0>", 5) == ("env_step_kernel", 5, "given", 1)
    assert c("mcn::env_step_kernel<64, 10, 0, 0, 2>") == ("env_step_kernel", 10, "orca", 1)
    assert c("mcn::env_step_quad_kernel<5, 0, false>") == ("env_step_quad_kernel", 5, "orca", 4)
    assert c("mcn::env_rollout_quad_kernel<5, 0, false, true>") == ("env_rollout_quad_kernel", 5, "orca", 8)
    assert c("mcn::sarl_value_kernel", 5)[0] == "sarl_value_kernel"
    # the dispatcher's choices (csrc/mcn_api.hip, env_step.hip)
    e = bench.expected_kernel
    assert e(4096, 5, False) == "env_step_quad_kernel" and e(1 << 20, 5, False) == "env_step_kernel"
    assert e(4096, 5, True) == "env_step_kernel" and e(1 << 20, 5, True) == "env_pair_kernel"
    # multi-step launches: the quad rollout kernel up to 5 humans, the looped one-wavefront step kernel for 6-10 humans
    # while the batch is latency-bound (launch_env_step_loop), T step launches beyond
    assert c("mcn::env_step_loop_kernel<10, 0>") == ("env_step_loop_kernel", 10, "orca", 1)
    assert e(4096, 5, False, 1000) == "env_rollout_quad_kernel"
    assert e(4096, 10, False, 500) == "env_step_loop_kernel" and e(16384, 10, False, 500) == "env_step_loop_kernel"
    assert e(32768, 10, False, 500) == "env_step_kernel"          # 5462 wavefronts: throughput-bound, T launches
    assert e(4096, 7, False, 100) == "env_step_loop_kernel" and e(65536, 7, False, 100) == "env_step_kernel"
    assert e(1 << 18, 10, False, 100) == "env_step_kernel"
    assert e(4096, 10, False) == "env_step_kernel" and e(4096, 5, False, 20) == "env_rollout_quad_kernel"


def test_traffic_lookup_exact_and_affine_in_launch_length():
    import bench
    files, tag = _latest("pmc_env_rollout.json")
    roll = json.load(open(files[0]))["kernels"]
    by_t = {k["steps_per_launch"]: k["traffic_bytes_per_launch"] for k in roll}
    assert set(by_t) >= {20, 100, 1000}
    t20, src, kn = bench.pmc_traffic(4096, False, 20)
    assert t20 == by_t[20] and tag + "_pmc_env_rollout" in src and "env_rollout_quad_kernel" in kn
    t50, src50, _ = bench.pmc_traffic(4096, False, 50)            # between two measured launch lengths
    assert by_t[20] < t50 < by_t[100] and "affine" in src50
    want = by_t[20] + (by_t[100] - by_t[20]) * (50 - 20) / 80
    assert abs(t50 - want) <= 1
    t1, src1, kn1 = bench.pmc_traffic(4096, False, 1)             # one mcn_env_step launch per step (quad kernel)
    assert t1 and "pmc_env_step" in src1 and "env_step_quad_kernel" in kn1
    assert bench.pmc_traffic(12345, False, 1) == (None, None, None)   # no such run: null, not a guess
    # real traffic of a fused launch is a fraction of the algorithmic bytes (state stays in registers)
    assert t20 < 0.5 * bench.algorithmic_bytes_per_env_step(5) * 4096 * 20


def test_traffic_is_only_attached_to_the_kernel_it_was_measured_on():
    """the pair kernel's counters once landed on the fused-ORCA rows.  Every row's traffic must
    come from a same-round file entry whose kernel family, mode, humans and envs are the row's own."""
    import bench
    files, tag = _latest("pmc_env_*.json")
    entries = [k for f in files for k in json.load(open(f))["kernels"]]
    for E, N, given, spl in [(4096, 5, False, 1), (1 << 20, 5, False, 1), (1 << 20, 5, True, 1), (4096, 5, True, 1),
                             (4096, 5, False, 20), (4096, 10, False, 1), (1 << 18, 10, False, 1), (1 << 16, 5, False, 1)]:
        row = bench.roofline_entry(E, N, 0.1, given=given, steps_per_launch=spl)
        if row["traffic"] is None:
            assert row["traffic_source"] is None and row["traffic_kernel"] is None
            continue
        assert tag + "_" in row["traffic_source"], row["traffic_source"]
        fam, n, mode, _ = bench.classify_kernel(row["traffic_kernel"], N)
        assert "mcn::" + fam == row["kernel"] and n == N and mode == ("given" if given else "orca")
        if "affine" not in row["traffic_source"]:
            src = [k for k in entries if k["kernel"] == row["traffic_kernel"] and k["envs"] == E
                   and int(k.get("steps_per_launch", 1)) == spl]
            assert src and src[0]["traffic_bytes_per_launch"] == row["traffic"]
    # the two 2^20 rows are different kernels with different traffic
    a = bench.roofline_entry(1 << 20, 5, 0.2)
    b = bench.roofline_entry(1 << 20, 5, 0.13, given=True)
    if a["traffic"] and b["traffic"]:
        assert a["traffic_kernel"] != b["traffic_kernel"] and a["traffic"] != b["traffic"]


def test_roofline_entries_carry_traffic_and_valu():
    import bench
    e = bench.roofline_entry(4096, 5, 0.0745, steps_per_launch=20)
    assert e["traffic"] and e["kernel"] == "mcn::env_rollout_quad_kernel" and e["bound"] == "hbm"
    assert abs(e["achieved"] - 622 * 4096 * 20 / 74.5e-6 / 1e9) < 1.0 and e["frac"] == round(e["achieved"] / 8000.0, 5)
    v = bench.valu_roofline(4096, 5, 0.0745, 20, rollout=True)
    cyc, src = bench.valu_cycles_per_instruction()
    assert v["bound"] == "valu-issue" and 0.02 < v["frac"] < 1.0 and abs(v["peak"] - 1024 * 2.4 / cyc) < 0.1
    assert v["peak_cycles_per_instruction"] == cyc and 1.9 <= cyc <= 4.1
    files, _ = _latest("pmc_sq.json")
    sq = {(bench.classify_kernel(k["kernel"], k["humans"])[0], k["envs"], k["humans"]): k
          for k in json.load(open(files[0]))["kernels"]}
    assert abs(v["valu_wave_instructions_per_env_step"] - sq[("env_rollout_quad_kernel", 4096, 5)]["valu_per_env_step"]) < 0.01
    f = bench.valu_roofline(1 << 20, 5, 0.2, 1, rollout=False)
    p = bench.valu_roofline(1 << 20, 5, 0.13, 1, rollout=False, given=True)
    assert f["kernel"].startswith("mcn::env_step_kernel") and p["kernel"].startswith("mcn::env_pair_kernel")
    assert bench.valu_roofline(777, 5, 1.0, 1, rollout=False) is None


def test_pmc_summariser_classifies_by_template_arguments(tmp_path):
    """tools/pmc_summary.py on a synthetic pair of counter passes: the streaming pair kernel (whose name carries no MODE
    argument) is given-velocity traffic with the 566-B accounting, `env_step_kernel<256, 5, 0, 0, 2>` (HH = 2, MODE = 0)
    is fused-ORCA traffic, and each launch is matched to the batch size its grid covers."""
    import csv
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import pmc_summary
    hdr = ["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"]
    rows = [("void mcn::env_pair_kernel<5, true>(mcn::StepParams)", 5592576, 256),
            ("void mcn::env_step_kernel<256, 5, 0, 0, 2>(mcn::StepParams)", 5592576, 256),
            ("void mcn::env_step_kernel<64, 0, 0, 2, 0>(mcn::StepParams)", 21888, 64),
            ("void mcn::env_step_quad_kernel<5, 0, false>(mcn::StepParams)", 87424, 64),
            ("void at::native::something(int)", 1024, 256)]
    for ctr, d, val in (("FETCH_SIZE", "f", 1000.0), ("WRITE_SIZE", "w", 500.0)):
        os.makedirs(tmp_path / d)
        with open(tmp_path / d / "x_counter_collection.csv", "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(hdr)
            for name, grid, wg in rows:
                for rep in range(2):
                    wr.writerow([name, grid, wg, ctr, val + rep])
    alg = lambda n, given: bench.pairwise_bytes_per_env_step(n) if given else bench.algorithmic_bytes_per_env_step(n)
    out = pmc_summary.summarise((str(tmp_path / "f"), str(tmp_path / "w"), 5, [4096, 1048576], 1), alg)
    by = {k["kernel"]: k for k in out}
    assert len(out) == 4                                                     # the torch kernel is dropped
    pair = by["mcn::env_pair_kernel<5, true>"]
    assert (pair["family"], pair["mode"], pair["humans"], pair["envs"]) == ("env_pair_kernel", "given", 5, 1048576)
    assert pair["algorithmic_bytes_per_launch"] == 566 * 1048576
    fused = by["mcn::env_step_kernel<256, 5, 0, 0, 2>"]
    assert (fused["family"], fused["mode"], fused["envs"]) == ("env_step_kernel", "orca", 1048576)
    assert fused["algorithmic_bytes_per_launch"] == 622 * 1048576
    small = by["mcn::env_step_kernel<64, 0, 0, 2, 0>"]
    assert (small["mode"], small["humans"], small["envs"]) == ("given", 5, 4096)
    assert by["mcn::env_step_quad_kernel<5, 0, false>"]["envs"] == 4096
    # reads doubled (gfx950 FETCH_SIZE), writes as they are; KB -> bytes; mean of the two launches
    assert pair["traffic_bytes_per_launch"] == int(2 * 1000.5 * 1024 + 500.5 * 1024)


def test_python_scalar_baseline_step_equals_the_c_oracle():
    """bench.py's `cpu_baseline.python_scalar` times tools/py_scalar_step.py (an object-per-agent Python step in the
    shape of crowd_sim.py:331-434 over the oracle's ORCA solve): before timing it, pin it -- 3 envs x 5 humans x 110
    steps (collisions, goals, time-outs with auto-reset) against oracle/mcn_oracle.c's batched step, all outputs exact."""
    import numpy as np
    from oracle import cport
    from tools import py_scalar_step as P
    import importlib.util
    spec = importlib.util.spec_from_file_location("mcn_scen", os.path.join(ROOT, "modelcrowdnav_amd", "envs", "scenarios.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    pool = S.scenario_pool(S.ScenarioSpec(), "test", range(3), 5, "circle_crossing")
    st, cfg = cport.EnvState(3, 5), cport.default_cfg()
    st.hpx[:], st.hpy[:], st.hgx[:], st.hgy[:] = pool[..., 0], pool[..., 1], pool[..., 2], pool[..., 3]
    st.hr[:], st.hvpref[:] = pool[..., 7], pool[..., 8]
    st.rpy[:], st.rgy[:], st.rr[:] = -4.0, 4.0, 0.3
    fresh = st.copy()
    envs = [P.ScalarCrowdSim(pool[e], cfg) for e in range(3)]
    rng = np.random.RandomState(3)
    seen = set()
    for t in range(110):
        a = rng.uniform(-1, 1, (3, 2)) * (0.2 if t % 7 else 1.0)
        a[0] = (0.0, 1.0)                                       # env 0 walks to its goal through the crossing
        ref = cport.env_step(cfg, st, a[:, 0].copy(), a[:, 1].copy())
        for e, env in enumerate(envs):
            ob, reward, done, info, overlaps = env.step(P.ActionXY(a[e, 0], a[e, 1]))
            assert reward == ref["reward"][e] and bool(done) == bool(ref["done"][e]) and info == ref["info"][e], (t, e)
            assert overlaps == ref["hh_count"][e]
            assert [h.px for h in env.humans] == list(st.hpx[e]) and [h.vy for h in env.humans] == list(st.hvy[e])
            assert (env.robot.px, env.robot.py) == (st.rpx[e], st.rpy[e]) and env.global_time == st.gtime[e]
            assert env.human_times == list(st.human_times[e])
            seen.add(int(info))
            if done:
                env.reset()
                for k in cport.EnvState.FIELDS_H + cport.EnvState.FIELDS_R + ("gtime", "human_times"):
                    getattr(st, k)[e] = getattr(fresh, k)[e]
    assert {P.TIMEOUT, P.NOTHING} <= seen and len(seen) >= 4, seen
