"""CPU: bench.py's roofline bookkeeping against the committed PMC summaries (profiles/rNN_pmc_*.json) -- the fields
the driver's JSON line must carry (`roofline.traffic` for whatever launch length ran, `roofline_valu`)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_follow_survey_8d():
    import bench
    assert bench.algorithmic_bytes_per_env_step(5) == 622 and bench.algorithmic_bytes_per_env_step(10) == 1102
    assert bench.pairwise_bytes_per_env_step(5) == 566


def test_traffic_lookup_exact_and_affine_in_launch_length():
    import bench
    roll = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_env_rollout.json")))["kernels"]
    by_t = {k["steps_per_launch"]: k["traffic_bytes_per_launch"] for k in roll}
    assert set(by_t) >= {20, 100, 1000}
    t20, src = bench.pmc_traffic(4096, False, 20)
    assert t20 == by_t[20] and "r02_pmc_env_rollout" in src
    t50, src50 = bench.pmc_traffic(4096, False, 50)            # between two measured launch lengths
    assert by_t[20] < t50 < by_t[100] and "affine" in src50
    want = by_t[20] + (by_t[100] - by_t[20]) * (50 - 20) / 80
    assert abs(t50 - want) <= 1
    t1, src1 = bench.pmc_traffic(4096, False, 1)               # one mcn_env_step launch per step (quad kernel)
    assert t1 and "pmc_env_step" in src1
    assert bench.pmc_traffic(12345, False, 1) == (None, None)   # no such run: null, not a guess
    # real traffic of a fused launch is a fraction of the algorithmic bytes (state stays in registers)
    assert t20 < 0.5 * bench.algorithmic_bytes_per_env_step(5) * 4096 * 20


def test_roofline_entries_carry_traffic_and_valu():
    import bench
    e = bench.roofline_entry(4096, 5, 0.0745, steps_per_launch=20)
    assert e["traffic"] and e["kernel"] == "mcn::env_rollout_quad_kernel" and e["bound"] == "hbm"
    assert abs(e["achieved"] - 622 * 4096 * 20 / 74.5e-6 / 1e9) < 1.0 and e["frac"] == round(e["achieved"] / 8000.0, 5)
    v = bench.valu_roofline(4096, 5, 0.0745, 20, rollout=True)
    assert v["bound"] == "valu-issue" and 0.05 < v["frac"] < 1.0 and v["peak"] == 614.4
    sq = {k["kind"]: k for k in json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_sq.json")))["kernels"]}
    assert abs(v["valu_wave_instructions_per_env_step"] - sq["rollout"]["valu_per_env_step"]) < 0.01
    f = bench.valu_roofline(1 << 20, 5, 0.2, 1, rollout=False)
    p = bench.valu_roofline(1 << 20, 5, 0.13, 1, rollout=False, given=True)
    assert f["kernel"].startswith("mcn::env_step_kernel") and p["kernel"].startswith("mcn::env_pair_kernel")
    assert bench.valu_roofline(777, 5, 1.0, 1, rollout=False) is None
