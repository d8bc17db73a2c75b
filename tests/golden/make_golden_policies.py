"""Build-container only: golden vectors for the heuristic POLICIES.

The REAL reference agents (agents/dqn/dqn_evaluation_results/greedy_agents.py: NearestSensorGreedy,
MaxThroughputGreedyV2) drive the REAL reference environment under the noise tape (slot zP = the
agent's own is_in_range() samples).  Recorded: the action the agent chose at every step + the step's
observation / reward / truncation.  The C oracle's restatement of the policies is checked alongside;
seeds on which the reference trajectory falls inside its own 1-ulp log10 fuzz are skipped as in
make_golden.py.  Writes tests/golden/policy_*.npz.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import check_oracle_vs_reference as K  # noqa: E402
import ref_harness as R  # noqa: E402
import tape as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

POLICY_ID = {"nearest": O.POLICY_NEAREST, "max_throughput_v2": O.POLICY_MAX_THROUGHPUT_V2}
CASES = [
    dict(name="policy_nearest_n5_g1000", kind="nearest", n=5, grid=(1000, 1000), steps=600, kw={}),
    dict(name="policy_nearest_n10_g300_short", kind="nearest", n=10, grid=(300, 300), steps=300,
         kw=dict(max_steps=120, sensor_duty_cycle=40.0, uav_start_position=(290.0, 5.0))),
    dict(name="policy_v2_n20_g500", kind="max_throughput_v2", n=20, grid=(500, 500), steps=420, kw={}),
    dict(name="policy_v2_n50_g400_lowbatt", kind="max_throughput_v2", n=50, grid=(400, 400), steps=260,
         kw=dict(max_battery=30.0, sensor_duty_cycle=30.0, uav_start_position=(200.0, 10.0))),
    dict(name="policy_v2_n10_g1000", kind="max_throughput_v2", n=10, grid=(1000, 1000), steps=400,
         kw=dict(max_steps=300)),
]


def record(case, tape_seed):
    n, grid, steps, kw = case["n"], case["grid"], case["steps"], case["kw"]
    cfg = K.oracle_config(n, grid, kw)
    pid = POLICY_ID[case["kind"]]
    obs_l, rew_l, tr_l, act_l, reset_obs = [], [], [], [], []
    with R.TapedReference(n, grid, tape_seed, **kw) as ref:
        orc = O.OracleEnv(cfg, 0, ref.pos_x, ref.pos_y)
        agent = ref.make_agent(case["kind"])
        episode = 0
        ro, _ = ref.reset()
        oo = orc.reset_tape(T.reset_tape(tape_seed, 0, episode, n))
        assert np.array_equal(ro, oo)
        reset_obs.append(ro)
        for s in range(steps):
            tp = T.step_tape(tape_seed, 0, s, n)
            ra, (ro, rr, _, rtr, _) = ref.policy_step(agent, ro)
            oa, oo, orr, otr = orc.step_policy_tape(pid, tp)
            bad = K.compare_states(ref.state(), orc.state(), s)
            if ra != oa or bad or not np.array_equal(ro, oo) or rr != orr or rtr != otr:
                return None, dict(step=s, ref_action=ra, oracle_action=oa, keys=[b[1] for b in bad])
            obs_l.append(ro); rew_l.append(rr); tr_l.append(rtr); act_l.append(ra)
            if rtr:
                episode += 1
                ro, _ = ref.reset()
                oo = orc.reset_tape(T.reset_tape(tape_seed, 0, episode, n))
                assert np.array_equal(ro, oo)
                reset_obs.append(ro)
                agent = ref.make_agent(case["kind"])       # a fresh agent per episode, like dqn.py:517
        final = ref.state()
    out = dict(meta=np.array(json.dumps(dict(name=case["name"], policy=case["kind"], policy_id=pid, n=n, grid=list(grid),
                                             steps=steps, tape_seed=tape_seed, sigma=None, kwargs=kw))),
               actions=np.array(act_l, np.int8), obs=np.array(obs_l, np.float32), reward=np.array(rew_l, np.float64),
               truncated=np.array(tr_l, np.uint8), reset_obs=np.array(reset_obs, np.float32))
    for k, v in final.items():
        out["final_" + k] = np.asarray(v)
    return out, None


if __name__ == "__main__":
    skipped = []
    for ci, case in enumerate(CASES):
        for attempt in range(8):
            seed = 515100 + 100 * ci + attempt
            out, div = record(case, seed)
            if out is not None:
                break
            skipped.append(dict(case=case["name"], tape_seed=seed, **div))
            print("  skipped", case["name"], seed, div)
        else:
            raise RuntimeError("no reproducible seed for " + case["name"])
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{case['name']}: seed {seed} actions {np.bincount(out['actions'], minlength=5).tolist()} "
              f"truncations {int(out['truncated'].sum())} collected {float(out['final_total_collected']):.0f} B, "
              f"{os.path.getsize(path)} bytes")
    json.dump(dict(skipped=skipped), open(os.path.join(HERE, "manifest_policies.json"), "w"), indent=1)
