"""Generate the committed golden fixtures from the REAL reference (build container only).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz + manifest.json

For every case the real reference environment (/root/reference/src/environment/uav_env.py, imported
behind tests/oracle_stub/gymnasium) is stepped under the integer noise tape of tests/tape.py with
auto-reset on truncation (what SB3's DummyVecEnv does around it).  Recorded per case:

    inputs : reference constructor kwargs, N, grid, tape_seed, sigma, actions (int8[steps])
    outputs: reset_obs float32[episodes, D], obs float32[steps, D] (the step's own observation, i.e.
             the terminal one on a truncated step), reward float64[steps], truncated uint8[steps],
             sf int8[steps, N], final per-sensor / per-env state

A fixture is data only.  The tape itself is not stored (tests/tape.py regenerates it bit-exactly
from `tape_seed`), sensor positions likewise.

Seeds: the reference's float32 log10/pow are platform dependent at the 1-ulp level (see
oracle/uavenv_oracle.h), so on rare seeds a threshold compare lands inside that fuzz and the
reference trajectory is not reproducible by ANY idealised restatement.  Such seeds are detected by
running the C oracle alongside; they are skipped only when the first divergence is a hard compare
whose margin is below 1e-4 dB, and are listed in manifest.json ("fuzz_skipped").
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

CASES = []
for n in (10, 20, 50):
    for g in (250, 500, 1000):
        centre = (float(g // 2), float(g // 2))
        CASES.append(dict(name=f"base_n{n}_g{g}", n=n, grid=(g, g), steps=160 if n == 50 else 240,
                          p_collect=0.25, kw=dict(uav_start_position=centre) if g != 500 else {}))
for i, c in enumerate(K.STRESS):
    d = dict(c)
    d["name"] = f"stress{i}_n{c['n']}"
    if c["n"] == 50:
        d["steps"] = 160
    CASES.append(d)


def threshold_margin(ref_state, orc_state, cfg):
    """Smallest distance of the reference's EMA RSSI to an SF threshold over sensors whose SF differs."""
    thr = np.array(list(cfg.sf_thresholds))
    m = np.inf
    for i in np.nonzero(ref_state["sf"] != orc_state["sf"])[0]:
        m = min(m, float(np.min(np.abs(ref_state["avg_rssi"][i] - thr))))
    return m


def record(case, tape_seed):
    n, grid, steps = case["n"], case["grid"], case["steps"]
    kw, sigma = case["kw"], case.get("sigma")
    cfg = K.oracle_config(n, grid, kw)
    if sigma is not None:
        cfg.shadowing_std_db = sigma
    acts = T.actions(tape_seed, 0, steps, case["p_collect"])
    obs, rew, trunc, sfs, reset_obs = [], [], [], [], []
    with R.TapedReference(n, grid, tape_seed, sigma=sigma, **kw) as ref:
        orc = O.OracleEnv(cfg, 0, ref.pos_x, ref.pos_y)
        episode = 0
        ro, _ = ref.reset()
        oo = orc.reset_tape(T.reset_tape(tape_seed, 0, episode, n))
        assert np.array_equal(ro, oo)
        reset_obs.append(ro)
        for s, a in enumerate(acts):
            ro, rr, rte, rtr, _ = ref.step(a)
            assert rte is False
            oo, orr, otr = orc.step_tape(a, T.step_tape(tape_seed, 0, s, n))
            rs, os_ = ref.state(), orc.state()
            bad = K.compare_states(rs, os_, s)
            if bad or not np.array_equal(ro, oo) or rr != orr or rtr != otr:
                margin = threshold_margin(rs, os_, cfg)
                return None, dict(step=s, margin_db=margin, keys=[b[1] for b in bad])
            obs.append(ro); rew.append(rr); trunc.append(rtr); sfs.append(rs["sf"].astype(np.int8))
            if rtr:
                episode += 1
                ro, _ = ref.reset()
                oo = orc.reset_tape(T.reset_tape(tape_seed, 0, episode, n))
                assert np.array_equal(ro, oo)
                reset_obs.append(ro)
        final = ref.state()
    out = dict(
        meta=np.array(json.dumps(dict(name=case["name"], n=n, grid=list(grid), steps=steps, tape_seed=tape_seed,
                                      p_collect=case["p_collect"], sigma=sigma, kwargs=kw))),
        actions=acts, obs=np.array(obs, np.float32), reward=np.array(rew, np.float64),
        truncated=np.array(trunc, np.uint8), sf=np.array(sfs, np.int8), reset_obs=np.array(reset_obs, np.float32),
    )
    for k, v in final.items():
        out["final_" + k] = np.asarray(v)
    return out, None


def main():
    manifest = dict(numpy=np.__version__, cases=[], fuzz_skipped=[])
    for f in os.listdir(HERE):
        if f.endswith(".npz") and not f.startswith(("policy_", "domainrand_")):     # the other generators' files stay
            os.remove(os.path.join(HERE, f))
    for ci, case in enumerate(CASES):
        for attempt in range(8):
            tape_seed = 424200 + 100 * ci + attempt
            out, div = record(case, tape_seed)
            if out is not None:
                break
            assert div["margin_db"] < 1e-4, ("NOT a fuzz flip -- oracle bug?", case["name"], tape_seed, div)
            manifest["fuzz_skipped"].append(dict(case=case["name"], tape_seed=tape_seed, **div))
            print("  fuzz-skipped", case["name"], tape_seed, div)
        else:
            raise RuntimeError("no reproducible seed for " + case["name"])
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **out)
        ntr = int(out["truncated"].sum())
        coll = float(out["final_total_collected"])
        manifest["cases"].append(dict(name=case["name"], tape_seed=tape_seed, steps=case["steps"], truncations=ntr,
                                      bytes_collected_last_episode=coll, file_bytes=os.path.getsize(path)))
        print(f"{case['name']}: seed {tape_seed}, {ntr} truncations, collected {coll:.1f} B, {os.path.getsize(path)} bytes")
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("total bytes", sum(c["file_bytes"] for c in manifest["cases"]), "skipped", len(manifest["fuzz_skipped"]))


if __name__ == "__main__":
    main()
