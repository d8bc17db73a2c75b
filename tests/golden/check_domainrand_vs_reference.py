"""Build-container-only: how often does a run of the REAL DomainRandEnv (make_golden_domainrand.py's harness, keyed noise) differ
from the C oracle?  Runs every fixture case on fresh seeds and reports identical / diverged runs with the margin of the first
divergence.   python tests/golden/check_domainrand_vs_reference.py [seeds_per_case]

Not collected by pytest (no test_ prefix): /root/reference does not exist on the GPU box."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_domainrand as M  # noqa: E402

if __name__ == "__main__":
    per_case = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    total = bad = 0
    for ci, case in enumerate(M.CASES):
        for k in range(per_case):
            seed, env_index = 555000 + 37 * ci + 1009 * k, 11 + 5 * ci + k
            out, div = M.record(case, seed, env_index)
            total += 1
            if out is None:
                bad += 1
                print(f"{case['name']} seed {seed} env {env_index}: DIVERGED {div}")
            else:
                print(f"{case['name']} seed {seed} env {env_index}: identical ({int(out['truncated'].sum())} finished episodes, "
                      f"grids {sorted(set(int(g[0]) for g in out['ep_grid']))})", flush=True)
    print(f"{bad} of {total} runs diverged")
