"""bench.py's domain-decomposition leg, rehearsed on one GPU: BENCH_DD_SELF_LINKS=xyz gives the single rank a real halo (its own periodic
images), real RCCL groups and non-empty non-local lists, i.e. the code path the ranks of a multi-GPU run take — list upload (merged or
two localities), the C++ step, the pair count off the device lists, the first-step parity check against the single-domain forces and
the dd_leg_ok field at the top level of the line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("merged", ["1", "0"])
def test_decomposition_leg_rehearsal_with_self_links(merged):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(BENCH_DD_MERGED=merged, BENCH_DD_SELF_LINKS="xyz", BENCH_DD_CONDITION_STEPS="20")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dd", "--atoms", "24k", "--steps", "20", "--warmup", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    leg = rec["domain_decomposition"]
    assert rec["dd_leg_ok"] is True and rec["scaling"] == "strong"
    assert leg["parity_of_first_step"]["ok"] and leg["parity_of_first_step"]["max_err_over_tolerance"] < 1.0
    assert leg["halo_atoms_per_rank_mean"] > 0 and leg["cluster_pairs_all_ranks"] > 200000
    assert ("merged" in leg["schedule"]) == (merged == "1")
