"""`python bench.py --gpus N` has to start its N ranks itself (the driver calls exactly that).  BENCH_DRY_RUN=1 walks through the
launcher, the torch.distributed rendezvous (gloo here, RCCL on GPUs), the barriers and the one-JSON-line contract without any
GPU work."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None):
    env = dict(os.environ, BENCH_DRY_RUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


@pytest.mark.parametrize("n", [2, 3])
def test_launcher_starts_n_ranks_and_prints_one_line(n):
    p, lines = run_bench("--gpus", str(n), "--steps", "4", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["rccl_ranks"] == n and rec["steps"] == 4 and rec["warmup"] == 1
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in rec


def test_single_rank_needs_no_launcher():
    p, lines = run_bench("--gpus", "1", "--steps", "2", "--warmup", "0")
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(lines[0])["n_gpus"] == 1


def test_rank_count_mismatch_is_an_error():
    # a launcher that started 1 rank for --gpus 2 must not produce a line labelled 2
    p, lines = run_bench("--gpus", "2", env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not lines


@pytest.mark.parametrize("how", ["error", "wrong", "hang"])
def test_a_failed_decomposition_leg_is_not_a_green_run(how):
    """The leg failed, hung (watchdog) or computed other forces than the single domain: the complete line is still printed and
    forwarded by the launcher, with "dd_leg_ok": false — and the job's exit code is 3.  BENCH_DD_LENIENT=1 restores exit code 0."""
    p, lines = run_bench("--gpus", "2", "--steps", "2", "--warmup", "0", env_extra={"BENCH_DRY_RUN_DD_LEG": how})
    assert p.returncode != 0, "a failed leg exited 0"
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dd_leg_ok"] is False and rec["dd_leg_error"]
    assert "DOMAIN-DECOMPOSITION LEG FAILED" in p.stderr
    if how == "error":
        p, lines = run_bench("--gpus", "2", "--steps", "2", "--warmup", "0", env_extra={"BENCH_DRY_RUN_DD_LEG": how, "BENCH_DD_LENIENT": "1"})
        assert p.returncode == 0 and json.loads(lines[0])["dd_leg_ok"] is False


def test_decomposition_leg_state_is_at_the_top_level_of_the_line(capsys):
    """A failed, timed-out or wrong (first step differs from the single-domain forces) decomposition leg must show as "dd_leg_ok": false
    with its reason at the TOP level of the JSON line and on stderr — not only inside the nested record."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    out = {}
    bench.mark_dd_leg(out, {"error": "no result within 150 s"})
    assert out["dd_leg_ok"] is False and "150 s" in out["dd_leg_error"]
    assert "DOMAIN-DECOMPOSITION LEG FAILED" in capsys.readouterr().err
    out = {}
    bench.mark_dd_leg(out, None)          # (a rank other than 0 hands None: never printed, but must not raise)
    assert out["dd_leg_ok"] is False
    out = {}
    bench.mark_dd_leg(out, {"ms_per_step": 0.1, "parity_of_first_step": {"ok": False, "max_err_over_tolerance": 37.0}})
    assert out["dd_leg_ok"] is False and "single-domain" in out["dd_leg_error"]
    out = {}
    bench.mark_dd_leg(out, {"ms_per_step": 0.1, "parity_of_first_step": {"ok": True, "max_err_over_tolerance": 0.01}})
    assert out["dd_leg_ok"] is True and out["dd_leg_error"] is None
