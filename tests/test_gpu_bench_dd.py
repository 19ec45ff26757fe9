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


@pytest.mark.parametrize("ncells,merged", [((2, 2, 2), "1"), ((2, 2, 2), "0"), ((4, 2, 1), "1"), ((2, 1, 1), "1")])
def test_decomposition_leg_with_several_ranks_as_threads(ncells, merged, monkeypatch):
    """bench_dd.measure — the function every rank of bench.py's decomposition leg runs — with 2 and 8 ranks at once: host threads of this
    process over the in-process peer-copy transport, with a thread-based stand-in for torch.distributed (domdec.ThreadRanks).  What a
    one-rank rehearsal cannot reach: per-rank plans and lists on a real domain grid, the first-step parity check with its broadcast of
    the single-domain forces, the sums over the ranks.  The pair total of all ranks must equal the single-domain count of the same box
    up to the cluster pairs that straddle a domain border differently."""
    import importlib
    import types
    import numpy as np
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    pkg = load_package()
    import torch
    domdec = importlib.import_module("gromacs_fep_gpu_amd.domdec")
    bench_dd = importlib.import_module("gromacs_fep_gpu_amd.bench_dd")
    monkeypatch.setenv("BENCH_DD_MERGED", merged)
    monkeypatch.setenv("BENCH_DD_CONDITION_STEPS", "3")
    monkeypatch.setenv("HALO_GPU_PEER_TIMEOUT", "120")
    world = int(np.prod(ncells))
    ranks = domdec.ThreadRanks(pkg, world)
    args = types.SimpleNamespace(dd_grid="%dx%dx%d" % ncells, max_cjpacked_per_sci=16, no_prune=False)
    recs = [None] * world

    def rank_call(r):
        def run():
            recs[r] = bench_dd.measure(args, r, world, ranks.view(r), torch, (20, 20, 20), 3, "cuda", 5, 2, check_parity=True)
        return run

    domdec.run_ranks_in_threads([rank_call(r) for r in range(world)])
    rec = recs[0]
    assert all(r is None for r in recs[1:]) and rec is not None
    assert rec["domain_grid"] == "%dx%dx%d" % ncells and rec["atoms"] == 24000
    assert rec["parity_of_first_step"]["ok"], rec["parity_of_first_step"]
    assert abs(rec["home_atoms_per_rank_mean"] * world - 24000) < 1e-6
    assert rec["halo_atoms_per_rank_mean"] > 0
    assert 0.9 * 271960 < rec["cluster_pairs_all_ranks"] < 1.5 * 271960     # 271,960 cluster pairs as one domain (profiles/r03/sizes.txt)


def _ipc_rank(rank, world, port, out_path, ncells):
    """one PROCESS per rank (all on cuda:0), gloo for the out-of-band channel: the shape of a node run with the one-sided transport"""
    import importlib
    import types
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BENCH_DD_CONDITION_STEPS"] = "3"
    os.environ["HALO_GPU_PEER_TIMEOUT"] = "120"
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    load_package()
    bench_dd = importlib.import_module("gromacs_fep_gpu_amd.bench_dd")
    args = types.SimpleNamespace(dd_grid="%dx%dx%d" % ncells, max_cjpacked_per_sci=16, no_prune=False)
    rec = bench_dd.measure(args, rank, world, dist, torch, (20, 20, 20), 3, "cpu", 5, 2, check_parity=True, transport="push")
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(rec, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ncells", [(2, 1, 1), (2, 2, 1)])
def test_one_sided_transport_between_processes(ncells, tmp_path):
    """HALO_GPU_TRANSPORT_IPC_PUSH as a node run uses it: one PROCESS per rank, the ranks' buffers opened through hipIpc handles that
    travel over the process group (halo_gpu_push_export / _import at every halo_gpu_reinit), coordinates stored into the peers' receive
    buffers and forces into the owners', sequence flags across processes.  All ranks share this box's one GPU (the device time-slices
    the processes), so only correctness is looked at: the first step of bench_dd.measure against the single-domain forces."""
    import socket
    import torch.multiprocessing as mp
    world = ncells[0] * ncells[1] * ncells[2]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rec.json")
    mp.spawn(_ipc_rank, args=(world, port, out, ncells), nprocs=world, join=True)
    rec = json.load(open(out))
    assert rec["domain_grid"] == "%dx%dx%d" % ncells and rec["atoms"] == 24000
    assert "one-sided" in rec["transport"] and rec["one_sided_status"] == 0
    assert rec["parity_of_first_step"]["ok"], rec["parity_of_first_step"]
    assert rec["halo_atoms_per_rank_mean"] > 0
