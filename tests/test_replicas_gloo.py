"""N > 1 path on CPU: two gloo ranks run as lambda replicas (BASELINE config 4).  Each rank evaluates ITS
lambda window of the same box with the CPU oracle (there is no GPU here; the sharding logic is what is under
test: rank -> lambda, barrier, max-over-ranks time, whole-job throughput, no data-path collective)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fep_testlib as tl
    from __graft_entry__ import load_package
    load_package()
    import importlib
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")
    lam = replica.replica_lambda(rank, world)
    case = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=2, elec="rf", seed=31, lambda_coul=lam, lambda_vdw=lam)
    dist.barrier()
    res = tl.run_oracle(case, energy=True)
    elapsed = 0.5 + 0.25 * rank                       # synthetic per-rank times: the max must win
    t_max = replica.max_over_ranks(elapsed, dist)
    value = replica.aggregate_throughput(1000.0, 10, t_max, world)
    # the replicas are independent: gather only to CHECK that they differ as lambda windows must
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(rank=rank, lam=lam, dvdl=res["dvdl_coul"] + res["dvdl_vdw"], e=res["e_el"]))
    if rank == 0:
        np.save(os.path.join(out_dir, "result.npy"), np.array([t_max, value] + [g["lam"] for g in gathered]
                                                               + [g["dvdl"] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_over_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = np.load(os.path.join(str(tmp_path), "result.npy"))
    t_max, value, lam0, lam1, dvdl0, dvdl1 = r
    assert t_max == pytest.approx(0.75)               # max over ranks, not rank 0's 0.5
    assert value == pytest.approx(2 * 1000.0 * 10 / 0.75)
    assert (lam0, lam1) == (0.0, pytest.approx(0.1))  # window = rank mod 11
    assert dvdl0 != dvdl1                             # different windows, different dH/dlambda


def test_replica_lambda_assignment():
    from __graft_entry__ import load_package
    load_package()
    import importlib
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")
    assert replica.replica_lambda(0, 1) == 0.5
    lams = [replica.replica_lambda(r, 8) for r in range(8)]
    assert lams == pytest.approx([0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7])
    assert replica.replica_lambda(11, 16) == 0.0


@pytest.mark.parametrize("world,n", [(1, 11), (2, 11), (4, 11), (8, 11), (8, 5), (16, 11)])
def test_every_window_of_the_set_is_on_exactly_one_rank(world, n):
    """BASELINE configs[3]: 11 lambda windows on the 8 GPUs of a node — a rank with more than one window runs them as one object
    (replica.batch_windows); the assignment has to cover every window once and be balanced to within one window."""
    sys.path.insert(0, os.path.dirname(HERE))
    from __graft_entry__ import load_package
    load_package()
    import importlib
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")
    held = [replica.windows_of_rank(r, world, n) for r in range(world)]
    assert sorted(w for h in held for w in h) == list(range(n))
    sizes = [len(h) for h in held]
    assert max(sizes) - min(sizes) <= 1 and max(sizes) == -(-n // world)
