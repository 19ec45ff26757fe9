"""bench.py's domain-decomposition leg (BASELINE configs[4]): one box decomposed over the ranks on a 3-D domain grid
(gromacs-fep-gpu_amd/domdec.py), halo exchange over RCCL (include/halo_hip.h); kept out of bench.py so that the contract file
stays short."""
import importlib
import os
import time

import numpy as np


def parse_grid(text, world, box=None, rcomm=1.2):
    """--dd-grid AxBxC, or the grid with the smallest halo for this box (domdec.default_grid)"""
    domdec = importlib.import_module("gromacs_fep_gpu_amd.domdec")
    if text:
        g = tuple(int(v) for v in text.lower().split("x"))
        assert len(g) == 3 and int(np.prod(g)) == world, "--dd-grid must multiply to the number of ranks"
        return g
    return domdec.default_grid(world, box, rcomm)


def measure(args, rank, world, dist, torch, nm, npert, reduce_device, steps, warmup, grid_text=None):
    """A step = halo x (pack, RCCL send / receive) on the non-local stream beside the local kernel, x -> xq per locality, local
    and non-local fused cluster kernels, forces to atom order per locality, halo f.  Returns the record of the leg (rank 0) —
    pair interactions of the WHOLE system per second (strong scaling)."""
    pkg = importlib.import_module("gromacs_fep_gpu_amd")
    wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
    domdec = importlib.import_module("gromacs_fep_gpu_amd.domdec")
    replica = importlib.import_module("gromacs_fep_gpu_amd.replica")
    t0 = time.time()
    # the synthetic system (topology-order arrays) is generated on every rank; grid and lists are built for the rank's own domain only
    case = wl.make_case(nm=nm, num_perturbed_molecules=npert, elec="ewald", seed=2026, n_lambda=11, build_lists=False)
    ncells = parse_grid(grid_text if grid_text is not None else args.dd_grid, world, case.sys["box"], case.rlist + 0.1)
    dd = domdec.DomainDecomposition(case.sys["x"], case.sys["box"], case.sys["molId"], ncells, case.rlist)
    plan = dd.plan(rank)
    t_plan = time.time() - t0
    t0 = time.time()
    system = domdec.RankSystem(pkg, plan, case.sys["box"], case.sys["qA"], case.sys["qB"], case.sys["typeA"], case.sys["typeB"], case.ntype,
                               case.sys["molId"], case.rlist, perturbed=case.perturbed, max_cjpacked_per_sci=args.max_cjpacked_per_sci)
    t_lists = time.time() - t0
    nb = domdec.make_rank_gpu(pkg, wl, case, system, use_dynamic_pruning=not args.no_prune)
    nb.set_timing(False)
    halo = domdec.RcclHalo(pkg, dist, rank, world, nb.stream(pkg.NONLOCAL))
    st = domdec.DomainStep(pkg, nb, system, halo)
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    for _ in range(1 + warmup):
        st.step(sw)
    torch.cuda.synchronize()
    pairs = 0
    for pl, iloc in ((system.local, pkg.LOCAL), (system.nonlocal_, pkg.NONLOCAL)):
        if len(pl.cjPacked):
            cj = pkg.download_cjpacked(nb, len(pl.cjPacked), iloc)
            pairs += int(np.unpackbits(np.ascontiguousarray(cj["imei"][:, 0]["imask"]).view(np.uint8)).sum())
    tot = torch.tensor([float(pairs), float(plan.num_halo), float(plan.num_home)], device=reduce_device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(steps):
        st.step(sw)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = replica.max_over_ranks(time.perf_counter() - t_start, dist if world > 1 else None, device=reduce_device)
    rec = None
    if rank == 0:
        rec = {"pair_interactions_per_s": 64.0 * float(tot[0].item()) * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
               "domain_grid": "%dx%dx%d" % ncells, "atoms": int(case.natoms), "cluster_pairs_all_ranks": int(tot[0].item()),
               "halo_atoms_per_rank_mean": float(tot[1].item()) / world, "home_atoms_per_rank_mean": float(tot[2].item()) / world,
               "halo_bytes_sent_and_received_rank0_per_step": halo.bytes_per_step(), "transport": "RCCL ncclSend/ncclRecv groups (halo_hip.h)",
               "local_launch": ("two parts, the second behind the non-local kernel (HALO_GPU_LOCAL_PARTS)" if world > 1 and os.environ.get("HALO_GPU_LOCAL_PARTS", "2") == "2"
                                else "one launch"),
               "host_plan_s": t_plan, "host_rank_lists_s": t_lists}
    halo.free()
    nb.free()
    return rec
