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


def step_parity(pkg, wl, case_args, st, plan, rank, world, dist, torch, reduce_device):
    """Validates the exchange, not only times it: after one step of the decomposed box, every rank's home forces against the forces
    of the SAME box evaluated as a single domain on rank 0's GPU (same kernels, fused mode), broadcast over the process group.
    Per atom |delta f| <= 1e-4 max(|f|, rms |f|) — the bar of the parity tests; both sides are fp32 sums in different orders.
    Also Newton's third law over all ranks.  Returns {"max_err_over_tolerance", "net_force_over_rms", "ok"} on rank 0."""
    f_home = torch.from_numpy(st.home_forces().astype(np.float64)).to(reduce_device)
    natoms = int(case_args["natoms"])
    f_ref = torch.zeros((natoms, 3), dtype=torch.float64, device=reduce_device)
    if rank == 0:
        c = wl.make_case(build_lists=True, **case_args["make_case"])
        got = wl.run_gpu(c, energy=False, fused=True)
        g = c.grid
        real = g.atomIndices >= 0
        ref = np.zeros((natoms, 3))
        ref[g.atomIndices[real]] = got["f"][real]
        f_ref = torch.from_numpy(ref).to(reduce_device)
    if world > 1:
        dist.broadcast(f_ref, 0)
    mine = f_ref[torch.from_numpy(np.asarray(plan.home, np.int64)).to(reduce_device)]
    frms = float(torch.sqrt(torch.mean(torch.sum(f_ref ** 2, dim=1))).item())
    err = torch.sqrt(torch.sum((f_home - mine) ** 2, dim=1))
    tol = 1e-4 * torch.clamp(torch.sqrt(torch.sum(mine ** 2, dim=1)), min=frms)
    worst = torch.max(err / tol).reshape(1)
    net = torch.sum(f_home, dim=0)
    if world > 1:
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        dist.all_reduce(net)
    net_over_rms = float(torch.max(torch.abs(net)).item()) / (frms * np.sqrt(natoms))
    return {"max_err_over_tolerance": float(worst.item()), "net_force_over_rms_sqrt_n": net_over_rms,
            "ok": bool(worst.item() <= 1.0 and net_over_rms <= 1e-3),
            "how": "home forces of every rank after one decomposed step against the same box as a single domain on rank 0 (1e-4 of max(|f_i|, rms |f|)); sum of all forces"}


def measure(args, rank, world, dist, torch, nm, npert, reduce_device, steps, warmup, grid_text=None, check_parity=False, transport="rccl"):
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
    # BENCH_DD_SELF_LINKS=x|xyz (one-GPU rehearsal): a dimension with one cell is its own neighbour through the periodic boundary, so that
    # a single rank has a real halo, real RCCL groups (to itself) and non-empty non-local lists — the code path of a multi-rank leg
    links = os.environ.get("BENCH_DD_SELF_LINKS", "")
    self_links = tuple(d in links for d in "xyz")
    dd = domdec.DomainDecomposition(case.sys["x"], case.sys["box"], case.sys["molId"], ncells, case.rlist, self_links=self_links)
    plan = dd.plan(rank)
    t_plan = time.time() - t0
    t0 = time.time()
    system = domdec.RankSystem(pkg, plan, case.sys["box"], case.sys["qA"], case.sys["qB"], case.sys["typeA"], case.sys["typeB"], case.ntype,
                               case.sys["molId"], case.rlist, perturbed=case.perturbed, max_cjpacked_per_sci=args.max_cjpacked_per_sci)
    t_lists = time.time() - t0
    # merged localities (default): one device list, one cluster-kernel launch and one stream per rank and step;
    # BENCH_DD_MERGED=0: the reference's two-locality, two-stream schedule
    merged = os.environ.get("BENCH_DD_MERGED", "1") != "0"
    nb = domdec.make_rank_gpu(pkg, wl, case, system, use_dynamic_pruning=not args.no_prune, merged=merged)
    nb.set_timing(False)
    # (dist.peer_copy_id: the ranks are threads of this process — domdec.ThreadRanks, a rehearsal — and use the in-process transport)
    # transport: RCCL send / receive groups, or (transport="push", merged localities only) the one-sided transport between the ranks'
    # processes — stores into the peers' buffers opened over hipIpc, sequence flags, no transfer kernel (halo_hip.h)
    ipc_push = (transport == "push" and merged and getattr(dist, "peer_copy_id", None) is None)
    halo = domdec.RcclHalo(pkg, dist, rank, world, nb.stream(pkg.LOCAL if merged else pkg.NONLOCAL), unique_id=getattr(dist, "peer_copy_id", None),
                           ipc_push=ipc_push)
    st = domdec.DomainStep(pkg, nb, system, halo)
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    st.step(sw)
    if ipc_push:
        # a peer that never stores its side makes the waiting kernels give up after seconds (they do not hang): found here, not 1,200 steps later
        torch.cuda.synchronize()
        bad = torch.tensor([float(halo.push_status())], device=reduce_device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if bad.item() != 0:
            halo.free()
            nb.free()
            raise RuntimeError("one-sided transport: a kernel gave up waiting for a peer's flag in the first step (status %d)" % int(bad.item()))
    parity = None
    if check_parity:
        parity = step_parity(pkg, wl, dict(natoms=case.natoms, make_case=dict(nm=nm, num_perturbed_molecules=npert, elec="ewald", seed=2026,
                                                                                 n_lambda=11, max_cjpacked_per_sci=args.max_cjpacked_per_sci)),
                             st, plan, rank, world, dist, torch, reduce_device)
    # the same warm-up + timed steps with NO conditioning ahead of them: what a short protocol reads by itself (bench.py does the same)
    for _ in range(warmup):
        st.step(sw)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_cold = time.perf_counter()
    for _ in range(steps):
        st.step(sw)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed_cold = replica.max_over_ranks(time.perf_counter() - t_cold, dist if world > 1 else None, device=reduce_device)
    # untimed device conditioning ahead of the warm-up (the host has been building plans and lists for seconds with the GPU idle: the
    # device's clock needs a few hundred ms of work to settle, see bench.py)
    for _ in range(int(os.environ.get("BENCH_DD_CONDITION_STEPS", "1000"))):
        st.step(sw)
    for _ in range(warmup):
        st.step(sw)
    torch.cuda.synchronize()
    pairs = 0
    # (merged localities: the local DEVICE list holds both lists, local entries first; the non-local device list is empty)
    device_lists = ([(len(system.local.cjPacked) + len(system.nonlocal_.cjPacked), pkg.LOCAL)] if merged
                    else [(len(system.local.cjPacked), pkg.LOCAL), (len(system.nonlocal_.cjPacked), pkg.NONLOCAL)])
    for ncj, iloc in device_lists:
        if ncj:
            cj = pkg.download_cjpacked(nb, ncj, iloc)
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
    status_all = None
    if ipc_push:
        # ... and at the end: a kernel that gave up in a later step made the object skip the steps behind it (halo_exchange.hip), so the time
        # above would be meaningless — every rank learns it, the record says so and carries no rate
        bad = torch.tensor([float(halo.push_status())], device=reduce_device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        status_all = int(bad.item())
        if status_all != 0:
            halo.free()
            nb.free()
            raise RuntimeError("one-sided transport: a kernel gave up waiting for a peer's flag during the timed steps (status %d)" % status_all)
    rec = None
    if rank == 0:
        rec = {"pair_interactions_per_s": 64.0 * float(tot[0].item()) * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
               "ms_per_step_cold": 1e3 * elapsed_cold / steps,
               "device_conditioning_steps_before_warmup": int(os.environ.get("BENCH_DD_CONDITION_STEPS", "1000")),
               "domain_grid": "%dx%dx%d" % ncells, "atoms": int(case.natoms), "cluster_pairs_all_ranks": int(tot[0].item()),
               "halo_atoms_per_rank_mean": float(tot[1].item()) / world, "home_atoms_per_rank_mean": float(tot[2].item()) / world,
               "halo_bytes_sent_and_received_rank0_per_step": halo.bytes_per_step(),
               "transport": ("in-process peer copies between rank threads (halo_hip.h; a rehearsal)" if getattr(dist, "peer_copy_id", None) is not None
                             else "one-sided: stores into the peers' buffers (hipIpc), sequence flags, no transfer kernel (halo_hip.h)" if ipc_push
                             else "RCCL ncclSend/ncclRecv groups (halo_hip.h)"),
               "one_sided_status": status_all,
               "schedule": ("merged localities: one list, one launch, one stream per rank" if merged else "two localities on two streams"),
               "local_launch": ("one launch" if merged else
                                "two parts, the second behind the non-local kernel (HALO_GPU_LOCAL_PARTS)" if world > 1 and os.environ.get("HALO_GPU_LOCAL_PARTS", "2") == "2"
                                else "one launch"),
               "host_plan_s": t_plan, "host_rank_lists_s": t_lists, "parity_of_first_step": parity}
    halo.free()
    nb.free()
    return rec
