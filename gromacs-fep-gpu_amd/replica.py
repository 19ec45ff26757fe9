"""Multi-GPU sharding of the FEP path: independent lambda replicas, one process per GPU (BASELINE config 4).

The reference runs lambda windows as independent simulations (`mdrun -multidir`, mdrunutility/multisim.cpp);
nothing is exchanged on the data path, so the only collective work is initialisation, a barrier and the
max-over-ranks step time.  Backend: "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np


def lambda_schedule(n_lambda=11):
    """Foreign/replica lambda set {0, 0.1, ..., 1.0}."""
    return np.linspace(0.0, 1.0, n_lambda)


def replica_lambda(rank, world_size, n_lambda=11, single_gpu_lambda=0.5):
    """Lambda of the replica that `rank` runs.  One GPU: the headline lambda 0.5; N GPUs: window rank mod n_lambda."""
    if world_size == 1:
        return float(single_gpu_lambda)
    return float(lambda_schedule(n_lambda)[rank % n_lambda])


def windows_of_rank(rank, world_size, n_windows=11):
    """The lambda windows rank `rank` holds when ALL n_windows windows of a set are spread over world_size ranks (BASELINE configs[3]:
    11 windows on the 8 GPUs of a node): windows rank, rank + world_size, ... — at most ceil(n_windows / world_size) per rank, every
    window on exactly one rank.  A rank with several windows runs them as ONE object (batch_windows): one launch for all of them."""
    return list(range(int(rank), int(n_windows), int(world_size)))


def max_over_ranks(value, dist=None, device="cpu"):
    """Max of a python float over all ranks (the bench's timed interval)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(units_per_step_per_rank, steps, elapsed_max, world_size):
    """Whole-job throughput of a weak-scaling replica set: every rank processes the same number of units."""
    return world_size * units_per_step_per_rank * steps / elapsed_max


def batch_windows(grid, plist, num_windows):
    """R lambda windows of one system as ONE set of non-bonded inputs (nbnxm_gpu_set_window_lambdas): the grid
    arrays tiled R times, the fused-mode list concatenated with cluster, group and exclusion indices shifted per window — no pair
    connects two windows.  The windows start from the same list (same search); their coordinates are free to differ.
    Returns a dict with the arrays NbnxmGpu.init_atomdata / init_pairlist / init_fep_cluster_bits / init_x_to_nbat_x take and
    `clusters_per_window`, `slots_per_window`, `atoms_per_window`."""
    R = int(num_windows)
    ns = int(grid.num_atoms)
    ncl, nsc = ns // 8, ns // 64
    natoms = int(grid.natoms)
    sci = np.tile(plist.sci, R)
    cj = np.tile(plist.cjPacked, R)
    excl = np.tile(plist.excl, R)
    ng, nx, ne = len(plist.cjPacked), len(plist.excl), len(plist.sci)
    w_sci = np.repeat(np.arange(R), ne)
    w_cj = np.repeat(np.arange(R), ng)
    sci["sci"] += (w_sci * nsc).astype(sci["sci"].dtype)
    sci["cjPackedBegin"] += (w_sci * ng).astype(np.int32)
    sci["cjPackedEnd"] += (w_sci * ng).astype(np.int32)
    cj["cj"] += (w_cj * ncl).astype(np.int32)[:, None]
    cj["imei"]["excl_ind"] += (w_cj * nx).astype(np.int32)[:, None]
    ai = np.tile(grid.atomIndices, R)
    shift = np.repeat(np.arange(R) * natoms, ns)
    ai = np.where(ai >= 0, ai + shift, ai).astype(np.int32)
    tile = lambda a: np.tile(np.asarray(a), R)
    return dict(sci=sci, cjPacked=cj, excl=excl, xq=np.tile(grid.xq.reshape(-1, 4), (R, 1)), type=tile(grid.type), qA=tile(grid.qA), qB=tile(grid.qB),
                typeA=tile(grid.typeA), typeB=tile(grid.typeB), fepBits=tile(grid.fepBits), atomIndices=ai, clusters_per_window=ncl,
                slots_per_window=ns, atoms_per_window=natoms, num_windows=R)
