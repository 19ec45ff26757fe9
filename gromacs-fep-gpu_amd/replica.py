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
