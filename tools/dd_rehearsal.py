#!/usr/bin/env python3
"""Rehearsal of `bench.py --dd` with 2 processes on ONE GPU (the pool gives single-GPU boxes; RCCL refuses two ranks on
one device): gloo process group, halo buffers staged through the host for the transfers, everything else — decomposition,
HIP pack / unpack, x -> xq, fused kernel per rank, force reduction, timing loop, JSON line — exactly as in the bench.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
      tools/dd_rehearsal.py --atoms 96k --steps 20 --warmup 3
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch
import torch.distributed as dist


class StagedGlooComm:
    def __init__(self, d):
        self.dist = d

    def exchange(self, send_bufs, recv_bufs):
        torch.cuda.current_stream().synchronize()
        host_recv = {q: torch.empty(b.shape, dtype=b.dtype) for q, b in recv_bufs.items() if b.numel()}
        host_send = {q: b.cpu() for q, b in send_bufs.items() if b.numel()}
        ops = [self.dist.P2POp(self.dist.irecv, t, q) for q, t in sorted(host_recv.items())]
        ops += [self.dist.P2POp(self.dist.isend, t, q) for q, t in sorted(host_send.items())]
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for q, t in host_recv.items():
            recv_bufs[q].copy_(t)

    def exchange_x(self, halo):
        self.exchange(halo.x_send, halo.x_recv)

    def exchange_f(self, halo):
        self.exchange(halo.f_send, halo.f_recv)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import importlib
    import bench
    import fep_testlib as tl
    domdec = importlib.import_module("gromacs_fep_gpu_amd.domdec")
    domdec.TorchDistComm = StagedGlooComm
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--atoms", default="96k")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    args = argparse.Namespace(steps=a.steps, warmup=a.warmup, no_prune=False, atoms=a.atoms)
    nm = {"24k": (20, 20, 20), "96k": (40, 40, 20), "768k": (80, 80, 40)}[a.atoms]
    bench.run_domain_decomposition(args, dist.get_rank(), dist.get_world_size(), dist, torch, tl, nm, 16 if a.atoms != "24k" else 3)


if __name__ == "__main__":
    main()
