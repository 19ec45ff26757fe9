#!/usr/bin/env python3
"""SETTLE / fused update with and without the virial (same-address atomics?): run under rocprofv3 --kernel-trace --stats"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fep_testlib as tl
pkg = tl.pkg
nw = 32000
rng = np.random.default_rng(0)
x = rng.uniform(0, 10, (3 * nw, 3)).astype(np.float32)
x[1::3] = x[0::3] + [0.09572, 0, 0]
x[2::3] = x[0::3] + [-0.024, 0.0927, 0]
xp = x + rng.normal(0, 0.003, x.shape).astype(np.float32)
d_x, d_xp, d_v = (torch.from_numpy(a.copy()).cuda() for a in (x, xp, np.zeros_like(x)))
sg = pkg.SettleGpu(15.9994, 1.008, 0.09572, 0.15139)
sg.set(np.arange(3 * nw, dtype=np.int32).reshape(-1, 3))
for vir in (False, True):
    for _ in range(50):
        sg.apply(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), 500.0, vir, 3, np.eye(3) * 10)
    torch.cuda.synchronize()
sg.free()
