"""GPU parity of the stochastic-dynamics update: HIP kernels behind include/update_hip.h against the CPU
oracle (oracle/langevin_ref.c; its random engine and normal table are pinned by the reference's known answers).  The random
numbers must be the same numbers (same Threefry block, same table entries): velocities and coordinates agree to fp32
rounding of a handful of operations."""
import numpy as np
import pytest

import fep_testlib as tl
import oracle_binding as ob

pytestmark = pytest.mark.gpu
pkg = tl.pkg


def _system(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 5, (n, 3)).astype(np.float32)
    v = rng.normal(0, 0.5, (n, 3)).astype(np.float32)
    f = rng.normal(0, 300, (n, 3)).astype(np.float32)
    im = (1.0 / rng.choice([1.008, 12.011, 15.999], n)).astype(np.float32)
    tc = rng.integers(0, 3, n).astype(np.uint16)
    return x, v, f, im, tc


@pytest.mark.parametrize("n", [1, 1000, 100003])
def test_langevin_update_matches_oracle(n):
    import torch
    x, v, f, im, tc = _system(n, 7)
    ref_t, tau_t, dt, seed = [300.0, 310.0, 0.0], [1.0, 0.1, 0.0], 0.002, 1993   # group 2: tau_t = 0 -> no friction, no noise
    lg = pkg.LangevinGpu(ref_t, tau_t, dt)
    lg.set(im, tc)
    d_x, d_v, d_f = (torch.from_numpy(a.copy()).cuda() for a in (x, v, f))
    d_xp = torch.zeros_like(d_x)
    xo, vo = x, v
    for step in (0, 1, 41):
        x_before = d_x.cpu().numpy()
        lg.integrate(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), dt, seed, step, pkg.LANGEVIN_FORCES_ONLY)
        xo1, xpo, vo1 = ob.langevin_update(0, xo, vo, f, im, tc, ref_t, tau_t, dt, seed, step)
        torch.cuda.synchronize()
        assert np.array_equal(d_xp.cpu().numpy(), x_before)                 # copy of the old coordinates: exact
        assert np.allclose(d_v.cpu().numpy(), vo1, rtol=2e-6, atol=1e-6)
        assert np.allclose(d_x.cpu().numpy(), xo1, rtol=1e-6, atol=1e-6)
        lg.integrate(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), dt, seed, step, pkg.LANGEVIN_FRICTION_AND_NOISE)
        xo2, _, vo2 = ob.langevin_update(1, xo1, vo1, f, im, tc, ref_t, tau_t, dt, seed, step)
        torch.cuda.synchronize()
        gv, gx = d_v.cpu().numpy(), d_x.cpu().numpy()
        assert np.allclose(gv, vo2, rtol=2e-6, atol=2e-6)                   # the same random numbers, fp32 rounding only
        assert np.allclose(gx, xo2, rtol=1e-6, atol=1e-6)
        frozen = tc == 2
        assert np.allclose(gv[frozen], vo1[frozen], rtol=1e-6, atol=1e-7)   # tau_t = 0: em = 1, sigmaV = 0
        xo, vo = xo2, vo2
    lg.free()


def test_langevin_noise_has_the_right_temperature():
    """many atoms, long coupling: the velocity increment of the noise step has variance kB T (1 - em^2) / m"""
    import torch
    n = 200000
    rng = np.random.default_rng(1)
    im = np.full(n, 1.0 / 12.011, np.float32)
    tc = np.zeros(n, np.uint16)
    dt, tau, T = 0.002, 0.05, 300.0
    lg = pkg.LangevinGpu([T], [tau], dt)
    lg.set(im, tc)
    d_x = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    d_v = torch.zeros_like(d_x)
    lg.integrate(d_x.data_ptr(), d_x.data_ptr(), d_v.data_ptr(), None, dt, 5, 3, pkg.LANGEVIN_FRICTION_AND_NOISE)
    torch.cuda.synchronize()
    v = d_v.cpu().numpy().astype(np.float64)
    em = np.exp(-dt / tau)
    want = 0.0083144626 * T * (1 - em * em) / 12.011
    assert abs(v.var() / want - 1.0) < 0.01 and abs(v.mean()) < 3 * np.sqrt(want / (3 * n)) * 3
    lg.free()
