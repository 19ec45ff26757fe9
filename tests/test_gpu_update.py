"""GPU parity of the stochastic-dynamics update: HIP kernels behind include/update_hip.h against the CPU
oracle (oracle/langevin_ref.c; its random engine and normal table are pinned by the reference's known answers).  The random
numbers must be the same numbers (same Threefry block, same table entries): velocities and coordinates agree to fp32
rounding of a handful of operations."""
import ctypes as C

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


# ---- leap-frog, SETTLE, LINCS and their composition -----------------------------------------------------------------------
import update_cases as uc


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


@pytest.mark.parametrize("idx", range(16))
def test_leapfrog_reproduces_reference_known_answers(idx):
    """the reference's own LeapFrogTest cases (mdlib/tests/leapfrog.cpp) through the C ABI, and the double oracle beside them"""
    import torch
    c = uc.leapfrog_cases()[idx]
    lf = pkg.LeapFrogGpu(c.num_tc)
    lf.set(c.invmass, c.groups)
    d_x, d_v, d_f = _dev(c.x0), _dev(c.v0), _dev(c.f)
    d_xp = torch.zeros_like(d_x)
    xo, vo = c.x0, c.v0
    pr_matrix = np.diag(c.pr_diag)
    for step in range(c.num_steps):
        pc = c.do_pressure_couple(step)
        x_before = d_x.clone()
        lf.integrate(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), c.dt, tc_lambdas=c.lambdas if c.num_tc > 0 else None,
                     pr_matrix=pr_matrix if pc else None, dt_pressure_couple=c.dt_pc)
        xo, _, vo = ob.leapfrog(xo, vo, c.f, c.invmass, c.dt, lambdas=c.lambdas if c.num_tc > 0 else None, groups=c.groups,
                                pr_diag=c.dt_pc * c.pr_diag if pc else None)
        torch.cuda.synchronize()
        assert torch.equal(d_xp, x_before)
    gx, gv = d_x.cpu().numpy(), d_v.cpu().numpy()
    assert np.max(np.abs(gx - c.final_x)) <= c.tolerance and np.max(np.abs(gv - c.final_v)) <= c.tolerance
    assert np.allclose(gx, xo, rtol=1e-6, atol=1e-6 * c.num_steps) and np.allclose(gv, vo, rtol=1e-6, atol=1e-6 * c.num_steps)
    lf.free()


def test_leapfrog_many_groups_and_large_system():
    """40 temperature-coupling groups (factors through a device buffer instead of kernel arguments), 100003 atoms"""
    import torch
    n = 100003
    x, v, f, im, _ = _system(n, 3)
    groups = (np.arange(n) % 40).astype(np.uint16)
    lambdas = 1.0 - 0.01 * np.arange(40)
    for ntc, lam, grp in ((40, lambdas, groups), (3, lambdas[:3], (groups % 3).astype(np.uint16)), (1, lambdas[5:6], None), (0, None, None)):
        lf = pkg.LeapFrogGpu(ntc)
        lf.set(im, grp)
        d_x, d_v, d_f = _dev(x), _dev(v), _dev(f)
        d_xp = torch.zeros_like(d_x)
        lf.integrate(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), 0.002, tc_lambdas=lam, pr_matrix=np.diag([0.3, -0.2, 0.1]),
                     dt_pressure_couple=0.02)
        xo, xpo, vo = ob.leapfrog(x, v, f, im, 0.002, lambdas=lam, groups=grp, pr_diag=0.02 * np.array([0.3, -0.2, 0.1]))
        torch.cuda.synchronize()
        assert np.array_equal(d_xp.cpu().numpy(), x)
        assert np.allclose(d_v.cpu().numpy(), vo, rtol=2e-6, atol=1e-6) and np.allclose(d_x.cpu().numpy(), xo, rtol=1e-6, atol=1e-6)
        lf.free()


@pytest.mark.parametrize("idx", range(13))
def test_settle_reproduces_reference_known_answers(idx):
    """the reference's own SettleTest cases (mdlib/tests/settle.cpp)"""
    import torch
    c = uc.settle_cases()[idx]
    sg = pkg.SettleGpu(c.mO, c.mH, c.dOH, c.dHH)
    sg.set(c.atoms)
    d_x, d_xp, d_v = _dev(c.x), _dev(c.xp), _dev(c.v)
    vir = sg.apply(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr() if c.update_velocities else None, c.invdt, c.calc_virial, c.pbc_type, c.box)
    torch.cuda.synchronize()
    n = 3 * c.num_settles
    gxp, gv = d_xp.cpu().numpy().astype(np.float64), d_v.cpu().numpy()
    assert np.max(np.abs(gxp[:n] - c.final_x)) <= 1e-6                    # settle.cpp:363-371
    assert np.array_equal(gxp[n:].astype(np.float32), c.xp[n:].astype(np.float32))
    w = gxp[:n].reshape(-1, 3, 3)
    for a, b, d in ((0, 1, c.dOH), (0, 2, c.dOH), (1, 2, c.dHH)):
        assert np.max(np.abs(np.sum((w[:, a] - w[:, b]) ** 2, axis=1) - d * d)) <= 380 * 1.2e-7 * c.dOH * c.dOH
    if c.update_velocities:
        assert np.max(np.abs(gv[:n] - c.final_v)) <= 1e-4
    else:
        assert not gv.any()
    assert (vir != 0).all() == c.calc_virial and (vir != 0).any() == c.calc_virial
    if c.calc_virial:
        assert np.max(np.abs(vir - c.virial)) <= 1e-6 and np.max(np.abs(vir - vir.T)) <= 1e-6
    sg.free()


def _water_box(num_waters, seed, box=4.0, dOH=0.09572, dHH=0.15139):
    """rigid waters at random places and orientations in a periodic box (some straddle the faces), then a random displacement"""
    rng = np.random.default_rng(seed)
    h = np.sqrt(dOH * dOH - 0.25 * dHH * dHH)
    local = np.array([[0, 0, 0], [0.5 * dHH, h, 0], [-0.5 * dHH, h, 0]])
    q = rng.normal(size=(num_waters, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    a, b, c_, d = q.T
    rot = np.stack([np.stack([a*a+b*b-c_*c_-d*d, 2*(b*c_-a*d), 2*(b*d+a*c_)], 1), np.stack([2*(b*c_+a*d), a*a-b*b+c_*c_-d*d, 2*(c_*d-a*b)], 1),
                    np.stack([2*(b*d-a*c_), 2*(c_*d+a*b), a*a-b*b-c_*c_+d*d], 1)], 1)
    x = (rng.uniform(0, box, (num_waters, 1, 3)) + np.einsum("wij,aj->wai", rot, local)).reshape(-1, 3)
    x = x.astype(np.float32).astype(np.float64)
    xp = (x + rng.normal(0, 0.004, x.shape)).astype(np.float32).astype(np.float64)
    # atoms in the unit cell, molecules broken over the faces: the minimum image has to put them together again
    sh = np.floor(x / box) * box
    return x - sh, xp - sh, rng.normal(0, 0.5, x.shape).astype(np.float32).astype(np.float64)


def test_settle_large_periodic_box_matches_oracle():
    import torch
    nw, box = 33333, 4.0
    x, xp, v = _water_box(nw, 11, box)
    atoms = np.arange(3 * nw, dtype=np.int32).reshape(-1, 3)
    atoms[::2] = atoms[::2][:, [0, 2, 1]]       # the order of the hydrogens must not matter
    mO, mH, dOH, dHH, invdt = 15.9994, 1.008, 0.09572, 0.15139, 500.0
    bx = np.eye(3) * box
    sg = pkg.SettleGpu(mO, mH, dOH, dHH)
    sg.set(atoms)
    d_x, d_xp, d_v = _dev(x), _dev(xp), _dev(v)
    vir = sg.apply(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), invdt, True, 3, bx)
    oxp, ov, ovir = ob.settle(atoms, mO, mH, dOH, dHH, x, xp, v=v, invdt=invdt, compute_virial=True, pbc_type=3, box=bx)
    gxp, gv = d_xp.cpu().numpy(), d_v.cpu().numpy()
    assert np.max(np.abs(gxp - oxp)) <= 2e-6          # fp32 coordinates of size ~4
    assert np.max(np.abs(gv - ov)) <= 5e-4            # position error times 1/dt
    assert np.max(np.abs(vir - ovir)) <= 2e-4 * np.max(np.abs(ovir))
    sg.free()


@pytest.mark.parametrize("idx", range(14))
def test_lincs_reproduces_reference_known_answers(idx):
    """the reference's own ConstraintsTest cases (mdlib/tests/constr.cpp) with its tolerances, and the double oracle of the same
    algorithm beside them"""
    import torch
    c = uc.constraints_cases()[idx]
    lg = pkg.LincsGpu(c.n_iter, c.order)
    assert lg.set(c.iatoms, c.lengths, c.invmass)
    d_x, d_xp, d_v = _dev(c.x), _dev(c.xp), _dev(c.v)
    vir = lg.apply(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), c.invdt, True, c.pbc_type, c.box)
    torch.cuda.synchronize()
    gxp, gv = d_xp.cpu().numpy().astype(np.float64), d_v.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(gxp - c.final_x)) <= c.tol_x, c.title
    assert np.max(np.abs(gv - c.final_v)) <= c.tol_v, c.title
    assert np.max(np.abs(vir - c.virial)) <= c.tol_virial, c.title
    oxp, ov, ovir = ob.lincs(c.iatoms, c.lengths, c.invmass, c.n_iter, c.order, c.x, c.xp, v=c.v, invdt=c.invdt, compute_virial=True,
                             pbc_type=c.pbc_type, box=c.box)
    scale = max(1.0, np.max(np.abs(c.xp)))
    assert np.max(np.abs(gxp - oxp)) <= 2e-6 * scale, c.title
    assert np.max(np.abs(gv - ov)) <= 2e-6 * scale * c.invdt, c.title
    assert np.max(np.abs(vir - ovir)) <= 1e-4 * max(np.max(np.abs(ovir)), 1e-6), c.title
    for t, i, j in c.iatoms:
        d1 = gxp[i] - gxp[j]
        if c.pbc_type == 3:
            d1 -= np.rint(d1 / np.diag(c.box)) * np.diag(c.box)
        assert abs(np.linalg.norm(d1) - c.lengths[t]) <= 0.002 * c.lengths[t] + 1e-12     # constr.cpp:655
    lg.free()


def _molecule_soup(seed, num_molecules=4000, box=6.0, chain=0):
    """a mix of OH (1 constraint), CH2 (2), CH3 (3), triangles (3) and, if chain, one all-bonds chain of that many atoms"""
    rng = np.random.default_rng(seed)
    templ = [
        (np.array([[0, 0, 0], [0.1, 0, 0]]), [12.0, 1.0], [(0, 0, 1)]),
        (np.array([[0, 0, 0], [0.109, 0, 0], [-0.036, 0.103, 0]]), [12.0, 1.0, 1.0], [(1, 0, 1), (1, 0, 2)]),
        (np.array([[0, 0, 0], [0.109, 0, 0], [-0.036, 0.103, 0], [-0.036, -0.051, 0.089]]), [12.0, 1.0, 1.0, 1.0], [(1, 0, 1), (1, 0, 2), (1, 0, 3)]),
        (np.array([[0, 0, 0], [0.1, 0, 0], [0.05, 0.0866025, 0]]), [16.0, 1.0, 1.0], [(0, 0, 1), (0, 0, 2), (0, 1, 2)]),
    ]
    lengths = np.array([0.1, 0.109, 0.153])
    xs, ms, cons = [], [], []
    for m in range(num_molecules):
        pos, mass, cs = templ[rng.integers(0, 4)]
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        a, b, c_, d = q
        rot = np.array([[a*a+b*b-c_*c_-d*d, 2*(b*c_-a*d), 2*(b*d+a*c_)], [2*(b*c_+a*d), a*a-b*b+c_*c_-d*d, 2*(c_*d-a*b)],
                        [2*(b*d-a*c_), 2*(c_*d+a*b), a*a-b*b-c_*c_+d*d]])
        off = sum(len(p) for p in xs)
        xs.append(rng.uniform(0, box, 3) + pos @ rot.T)
        ms += mass
        for t, i, j in cs:
            cons.append((t, off + j, off + i) if rng.random() < 0.5 else (t, off + i, off + j))   # both orientations
    if chain:
        off = sum(len(p) for p in xs)
        ang = np.deg2rad(111.0)
        pos = np.zeros((chain, 3))
        for k in range(1, chain):
            pos[k] = pos[k - 1] + 0.153 * np.array([np.cos((k % 2) * (np.pi - ang)), np.sin((k % 2) * (np.pi - ang)), 0.0])
        xs.append(np.array([1.0, 1.0, 1.0]) + pos)
        ms += [12.0] * chain
        cons += [(2, off + k, off + k + 1) for k in range(chain - 1)]
    x = np.concatenate(xs).astype(np.float32).astype(np.float64)
    # lengths as they are in x (the templates are only approximately at the target lengths)
    xp = (x + rng.normal(0, 0.003, x.shape)).astype(np.float32).astype(np.float64)
    sh = np.floor(x / box) * box
    perm = rng.permutation(len(cons))
    return x - sh, xp - sh, rng.normal(0, 0.5, x.shape), 1.0 / np.array(ms), np.array(cons, np.int32)[perm], lengths


@pytest.mark.parametrize("chain,block", [(0, 64), (100, 128), (600, 1024)])
def test_lincs_mixed_system_matches_oracle(chain, block):
    """thousands of small coupled groups packed into one-wave work-groups, and one long all-bonds chain that needs a larger one"""
    import torch
    box = 6.0
    x, xp, v, im, iatoms, lengths = _molecule_soup(5 + chain, 4000, box, chain)
    bx = np.eye(3) * box
    n_iter, order, invdt = (2, 6, 500.0) if chain else (1, 4, 500.0)
    lg = pkg.LincsGpu(n_iter, order)
    assert lg.set(iatoms, lengths, im)
    d_x, d_xp, d_v = _dev(x), _dev(xp), _dev(v)
    vir = lg.apply(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), invdt, True, 3, bx)
    oxp, ov, ovir = ob.lincs(iatoms, lengths, im, n_iter, order, x, xp, v=v, invdt=invdt, compute_virial=True, pbc_type=3, box=bx)
    gxp, gv = d_xp.cpu().numpy(), d_v.cpu().numpy()
    assert np.max(np.abs(gxp - oxp)) <= 3e-6
    assert np.max(np.abs(gv - ov)) <= 3e-6 * invdt
    assert np.max(np.abs(vir - ovir)) <= 1e-3 * np.max(np.abs(ovir))
    # atoms without a constraint are not touched; a second application with the same object gives the same answer
    free = np.setdiff1d(np.arange(len(im)), iatoms[:, 1:].ravel())
    assert np.array_equal(gxp[free], xp[free].astype(np.float32))
    d_xp2, d_v2 = _dev(xp), _dev(v)
    lg.apply(d_x.data_ptr(), d_xp2.data_ptr(), d_v2.data_ptr(), invdt, False, 3, bx)
    torch.cuda.synchronize()      # the object's own stream: nothing waits for it without the virial
    assert np.max(np.abs(d_xp2.cpu().numpy() - gxp)) <= 1e-6
    lg.free()


def test_lincs_refuses_too_many_coupled_constraints():
    n = 1100                                           # 1099 coupled constraints > the largest work-group
    iatoms = np.array([(0, k, k + 1) for k in range(n - 1)], np.int32)
    lg = pkg.LincsGpu(1, 4)
    assert not lg.set(iatoms, [0.15], np.ones(n))
    assert lg.set(iatoms[:1000], [0.15], np.ones(n))
    lg.free()


@pytest.mark.parametrize("sd", [False, True])
def test_update_constrain_composition(sd):
    """UpdateConstrainGpu::integrate against the same sequence of oracle calls: integrator, LINCS, SETTLE and, for stochastic
    dynamics, friction + noise and the second constraint pass; the virial comes out scaled by 0.5 / dt^2"""
    import torch
    box, dt, seed = 5.0, 0.002, 77
    nw = 3000
    xw, _, vw = _water_box(nw, 21, box)
    xm, _, vm, imm, iatoms, lengths = _molecule_soup(9, 800, box, 0)
    x = np.concatenate([xw, xm]).astype(np.float32)
    v = np.concatenate([vw, vm]).astype(np.float32)
    mO, mH, dOH, dHH = 15.9994, 1.008, 0.09572, 0.15139
    im = np.concatenate([np.tile([1 / mO, 1 / mH, 1 / mH], nw), imm]).astype(np.float32)
    iatoms = iatoms.copy()
    iatoms[:, 1:] += 3 * nw
    # start from satisfied constraints: one projection of x onto itself-ish lengths
    x64 = x.astype(np.float64)
    x64, _, _ = ob.lincs(iatoms, lengths, im, 4, 8, x64, x64, pbc_type=3, box=np.eye(3) * box)
    x = x64.astype(np.float32)
    n = len(im)
    rng = np.random.default_rng(2)
    f = rng.normal(0, 300, (n, 3)).astype(np.float32)
    tc = (np.arange(n) % 2).astype(np.uint16)
    settles = np.arange(3 * nw, dtype=np.int32).reshape(-1, 3)
    ref_t, tau_t = [300.0, 250.0], [0.5, 2.0]
    up = pkg.UpdateConstrainGpu(dt, num_temp_coupl_groups=2, stochastic_dynamics=sd, ref_t=ref_t, tau_t=tau_t, n_lincs_iter=1, n_proj_order=4,
                                settle=(mO, mH, dOH, dHH))
    d_x, d_v, d_f = _dev(x), _dev(v), _dev(f)
    assert up.set(d_x.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), im, tc, iatoms, lengths, settles)
    bx = np.eye(3) * box
    up.set_pbc(3, bx)
    lambdas = [0.98, 1.03]
    xo, vo = x.astype(np.float64), v.astype(np.float64)
    for step in (0, 1):
        vir = up.integrate(dt, update_velocities=True, compute_virial=True, tc_lambdas=None if sd else lambdas, seed=seed, step=step)
        if sd:
            x1, xb, v1 = ob.langevin_update(0, xo, vo, f, im, tc, ref_t, tau_t, dt, seed, step)
            x1, xb, v1 = (a.astype(np.float64) for a in (x1, xb, v1))
        else:
            x1, xb, v1 = ob.leapfrog(xo, vo, f, im, dt, lambdas=lambdas, groups=tc)
        x2, v2, vl = ob.lincs(iatoms, lengths, im, 1, 4, xb, x1, v=v1, invdt=1 / dt, compute_virial=True, pbc_type=3, box=bx)
        x3, v3, vs = ob.settle(settles, mO, mH, dOH, dHH, xb, x2, v=v2, invdt=1 / dt, compute_virial=True, pbc_type=3, box=bx)
        if sd:
            x4, _, v4 = ob.langevin_update(1, x3, v3, f, im, tc, ref_t, tau_t, dt, seed, step)
            x4, v4 = x4.astype(np.float64), v4.astype(np.float64)
            x5, _, _ = ob.lincs(iatoms, lengths, im, 1, 4, xb, x4, pbc_type=3, box=bx)
            x3, _, _ = ob.settle(settles, mO, mH, dOH, dHH, xb, x5, pbc_type=3, box=bx)
            v3 = v4
        torch.cuda.synchronize()
        gx, gv = d_x.cpu().numpy(), d_v.cpu().numpy()
        assert np.max(np.abs(gx - x3)) <= 4e-6
        assert np.max(np.abs(gv - v3)) <= 2e-3
        want_vir = 0.5 / (dt * dt) * (vl + vs)
        assert np.max(np.abs(vir - want_vir)) <= 2e-3 * np.max(np.abs(want_vir))
        xo, vo = gx.astype(np.float64), gv.astype(np.float64)     # continue from the GPU state: errors do not accumulate in the check
    # scaling of coordinates and velocities (pressure coupling)
    mu = np.array([[1.01, 0, 0], [0.002, 0.99, 0], [-0.001, 0.003, 1.02]])
    before = d_x.cpu().numpy().astype(np.float64)
    up.scale_coordinates(mu)
    want = np.stack([mu[0, 0] * before[:, 0] + mu[1, 0] * before[:, 1] + mu[2, 0] * before[:, 2], mu[1, 1] * before[:, 1] + mu[2, 1] * before[:, 2],
                     mu[2, 2] * before[:, 2]], axis=1)
    assert np.allclose(d_x.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    vb = d_v.cpu().numpy().astype(np.float64)
    up.scale_velocities(np.diag([0.5, 2.0, 1.0]))
    assert np.allclose(d_v.cpu().numpy(), vb * [0.5, 2.0, 1.0], rtol=1e-6, atol=1e-6)
    assert up.x_updated_event()
    up.free()


def test_gpu_resident_md_steps_match_the_oracle_schedule():
    """x -> xq, cluster-pair + FEP kernels, force reduction, leap-frog, SETTLE over several steps with nothing leaving HBM, against
    the same schedule made of oracle calls (forces from the CPU kernels on the same pair list)."""
    import importlib
    import torch
    mdloop = importlib.import_module("gromacs_fep_gpu_amd.mdloop")
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=3, elec="rf", seed=12)
    g = c.grid
    nb = tl.setup_gpu(c, fused=True)
    n = c.natoms
    mO, mH, dOH, dHH = 15.9994, 1.008, 0.1, 0.16330
    im = np.tile([1 / mO, 1 / mH, 1 / mH], n // 3)
    settles = np.arange(n, dtype=np.int32).reshape(-1, 3)
    rng = np.random.default_rng(4)
    v0 = np.zeros((n, 3))
    v0 += np.repeat(rng.normal(0, 0.3, (n // 3, 3)), 3, axis=0)          # rigid-body translation: satisfies the constraints
    dt, bx = 0.001, np.diag(g.box.astype(np.float64))
    loop = mdloop.ShortRangeMdLoop(nb, g, g.x_wrapped, v0, im, dt, bx, settles=settles, settle_params=(mO, mH, dOH, dHH))
    ai = g.atomIndices
    real = ai >= 0
    xq0, xw0 = g.xq.copy(), g.x_wrapped.copy()
    try:
        for step in range(4):
            x_start = loop.d_x.cpu().numpy()
            v_start = loop.d_v.cpu().numpy()
            loop.step(step)
            loop.synchronize()
            # the oracle on the coordinates the step started from
            g.xq.reshape(-1, 4)[real, :3] = x_start[ai[real]]
            g.x_wrapped[:] = x_start
            want = tl.run_oracle(c, energy=False)
            f = np.zeros((n, 3))
            f[ai[real]] = want["f"][real]
            frms = np.sqrt(np.mean(f ** 2))
            assert np.max(np.abs(loop.d_f.cpu().numpy() - f)) <= 1e-4 * max(frms, np.max(np.abs(f)) * 0.05)
            x1, xb, v1 = ob.leapfrog(x_start, v_start, f, im, dt)
            x2, v2, _ = ob.settle(settles, mO, mH, dOH, dHH, xb, x1, v=v1, invdt=1 / dt, pbc_type=3, box=bx)
            assert np.max(np.abs(loop.d_x.cpu().numpy() - x2)) <= 2e-6
            assert np.max(np.abs(loop.d_v.cpu().numpy() - v2)) <= 2e-3
            w = loop.d_x.cpu().numpy().astype(np.float64).reshape(-1, 3, 3)
            d = w[:, 0] - w[:, 1]
            d -= np.rint(d / np.diag(bx)) * np.diag(bx)
            assert np.max(np.abs(np.linalg.norm(d, axis=1) - dOH)) <= 2e-6
    finally:
        g.xq[:] = xq0
        g.x_wrapped[:] = xw0
    loop.free()
    nb.free()


@pytest.mark.parametrize("sd", [False, True])
def test_fused_update_equals_the_kernel_sequence(sd):
    """MI355X extension: force gather + integrator + SETTLE + clear + xq write in one kernel must give the trajectory of the
    reference's sequence x -> xq, clear, kernels, reduction, integrator, SETTLE (waters and, for some molecules left out of the
    SETTLE list, unconstrained atoms; temperature coupling or stochastic dynamics)."""
    import importlib
    import torch
    mdloop = importlib.import_module("gromacs_fep_gpu_amd.mdloop")
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=3, elec="ewald", seed=31)
    g = c.grid
    n = c.natoms
    mO, mH = 15.9994, 1.008
    im = np.tile([1 / mO, 1 / mH, 1 / mH], n // 3)
    settles = np.arange(n, dtype=np.int32).reshape(-1, 3)
    settles = np.delete(settles, np.arange(5, len(settles), 17), axis=0)      # these waters are integrated atom by atom
    rng = np.random.default_rng(8)
    v0 = np.repeat(rng.normal(0, 0.3, (n // 3, 3)), 3, axis=0)
    tc = (np.arange(n) // 3 % 2).astype(np.uint16)
    kw = dict(settles=settles, settle_params=(mO, mH, 0.1, 0.16330), temp_coupl_groups=tc, num_temp_coupl_groups=2, stochastic_dynamics=sd,
              ref_t=[300.0, 280.0], tau_t=[0.1, 1.0])
    dt, bx = 0.001, np.diag(g.box.astype(np.float64))
    out = {}
    for fused_update in (False, True):
        nb = tl.setup_gpu(c, fused=True)
        loop = mdloop.ShortRangeMdLoop(nb, g, g.x_wrapped, v0, im, dt, bx, fused_update=fused_update, **kw)
        vir = None
        for step in range(3):
            sw = pkg.step_workload(energy=(step == 1), virial=(step == 1), dhdl=False)
            vir = loop.step(step, step_work=sw, seed=5, compute_virial=(step == 2), tc_lambdas=None if sd else [0.97, 1.02])
        loop.synchronize()
        xq = np.zeros((g.num_atoms, 4), np.float32)
        pkg.hip_lib().nbnxm_gpu_debug_download(nb.h, C.c_void_p(pkg.hip_lib().nbnxm_gpu_get_xq(nb.h)), xq.ctypes.data_as(C.c_void_p),
                                               C.c_size_t(xq.nbytes))
        out[fused_update] = (loop.d_x.cpu().numpy(), loop.d_v.cpu().numpy(), vir, xq)
        loop.free()
        nb.free()
    (x0, vv0, vir0, _), (x1, vv1, vir1, xq1) = out[False], out[True]
    assert np.max(np.abs(x1 - x0)) <= 2e-6
    assert np.max(np.abs(vv1 - vv0)) <= 2e-3
    assert np.max(np.abs(vir1 - vir0)) <= 1e-3 * np.max(np.abs(vir0)) and np.max(np.abs(vir0)) > 0
    # the non-bonded coordinates of the next step are the updated coordinates, charges untouched
    ai = g.atomIndices
    real = ai >= 0
    assert np.array_equal(xq1[real, :3], x1[ai[real]])
    assert np.array_equal(xq1[:, 3], g.xq.reshape(-1, 4)[:, 3])


def test_nve_energy_conservation_of_the_gpu_resident_loop():
    """Forces and energies of the kernels must belong to one Hamiltonian, and the update + SETTLE must integrate it: total energy
    (non-bonded potential from energy steps + kinetic energy, average of the two half-step values) over 150 steps of 0.1 fs of a
    water box with a half-decoupled solute (reaction field, soft-core at lambda = 0.5) stays constant while kinetic and potential
    energy exchange ten thousand times the allowed drift (measured: 0.3 kJ/mol of 10,300, tools/nve_probe.py).  15 fs only: the
    start from a lattice heats the box to several hundred kelvin, and the pair list, built once with a 0.1 nm buffer, must stay
    valid."""
    import importlib
    import torch
    mdloop = importlib.import_module("gromacs_fep_gpu_amd.mdloop")
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=3, elec="rf", seed=5)
    g = c.grid
    nb = tl.setup_gpu(c, fused=True)
    n = c.natoms
    mO, mH = 15.9994, 1.008
    mass = np.tile([mO, mH, mH], n // 3)
    settles = np.arange(n, dtype=np.int32).reshape(-1, 3)
    dt = 0.0001     # the start from a lattice is violent: the half-step average of the kinetic energy is off by dt^2 m a^2 / 8
    loop = mdloop.ShortRangeMdLoop(nb, g, g.x_wrapped, np.zeros((n, 3)), 1.0 / mass, dt, np.diag(g.box.astype(np.float64)), settles=settles,
                                   settle_params=(mO, mH, 0.1, 0.16330), fused_update=True)
    sw_e = pkg.step_workload(energy=True, virial=False, dhdl=False)

    def kinetic():
        loop.synchronize()          # the loop runs on the non-bonded object's own stream: torch's copy alone does not wait for it
        v = loop.d_v.cpu().numpy().astype(np.float64)
        return 0.5 * float(np.sum(mass[:, None] * v * v))

    etot, ekin, epot = [], [], []
    for step in range(150):
        if step % 5 == 0:
            k0 = kinetic()
            loop.step(step, step_work=sw_e)
            f = np.zeros((g.num_atoms, 3), np.float32)
            nb.launch_cpyback(f, sw_e)                              # energies of x(t); the forces are consumed (and cleared) already
            res = nb.wait_finish_task(sw_e, c.have_soft_core)
            k1 = kinetic()
            ekin.append(0.5 * (k0 + k1))
            epot.append(res["e_lj"] + res["e_el"])
            etot.append(ekin[-1] + epot[-1])
        else:
            loop.step(step)
    loop.synchronize()
    etot, ekin, epot = np.array(etot), np.array(ekin), np.array(epot)
    exchange = ekin.max() - ekin.min()
    drift = np.max(np.abs(etot - etot[0]))
    assert exchange > 50.0                       # kJ/mol: the lattice start relaxes, plenty of energy moves
    assert drift < 1e-3 * exchange, (drift, exchange, etot.tolist())
    loop.free()
    nb.free()


def test_update_edge_cases():
    """empty constraint sets are no-ops, an update without any constraint is plain leap-frog (also through the fused path), lists
    can be replaced by larger and smaller ones on the same objects (search steps)"""
    import torch
    n = 1000
    x, v, f, im, tc = _system(n, 4)
    d_x, d_xp, d_v = _dev(x), _dev(x + 0.01), _dev(v)
    sg = pkg.SettleGpu(15.9994, 1.008, 0.09572, 0.15139)
    sg.set(np.zeros((0, 3), np.int32))
    vir = sg.apply(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), 500.0, True, 3, np.eye(3) * 5)
    lg = pkg.LincsGpu(1, 4)
    assert lg.set(np.zeros((0, 3), np.int32), [0.1], im)
    vir2 = lg.apply(d_x.data_ptr(), d_xp.data_ptr(), d_v.data_ptr(), 500.0, True, 3, np.eye(3) * 5)
    torch.cuda.synchronize()
    assert not vir.any() and not vir2.any()
    assert np.array_equal(d_xp.cpu().numpy(), (x + 0.01).astype(np.float32)) and np.array_equal(d_v.cpu().numpy(), v)
    # growing and shrinking lists on the same objects
    for count in (300, 40, 333):
        iat = np.stack([np.zeros(count, np.int32), np.arange(count, dtype=np.int32) * 3, np.arange(count, dtype=np.int32) * 3 + 1], axis=1)
        assert lg.set(iat, [0.1], im)
        xx = x.copy()
        xx[iat[:, 2]] = xx[iat[:, 1]] + np.array([0.1, 0, 0], np.float32)
        xp = xx + np.random.default_rng(count).normal(0, 0.004, xx.shape).astype(np.float32)
        d_a, d_b = _dev(xx), _dev(xp)
        lg.apply(d_a.data_ptr(), d_b.data_ptr(), None, 500.0, False, 0, None)
        torch.cuda.synchronize()
        oxp, _, _ = ob.lincs(iat, [0.1], im, 1, 4, xx.astype(np.float64), xp.astype(np.float64), pbc_type=0)
        assert np.max(np.abs(d_b.cpu().numpy() - oxp)) <= 2e-6 * max(1.0, np.abs(xp).max())
    sg.free()
    lg.free()
    # no constraints at all: the composite is the integrator
    up = pkg.UpdateConstrainGpu(0.002, num_temp_coupl_groups=3)
    d_x, d_v, d_f = _dev(x), _dev(v), _dev(f)
    assert up.set(d_x.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), im, tc)
    up.set_pbc(0, None)
    lam = [0.9, 1.0, 1.1]
    vir = up.integrate(0.002, compute_virial=True, tc_lambdas=lam)
    xo, _, vo = ob.leapfrog(x, v, f, im, 0.002, lambdas=lam, groups=tc)
    torch.cuda.synchronize()
    assert not vir.any()
    assert np.allclose(d_x.cpu().numpy(), xo, rtol=1e-6, atol=1e-6) and np.allclose(d_v.cpu().numpy(), vo, rtol=2e-6, atol=1e-6)
    # the same through the fused kernel: grid slots = a permutation, forces in grid order
    perm = np.random.default_rng(0).permutation(n).astype(np.int32)
    d_xq = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    f_grid = np.zeros((n, 3), np.float32)
    f_grid[perm] = f
    d_fg = _dev(f_grid)
    d_x, d_v = _dev(x), _dev(v)
    assert up.set(d_x.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), im, tc)
    assert not up.can_fuse()                      # the coupling belongs to a search: it is gone after set()
    up.set_nbat_coupling(perm, d_xq.data_ptr(), d_fg.data_ptr())
    assert up.can_fuse()
    up.integrate_fused(0.002, tc_lambdas=lam)
    torch.cuda.synchronize()
    assert np.allclose(d_x.cpu().numpy(), xo, rtol=1e-6, atol=1e-6) and np.allclose(d_v.cpu().numpy(), vo, rtol=2e-6, atol=1e-6)
    assert not d_fg.cpu().numpy().any()                                   # forces consumed and cleared
    assert np.array_equal(d_xq.cpu().numpy()[perm, :3], d_x.cpu().numpy())  # next step's coordinates in grid order
    up.free()


def test_fused_update_with_lincs_molecules():
    """waters (SETTLE in registers), molecules with LINCS constraints (old coordinates kept for the LINCS kernel, which also writes
    xq) and free atoms: the fused path must reproduce the kernel sequence, including the virial and the next step's xq"""
    import torch
    box, dt = 5.0, 0.002
    nw = 2000
    xw, _, vw = _water_box(nw, 5, box)
    xm, _, vm, imm, iatoms, lengths = _molecule_soup(6, 600, box, 0)
    nfree = 500
    rng = np.random.default_rng(7)
    x = np.concatenate([xw, xm, rng.uniform(0, box, (nfree, 3))]).astype(np.float32)
    v = np.concatenate([vw, vm, rng.normal(0, 0.5, (nfree, 3))]).astype(np.float32)
    mO, mH, dOH, dHH = 15.9994, 1.008, 0.09572, 0.15139
    im = np.concatenate([np.tile([1 / mO, 1 / mH, 1 / mH], nw), imm, np.full(nfree, 1 / 12.0)]).astype(np.float32)
    iatoms = iatoms.copy()
    iatoms[:, 1:] += 3 * nw
    x64, _, _ = ob.lincs(iatoms, lengths, im, 4, 8, x.astype(np.float64), x.astype(np.float64), pbc_type=3, box=np.eye(3) * box)
    x = x64.astype(np.float32)
    n = len(im)
    f = rng.normal(0, 300, (n, 3)).astype(np.float32)
    tc = (np.arange(n) % 2).astype(np.uint16)
    settles = np.arange(3 * nw, dtype=np.int32).reshape(-1, 3)
    perm = rng.permutation(n + 64)[:n].astype(np.int32)          # grid slots, with holes (filler slots)
    out = {}
    for fused in (False, True):
        up = pkg.UpdateConstrainGpu(dt, num_temp_coupl_groups=2, n_lincs_iter=1, n_proj_order=4, settle=(mO, mH, dOH, dHH))
        d_x, d_v, d_f = _dev(x), _dev(v), _dev(f)
        d_xq = torch.full((n + 64, 4), 7.0, dtype=torch.float32, device="cuda")
        f_grid = np.zeros((n + 64, 3), np.float32)
        f_grid[perm] = f
        d_fg = _dev(f_grid)
        assert up.set(d_x.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), im, tc, iatoms, lengths, settles)
        up.set_pbc(3, np.eye(3) * box)
        if fused:
            up.set_nbat_coupling(perm, d_xq.data_ptr(), d_fg.data_ptr())
            assert up.can_fuse()
            vir = up.integrate_fused(dt, compute_virial=True, tc_lambdas=[0.98, 1.03])
        else:
            vir = up.integrate(dt, compute_virial=True, tc_lambdas=[0.98, 1.03])
        torch.cuda.synchronize()
        out[fused] = (d_x.cpu().numpy(), d_v.cpu().numpy(), vir, d_xq.cpu().numpy(), d_fg.cpu().numpy())
        up.free()
    (x0, v0, vir0, _, _), (x1, v1, vir1, xq1, fg1) = out[False], out[True]
    assert np.max(np.abs(x1 - x0)) <= 2e-6 and np.max(np.abs(v1 - v0)) <= 2e-3
    assert np.max(np.abs(vir1 - vir0)) <= 1e-3 * np.max(np.abs(vir0))
    assert np.array_equal(xq1[perm, :3], x1) and (xq1[:, 3] == 7.0).all()
    holes = np.setdiff1d(np.arange(n + 64), perm)
    assert (xq1[holes] == 7.0).all() and not fg1.any()
    # stochastic dynamics with LINCS constraints is the one combination the fused path leaves to the kernel sequence
    up = pkg.UpdateConstrainGpu(dt, num_temp_coupl_groups=2, stochastic_dynamics=True, ref_t=[300.0, 300.0], tau_t=[1.0, 1.0],
                                settle=(mO, mH, dOH, dHH))
    d_x, d_v, d_f = _dev(x), _dev(v), _dev(f)
    assert up.set(d_x.data_ptr(), d_v.data_ptr(), d_f.data_ptr(), im, tc, iatoms, lengths, settles)
    up.set_nbat_coupling(perm, d_xq.data_ptr(), d_fg.data_ptr())
    assert not up.can_fuse()
    up.free()


def test_md_loop_with_listed_forces_on_the_nonbonded_buffers():
    """The whole short-range schedule on one stream: cluster-pair + perturbed-pair kernels, the listed-forces kernel working on
    the non-bonded xq / f buffers in grid order (perturbed bonds, angles and soft-core 1-4 pairs with the A/B charges of q4),
    force reduction or fused update.  Forces against the sum of the oracles; fused and kernel-sequence trajectories agree."""
    import importlib
    import torch
    mdloop = importlib.import_module("gromacs_fep_gpu_amd.mdloop")
    c = tl.make_case(nm=(8, 8, 8), num_perturbed_molecules=3, elec="rf", seed=19)
    g = c.grid
    n = c.natoms
    ai = g.atomIndices
    real = ai >= 0
    cell = np.full(n, -1, np.int32)
    cell[ai[real]] = np.nonzero(real)[0]
    x0 = g.x_wrapped.astype(np.float64)
    bx = np.diag(g.box.astype(np.float64))
    # listed interactions among the oxygens of the first six waters (three of them perturbed) and 1-4 pairs between hydrogens
    ox = np.arange(0, 18, 3)

    def dist(a, b):
        d = x0[a] - x0[b]
        d -= np.rint(d / g.box) * g.box
        return float(np.linalg.norm(d))

    prm = np.zeros(3, ob.LISTED_IPARAMS)
    prm["p"][0, :4] = [dist(ox[0], ox[1]) * 0.97, 3000.0, dist(ox[0], ox[1]) * 1.02, 2000.0]     # bond, A != B
    prm["p"][1, :4] = [100.0, 300.0, 110.0, 200.0]                                                  # angle
    prm["p"][2, :4] = [2.0e-3, 2.0e-6, 0.0, 0.0]                                                    # 1-4 pair vanishing in B
    lists = {"bonds": np.array([[0, ox[i], ox[i + 1]] for i in range(5)], np.int32),
             "angles": np.array([[1, ox[i], ox[i + 1], ox[i + 2]] for i in range(4)], np.int32),
             "lj14": np.array([[2, 1, 4], [2, 2, 7], [2, 5, 8], [2, 10, 13]], np.int32)}
    lam = dict(bonded=0.4, coul=0.5, vdw=0.5)
    fep = pkg.ListedFepParams(c.sc_alpha, c.sc_alpha, c.sc_power, c.sc_sigma ** 6, c.sc_sigma ** 6, lam["bonded"], lam["coul"], lam["vdw"], 0.0)
    elec_scale = c.epsfac * 0.8333
    mO, mH = 15.9994, 1.008
    im = np.tile([1 / mO, 1 / mH, 1 / mH], n // 3)
    settles = np.arange(n, dtype=np.int32).reshape(-1, 3)
    dt = 0.0005
    gp = np.zeros(3, pkg.LISTED_IPARAMS)
    gp["p"], gp["mult"] = prm["p"].astype(np.float32), prm["mult"]
    traj = {}
    for fused_update in (False, True):
        nb = tl.setup_gpu(c, fused=True)
        lg = pkg.ListedGpu(stream=nb.stream())
        lg.set_force_params(gp)
        for name, ia in lists.items():
            ia_grid = ia.copy()
            ia_grid[:, 1:] = cell[ia[:, 1:]]
            lg.update_interaction_list(name, ia_grid, g.num_atoms)
        loop = mdloop.ShortRangeMdLoop(nb, g, g.x_wrapped, np.zeros((n, 3)), im, dt, bx, settles=settles, settle_params=(mO, mH, 0.1, 0.16330),
                                       fused_update=fused_update, listed=lg, listed_fep=fep, listed_elec_scale=elec_scale)
        if not fused_update:
            loop.compute_forces()
            loop.synchronize()
            want = tl.run_oracle(c, energy=False)
            f = np.zeros((n, 3))
            f[ai[real]] = want["f"][real]
            f_listed = np.zeros((n, 3))
            for name in ("bonds", "angles"):
                f_listed += ob.listed(name, lists[name], prm, x0, g.box.astype(np.float64), 3, lam["bonded"])["f"]
            pf = ob.ListedPairsFep(c.sc_alpha, c.sc_alpha, c.sc_power, 0, c.sc_sigma ** 6, c.sc_sigma ** 6, lam["coul"], lam["vdw"])
            f_listed += ob.listed_pairs(lists["lj14"], prm, x0, c.sys["qA"], c.sys["qB"], g.box.astype(np.float64), 3, pf, elec_scale)["f"]
            assert np.abs(f_listed).max() > 100.0                       # the listed part is not a rounding error of the test
            got = loop.d_f.cpu().numpy()
            tot = f + f_listed
            frms = np.sqrt(np.mean(tot ** 2))
            assert np.max(np.abs(got - tot)) <= 1e-4 * max(frms, 0.05 * np.abs(tot).max())
        for step in range(3):
            loop.step(step)
        loop.synchronize()
        traj[fused_update] = (loop.d_x.cpu().numpy(), loop.d_v.cpu().numpy())
        loop.free()
        lg.free()
        nb.free()
    assert np.max(np.abs(traj[True][0] - traj[False][0])) <= 2e-6
    assert np.max(np.abs(traj[True][1] - traj[False][1])) <= 2e-3


@pytest.mark.parametrize("ncells", [(2, 1, 1), (2, 2, 2), "rccl-self"])
def test_domain_decomposed_md_steps_with_whole_molecules(ncells):
    """Config 5 end to end on one GPU (all ranks in this process, in-process halo double): per step halo x, local and non-local
    kernels on the rank's own lists, halo f, then leap-frog + SETTLE on the rank's home molecules — rows [0, num_home) of the
    rank's own arrays.  The trajectory of every atom, taken from its owner, equals the single-domain GPU-resident loop (positions
    modulo the box: a rank keeps its molecules whole and unwrapped along the decomposed dimensions)."""
    import importlib
    import torch
    domdec = importlib.import_module("gromacs_fep_gpu_amd.domdec")
    mdloop = importlib.import_module("gromacs_fep_gpu_amd.mdloop")
    wl = importlib.import_module("gromacs_fep_gpu_amd.workload")
    c = tl.make_case(nm=(10, 10, 10), num_perturbed_molecules=3, elec="ewald", seed=52)
    g = c.grid
    n = c.natoms
    mO, mH = 15.9994, 1.008
    im = np.tile([1 / mO, 1 / mH, 1 / mH], n // 3)
    settles = np.arange(n, dtype=np.int32).reshape(-1, 3)
    rng = np.random.default_rng(3)
    v0 = np.repeat(rng.normal(0, 0.3, (n // 3, 3)), 3, axis=0)
    dt, bx = 0.001, np.diag(g.box.astype(np.float64))
    sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
    # single domain
    nb0 = tl.setup_gpu(c, fused=True)
    ref = mdloop.ShortRangeMdLoop(nb0, g, g.x_wrapped, v0, im, dt, bx, settles=settles, settle_params=(mO, mH, 0.1, 0.16330))
    for step in range(3):
        ref.step(step)
    ref.synchronize()
    x_ref, v_ref = ref.d_x.cpu().numpy(), ref.d_v.cpu().numpy()
    ref.free()
    nb0.free()
    # decomposed
    # "rccl-self": ONE rank that is its own neighbour in every dimension, the real RCCL transport and the one-call C++ force step
    # (halo_gpu_domain_force_step) with the update's x-updated event as the dependency of the next step's coordinate reads
    rccl = ncells == "rccl-self"
    dd = domdec.DomainDecomposition(c.sys["x"], c.sys["box"], c.sys["molId"], (1, 1, 1) if rccl else ncells, c.rlist,
                                    self_links=(True, True, True) if rccl else (False, False, False))
    steps, halos = [], []
    for r in range(dd.num_ranks):
        plan = dd.plan(r)
        system = domdec.RankSystem(pkg, plan, c.sys["box"], c.sys["qA"], c.sys["qB"], c.sys["typeA"], c.sys["typeB"], c.ntype,
                                   c.sys["molId"], c.rlist, perturbed=c.perturbed)
        nb = domdec.make_rank_gpu(pkg, wl, c, system, use_dynamic_pruning=False)
        halo = domdec.RcclHalo(pkg, None, 0, 1, nb.stream(pkg.NONLOCAL)) if rccl else domdec.TensorHalo(peers={})
        steps.append(domdec.DomainMdStep(pkg, nb, system, halo, v0, im, dt, bx, settles=settles, settle_params=(mO, mH, 0.1, 0.16330)))
        halos.append(halo)
    for step in range(3):
        if rccl:
            steps[0].md_step(sw, step)       # no host synchronisation between the steps
            continue
        torch.cuda.synchronize()
        domdec.loopback_exchange_coordinates(halos)
        torch.cuda.synchronize()
        for s in steps:
            s.launch(sw)
            s.reduce_halo_forces()
        torch.cuda.synchronize()
        domdec.loopback_exchange_forces(halos)
        torch.cuda.synchronize()
        for s in steps:
            s.reduce_home_forces()
            s.integrate(step)
    torch.cuda.synchronize()
    x_dd, v_dd = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    for s in steps:
        home = s.sys.plan.home
        x_dd[home] = s.d_x.cpu().numpy()[:len(home)]
        v_dd[home] = s.d_v.cpu().numpy()
    dx = (x_dd - x_ref).astype(np.float64)
    dx -= g.box.astype(np.float64) * np.rint(dx / g.box.astype(np.float64))
    assert np.max(np.abs(dx)) <= 5e-6
    assert np.max(np.abs(v_dd - v_ref)) <= 5e-3
    for s, h in zip(steps, halos):
        s.update.free()
        h.free()
        s.nb.free()
