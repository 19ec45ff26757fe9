"""The reference's nblib known-answer systems (tests/golden/nblib_refdata.json, made by golden/make_nblib_golden.py)
as inputs of the cluster-pair path: grid + GPU-layout list from libnbnxm_host, parameters as nblib's
createInteractionConst() sets them (api/nblib/nbnxmsetuphelpers.cpp:232-292).  Shared by the oracle test (CPU) and
the HIP test (GPU)."""
import json
import os
from types import SimpleNamespace

import numpy as np

import fep_testlib as tl
import oracle_binding as ob

pkg = tl.pkg
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nblib_refdata.json")


def load():
    with open(GOLDEN) as fh:
        return json.load(fh)


def build(data, case):
    s = case["system"]
    names = sorted(set(s["type_names"]), key=s["type_names"].index)
    ntype = len(names)
    atype = np.array([names.index(t) for t in s["type_names"]], np.int32)
    c6 = np.array([data["c6"][t] for t in names])
    c12 = np.array([data["c12"][t] for t in names])
    nbfp = np.zeros((ntype, ntype, 2))
    nbfp[..., 0] = 6.0 * np.sqrt(np.outer(c6, c6))       # interactions.cpp:83-88, nbnxmsetuphelpers.cpp:152-178
    nbfp[..., 1] = 12.0 * np.sqrt(np.outer(c12, c12))
    nbfp64 = nbfp.reshape(-1).copy()
    nbfp = nbfp.reshape(-1).astype(np.float32)
    x = np.array(s["x"], np.float32)
    q = np.array(s["q"], np.float32)
    box = np.full(3, s["box"], np.float32)
    grid = pkg.Grid(x, box, q, q, atype, atype, ntype)
    excl_index, excl_atoms = pkg.exclusions_from_groups(np.array(s["molecule"]))
    rc = float(data["cutoff"])
    plist = grid.build_pairlist(excl_index, excl_atoms, rc, 0, carve_fep=False)
    c = SimpleNamespace(grid=grid, plist=plist, nbfp=nbfp, ntype=ntype, x=x, q=q, box=box, rc=rc, natoms=len(q))
    # the same inputs in double (the reference's tighter known answers come from a double build): all coordinates lie inside
    # the box, so the grid's float xq are the rounded inputs and the double ones can be put in their slots
    real = grid.atomIndices >= 0
    c.xq64 = grid.xq.reshape(-1, 4).astype(np.float64)
    c.xq64[real, :3] = np.array(s["x"], np.float64)[grid.atomIndices[real]]
    c.xq64[real, 3] = np.array(s["q"], np.float64)[grid.atomIndices[real]]
    c.x64 = np.array(s["x"], np.float64)
    t = ntype
    c.nbfp64 = np.zeros((t + 1, t + 1, 2))
    c.nbfp64[:t, :t] = nbfp64.reshape(t, t, 2)
    c.nbfp64 = c.nbfp64.reshape(-1)
    c.epsfac = ob.ONE_4PI_EPS0
    c.k_rf = 0.0
    c.c_rf = 1.0 / rc
    c.beta = 0.0
    c.sh_ewald = 0.0     # nblib never sets it
    if case["coulomb"] == "pme":
        c.beta = ob.lib().oracle_calc_ewaldcoeff_q(rc, data["ewald_rtol"])
        c.elec_type = pkg.ELEC_EWALD_ANA
    else:
        c.elec_type = pkg.ELEC_RF      # eeltype Cut runs the reaction-field kernel with k_rf = 0 (nbnxm/kerneldispatch.cpp)
    c.disp_shift = (0.0, 0.0, -rc ** -6)
    c.rep_shift = (0.0, 0.0, -rc ** -12)
    return c


def ref_params(c):
    p = ob.RefParams()
    p.elecType, p.vdwType = c.elec_type, pkg.VDW_CUT
    p.epsfac, p.c_rf, p.k_rf = c.epsfac, c.c_rf, c.k_rf
    p.ewaldcoeff_q, p.sh_ewald = c.beta, c.sh_ewald
    p.rcoulomb = p.rvdw = p.rlist = c.rc
    p.disp_c2, p.disp_c3, p.disp_cpot = c.disp_shift
    p.rep_c2, p.rep_c3, p.rep_cpot = c.rep_shift
    return p


def run_oracle(c, precision="f64"):
    g = c.grid
    dbl = precision == "f64"
    r = ob.nbnxm_ref(c.plist.sci, c.plist.cjPacked, c.plist.excl, c.xq64 if dbl else g.xq, g.type, g.num_types,
                     c.nbfp64 if dbl else g.nbat_nbfp(c.nbfp), ref_params(c), g.shift_vec, compute_energy=True, compute_fshift=True,
                     precision=precision)
    return finish(c, np.asarray(r["f"], np.float64), np.asarray(r["fshift"], np.float64), r["Vc"], r["Vv"], c.x64 if dbl else c.x)


def run_gpu(c):
    g = c.grid
    ic = pkg.make_interaction_params(c.elec_type, pkg.VDW_CUT, c.epsfac, c.rc, c.rc, c.rc, k_rf=c.k_rf, c_rf=c.c_rf,
                                     ewaldcoeff_q=c.beta, sh_ewald=c.sh_ewald, dispersion_shift=c.disp_shift,
                                     repulsion_shift=c.rep_shift)
    nb = pkg.NbnxmGpu(ic, g.num_types, g.nbat_nbfp(c.nbfp))
    nb.init_atomdata(g.num_atoms, g.type)
    nb.init_pairlist(c.plist.sci, c.plist.cjPacked, c.plist.excl)
    nb.upload_shiftvec(g.shift_vec)
    nb.copy_xq_to_gpu(g.xq)
    sw = pkg.step_workload(energy=True, virial=True)
    nb.clear_outputs(True)
    nb.launch_kernel(sw)
    f = np.zeros((g.num_atoms, 3), np.float32)
    nb.launch_cpyback(f, sw)
    res = nb.wait_finish_task(sw, False)
    nb.free()
    return finish(c, f.astype(np.float64), res["fshift"].astype(np.float64), res["e_el"], res["e_lj"])


def finish(c, f_grid, fshift, e_el, e_lj, x=None):
    """grid order -> topology order; virial as api/nblib/virials.cpp:53-84 computes it"""
    g = c.grid
    real = g.atomIndices >= 0
    f = np.zeros((c.natoms, 3))
    f[g.atomIndices[real]] = f_grid[real]
    vir = -0.5 * (np.einsum("si,sj->ij", g.shift_vec.astype(np.float64), fshift)
                  + np.einsum("ai,aj->ij", np.asarray(c.x if x is None else x, np.float64), f))
    return dict(f=f, energies=[e_el, e_lj, 0.0, 0.0, 0.0], virial=vir.reshape(-1))


def check(case, got, abs_tol=1e-6, rel_floor=0.0):
    """the reference's RefDataChecker (api/nblib/tests/testhelpers.h:70-84): passes within the relative tolerance of the test
    or within the absolute one (1e-6 in a float build, 1e-9 in a double build).  rel_floor: fp32 round-off floor for the HIP
    kernel where the reference's own test ran in double (the 1e-7 / 1e-8 argon cases)."""
    tol = max(case["tolerance"], rel_floor)
    for key in ("forces", "energies", "virial"):
        if key not in case:
            continue
        want = np.array(case[key], np.float64)
        have = np.asarray(got["f" if key == "forces" else key], np.float64).reshape(want.shape)
        bad = np.abs(have - want) > np.maximum(tol * np.abs(want), abs_tol)
        assert not bad.any(), "%s %s: got %s want %s" % (case["name"], key, have[bad], want[bad])
