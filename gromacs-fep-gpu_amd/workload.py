"""The synthetic FEP workloads of SURVEY §8d as inputs of the HIP path: water box + perturbed ligand, interaction
parameters as GROMACS' forcerec / interaction_const would set them, grid and pair lists from libnbnxm_host, and the
call sequence of one force step through the C ABI.  Used by bench.py and by the parity tests (tests/fep_testlib.py adds
the oracle side).  Nothing here touches the CPU oracle: without libnbnxm_hip.so and a GPU, setup_gpu / run_gpu fail loudly.
"""
import ctypes as C
import math
from types import SimpleNamespace

import numpy as np

import gromacs_fep_gpu_amd as pkg

ONE_4PI_EPS0 = 138.93545764438198  # gmx::c_one4PiEps0, api/legacy/include/gromacs/math/units.h:110


def calc_ewaldcoeff_q(rc, rtol):
    """ewald/ewald_utils.cpp:43-70 (libnbnxm_host)"""
    fn = pkg.host_lib().nbnxm_host_calc_ewaldcoeff_q
    fn.restype = C.c_double
    return fn(C.c_double(rc), C.c_double(rtol))


def calc_ewaldcoeff_lj(rc, rtol):
    """ewald/ewald_utils.cpp:72-112 (libnbnxm_host)"""
    fn = pkg.host_lib().nbnxm_host_calc_ewaldcoeff_lj
    fn.restype = C.c_double
    return fn(C.c_double(rc), C.c_double(rtol))


def count_pairs_within(case, rc=None):
    """interacting atom pairs within rc, each once (the reference benchmark's "useful" pairs)"""
    fn = pkg.host_lib().nbnxm_host_count_pairs_within
    fn.restype = C.c_longlong
    ei, ea = np.ascontiguousarray(case.excl_index, np.int32), np.ascontiguousarray(case.excl_atoms, np.int32)
    return int(fn(C.c_void_p(case.grid._h), C.c_float(case.rc if rc is None else rc), ei.ctypes.data_as(C.c_void_p),
                  ea.ctypes.data_as(C.c_void_p)))


def make_case(nm=(8, 8, 8), num_perturbed_molecules=3, elec="rf", vdw="cut", seed=2026, rc=1.0, rlist=1.1,
              rlist_fep=None, sc_alpha=0.5, sc_power=1, sc_sigma=0.3, sc_coul=True, lambda_coul=0.5,
              lambda_vdw=0.5, n_lambda=0, max_cjpacked_per_sci=0, identical_states=False, rvdw_switch=0.8,
              spacing=0.310736, jitter=0.03, num_extra_types=0, rvdw=None, softcore="beutler",
              gapsys=(0.85, 0.3, 0.3), build_lists=True):
    """elec: 'rf' | 'cut' | 'ewald' | 'ewald_tab' | 'ewald_tab_kept' (the tabulated kernels themselves instead of the analytical ones a
    tabulated pick runs by default: setup_gpu calls nbnxm_gpu_set_kernel_routing);  vdw: 'cut' | 'pswitch' | 'fswitch' | 'comb_geom' | 'comb_lb' |
    'ewald_geom' | 'ewald_lb' (LJ-PME real-space part; perturbed pairs as in the CPU kernel, with the grid correction)."""
    keep_kernel_pick = elec.endswith("_kept")
    elec = elec[:-len("_kept")] if keep_kernel_pick else elec
    sysd = pkg.make_water_box(nm[0], nm[1], nm[2], spacing=spacing, jitter=jitter, seed=seed,
                              num_perturbed_molecules=num_perturbed_molecules)
    n = len(sysd["qA"])
    perturbed = (sysd["qA"] != sysd["qB"]) | (sysd["typeA"] != sysd["typeB"])
    if identical_states:
        # keep the perturbed flags but make B == A (oracle cross-check in the non-perturbed limit)
        sysd["qB"] = sysd["qA"].copy()
        sysd["typeB"] = sysd["typeA"].copy()
    if num_extra_types > 0:
        add_oxygen_type_variants(sysd, num_extra_types, seed)
    ntype = sysd["ntype"]
    if rlist_fep is None:
        rlist_fep = rlist + 0.0886  # rlist_fep 1.1886 for rlist 1.1
    grid = plist = plist_fused = excl_index = excl_atoms = None
    if build_lists:     # (a domain-decomposed run builds grids and lists per rank instead: domdec.RankSystem)
        grid = pkg.Grid(sysd["x"], sysd["box"], sysd["qA"], sysd["qB"], sysd["typeA"], sysd["typeB"], ntype,
                        perturbed=perturbed.astype(np.uint8))
        excl_index, excl_atoms = pkg.exclusions_from_groups(sysd["molId"])
        plist = grid.build_pairlist(excl_index, excl_atoms, rlist, max_cjpacked_per_sci, carve_fep=True,
                                    rlist_fep=rlist_fep)
        plist_fused = grid.build_pairlist(excl_index, excl_atoms, rlist, max_cjpacked_per_sci, carve_fep=False,
                                          rlist_fep=rlist_fep)

    c = SimpleNamespace()
    c.sys, c.grid, c.plist, c.plist_fused = sysd, grid, plist, plist_fused
    c.natoms, c.ntype = n, ntype
    c.perturbed = perturbed
    c.excl_index, c.excl_atoms = excl_index, excl_atoms
    c.rc, c.rlist, c.rlist_fep = rc, rlist, rlist_fep
    # rvdw < rcoulomb: the twin-range flavours of the Ewald kernels (ElecType::EwaldAnaTwin / EwaldTabTwin)
    c.rvdw = rc if rvdw is None else float(rvdw)
    assert c.rvdw == rc or (c.rvdw < rc and elec in ("ewald", "ewald_tab"))
    c.elec, c.vdw = elec, vdw
    c.keep_kernel_pick = keep_kernel_pick
    c.epsfac = ONE_4PI_EPS0
    c.k_rf = c.c_rf = c.beta = c.sh_ewald = 0.0
    if elec == "rf":          # epsilon_rf = infinity
        c.k_rf = 1.0 / (2.0 * rc ** 3)
        c.c_rf = 1.0 / rc + c.k_rf * rc * rc
        c.elec_type = pkg.ELEC_RF
    elif elec == "cut":
        c.c_rf = 1.0 / rc
        c.elec_type = pkg.ELEC_CUT
    else:
        c.beta = calc_ewaldcoeff_q(rc, 1e-5)
        c.sh_ewald = math.erfc(c.beta * rc) / rc
        c.elec_type = pkg.ELEC_EWALD_ANA if elec == "ewald" else pkg.ELEC_EWALD_TAB
        if c.rvdw < rc:
            c.elec_type = pkg.ELEC_EWALD_ANA_TWIN if elec == "ewald" else pkg.ELEC_EWALD_TAB_TWIN
    # potential-shift modifier (GROMACS default): V(rvdw) = 0
    rv = c.rvdw
    c.disp_shift = (0.0, 0.0, -rv ** -6)
    c.rep_shift = (0.0, 0.0, -rv ** -12)
    c.vdw_switch = (0.0, 0.0, 0.0)
    c.rvdw_switch = 0.0
    c.beta_lj = c.sh_lj_ewald = 0.0
    c.vdw_type = pkg.VDW_CUT
    if vdw == "pswitch":
        c.vdw_type = pkg.VDW_PSWITCH
        c.rvdw_switch = rvdw_switch
        d = rv - rvdw_switch
        c.vdw_switch = (-10.0 / d ** 3, 15.0 / d ** 4, -6.0 / d ** 5)   # potential_switch_constants, forcerec/interaction_const
        c.disp_shift = (0.0, 0.0, 0.0)
        c.rep_shift = (0.0, 0.0, 0.0)
    elif vdw in ("comb_geom", "comb_lb"):
        c.vdw_type = pkg.VDW_CUT_COMB_GEOM if vdw == "comb_geom" else pkg.VDW_CUT_COMB_LB
    elif vdw in ("ewald_geom", "ewald_lb"):
        c.vdw_type = pkg.VDW_EWALD_GEOM if vdw == "ewald_geom" else pkg.VDW_EWALD_LB
        c.beta_lj = calc_ewaldcoeff_lj(rv, 1e-3)
        crc2 = (c.beta_lj * rv) ** 2
        # forcerec: the grid part's potential shift, sh_lj_ewald = (exp(-b^2 rc^2) (1 + b^2 rc^2 + b^4 rc^4 / 2) - 1) / rc^6
        c.sh_lj_ewald = (math.exp(-crc2) * (1 + crc2 + 0.5 * crc2 * crc2) - 1) / rv ** 6
    elif vdw == "fswitch":
        c.vdw_type = pkg.VDW_FSWITCH
        c.rvdw_switch = rvdw_switch
        c.disp_shift = force_switch_constants(6.0, rvdw_switch, rv)
        c.rep_shift = force_switch_constants(12.0, rvdw_switch, rv)
    c.sc_alpha, c.sc_power, c.sc_sigma, c.sc_coul = sc_alpha, sc_power, sc_sigma, sc_coul
    # softcore "gapsys": (sc-gapsys-scale-linpoint-lj, -q, sc-gapsys-sigma-lj); the alphas are not used then
    c.softcore, c.gapsys = softcore, tuple(gapsys)
    c.lambda_coul, c.lambda_vdw = lambda_coul, lambda_vdw
    c.n_lambda = n_lambda
    c.all_lambda = np.linspace(0.0, 1.0, n_lambda) if n_lambda > 0 else np.zeros(0)
    c.have_soft_core = (sc_alpha != 0) if softcore == "beutler" else (gapsys[0] != 0 or gapsys[1] != 0)
    return c

def add_oxygen_type_variants(sysd, k, seed):
    """Gives the water oxygens k extra atom types with slightly different LJ parameters (geometric mixing),
    to exercise the type-pair table with a realistic number of types (ntype = 3 + k)."""
    rng = np.random.default_rng(seed)
    nt_old, nt = sysd["ntype"], sysd["ntype"] + k
    old = np.asarray(sysd["nbfp"], np.float64).reshape(nt_old, nt_old, 2)
    c6 = np.zeros(nt)
    c12 = np.zeros(nt)
    c6[0], c12[0] = old[0, 0]
    for t in range(nt_old, nt):
        c6[t] = old[0, 0, 0] * (1.0 + 0.03 * (t - nt_old + 1))
        c12[t] = old[0, 0, 1] * (1.0 + 0.05 * (t - nt_old + 1))
    nbfp = np.zeros((nt, nt, 2))
    nbfp[..., 0] = np.sqrt(np.outer(c6, c6))
    nbfp[..., 1] = np.sqrt(np.outer(c12, c12))
    sysd["nbfp"] = nbfp.reshape(-1).astype(np.float32)
    sysd["ntype"] = nt
    ox = np.flatnonzero(sysd["typeA"] == 0)
    newt = rng.integers(0, k + 1, size=len(ox))
    newt = np.where(newt == 0, 0, newt + nt_old - 1).astype(np.int32)
    same = sysd["typeB"][ox] == sysd["typeA"][ox]
    sysd["typeA"][ox] = newt
    sysd["typeB"][ox] = np.where(same, newt, sysd["typeB"][ox])

def lj_grid_table(c):
    """ntype x ntype x 2 table of the grid C6 (the CPU kernel's fr->ljpme_c6grid, nb_free_energy.cpp:560-566; second component unused)"""
    if c.vdw not in ("ewald_geom", "ewald_lb"):
        return None
    t = lj_type_params(c).astype(np.float64)[:c.ntype]
    if c.vdw == "ewald_geom":
        c6 = np.outer(t[:, 0], t[:, 0])
    else:
        c6 = np.outer(t[:, 1], t[:, 1]) * (t[:, 0][:, None] + t[:, 0][None, :]) ** 6
    out = np.zeros((c.ntype, c.ntype, 2))
    out[..., 0] = c6
    return out.reshape(-1)

def lj_type_params(c):
    """The per-TYPE table of the same parameters (numTypes = ntype + 1 rows): NBParamGpu::nbfp_comb of the LJ-PME flavours"""
    return lj_comb_params(c, np.arange(c.ntype + 1))

def lj_comb_params(c, atype):
    """Per-atom combination-rule parameters in the kernels' convention (nbnxm_cuda_kernel.cuh:504-516):
    geometric: (sqrt(6 C6), sqrt(12 C12)); Lorentz-Berthelot: (sigma/2, sqrt(eps)) with
    c6 = eps sigma^6 = 6 C6 and c12 = c6 sigma^6 = 12 C12.  atype: grid-order (masked) types, numTypes = ntype + 1."""
    nt = c.ntype
    tab = np.zeros((nt + 1, 2))
    diag = np.asarray(c.sys["nbfp"], np.float64).reshape(nt, nt, 2)[np.arange(nt), np.arange(nt)]
    tab[:nt] = diag
    out = np.zeros((len(atype), 2), np.float32)
    c6, c12 = tab[atype, 0], tab[atype, 1]
    has = (c6 > 0) & (c12 > 0)
    if c.vdw in ("comb_geom", "ewald_geom"):
        out[:, 0] = np.sqrt(c6)
        out[:, 1] = np.sqrt(c12)
    else:
        sig6 = np.where(has, c12 / np.where(has, c6, 1.0), 0.0)
        out[:, 0] = 0.5 * sig6 ** (1.0 / 6.0)
        out[:, 1] = np.where(has, c6 / np.sqrt(np.where(has, c12, 1.0)), 0.0)
    return out

def force_switch_constants(p, rsw, rc):
    """force_switch_constants() of mdtypes/interaction_const.cpp: (c2, c3, cpot)."""
    c2 = ((p + 1) * rsw - (p + 4) * rc) / (rc ** (p + 2) * (rc - rsw) ** 2)
    c3 = -((p + 1) * rsw - (p + 3) * rc) / (rc ** (p + 2) * (rc - rsw) ** 3)
    cpot = -rc ** -p + p * c2 / 3 * (rc - rsw) ** 3 + p * c3 / 4 * (rc - rsw) ** 4
    return (c2, c3, cpot)

def gpu_interaction_params(c, use_dynamic_pruning=False):
    tab, scale = None, 0.0
    if c.elec == "ewald_tab":
        tab, scale = ewald_force_table(c.beta, c.rc + 0.3)
    return pkg.make_interaction_params(c.elec_type, c.vdw_type, c.epsfac, c.rc, c.rvdw, c.rlist, getattr(c, "rlist_inner", c.rlist),
                                       k_rf=c.k_rf, c_rf=c.c_rf, ewaldcoeff_q=c.beta, sh_ewald=c.sh_ewald,
                                       rvdw_switch=c.rvdw_switch, dispersion_shift=c.disp_shift,
                                       repulsion_shift=c.rep_shift, vdw_switch=c.vdw_switch,
                                       use_dynamic_pruning=use_dynamic_pruning, coulomb_tab=tab,
                                       coulomb_tab_scale=scale, ewaldcoeff_lj=c.beta_lj, sh_lj_ewald=c.sh_lj_ewald)

def ewald_force_table(beta, rmax, scale=2000.0):
    """Ewald correction force table with the meaning the kernels give coulombEwaldTables->tableF
    (nbnxm_cuda_kernel.cuh:624-628: F/r += qq (int_bit/r^2 - T(r)) / r):
    T(r) = -d/dr(erf(beta r)/r) = erf(beta r)/r^2 - 2 beta/sqrt(pi) exp(-beta^2 r^2)/r, at r = i/scale."""
    n = int(rmax * scale) + 2
    r = np.arange(n) / scale
    tab = np.zeros(n)
    r1 = r[1:]
    tab[1:] = (np.vectorize(math.erf)(beta * r1) / r1 ** 2
               - 2 * beta / math.sqrt(math.pi) * np.exp(-(beta * r1) ** 2) / r1)
    return tab.astype(np.float32), scale

def setup_gpu(c, fused=False, use_dynamic_pruning=False, list_override=None, keep_combination_kernels=False):
    """list_override: (sci, cjPacked, excl) to upload instead of the case's own list (domain decomposition)."""
    g = c.grid
    ic = gpu_interaction_params(c, use_dynamic_pruning)
    nb = pkg.NbnxmGpu(ic, g.num_types, g.nbat_nbfp(c.sys["nbfp"]),
                      nbfp_comb=lj_type_params(c) if c.vdw in ("ewald_geom", "ewald_lb") else None, fep=True, n_lambda=c.n_lambda)
    if getattr(c, "keep_kernel_pick", False) or keep_combination_kernels:
        nb.set_kernel_routing(keep_tabulated_kernels=getattr(c, "keep_kernel_pick", False), keep_combination_kernels=keep_combination_kernels)
    sig6 = c.sc_sigma ** 6
    alpha_coul = c.sc_alpha if c.sc_coul else 0.0
    sig6_min = sig6 if c.sc_coul else 0.0
    nb.copy_fepparams(alpha_coul, c.sc_alpha, c.sc_power, sig6, sig6_min, c.lambda_coul, c.lambda_vdw,
                      c.all_lambda, c.all_lambda)
    if c.softcore == "gapsys":
        nb.set_softcore(pkg.SOFTCORE_GAPSYS, *c.gapsys)
    ljc = lj_comb_params(c, g.type) if c.vdw in ("comb_geom", "comb_lb") else None
    nb.init_atomdata(g.num_atoms, g.type, lj_comb=ljc, qA=g.qA, qB=g.qB, typeA=g.typeA, typeB=g.typeB)
    pl = c.plist_fused if fused else c.plist
    if list_override is not None:
        nb.init_pairlist(*list_override)
    else:
        nb.init_pairlist(pl.sci, pl.cjPacked, pl.excl)
    if fused:
        # no atom-pair list at all: nbnxmFepClusterKernel covers forces, energies and foreign lambdas of the perturbed cluster pairs
        nb.init_fep_cluster_bits(g.fepBits)
        nb.set_fep_mode(True)
    else:
        nb.init_feppairlist(c.plist.fep, g.atomIndices)
    nb.upload_shiftvec(g.shift_vec)
    nb.copy_xq_to_gpu(g.xq)
    return nb

def run_gpu(c, energy=True, fused=False, dhdl=False, nb=None, prune=False, launch=None):
    """One step through the C ABI: clear, launch, copy back, finish.  launch(nb, sw) replaces the plain nbnxm_gpu_launch_kernel call
    (e.g. the two calls of nbnxm_gpu_launch_kernel_part)."""
    own = nb is None
    if own:
        nb = setup_gpu(c, fused=fused, use_dynamic_pruning=prune)
    sw = pkg.step_workload(energy=energy, virial=True, dhdl=dhdl)
    nb.clear_outputs(True)
    if launch is None:
        nb.launch_kernel(sw)
    else:
        launch(nb, sw)
    f = np.zeros((c.grid.num_atoms, 3), np.float32)
    nb.launch_cpyback(f, sw)
    res = nb.wait_finish_task(sw, c.have_soft_core)
    dv = res["dvdl_nonlin"] if c.have_soft_core else res["dvdl_lin"]
    out = dict(f=f.astype(np.float64), fshift=res["fshift"].astype(np.float64), e_lj=res["e_lj"], e_el=res["e_el"],
               dvdl_coul=dv[0], dvdl_vdw=dv[1], raw=res)
    if dhdl:
        out["foreign"] = dict(energies=res["foreign_energies"], dvdlCoul=res["foreign_dhdl_coul"],
                              dvdlVdw=res["foreign_dhdl_vdw"])
    if own:
        nb.free()
    return out
