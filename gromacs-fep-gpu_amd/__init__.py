"""Python face of the MI355X non-bonded FEP path.

The product is two in-tree shared libraries with a C ABI:
  lib/libnbnxm_hip.so   (include/nbnxm_hip.h)   HIP kernels + the Nbnxm GPU API drop-in
  lib/libnbnxm_host.so  (include/nbnxm_host.h)  synthetic box, cluster grid, pair-list builders
This module only binds them with ctypes for the tests and bench.py: names and argument
meaning mirror the reference's Nbnxm functions (nbnxm/nbnxm_gpu.h, nbnxm/gpu_data_mgmt.h).
There is NO CPU fallback: if libnbnxm_hip.so is missing, or there is no GPU when a GPU entry
point is called, this fails loudly.  (The directory name has a hyphen, so load it with
__graft_entry__.load_package(), which registers it as `gromacs_fep_gpu_amd`.)
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(PKG_DIR, "lib")
HIP_LIB_PATH = os.path.join(LIB_DIR, "libnbnxm_hip.so")
HOST_LIB_PATH = os.path.join(LIB_DIR, "libnbnxm_host.so")

# ---- ABI constants (include/nbnxm_hip.h) ----------------------------------------------------------
CLUSTER_SIZE = 8
NUM_CLUSTER_PER_SUPERCLUSTER = 8
JGROUP_SIZE = 4
NUM_SHIFT_VECTORS = 45
CENTRAL_SHIFT_INDEX = 22
ELEC_CUT, ELEC_RF, ELEC_EWALD_TAB, ELEC_EWALD_TAB_TWIN, ELEC_EWALD_ANA, ELEC_EWALD_ANA_TWIN = range(6)
SOFTCORE_BEUTLER, SOFTCORE_GAPSYS = 0, 1
VDW_CUT, VDW_CUT_COMB_GEOM, VDW_CUT_COMB_LB, VDW_FSWITCH, VDW_PSWITCH, VDW_EWALD_GEOM, VDW_EWALD_LB = range(7)
LOCAL, NONLOCAL = 0, 1

SCI_DTYPE = np.dtype([("sci", "<i4"), ("shift", "<i4"), ("cjPackedBegin", "<i4"), ("cjPackedEnd", "<i4")])
IMEI_DTYPE = np.dtype([("imask", "<u4"), ("excl_ind", "<i4")])
CJ_PACKED_DTYPE = np.dtype([("cj", "<i4", (4,)), ("imei", IMEI_DTYPE, (2,))])
EXCL_DTYPE = np.dtype([("pair", "<u4", (32,))])
assert SCI_DTYPE.itemsize == 16 and CJ_PACKED_DTYPE.itemsize == 32 and EXCL_DTYPE.itemsize == 128


class ShiftConsts(C.Structure):
    _fields_ = [("c2", C.c_float), ("c3", C.c_float), ("cpot", C.c_float)]


class SwitchConsts(C.Structure):
    _fields_ = [("c3", C.c_float), ("c4", C.c_float), ("c5", C.c_float)]


class InteractionParams(C.Structure):
    """nbnxm_interaction_params_t"""
    _fields_ = [
        ("elecType", C.c_int), ("vdwType", C.c_int),
        ("epsfac", C.c_float), ("c_rf", C.c_float), ("k_rf", C.c_float),
        ("ewaldcoeff_q", C.c_float), ("sh_ewald", C.c_float), ("sh_lj_ewald", C.c_float),
        ("ewaldcoeff_lj", C.c_float),
        ("rcoulomb", C.c_float), ("rvdw", C.c_float), ("rvdw_switch", C.c_float),
        ("rlistOuter", C.c_float), ("rlistInner", C.c_float),
        ("useDynamicPruning", C.c_int),
        ("dispersion_shift", ShiftConsts), ("repulsion_shift", ShiftConsts),
        ("vdw_switch", SwitchConsts),
        ("coulomb_tab_scale", C.c_float), ("coulomb_tab_size", C.c_int),
        ("coulomb_tab", C.POINTER(C.c_float)),
    ]


class StepWorkload(C.Structure):
    """nbnxm_step_workload_t"""
    _fields_ = [("computeForces", C.c_int), ("computeEnergy", C.c_int), ("computeVirial", C.c_int),
                ("computeDhdl", C.c_int), ("useGpuFBufferOps", C.c_int)]


class EnerData(C.Structure):
    """nbnxm_enerdata_t"""
    _fields_ = [("e_lj", C.c_double), ("e_el", C.c_double),
                ("dvdl_lin", C.c_double * 2), ("dvdl_nonlin", C.c_double * 2),
                ("n_lambda", C.c_int),
                ("foreign_energies", C.POINTER(C.c_double)),
                ("foreign_dhdl_coul", C.POINTER(C.c_double)),
                ("foreign_dhdl_vdw", C.POINTER(C.c_double))]


class GpuTimings(C.Structure):
    _fields_ = [("nb_k_ms", C.c_double), ("nb_k_count", C.c_int),
                ("fep_k_ms", C.c_double), ("fep_k_count", C.c_int),
                ("prune_k_ms", C.c_double), ("prune_k_count", C.c_int)]


# Every symbol include/nbnxm_hip.h declares (checked by tests/test_abi_symbols.py).
HIP_SYMBOLS = [
    "nbnxm_gpu_init", "nbnxm_gpu_free", "nbnxm_gpu_copy_fepparams", "nbnxm_gpu_set_softcore", "nbnxm_gpu_pme_loadbal_update_param",
    "nbnxm_gpu_set_kernel_routing",
    "nbnxm_gpu_init_atomdata", "nbnxm_gpu_init_pairlist", "nbnxm_gpu_init_feppairlist",
    "nbnxm_gpu_init_fep_cluster_bits", "nbnxm_gpu_upload_shiftvec", "nbnxm_gpu_copy_xq_to_gpu",
    "nbnxm_gpu_launch_kernel", "nbnxm_gpu_launch_kernel_pruneonly", "nbnxm_gpu_launch_cpyback",
    "nbnxm_gpu_try_finish_task", "nbnxm_gpu_wait_finish_task", "nbnxm_gpu_clear_outputs",
    "nbnxm_gpu_get_timings", "nbnxm_gpu_reset_timings", "nbnxm_gpu_set_timing",
    "nbnxm_gpu_min_ci_balanced", "nbnxm_gpu_is_kernel_ewald_analytical", "nbnxm_gpu_get_xq",
    "nbnxm_gpu_get_f", "nbnxm_gpu_get_fshift", "nbnxm_gpu_get_q4", "nbnxm_gpu_get_stream", "nbnxm_gpu_set_window_lambdas",
    "nbnxm_gpu_get_window_energies",
    "nbnxm_gpu_have_short_range_work", "nbnxm_gpu_set_fep_mode", "nbnxm_hip_abi_version",
    "nbnxm_hip_last_error", "nbnxm_gpu_debug_get_cjpacked", "nbnxm_gpu_debug_download", "nbnxm_gpu_debug_get_work_ranges",
    "nbnxm_gpu_debug_graph_steps", "nbnxm_gpu_debug_set_work_shares",
    "nbnxm_gpu_init_x_to_nbat_x", "nbnxm_gpu_x_to_nbat_x", "nbnxm_gpu_insert_nonlocal_dependency",
    "nbnxm_gpu_setup_short_range_work", "nbnxm_gpu_force_reduction_reinit", "nbnxm_gpu_force_reduction_execute",
    "nbnxm_gpu_halo_pack_x", "nbnxm_gpu_halo_unpack_f", "nbnxm_gpu_force_reduction_execute_range", "nbnxm_hip_query_launch_shape",
    "nbnxm_gpu_set_local_launch_parts", "nbnxm_gpu_launch_kernel_part", "nbnxm_hip_query_launch_plan",
    "nbnxm_gpu_set_merged_localities", "nbnxm_gpu_get_merged_localities",
]
HALO_SYMBOLS = [
    "halo_gpu_get_unique_id", "halo_gpu_get_unique_id_ex", "halo_gpu_create", "halo_gpu_free", "halo_gpu_last_error", "halo_gpu_reinit",
    "halo_gpu_communicate_coordinates", "halo_gpu_communicate_forces", "halo_gpu_coordinates_ready_event", "halo_gpu_forces_ready_event",
    "halo_gpu_bytes_per_step", "halo_gpu_pack_shifted", "halo_gpu_domain_force_step",
    "halo_gpu_push_export_bytes", "halo_gpu_push_export", "halo_gpu_push_import", "halo_gpu_push_status",
]
UPDATE_SYMBOLS = [
    "langevin_gpu_create", "langevin_gpu_free", "langevin_gpu_set", "langevin_gpu_integrate",
    "leapfrog_gpu_create", "leapfrog_gpu_free", "leapfrog_gpu_set", "leapfrog_gpu_integrate",
    "settle_gpu_create", "settle_gpu_free", "settle_gpu_set", "settle_gpu_apply",
    "lincs_gpu_create", "lincs_gpu_free", "lincs_gpu_set", "lincs_gpu_apply",
    "update_constrain_gpu_create", "update_constrain_gpu_free", "update_constrain_gpu_set", "update_constrain_gpu_set_pbc",
    "update_constrain_gpu_integrate", "update_constrain_gpu_scale_coordinates", "update_constrain_gpu_scale_velocities",
    "update_constrain_gpu_x_updated_event", "update_constrain_gpu_set_nbat_coupling", "update_constrain_gpu_can_fuse",
    "update_constrain_gpu_integrate_fused",
]
LISTED_SYMBOLS = [
    "listed_gpu_create", "listed_gpu_free", "listed_gpu_set_force_params", "listed_gpu_update_interaction_list",
    "listed_gpu_have_interactions", "listed_gpu_launch_kernel", "listed_gpu_launch_energy_transfer",
    "listed_gpu_wait_accumulate_energy_terms", "listed_gpu_clear_energies",
]
HOST_SYMBOLS = [
    "nbnxm_host_make_water_box", "nbnxm_host_grid_create", "nbnxm_host_grid_free",
    "nbnxm_host_grid_num_atoms", "nbnxm_host_grid_num_clusters", "nbnxm_host_grid_get",
    "nbnxm_host_grid_update_xq", "nbnxm_host_shift_vectors", "nbnxm_host_pairlist_build",
    "nbnxm_host_pairlist_free", "nbnxm_host_pairlist_sizes", "nbnxm_host_pairlist_get",
    "nbnxm_host_pairlist_get_fep", "nbnxm_host_abi_version", "nbnxm_host_count_pairs_within",
    "nbnxm_host_calc_ewaldcoeff_q", "nbnxm_host_calc_ewaldcoeff_lj", "nbnxm_host_grid_create_dd", "nbnxm_host_grid_num_atoms_home",
    "nbnxm_host_pairlist_build_dd",
]

_hip = None
_host = None


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _a(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def hip_lib():
    """Loads lib/libnbnxm_hip.so.  Raises if it has not been built — there is no fallback."""
    global _hip
    if _hip is None:
        path = os.environ.get("NBNXM_HIP_LIB", HIP_LIB_PATH)     # (kernel experiments: another build of the same library)
        if not os.path.exists(path):
            raise RuntimeError("HIP extension missing: %s (run __graft_entry__.build() / make -C %s)"
                               % (path, PKG_DIR))
        lib = C.CDLL(path)
        lib.nbnxm_gpu_init.restype = C.c_void_p
        lib.nbnxm_gpu_get_xq.restype = C.c_void_p
        lib.nbnxm_gpu_get_f.restype = C.c_void_p
        lib.nbnxm_gpu_get_fshift.restype = C.c_void_p
        lib.nbnxm_gpu_get_q4.restype = C.c_void_p
        lib.nbnxm_gpu_get_stream.restype = C.c_void_p
        lib.nbnxm_hip_last_error.restype = C.c_char_p
        _hip = lib
    return _hip


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError("host library missing: %s (run __graft_entry__.build())" % HOST_LIB_PATH)
        lib = C.CDLL(HOST_LIB_PATH)
        lib.nbnxm_host_grid_create.restype = C.c_void_p
        lib.nbnxm_host_pairlist_build.restype = C.c_void_p
        _host = lib
    return _host


# ---- host side: synthetic system, grid, lists -------------------------------------------------------

def make_water_box(nmx, nmy, nmz, spacing=0.310736, jitter=0.03, seed=2026, num_perturbed_molecules=0):
    """SPC/E-like water box (nbnxm/benchmark/bench_system.cpp recipe, 100 atoms/nm^3 at the default
    spacing: 3000 atoms in 3.10736 nm = 10 molecules per edge).  Returns a dict of topology-order arrays."""
    n = 3 * nmx * nmy * nmz
    out = dict(x=np.zeros((n, 3), np.float32), qA=np.zeros(n, np.float32), qB=np.zeros(n, np.float32),
               typeA=np.zeros(n, np.int32), typeB=np.zeros(n, np.int32), molId=np.zeros(n, np.int32),
               box=np.zeros(3, np.float32))
    host_lib().nbnxm_host_make_water_box(C.c_int(nmx), C.c_int(nmy), C.c_int(nmz), C.c_double(spacing),
                                         C.c_double(jitter), C.c_uint(seed), C.c_int(num_perturbed_molecules),
                                         _p(out["x"]), _p(out["qA"]), _p(out["qB"]), _p(out["typeA"]),
                                         _p(out["typeB"]), _p(out["molId"]), _p(out["box"]))
    out["ntype"] = 3
    # 6*C6, 12*C12 (bench_system.cpp:76-84: OW-OW only), ntype x ntype
    nbfp = np.zeros((3, 3, 2), np.float64)
    nbfp[0, 0] = (6.0 * 0.0026173456, 12.0 * 2.634129e-6)
    out["nbfp"] = nbfp.reshape(-1).astype(np.float32)  # the table every consumer sees (float, like nbat)
    return out


def exclusions_from_groups(group_id):
    """CSR exclusion lists: atoms with the same group id exclude each other (rigid waters)."""
    group_id = np.asarray(group_id)
    order = np.argsort(group_id, kind="stable")
    sorted_ids = group_id[order]
    starts = np.flatnonzero(np.r_[True, sorted_ids[1:] != sorted_ids[:-1]])
    ends = np.r_[starts[1:], len(order)]
    n = len(group_id)
    counts = np.zeros(n, np.int64)
    for s, e in zip(starts, ends):
        counts[order[s:e]] = e - s
    index = np.zeros(n + 1, np.int32)
    index[1:] = np.cumsum(counts)
    atoms = np.zeros(index[-1], np.int32)
    for s, e in zip(starts, ends):
        members = order[s:e]
        for a in members:
            atoms[index[a]:index[a + 1]] = members
    return index, atoms


class Grid:
    """Cluster grid of one system (nbnxm/grid.cpp role)."""

    def __init__(self, x, box, qA, qB, typeA, typeB, ntype, perturbed=None, num_home=None, periodic=(True, True, True)):
        """num_home: the grid of ONE DOMAIN of a decomposed run — the first num_home atoms are home atoms, the rest halo atoms
        (two zones); periodic[d] False for a decomposed dimension (nbnxm_host_grid_create_dd)"""
        self.natoms = len(qA)
        self.ntype = int(ntype)
        self.box = _a(box, np.float32)
        x = _a(x, np.float32)
        pert = _a(perturbed, np.uint8)
        lib = host_lib()
        if num_home is None:
            self._h = lib.nbnxm_host_grid_create(C.c_int(self.natoms), _p(x), _p(self.box),
                                                 _p(_a(qA, np.float32)), _p(_a(qB, np.float32)),
                                                 _p(_a(typeA, np.int32)), _p(_a(typeB, np.int32)),
                                                 C.c_int(self.ntype), _p(pert))
        else:
            lib.nbnxm_host_grid_create_dd.restype = C.c_void_p
            per = _a([1 if p else 0 for p in periodic], np.int32)
            self._h = lib.nbnxm_host_grid_create_dd(C.c_int(int(num_home)), C.c_int(self.natoms - int(num_home)), _p(x), _p(self.box), _p(per),
                                                    _p(_a(qA, np.float32)), _p(_a(qB, np.float32)),
                                                    _p(_a(typeA, np.int32)), _p(_a(typeB, np.int32)),
                                                    C.c_int(self.ntype), _p(pert))
        h = C.c_void_p(self._h)
        self.num_atoms_home = lib.nbnxm_host_grid_num_atoms_home(h)      # padded slots of the home zone
        self.num_atoms = host_lib().nbnxm_host_grid_num_atoms(h)
        self.num_clusters = host_lib().nbnxm_host_grid_num_clusters(h)
        np_ = self.num_atoms
        self.xq = np.zeros((np_, 4), np.float32)
        self.type = np.zeros(np_, np.int32)
        self.qA = np.zeros(np_, np.float32)
        self.qB = np.zeros(np_, np.float32)
        self.typeA = np.zeros(np_, np.int32)
        self.typeB = np.zeros(np_, np.int32)
        self.atomIndices = np.zeros(np_, np.int32)
        self.fepBits = np.zeros(self.num_clusters, np.uint8)
        self.x_wrapped = np.zeros((self.natoms, 3), np.float32)
        host_lib().nbnxm_host_grid_get(h, _p(self.xq), _p(self.type), _p(self.qA), _p(self.qB), _p(self.typeA),
                                       _p(self.typeB), _p(self.atomIndices), _p(self.fepBits), _p(self.x_wrapped))
        self.shift_vec = np.zeros((NUM_SHIFT_VECTORS, 3), np.float32)
        host_lib().nbnxm_host_shift_vectors(_p(self.box), _p(self.shift_vec))

    @property
    def num_types(self):
        """numTypes of the grid-order arrays: topology types + the zero filler type."""
        return self.ntype + 1

    def nbat_nbfp(self, nbfp):
        """(ntype+1)^2 x 2 table with a zero last row/column (nbnxn_atomdata_t params().nbfp)."""
        t = self.ntype
        out = np.zeros((t + 1, t + 1, 2), np.float32)
        out[:t, :t] = np.asarray(nbfp, np.float32).reshape(t, t, 2)
        return out.reshape(-1)

    def update_xq(self, x):
        xq = self.xq.copy()
        host_lib().nbnxm_host_grid_update_xq(C.c_void_p(self._h), _p(_a(x, np.float32)), _p(xq))
        return xq

    def build_pairlist(self, excl_index, excl_atoms, rlist, max_cjpacked_per_sci=0, carve_fep=True,
                       rlist_fep=None):
        return Pairlist(self, excl_index, excl_atoms, rlist, max_cjpacked_per_sci, carve_fep,
                        rlist if rlist_fep is None else rlist_fep)

    def build_pairlist_dd(self, non_local, excl_index, excl_atoms, rlist, max_cjpacked_per_sci=0):
        """one of the two lists of a domain: home x home (non_local False) or home x halo (True); perturbed pairs stay in the list"""
        return Pairlist(self, excl_index, excl_atoms, rlist, max_cjpacked_per_sci, False, rlist, dd_non_local=bool(non_local))

    def __del__(self):
        # at interpreter shutdown module globals may already be gone: the process is ending, nothing to free
        try:
            if getattr(self, "_h", None):
                host_lib().nbnxm_host_grid_free(C.c_void_p(self._h))
                self._h = None
        except Exception:
            pass


class Pairlist:
    """GPU-layout cluster pair list + atom-pair FEP list (NbnxnPairlistGpu + t_nblist)."""

    def __init__(self, grid, excl_index, excl_atoms, rlist, max_cjpacked_per_sci, carve_fep, rlist_fep, dd_non_local=None):
        ei = _a(excl_index, np.int32)
        ea = _a(excl_atoms, np.int32)
        if dd_non_local is None:
            h = host_lib().nbnxm_host_pairlist_build(C.c_void_p(grid._h), _p(ei), _p(ea), C.c_float(rlist),
                                                     C.c_int(max_cjpacked_per_sci), C.c_int(1 if carve_fep else 0),
                                                     C.c_float(rlist_fep))
        else:
            host_lib().nbnxm_host_pairlist_build_dd.restype = C.c_void_p
            h = host_lib().nbnxm_host_pairlist_build_dd(C.c_void_p(grid._h), C.c_int(1 if dd_non_local else 0), _p(ei), _p(ea),
                                                        C.c_float(rlist), C.c_int(max_cjpacked_per_sci))
        h = C.c_void_p(h)
        sizes = np.zeros(6, np.int64)
        host_lib().nbnxm_host_pairlist_sizes(h, _p(sizes))
        nsci, ncj, nexcl, nri, nrj, self.num_cluster_pairs = [int(v) for v in sizes]
        self.rlist = float(rlist)
        self.sci = np.zeros(nsci, SCI_DTYPE)
        self.cjPacked = np.zeros(ncj, CJ_PACKED_DTYPE)
        self.excl = np.zeros(nexcl, EXCL_DTYPE)
        host_lib().nbnxm_host_pairlist_get(h, _p(self.sci), _p(self.cjPacked), _p(self.excl))
        self.fep = dict(iinr=np.zeros(nri, np.int32), shift=np.zeros(nri, np.int32),
                        jindex=np.zeros(nri + 1, np.int32), jjnr=np.zeros(nrj, np.int32),
                        excl_fep=np.zeros(nrj, np.int32))
        host_lib().nbnxm_host_pairlist_get_fep(h, _p(self.fep["iinr"]), _p(self.fep["shift"]),
                                               _p(self.fep["jindex"]), _p(self.fep["jjnr"]),
                                               _p(self.fep["excl_fep"]))
        host_lib().nbnxm_host_pairlist_free(h)


# ---- GPU side: the Nbnxm GPU API ----------------------------------------------------------------------

def make_interaction_params(elec_type, vdw_type, epsfac, rcoulomb, rvdw, rlist_outer, rlist_inner=None,
                            k_rf=0.0, c_rf=0.0, ewaldcoeff_q=0.0, sh_ewald=0.0, ewaldcoeff_lj=0.0,
                            sh_lj_ewald=0.0, rvdw_switch=0.0, dispersion_shift=(0, 0, 0),
                            repulsion_shift=(0, 0, 0), vdw_switch=(0, 0, 0), use_dynamic_pruning=False,
                            coulomb_tab=None, coulomb_tab_scale=0.0):
    ic = InteractionParams()
    ic.elecType, ic.vdwType = int(elec_type), int(vdw_type)
    ic.epsfac, ic.c_rf, ic.k_rf = epsfac, c_rf, k_rf
    ic.ewaldcoeff_q, ic.sh_ewald, ic.sh_lj_ewald, ic.ewaldcoeff_lj = ewaldcoeff_q, sh_ewald, sh_lj_ewald, ewaldcoeff_lj
    ic.rcoulomb, ic.rvdw, ic.rvdw_switch = rcoulomb, rvdw, rvdw_switch
    ic.rlistOuter = rlist_outer
    ic.rlistInner = rlist_outer if rlist_inner is None else rlist_inner
    ic.useDynamicPruning = 1 if use_dynamic_pruning else 0
    ic.dispersion_shift = ShiftConsts(*dispersion_shift)
    ic.repulsion_shift = ShiftConsts(*repulsion_shift)
    ic.vdw_switch = SwitchConsts(*vdw_switch)
    if coulomb_tab is not None:
        tab = _a(coulomb_tab, np.float32)
        ic._tab_keepalive = tab
        ic.coulomb_tab = tab.ctypes.data_as(C.POINTER(C.c_float))
        ic.coulomb_tab_size = len(tab)
        ic.coulomb_tab_scale = coulomb_tab_scale
    return ic


def step_workload(energy=False, virial=False, dhdl=False):
    return StepWorkload(1, 1 if energy else 0, 1 if virial else 0, 1 if dhdl else 0, 0)


class NbnxmGpu:
    """Owner of one NbnxmGpu object behind the C ABI; method names follow Nbnxm::gpu_* ."""

    def __init__(self, ic, num_types, nbfp, nbfp_comb=None, local_and_nonlocal=False, fep=False, n_lambda=0):
        lib = hip_lib()
        self._lib = lib
        self.n_lambda = int(n_lambda)
        self._ic = ic
        nbfp = _a(nbfp, np.float32)
        assert nbfp.size == 2 * num_types * num_types, "nbfp must hold 2*numTypes^2 floats"
        comb = _a(nbfp_comb, np.float32)
        self._h = lib.nbnxm_gpu_init(C.byref(ic), C.c_int(num_types), _p(nbfp), _p(comb),
                                     C.c_int(1 if local_and_nonlocal else 0), C.c_int(1 if fep else 0),
                                     C.c_int(n_lambda), None, None)
        if not self._h:
            raise RuntimeError("nbnxm_gpu_init failed: %s" % lib.nbnxm_hip_last_error().decode())
        self.num_atoms = 0

    @property
    def h(self):
        return C.c_void_p(self._h)

    def pme_loadbal_update_param(self, ic):
        """Nbnxm::gpu_pme_loadbal_update_param: new Coulomb cut-off / Ewald coefficient (and table) from PME load balancing"""
        self._ic = ic
        self._lib.nbnxm_gpu_pme_loadbal_update_param(self.h, C.byref(ic))

    def copy_fepparams(self, alpha_coul, alpha_vdw, lam_power, sc_sigma6_def, sc_sigma6_min, lambda_q,
                       lambda_v, all_lambda_coul=(), all_lambda_vdw=()):
        alc = _a(all_lambda_coul, np.float64)
        alv = _a(all_lambda_vdw, np.float64)
        assert len(alc) == len(alv) == self.n_lambda
        self._lib.nbnxm_gpu_copy_fepparams(self.h, C.c_int(1), C.c_float(alpha_coul), C.c_float(alpha_vdw),
                                           C.c_int(lam_power), C.c_float(sc_sigma6_def), C.c_float(sc_sigma6_min),
                                           C.c_float(lambda_q), C.c_float(lambda_v), C.c_int(self.n_lambda),
                                           _p(alc), _p(alv))

    def set_kernel_routing(self, keep_tabulated_kernels=False, keep_combination_kernels=False):
        """run exactly the tabulated-Ewald / combination-rule kernels the interaction parameters pick instead of their faster equivalents"""
        self._lib.nbnxm_gpu_set_kernel_routing(self.h, C.c_int(1 if keep_tabulated_kernels else 0), C.c_int(1 if keep_combination_kernels else 0))

    def set_softcore(self, softcore_type, gapsys_scale_linpoint_vdw=0.85, gapsys_scale_linpoint_coul=0.3, gapsys_sigma_vdw=0.3):
        """SOFTCORE_BEUTLER (default after init) | SOFTCORE_GAPSYS; the Gapsys parameters are mdp's sc-gapsys-scale-linpoint-lj / -q
        and sc-gapsys-sigma-lj (the C ABI takes sigma^6, as interaction_const_t does)"""
        self._lib.nbnxm_gpu_set_softcore(self.h, C.c_int(softcore_type), C.c_float(gapsys_scale_linpoint_vdw),
                                         C.c_float(gapsys_scale_linpoint_coul), C.c_float(gapsys_sigma_vdw ** 6))

    def init_atomdata(self, num_atoms, atype, lj_comb=None, qA=None, qB=None, typeA=None, typeB=None,
                      lj_combA=None, lj_combB=None, num_atoms_local=None):
        self.num_atoms = int(num_atoms)
        nl = self.num_atoms if num_atoms_local is None else int(num_atoms_local)
        self._lib.nbnxm_gpu_init_atomdata(self.h, C.c_int(self.num_atoms), C.c_int(nl), _p(_a(atype, np.int32)),
                                          _p(_a(lj_comb, np.float32)), _p(_a(qA, np.float32)),
                                          _p(_a(qB, np.float32)), _p(_a(typeA, np.int32)),
                                          _p(_a(typeB, np.int32)), _p(_a(lj_combA, np.float32)),
                                          _p(_a(lj_combB, np.float32)))

    def init_pairlist(self, sci, cjPacked, excl, iloc=LOCAL):
        sci = np.ascontiguousarray(sci)
        cjPacked = np.ascontiguousarray(cjPacked)
        excl = np.ascontiguousarray(excl)
        assert sci.dtype == SCI_DTYPE and cjPacked.dtype == CJ_PACKED_DTYPE and excl.dtype == EXCL_DTYPE
        self._lib.nbnxm_gpu_init_pairlist(self.h, C.c_int(iloc), C.c_int(CLUSTER_SIZE), C.c_int(len(sci)), _p(sci),
                                          C.c_int(len(cjPacked)), _p(cjPacked), C.c_int(len(excl)), _p(excl))

    def init_feppairlist(self, fep, atom_indices, iloc=LOCAL):
        iinr, shift, jindex = (_a(fep[k], np.int32) for k in ("iinr", "shift", "jindex"))
        jjnr = _a(fep["jjnr"], np.int32)
        excl = _a(fep.get("excl_fep"), np.int32)
        ai = _a(atom_indices, np.int32)
        self._lib.nbnxm_gpu_init_feppairlist(self.h, C.c_int(iloc), C.c_int(len(iinr)), _p(iinr), _p(shift),
                                             _p(jindex), C.c_int(len(jjnr)), _p(jjnr), _p(excl),
                                             C.c_int(0 if ai is None else len(ai)), _p(ai))

    def init_fep_cluster_bits(self, fep_bits):
        fb = _a(fep_bits, np.uint8)
        self._lib.nbnxm_gpu_init_fep_cluster_bits(self.h, C.c_int(len(fb)), _p(fb))

    def set_window_lambdas(self, clusters_per_window, lambda_q, lambda_v):
        """several lambda windows batched into this object (see include/nbnxm_hip.h); empty arrays switch it off"""
        lq, lv = _a(lambda_q, np.float32), _a(lambda_v, np.float32)
        assert lq.size == lv.size
        self._lib.nbnxm_gpu_set_window_lambdas(self.h, C.c_int(lq.size), C.c_int(clusters_per_window), _p(lq), _p(lv))

    def get_window_energies(self, window, have_soft_core):
        """one window's share of the last finished energy / dH/dlambda step of an object with batched windows"""
        nl = self.n_lambda
        fe, fc, fv = np.zeros(nl + 1), np.zeros(nl + 1), np.zeros(nl + 1)
        ed = EnerData()
        ed.n_lambda = nl
        ed.foreign_energies = fe.ctypes.data_as(C.POINTER(C.c_double))
        ed.foreign_dhdl_coul = fc.ctypes.data_as(C.POINTER(C.c_double))
        ed.foreign_dhdl_vdw = fv.ctypes.data_as(C.POINTER(C.c_double))
        rc = self._lib.nbnxm_gpu_get_window_energies(self.h, C.c_int(window), C.byref(ed), C.c_int(1 if have_soft_core else 0))
        if rc != 0:
            raise IndexError("no such window")
        return dict(e_lj=ed.e_lj, e_el=ed.e_el, dvdl_lin=list(ed.dvdl_lin), dvdl_nonlin=list(ed.dvdl_nonlin), foreign_energies=fe,
                    foreign_dhdl_coul=fc, foreign_dhdl_vdw=fv)

    def set_fep_mode(self, fused):
        self._lib.nbnxm_gpu_set_fep_mode(self.h, C.c_int(1 if fused else 0))

    def upload_shiftvec(self, shift_vec):
        sv = _a(shift_vec, np.float32)
        assert sv.size == 3 * NUM_SHIFT_VECTORS
        self._lib.nbnxm_gpu_upload_shiftvec(self.h, _p(sv))

    def copy_xq_to_gpu(self, xq, aloc=LOCAL):
        xq = _a(xq, np.float32)
        assert xq.size == 4 * self.num_atoms, "xq must hold 4 floats for each of the %d atoms" % self.num_atoms
        self._lib.nbnxm_gpu_copy_xq_to_gpu(self.h, _p(xq), C.c_int(aloc))

    def xq_device_pointer(self):
        """gpuGetNBAtomData()->xq: float4 per grid slot"""
        return self._lib.nbnxm_gpu_get_xq(self.h)

    def f_device_pointer(self):
        """gpu_get_f: float3 per grid slot"""
        return self._lib.nbnxm_gpu_get_f(self.h)

    def fshift_device_pointer(self):
        return self._lib.nbnxm_gpu_get_fshift(self.h)

    def q4_device_pointer(self):
        """gpuGetNBAtomData()->q4: (qA, qB, -, -) per grid slot"""
        return self._lib.nbnxm_gpu_get_q4(self.h)

    def stream(self, iloc=LOCAL):
        """hipStream_t of a locality as an integer (e.g. for torch.cuda.ExternalStream)."""
        return int(self._lib.nbnxm_gpu_get_stream(self.h, C.c_int(iloc)) or 0)

    # ---- coordinate / force buffer operations (device pointers: ints, e.g. torch.Tensor.data_ptr()) ----
    def init_x_to_nbat_x(self, atom_indices):
        ai = _a(atom_indices, np.int32)
        self._lib.nbnxm_gpu_init_x_to_nbat_x(self.h, C.c_int(ai.size), _p(ai))

    def x_to_nbat_x(self, d_x, slot_begin, slot_end, aloc=LOCAL, x_ready_event=None, insert_nonlocal_dependency=False):
        self._lib.nbnxm_gpu_x_to_nbat_x(self.h, C.c_void_p(d_x), C.c_void_p(x_ready_event), C.c_int(aloc),
                                        C.c_int(slot_begin), C.c_int(slot_end), C.c_int(1 if insert_nonlocal_dependency else 0))

    def insert_nonlocal_dependency(self, iloc=LOCAL):
        self._lib.nbnxm_gpu_insert_nonlocal_dependency(self.h, C.c_int(iloc))

    def setup_short_range_work(self, have_listed_forces=False, iloc=LOCAL):
        self._lib.nbnxm_gpu_setup_short_range_work(self.h, C.c_int(1 if have_listed_forces else 0), C.c_int(iloc))

    def force_reduction_reinit(self, cell, atom_start=0, accumulate=False):
        cell = _a(cell, np.int32)
        self._lib.nbnxm_gpu_force_reduction_reinit(self.h, C.c_int(cell.size), _p(cell), C.c_int(atom_start),
                                                   C.c_int(1 if accumulate else 0))

    def force_reduction_execute(self, d_base_force, d_rvec_force=None, stream=None):
        self._lib.nbnxm_gpu_force_reduction_execute(self.h, C.c_void_p(d_base_force), C.c_void_p(d_rvec_force), C.c_void_p(stream))

    def force_reduction_execute_range(self, d_base_force, atom_begin, atom_end, accumulate=False, stream=None):
        self._lib.nbnxm_gpu_force_reduction_execute_range(self.h, C.c_void_p(d_base_force), C.c_int(atom_begin), C.c_int(atom_end),
                                                          C.c_int(1 if accumulate else 0), C.c_void_p(stream))

    def clear_outputs(self, compute_virial=True):
        self._lib.nbnxm_gpu_clear_outputs(self.h, C.c_int(1 if compute_virial else 0))

    def launch_kernel(self, step_work, iloc=LOCAL):
        self._lib.nbnxm_gpu_launch_kernel(self.h, C.byref(step_work), C.c_int(iloc))

    def set_local_launch_parts(self, num_parts, first_part_fraction=0.65):
        """nbnxm_gpu_set_local_launch_parts: the local list partitioned for a launch in two parts (domain decomposition)"""
        self._lib.nbnxm_gpu_set_local_launch_parts(self.h, C.c_int(num_parts), C.c_float(first_part_fraction))

    def set_merged_localities(self, merged=True):
        """nbnxm_gpu_set_merged_localities: before the lists are uploaded"""
        self._lib.nbnxm_gpu_set_merged_localities(self.h, C.c_int(1 if merged else 0))

    def launch_kernel_part(self, step_work, part, iloc=LOCAL):
        self._lib.nbnxm_gpu_launch_kernel_part(self.h, C.byref(step_work), C.c_int(iloc), C.c_int(part))

    def launch_kernel_pruneonly(self, iloc=LOCAL, num_parts=1):
        self._lib.nbnxm_gpu_launch_kernel_pruneonly(self.h, C.c_int(iloc), C.c_int(num_parts))

    def launch_cpyback(self, f_out, step_work, aloc=LOCAL):
        assert f_out.dtype == np.float32 and f_out.size == 3 * self.num_atoms and f_out.flags["C_CONTIGUOUS"]
        self._lib.nbnxm_gpu_launch_cpyback(self.h, _p(f_out), C.byref(step_work), C.c_int(aloc))

    def _finish_task(self, step_work, have_soft_core, aloc, wait):
        nl = self.n_lambda
        fe = np.zeros(nl + 1)
        fc = np.zeros(nl + 1)
        fv = np.zeros(nl + 1)
        ed = EnerData()
        ed.n_lambda = nl
        ed.foreign_energies = fe.ctypes.data_as(C.POINTER(C.c_double))
        ed.foreign_dhdl_coul = fc.ctypes.data_as(C.POINTER(C.c_double))
        ed.foreign_dhdl_vdw = fv.ctypes.data_as(C.POINTER(C.c_double))
        fshift = np.zeros((NUM_SHIFT_VECTORS, 3), np.float32)
        fn = self._lib.nbnxm_gpu_wait_finish_task if wait else self._lib.nbnxm_gpu_try_finish_task
        done = fn(self.h, C.byref(step_work), C.c_int(aloc), C.c_int(1 if have_soft_core else 0), C.byref(ed), _p(fshift))
        if not wait and not done:
            return None
        return dict(e_lj=ed.e_lj, e_el=ed.e_el, dvdl_lin=list(ed.dvdl_lin), dvdl_nonlin=list(ed.dvdl_nonlin),
                    foreign_energies=fe, foreign_dhdl_coul=fc, foreign_dhdl_vdw=fv, fshift=fshift)

    def wait_finish_task(self, step_work, have_soft_core, aloc=LOCAL):
        """Returns dict(e_lj, e_el, dvdl_lin, dvdl_nonlin, foreign_energies, foreign_dhdl_coul,
        foreign_dhdl_vdw, fshift) — the caller-owned accumulators, zero-initialised here."""
        return self._finish_task(step_work, have_soft_core, aloc, True)

    def try_finish_task(self, step_work, have_soft_core, aloc=LOCAL):
        """gpu_try_finish_task: None while the locality's work is still running, else the dict of wait_finish_task"""
        return self._finish_task(step_work, have_soft_core, aloc, False)

    def set_timing(self, enable):
        self._lib.nbnxm_gpu_set_timing(self.h, C.c_int(1 if enable else 0))

    def get_timings(self):
        t = GpuTimings()
        self._lib.nbnxm_gpu_get_timings(self.h, C.byref(t))
        return t

    def reset_timings(self):
        self._lib.nbnxm_gpu_reset_timings(self.h)

    def free(self):
        if getattr(self, "_h", None):
            self._lib.nbnxm_gpu_free(self.h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_PINNED_KEEPALIVE = []


def pinned_copy(arr):
    """A copy of a numpy array (structured dtypes too) in page-locked host memory, as the reference keeps its pair lists and atom data
    (HostVector with the pinning allocator): nbnxm_gpu_init_pairlist reads such arrays in place instead of staging them.
    Needs torch with a GPU (the tensor that owns the memory is kept alive for the life of the process)."""
    import torch
    a = np.ascontiguousarray(arr)
    t = torch.empty(max(1, a.nbytes), dtype=torch.uint8).pin_memory()
    _PINNED_KEEPALIVE.append(t)
    out = t.numpy()[:a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def download_cjpacked(nb, ncj, iloc=LOCAL):
    """Test helper: reads the device copy of the packed j-list back (after pruning)."""
    lib = hip_lib()
    lib.nbnxm_gpu_debug_get_cjpacked.restype = C.c_void_p
    ptr = lib.nbnxm_gpu_debug_get_cjpacked(nb.h, C.c_int(iloc))
    out = np.zeros(ncj, CJ_PACKED_DTYPE)
    lib.nbnxm_gpu_debug_download(nb.h, C.c_void_p(ptr), _p(out), C.c_size_t(out.nbytes))
    return out


def halo_pack_x(stream, d_x, d_map, map_size, d_send_buf, coordinate_shift=None):
    """sendBuf[i] = x[map[i]] (+ shift); all buffers are device pointers (ints), stream a hipStream_t value (int or None)."""
    sh = _a(coordinate_shift, np.float32) if coordinate_shift is not None else None
    hip_lib().nbnxm_gpu_halo_pack_x(C.c_void_p(stream), C.c_void_p(d_x), C.c_void_p(d_map), C.c_int(map_size),
                                    _p(sh) if sh is not None else None, C.c_void_p(d_send_buf))


def halo_unpack_f(stream, d_f, d_map, map_size, d_recv_buf, accumulate=True):
    """f[map[i]] (+)= recvBuf[i]; device pointers as in halo_pack_x."""
    hip_lib().nbnxm_gpu_halo_unpack_f(C.c_void_p(stream), C.c_void_p(d_f), C.c_void_p(d_map), C.c_int(map_size),
                                      C.c_void_p(d_recv_buf), C.c_int(1 if accumulate else 0))


# ---- listed (bonded) interactions with FEP: include/listed_hip.h ---------------------------------------------
LISTED_TYPES = {"bonds": 0, "angles": 1, "urey_bradley": 2, "pdihs": 3, "rbdihs": 4, "idihs": 5, "lj14": 6, "ljc14_q": 7,
                "ljc_pairs_nb": 8, "restrbonds": 9, "angres": 10, "dihres": 11}
LISTED_NRAL = {"bonds": 2, "angles": 3, "urey_bradley": 3, "pdihs": 4, "rbdihs": 4, "idihs": 4, "lj14": 2, "ljc14_q": 2,
               "ljc_pairs_nb": 2, "restrbonds": 2, "angres": 4, "dihres": 4}
LISTED_IPARAMS = np.dtype([("p", np.float32, 12), ("mult", np.int32)])
LISTED_NUM_ENERGY_TERMS = 14   # one per function type + Coulomb-14 + Coulomb of F_LJC_PAIRS_NB
LISTED_ENERGY_COULOMB14 = 12
LISTED_ENERGY_COULOMB_PAIRS_NB = 13
LISTED_DVDL = {"bonded": 0, "coul": 1, "vdw": 2, "restraint": 3}


class ListedFepParams(C.Structure):
    """listed_gpu_fep_params_t (gmx::BondedFepParameters)"""
    _fields_ = [("alphaCoul", C.c_float), ("alphaVdw", C.c_float), ("lambdaPower", C.c_int), ("sc_sigma6", C.c_float),
                ("sc_sigma6_min", C.c_float), ("lambdaBonded", C.c_float), ("lambdaCoul", C.c_float), ("lambdaVdw", C.c_float),
                ("lambdaRestraint", C.c_float)]


class ListedGpu:
    """ctypes mirror of gmx::ListedForcesGpu for the perturbed function types (listed_forces_gpu.h:120-190)."""

    def __init__(self, stream=None):
        self._lib = hip_lib()
        self._lib.listed_gpu_create.restype = C.c_void_p
        self._h = self._lib.listed_gpu_create(C.c_void_p(stream))
        if not self._h:
            raise RuntimeError("listed_gpu_create failed")

    @property
    def h(self):
        return C.c_void_p(self._h)

    def set_force_params(self, params):
        prm = np.ascontiguousarray(params, dtype=LISTED_IPARAMS)
        self._lib.listed_gpu_set_force_params(self.h, C.c_int(prm.size), _p(prm))

    def update_interaction_list(self, type_name, iatoms, num_atoms):
        ia = _a(iatoms, np.int32)
        n = ia.size // (1 + LISTED_NRAL[type_name])
        self._lib.listed_gpu_update_interaction_list(self.h, C.c_int(LISTED_TYPES[type_name]), C.c_int(n), _p(ia), C.c_int(num_atoms))

    def have_interactions(self):
        return bool(self._lib.listed_gpu_have_interactions(self.h))

    def launch_kernel(self, d_xq, d_f, d_fshift, box, pbc_type, fep, d_q4=None, elec_scale=0.0, compute_energy=True,
                      compute_virial=True, epsfac=0.0):
        """fep: ListedFepParams; d_q4: device float4[] with (qA, qB) per atom, needed with 1-4 pairs; elec_scale = epsfac * fudgeQQ
        for F_LJ14, epsfac for F_LJC14_Q and F_LJC_PAIRS_NB"""
        b = _a(box, np.float32)
        assert b.size == 9
        self._lib.listed_gpu_launch_kernel(self.h, C.c_void_p(d_xq), C.c_void_p(d_q4), C.c_void_p(d_f), C.c_void_p(d_fshift), _p(b),
                                           C.c_int(pbc_type), C.byref(fep), C.c_float(elec_scale), C.c_float(epsfac),
                                           C.c_int(1 if compute_energy else 0), C.c_int(1 if compute_virial else 0))

    def energies(self):
        """launch_energy_transfer + wait_accumulate_energy_terms: (energy terms[14], dV/dlambda[bonded, coul, vdw, restraint])"""
        epot = np.zeros(LISTED_NUM_ENERGY_TERMS, np.float64)
        dvdl = np.zeros(len(LISTED_DVDL), np.float64)
        self._lib.listed_gpu_launch_energy_transfer(self.h)
        self._lib.listed_gpu_wait_accumulate_energy_terms(self.h, _p(epot), _p(dvdl))
        return epot, dvdl

    def clear_energies(self):
        self._lib.listed_gpu_clear_energies(self.h)

    def free(self):
        if getattr(self, "_h", None):
            self._lib.listed_gpu_free(self.h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- stochastic-dynamics update: include/update_hip.h ---------------------------------------------------------
LANGEVIN_FORCES_ONLY, LANGEVIN_FRICTION_AND_NOISE = 0, 1


class LangevinGpu:
    """ctypes mirror of gmx::LangevinGpu (mdlib/langevin_gpu.h:90-160)"""

    def __init__(self, ref_t, tau_t, delta_t, stream=None):
        self._lib = hip_lib()
        self._lib.langevin_gpu_create.restype = C.c_void_p
        rt, tt = _a(ref_t, np.float32), _a(tau_t, np.float32)
        assert rt.size == tt.size
        self._h = self._lib.langevin_gpu_create(C.c_void_p(stream), C.c_int(rt.size), C.c_float(delta_t), _p(rt), _p(tt))

    @property
    def h(self):
        return C.c_void_p(self._h)

    def set(self, inverse_masses, temp_coupl_groups):
        im, tc = _a(inverse_masses, np.float32), _a(temp_coupl_groups, np.uint16)
        assert im.size == tc.size
        self._lib.langevin_gpu_set(self.h, C.c_int(im.size), _p(im), _p(tc))

    def integrate(self, d_x, d_xp, d_v, d_f, dt, seed, step, update_type):
        self._lib.langevin_gpu_integrate(self.h, C.c_void_p(d_x), C.c_void_p(d_xp), C.c_void_p(d_v), C.c_void_p(d_f), C.c_float(dt),
                                         C.c_int(seed), C.c_int(step), C.c_int(update_type))

    def free(self):
        if getattr(self, "_h", None):
            self._lib.langevin_gpu_free(self.h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _Handle:
    _free_name = None

    @property
    def h(self):
        return C.c_void_p(self._h)

    def free(self):
        if getattr(self, "_h", None):
            getattr(self._lib, self._free_name)(self.h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _box9(box):
    return _a(np.zeros((3, 3)) if box is None else box, np.float32).reshape(-1)


class LeapFrogGpu(_Handle):
    """ctypes mirror of gmx::LeapFrogGpu (mdlib/leapfrog_gpu.h:95-170)"""
    _free_name = "leapfrog_gpu_free"

    def __init__(self, num_temp_scale_values=0, stream=None):
        self._lib = hip_lib()
        self._lib.leapfrog_gpu_create.restype = C.c_void_p
        self._h = self._lib.leapfrog_gpu_create(C.c_void_p(stream), C.c_int(num_temp_scale_values))

    def set(self, inverse_masses, temp_scale_groups=None):
        im = _a(inverse_masses, np.float32)
        tc = _a(temp_scale_groups, np.uint16) if temp_scale_groups is not None else None
        self._lib.leapfrog_gpu_set(self.h, C.c_int(im.size), _p(im), _p(tc) if tc is not None else None)

    def integrate(self, d_x, d_xp, d_v, d_f, dt, tc_lambdas=None, pr_matrix=None, dt_pressure_couple=0.0):
        lam = _a(tc_lambdas, np.float32) if tc_lambdas is not None else None
        prm = _a(pr_matrix, np.float32).reshape(-1) if pr_matrix is not None else None
        self._lib.leapfrog_gpu_integrate(self.h, C.c_void_p(d_x), C.c_void_p(d_xp), C.c_void_p(d_v), C.c_void_p(d_f), C.c_float(dt),
                                         C.c_int(lam is not None), _p(lam) if lam is not None else None, C.c_int(prm is not None),
                                         C.c_float(dt_pressure_couple), _p(prm) if prm is not None else None)


class SettleGpu(_Handle):
    """ctypes mirror of gmx::SettleGpu (mdlib/settle_gpu.h:70-150)"""
    _free_name = "settle_gpu_free"

    def __init__(self, mO, mH, dOH, dHH, stream=None):
        self._lib = hip_lib()
        self._lib.settle_gpu_create.restype = C.c_void_p
        self._h = self._lib.settle_gpu_create(C.c_void_p(stream), C.c_float(mO), C.c_float(mH), C.c_float(dOH), C.c_float(dHH))

    def set(self, atoms):
        at = _a(atoms, np.int32).reshape(-1)
        self._lib.settle_gpu_set(self.h, C.c_int(at.size // 3), _p(at))

    def apply(self, d_x, d_xp, d_v=None, invdt=0.0, compute_virial=False, pbc_type=3, box=None):
        """returns the scaled virial contribution (3x3) when compute_virial"""
        vir = np.zeros(9, np.float32)
        self._lib.settle_gpu_apply(self.h, C.c_void_p(d_x), C.c_void_p(d_xp), C.c_int(d_v is not None), C.c_void_p(d_v), C.c_float(invdt),
                                   C.c_int(bool(compute_virial)), _p(vir), C.c_int(pbc_type), _p(_box9(box)))
        return vir.reshape(3, 3)


class LincsGpu(_Handle):
    """ctypes mirror of gmx::LincsGpu (mdlib/lincs_gpu.h:75-160)"""
    _free_name = "lincs_gpu_free"

    def __init__(self, num_iterations, expansion_order, stream=None):
        self._lib = hip_lib()
        self._lib.lincs_gpu_create.restype = C.c_void_p
        self._h = self._lib.lincs_gpu_create(C.c_void_p(stream), C.c_int(num_iterations), C.c_int(expansion_order))

    def set(self, iatoms, lengths, inverse_masses):
        """iatoms: (type, i, j) triples; returns False when a group of coupled constraints is too large for the GPU"""
        ia, ln, im = _a(iatoms, np.int32).reshape(-1), _a(lengths, np.float32), _a(inverse_masses, np.float32)
        return self._lib.lincs_gpu_set(self.h, C.c_int(ia.size // 3), _p(ia), _p(ln), C.c_int(im.size), _p(im)) == 0

    def apply(self, d_x, d_xp, d_v=None, invdt=0.0, compute_virial=False, pbc_type=3, box=None):
        vir = np.zeros(9, np.float32)
        self._lib.lincs_gpu_apply(self.h, C.c_void_p(d_x), C.c_void_p(d_xp), C.c_int(d_v is not None), C.c_void_p(d_v), C.c_float(invdt),
                                  C.c_int(bool(compute_virial)), _p(vir), C.c_int(pbc_type), _p(_box9(box)))
        return vir.reshape(3, 3)


class _UpdateConstrainParams(C.Structure):
    _fields_ = [("useStochasticDynamics", C.c_int), ("numTempCouplGroups", C.c_int), ("delta_t", C.c_float),
                ("ref_t", C.POINTER(C.c_float)), ("tau_t", C.POINTER(C.c_float)), ("nLincsIter", C.c_int), ("nProjOrder", C.c_int),
                ("haveSettle", C.c_int), ("mO", C.c_float), ("mH", C.c_float), ("dOH", C.c_float), ("dHH", C.c_float)]


class _UpdateConstrainTopology(C.Structure):
    _fields_ = [("numAtoms", C.c_int), ("inverseMasses", C.POINTER(C.c_float)), ("tempCouplGroups", C.POINTER(C.c_ushort)),
                ("numConstraints", C.c_int), ("constraints", C.POINTER(C.c_int)), ("constraintLengths", C.POINTER(C.c_float)),
                ("numSettles", C.c_int), ("settles", C.POINTER(C.c_int))]


class UpdateConstrainGpu(_Handle):
    """ctypes mirror of gmx::UpdateConstrainGpu (mdlib/update_constrain_gpu.h:70-185)"""
    _free_name = "update_constrain_gpu_free"

    def __init__(self, delta_t, num_temp_coupl_groups=0, stochastic_dynamics=False, ref_t=None, tau_t=None, n_lincs_iter=1, n_proj_order=4,
                 settle=None, stream=None):
        """settle: None or (mO, mH, dOH, dHH)"""
        self._lib = hip_lib()
        self._lib.update_constrain_gpu_create.restype = C.c_void_p
        self._lib.update_constrain_gpu_x_updated_event.restype = C.c_void_p
        rt = _a(ref_t if ref_t is not None else [0.0], np.float32)
        tt = _a(tau_t if tau_t is not None else [0.0], np.float32)
        p = _UpdateConstrainParams(int(stochastic_dynamics), num_temp_coupl_groups, delta_t, rt.ctypes.data_as(C.POINTER(C.c_float)),
                                   tt.ctypes.data_as(C.POINTER(C.c_float)), n_lincs_iter, n_proj_order, int(settle is not None),
                                   *(settle if settle is not None else (0.0, 0.0, 0.0, 0.0)))
        self._h = self._lib.update_constrain_gpu_create(C.c_void_p(stream), C.byref(p))

    def set(self, d_x, d_v, d_f, inverse_masses, temp_coupl_groups=None, constraints=None, constraint_lengths=None, settles=None):
        im = _a(inverse_masses, np.float32)
        tc = _a(temp_coupl_groups if temp_coupl_groups is not None else np.zeros(im.size), np.uint16)
        cs = _a(constraints if constraints is not None else [], np.int32).reshape(-1)
        cl = _a(constraint_lengths if constraint_lengths is not None else [0.0], np.float32)
        st = _a(settles if settles is not None else [], np.int32).reshape(-1)
        t = _UpdateConstrainTopology(im.size, im.ctypes.data_as(C.POINTER(C.c_float)), tc.ctypes.data_as(C.POINTER(C.c_ushort)),
                                     cs.size // 3, cs.ctypes.data_as(C.POINTER(C.c_int)), cl.ctypes.data_as(C.POINTER(C.c_float)),
                                     st.size // 3, st.ctypes.data_as(C.POINTER(C.c_int)))
        return self._lib.update_constrain_gpu_set(self.h, C.c_void_p(d_x), C.c_void_p(d_v), C.c_void_p(d_f), C.byref(t)) == 0

    def set_pbc(self, pbc_type, box):
        self._lib.update_constrain_gpu_set_pbc(self.h, C.c_int(pbc_type), _p(_box9(box)))

    def integrate(self, dt, update_velocities=True, compute_virial=False, tc_lambdas=None, pr_matrix=None, dt_pressure_couple=0.0, seed=0,
                  step=0, f_ready_event=None):
        """returns the constraint virial (3x3; zeros unless compute_virial)"""
        vir = np.zeros(9, np.float32)
        lam = _a(tc_lambdas, np.float32) if tc_lambdas is not None else None
        prm = _a(pr_matrix, np.float32).reshape(-1) if pr_matrix is not None else None
        self._lib.update_constrain_gpu_integrate(self.h, C.c_void_p(f_ready_event), C.c_float(dt), C.c_int(bool(update_velocities)),
                                                 C.c_int(bool(compute_virial)), _p(vir), C.c_int(lam is not None),
                                                 _p(lam) if lam is not None else None, C.c_int(prm is not None), C.c_float(dt_pressure_couple),
                                                 _p(prm) if prm is not None else None, C.c_int(seed), C.c_int(step))
        return vir.reshape(3, 3)

    def set_nbat_coupling(self, cell, d_xq, d_f_nbat):
        """MI355X extension: atom -> grid-slot map and the non-bonded xq / f device buffers, for integrate_fused"""
        self._lib.update_constrain_gpu_set_nbat_coupling(self.h, _p(_a(cell, np.int32)), C.c_void_p(d_xq), C.c_void_p(d_f_nbat))

    def can_fuse(self):
        return bool(self._lib.update_constrain_gpu_can_fuse(self.h))

    def integrate_fused(self, dt, compute_virial=False, tc_lambdas=None, pr_matrix=None, dt_pressure_couple=0.0, seed=0, step=0,
                        add_atom_order_forces=False, f_ready_event=None):
        vir = np.zeros(9, np.float32)
        lam = _a(tc_lambdas, np.float32) if tc_lambdas is not None else None
        prm = _a(pr_matrix, np.float32).reshape(-1) if pr_matrix is not None else None
        self._lib.update_constrain_gpu_integrate_fused(self.h, C.c_void_p(f_ready_event), C.c_float(dt), C.c_int(bool(compute_virial)), _p(vir),
                                                       C.c_int(lam is not None), _p(lam) if lam is not None else None,
                                                       C.c_int(prm is not None), C.c_float(dt_pressure_couple),
                                                       _p(prm) if prm is not None else None, C.c_int(seed), C.c_int(step),
                                                       C.c_int(bool(add_atom_order_forces)))
        return vir.reshape(3, 3)

    def scale_coordinates(self, matrix):
        self._lib.update_constrain_gpu_scale_coordinates(self.h, _p(_box9(matrix)))

    def scale_velocities(self, matrix):
        self._lib.update_constrain_gpu_scale_velocities(self.h, _p(_box9(matrix)))

    def x_updated_event(self):
        return self._lib.update_constrain_gpu_x_updated_event(self.h)
