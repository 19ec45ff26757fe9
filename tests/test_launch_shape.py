"""CPU-only: the workgroup shape the launch gives the cluster-pair kernel (host arithmetic behind nbnxm_hip_query_launch_shape,
include/nbnxm_hip.h) — resident workgroups must fit the CU's 160 KB of LDS for every type count and flavour, the shape falls back to
4 waves per SIMD with larger workgroups exactly when five 4-wave copies no longer fit, and beyond what one 16-wave workgroup can
hold occupancy falls, and when no workgroup fits at all the query says so (the launch aborts with a message there)."""
import ctypes as C

import pytest

import fep_testlib as tl

pkg = tl.pkg
LDS_PER_CU = 160 * 1024


def shape(elec, vdw, energy, ntypes, tab_size=0):
    lib = pkg.hip_lib()
    w, s, b = C.c_int(), C.c_int(), C.c_int()
    lib.nbnxm_hip_query_launch_shape(C.c_int(elec), C.c_int(vdw), C.c_int(1 if energy else 0), C.c_int(ntypes), C.c_int(tab_size), C.byref(w),
                                     C.byref(s), C.byref(b))
    return w.value, s.value, b.value


def test_headline_flavour_keeps_five_waves_up_to_28_types():
    for n in (1, 3, 16, 28):
        assert shape(pkg.ELEC_EWALD_ANA, pkg.VDW_CUT, False, n)[:2] == (4, 5)
    assert shape(pkg.ELEC_EWALD_ANA, pkg.VDW_CUT, False, 29)[1] == 4


@pytest.mark.parametrize("elec", [pkg.ELEC_CUT, pkg.ELEC_RF, pkg.ELEC_EWALD_ANA, pkg.ELEC_EWALD_ANA_TWIN, pkg.ELEC_EWALD_TAB, pkg.ELEC_EWALD_TAB_TWIN])
@pytest.mark.parametrize("vdw", [pkg.VDW_CUT, pkg.VDW_CUT_COMB_GEOM, pkg.VDW_CUT_COMB_LB, pkg.VDW_FSWITCH, pkg.VDW_PSWITCH, pkg.VDW_EWALD_GEOM,
                                 pkg.VDW_EWALD_LB])
@pytest.mark.parametrize("energy", [False, True])
def test_resident_workgroups_fit_the_lds(elec, vdw, energy):
    tab = 3000 if elec in (pkg.ELEC_EWALD_TAB, pkg.ELEC_EWALD_TAB_TWIN) else 0
    last = 6
    for n in range(1, 150):
        w, s, b = shape(elec, vdw, energy, n, tab)
        if s == 0:
            # nothing fits any more: from here on the launch refuses; the table alone is then close to the whole LDS
            assert vdw not in (pkg.VDW_CUT_COMB_GEOM, pkg.VDW_CUT_COMB_LB), "combination-rule kernels have no type table"
            assert 8 * n * n > 100 * 1024
            break
        assert w in (4, 8, 16) and 1 <= s <= 5
        if 8 * n * n <= 90 * 1024:
            assert s >= 4, "tables up to 90 KB keep 4 waves per SIMD (larger workgroups share a copy)"
        workgroups_per_cu = 4 * s // w
        assert workgroups_per_cu >= 1 and workgroups_per_cu * b <= LDS_PER_CU, (n, w, s, b)
        assert s <= last, "occupancy must not grow with the table"
        last = s
    else:
        assert vdw in (pkg.VDW_CUT_COMB_GEOM, pkg.VDW_CUT_COMB_LB)


def test_many_types_switch_to_larger_workgroups():
    assert shape(pkg.ELEC_EWALD_ANA, pkg.VDW_CUT, False, 40) [:2] == (4, 4)
    assert shape(pkg.ELEC_EWALD_ANA, pkg.VDW_CUT, False, 64)[:2] == (8, 4)
    assert shape(pkg.ELEC_EWALD_ANA, pkg.VDW_CUT, False, 100)[:2] == (16, 4)
    # the energy flavours carry the 30 KB force + potential table (1,920 entries: round 4): four 4-wave workgroups fit a CU with a small
    # type table, two 8-wave ones beyond 11 types
    assert shape(pkg.ELEC_EWALD_ANA, pkg.VDW_CUT, True, 3)[:2] == (4, 4)
    assert shape(pkg.ELEC_EWALD_ANA, pkg.VDW_CUT, True, 11)[:2] == (4, 4)
    assert shape(pkg.ELEC_EWALD_ANA, pkg.VDW_CUT, True, 12)[:2] == (8, 4)
    assert shape(pkg.ELEC_RF, pkg.VDW_CUT, True, 3)[:2] == (4, 4)
