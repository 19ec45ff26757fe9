"""Pins oracle/nbnxm_ref.c (the CPU twin of the cluster-pair kernel) with the reference's own nblib known answers:
forces, energies and the virial of the SPC-methanol and argon systems (api/nblib/tests/refdata/NBlibTest_*.xml,
transcribed by golden/make_nblib_golden.py), through libnbnxm_host's grid + GPU-layout pair list."""
import numpy as np
import pytest

import nblib_cases as nc

DATA = nc.load()


@pytest.mark.parametrize("case", DATA["cases"], ids=[c["name"] for c in DATA["cases"]])
def test_oracle_reproduces_nblib_known_answers(case):
    c = nc.build(DATA, case)
    got = nc.run_oracle(c, "f64")
    # double-precision oracle on the double inputs: the reference's own tolerances, with the absolute bar of its double build
    nc.check(case, got, abs_tol=1e-6 if "OnGpu" in case["name"] else 1e-9)


@pytest.mark.parametrize("case", DATA["cases"], ids=[c["name"] for c in DATA["cases"]])
def test_float_oracle_reproduces_nblib_known_answers(case):
    c = nc.build(DATA, case)
    got = nc.run_oracle(c, "f32")
    # float arithmetic on float coordinates: the argon pair sits at the LJ minimum (r = 0.385 nm), where the force is the small
    # difference of two terms, so the reference's double-build bars of 1e-7 / 1e-8 give way to its float bar of 5e-5
    nc.check(case, got, abs_tol=1e-6, rel_floor=5e-5)


def test_list_covers_every_pair_within_the_cutoff_once():
    """all 15 pairs of the SPC-methanol system are within 1.0 nm: 9 interact, 6 are excluded (intramolecular)"""
    case = DATA["cases"][0]
    c = nc.build(DATA, case)
    r = nc.ob.nbnxm_ref(c.plist.sci, c.plist.cjPacked, c.plist.excl, c.grid.xq, c.grid.type, c.grid.num_types,
                        c.grid.nbat_nbfp(c.nbfp), nc.ref_params(c), c.grid.shift_vec)
    assert r["npairs"] == 9
