"""The HIP cluster-pair kernel against the reference's nblib known answers (api/nblib/tests/refdata/NBlibTest_*.xml,
fixture tests/golden/nblib_refdata.json): forces, energies and virial of the SPC-methanol and argon systems through the
C ABI, at the reference's float-build tolerance (5e-5 relative or 1e-6 absolute, api/nblib/tests/testhelpers.h:70-84)."""
import pytest

import nblib_cases as nc

DATA = nc.load()


@pytest.mark.gpu
@pytest.mark.parametrize("case", DATA["cases"], ids=[c["name"] for c in DATA["cases"]])
def test_hip_kernel_reproduces_nblib_known_answers(case):
    c = nc.build(DATA, case)
    got = nc.run_gpu(c)
    nc.check(case, got, abs_tol=1e-6, rel_floor=5e-5)
    # and the float oracle on the same float inputs, everything it produces (forces, energies, virial) at the 1e-4 bar of the path
    ref = nc.run_oracle(c, "f32")
    nc.check(dict(case, tolerance=1e-4, forces=ref["f"].tolist(), energies=ref["energies"], virial=ref["virial"].tolist()), got,
             abs_tol=1e-6)
