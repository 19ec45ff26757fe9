"""CPU-only checks of the drop-in boundary: the C-ABI libraries load without a GPU and export every
symbol that include/*.h declare (no compute call is made here)."""
import ctypes
import os
import re

import fep_testlib as tl

pkg = tl.pkg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_hip_library_exports_every_declared_symbol():
    lib = pkg.hip_lib()
    names = [n for n in declared_functions("nbnxm_hip.h", "nbnxm_") if not n.startswith("nbnxm_host_")]
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libnbnxm_hip.so does not export %s" % n
    assert sorted(names) == sorted(pkg.HIP_SYMBOLS)
    assert lib.nbnxm_hip_abi_version() == 1


def test_hip_library_exports_every_listed_forces_symbol():
    lib = pkg.hip_lib()
    names = declared_functions("listed_hip.h", "listed_gpu_")
    assert len(names) == 9
    for n in names:
        assert hasattr(lib, n), "libnbnxm_hip.so does not export %s" % n
    assert sorted(names) == sorted(pkg.LISTED_SYMBOLS)
    assert pkg.LISTED_IPARAMS.itemsize == 52
    assert ctypes.sizeof(pkg.ListedFepParams) == 36


def test_hip_library_exports_every_update_symbol():
    lib = pkg.hip_lib()
    names = declared_functions("update_hip.h", "(?:langevin|leapfrog|settle|lincs|update_constrain)_gpu_")
    assert len(names) == 27
    for n in names:
        assert hasattr(lib, n), "libnbnxm_hip.so does not export %s" % n
    assert sorted(names) == sorted(pkg.UPDATE_SYMBOLS)


def test_hip_library_exports_every_halo_exchange_symbol():
    lib = pkg.hip_lib()
    names = declared_functions("halo_hip.h", "halo_gpu_")
    assert len(names) == 17
    for n in names:
        assert hasattr(lib, n), "libnbnxm_hip.so does not export %s" % n
    assert sorted(names) == sorted(pkg.HALO_SYMBOLS)


def test_host_library_exports_every_declared_symbol():
    lib = pkg.host_lib()
    names = declared_functions("nbnxm_host.h", "nbnxm_host_")
    for n in names:
        assert hasattr(lib, n), "libnbnxm_host.so does not export %s" % n
    assert sorted(names) == sorted(pkg.HOST_SYMBOLS)
    assert lib.nbnxm_host_abi_version() == 1


def test_struct_layouts_match_the_header():
    # sizes the kernels rely on (nbnxm/pairlist.h:198-280 with cluster-pair split 2)
    assert pkg.SCI_DTYPE.itemsize == 16
    assert pkg.CJ_PACKED_DTYPE.itemsize == 32
    assert pkg.EXCL_DTYPE.itemsize == 128
    assert ctypes.sizeof(pkg.StepWorkload) == 20
    assert pkg.CJ_PACKED_DTYPE.fields["imei"][1] == 16


def test_product_package_does_not_import_the_oracle():
    # the oracle is test infrastructure: nothing under the package may reference it
    pkg_dir = os.path.join(ROOT, "gromacs-fep-gpu_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        if os.sep + "build" in dirpath or os.sep + "lib" in dirpath:
            continue
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle_binding" not in text and "liboracle" not in text and "fep_oracle" not in text, fn
