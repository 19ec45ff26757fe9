"""The reference-side binding (integration/nbnxm_hip_shim.cpp) against the reference's own headers, by the compiler's word.

Build container only (skipped where /root/reference is absent).  `g++ -std=c++17 -fsyntax-only` in the CPU-only configuration
of the reference's headers; the reference's build system is not run.  Its three cmake-generated headers (config.h,
gmxpre-config.h, gromacs/libgromacs_export.h) are created EMPTY in a temporary directory and the configuration macros the
included headers read are given as -D flags (the CUDA defaults of the GPU pair list: 8-atom clusters, 2 x 2 x 2 clusters per
cell, cluster-pair split on).  In that mode the shim defines its functions in namespace NbnxmShim and static_asserts each one
against the type of its Nbnxm:: declaration (see the file header)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SHIM = os.path.join(ROOT, "integration", "nbnxm_hip_shim.cpp")

DEFINES = ["NBNXM_SHIM_CHECK_SIGNATURES", "LIBGROMACS_EXPORT=", "GMX_DOUBLE=0", "GMX_GPU=0", "GMX_GPU_CUDA=0", "GMX_GPU_OPENCL=0",
           "GMX_GPU_SYCL=0", "GMX_GPU_NB_CLUSTER_SIZE=8", "GMX_GPU_NB_NUM_CLUSTER_PER_CELL_X=2", "GMX_GPU_NB_NUM_CLUSTER_PER_CELL_Y=2",
           "GMX_GPU_NB_NUM_CLUSTER_PER_CELL_Z=2", "GMX_GPU_NB_DISABLE_CLUSTER_PAIR_SPLIT=0"]


def compile_shim(source, tmp_path):
    cfg = tmp_path / "generated"
    (cfg / "gromacs").mkdir(parents=True, exist_ok=True)
    for name in ("config.h", "gmxpre-config.h", os.path.join("gromacs", "libgromacs_export.h")):
        (cfg / name).write_text("")
    inc = [str(cfg), REF + "/src", REF + "/src/include", REF + "/api/legacy/include", REF + "/src/external",
           REF + "/src/external/thread_mpi/include"] + sorted(glob.glob(REF + "/src/gromacs/*/include")) + [os.path.join(ROOT, "include")]
    cmd = ["g++", "-std=c++17", "-fsyntax-only"] + ["-I" + i for i in inc] + ["-D" + d for d in DEFINES] + [source]
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


needs_reference = pytest.mark.skipif(not os.path.isdir(REF + "/src/gromacs/nbnxm"), reason="the reference tree is only in the build container")


@needs_reference
def test_shim_compiles_against_the_reference_headers(tmp_path):
    p = compile_shim(SHIM, tmp_path)
    assert p.returncode == 0, p.stderr[-4000:]


@needs_reference
@pytest.mark.parametrize("old,new", [
    ("void gpu_launch_kernel(NbnxmGpu* nb, const gmx::StepWorkload& stepWork", "void gpu_launch_kernel(NbnxmGpu* nb, gmx::StepWorkload& stepWork"),
    ("nbat->natoms_local", "nbat->numAtomsLocal"),
    ("h_nblist->cjPacked.list_.data()", "h_nblist->cjPacked.data()"),
])
def test_a_wrong_signature_or_member_name_does_not_compile(tmp_path, old, new):
    text = open(SHIM).read()
    assert old in text
    bad = tmp_path / "shim_bad.cpp"
    bad.write_text(text.replace(old, new))
    p = compile_shim(str(bad), tmp_path)
    assert p.returncode != 0
