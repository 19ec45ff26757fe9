import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # pytest captures file descriptor 2, so the message of a fatal error of the library (which aborts the process) would be lost
    # with the capture file: the library appends it to this file too (csrc/device_utils.h: fatal()).
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out) and os.access(out, os.W_OK):
        os.environ.setdefault("NBNXM_HIP_FATAL_LOG", os.path.join(out, "nbnxm_hip_fatal.log"))
    # (no switch of the library is set here: every test runs the product's default routing unless it asks for something else itself —
    # elec="ewald_tab_kept" for the tabulated Ewald kernels, NBNXM_HIP_DIAGNOSTICS=1 plus a switch through monkeypatch for an A/B form)


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if someone runs everything on a CPU-only host, skip them.
    if have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
