import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_hw: a gpu test the SIMT emulation cannot stand in for (skipped under TKAMD_SIMT=1)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# TKAMD_SIMT=1 (no GPU here): every -m gpu test RUNS, against the host build of the kernels under the SIMT shim (tests/harness/simt/,
# tests/harness/simt_build.py) instead of being skipped -- `TKAMD_SIMT=1 python -m pytest tests -m gpu` is the CPU rehearsal of the
# hardware gate (tools/simt_check.sh).  Sizes shrink through tests.helpers.N(); what cannot run there at all (torch device tensors,
# RCCL) is marked `needs_hw`.
SIMT = os.environ.get("TKAMD_SIMT") == "1"


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    skip_hw = pytest.mark.skip(reason="needs the real device (torch device memory / RCCL), not the SIMT emulation")
    for item in items:
        if "gpu" in item.keywords:
            if not SIMT:
                item.add_marker(skip)
            elif "needs_hw" in item.keywords:
                item.add_marker(skip_hw)


if SIMT and not _has_gpu():
    from tests.harness import simt_env
    simt_env.install()


@pytest.fixture(scope="session")
def ref_tokenizers():
    """The reference itself (prebuilt wheel), used only as the checker."""
    return pytest.importorskip("tokenizers")
