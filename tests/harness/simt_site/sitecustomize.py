"""On PYTHONPATH only while TKAMD_SIMT=1 tests run (tests/harness/simt_env.py puts it there): python subprocesses started by a test
get the same SIMT library swap and corpus scaling as the pytest process.  Test infrastructure only."""
import os
import sys

if os.environ.get("TKAMD_SIMT") == "1":
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    if _root not in sys.path:
        sys.path.insert(0, _root)
    try:
        import torch
        _gpu = torch.cuda.is_available()
    except Exception:
        _gpu = False
    if not _gpu:
        from tests.harness import simt_env
        simt_env.install()
