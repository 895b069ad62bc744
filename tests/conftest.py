import faulthandler
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fatal signal (SIGABRT included) leaves every thread's Python stack on stderr; on a GPU box the native stack goes to
    # gpurun_out/abort_trace.txt as well (tests/csrc/abort_trace.c), so that an abort can be read afterwards instead of re-run.
    # (pytest's own faulthandler plugin does this as well; the native tracer is installed by the session fixture below, after
    # every plugin's configure hook, so that it is the first handler to run and faulthandler's the second)
    if not faulthandler.is_enabled():
        faulthandler.enable(all_threads=True)


def _install_native_abort_trace():
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tests", "csrc", "abort_trace.c")
    so = os.path.join(ROOT, "tests", "csrc", "libabort_trace.so")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", src, "-o", so])
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        L = ctypes.CDLL(so)
        L.abort_trace_install.argtypes = [ctypes.c_char_p]
        L.abort_trace_install(os.path.join(out_dir, "abort_trace_%d.txt" % os.getpid()).encode())
    except Exception as e:   # evidence gathering only: never in the way of the tests
        print("abort trace not installed: %r" % (e,), file=sys.stderr)


@pytest.fixture(autouse=True, scope="session")
def _deterministic_teardown():
    """Teardown in a defined order while the HIP runtime is certainly alive: every context the tests left open is closed
    (cabac_hip_destroy waits for its streams first), then every pinned array, then the device is idle — and only then do the
    interpreter and the loaded libraries finalize.  (Round 2 ended GPU runs with os._exit() after one unexplained SIGABRT
    at the end of a run; that hid library teardown from every test.  DESIGN.md section 4 has the analysis.)"""
    _install_native_abort_trace()
    yield
    try:
        from entropy_coding_amd import capi
    except Exception:
        return
    capi.close_all()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
