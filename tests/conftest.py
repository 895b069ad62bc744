import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True, scope="session")
def _torch_fills_land_before_library_launches():
    """The library launches on its own non-blocking stream.  The tests allocate and fill their device buffers with torch
    (torch's stream) right before a *_device call, so every such call first waits for torch's work — what a caller of the
    device-pointer API has to do with an event or by handing the library its stream (cabac_hip_set_stream).  Found by the
    full-size C5 round trip failing once: the zero fill of its 300 MB byte buffer was still running when the encoder wrote."""
    try:
        import torch
        from entropy_coding_amd import capi
    except Exception:
        yield
        return
    if not torch.cuda.is_available():
        yield
        return
    patched = {}
    for name in dir(capi.CabacHip):
        if name.endswith("_device") and callable(getattr(capi.CabacHip, name)):
            fn = getattr(capi.CabacHip, name)
            patched[name] = fn

            def wrapper(self, *a, __fn=fn, **kw):
                torch.cuda.synchronize()
                return __fn(self, *a, **kw)

            setattr(capi.CabacHip, name, wrapper)
    yield
    for name, fn in patched.items():
        setattr(capi.CabacHip, name, fn)
