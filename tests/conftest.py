import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_FAULT_LOG = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # A fatal signal in a test process (a GPU memory fault aborts through the runtime) leaves its Python stacks in gpurun_out/, which
    # travels back from the GPU box: the tail of stderr that a driver keeps does not reach the top of that dump.
    global _FAULT_LOG
    try:
        import faulthandler
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        _FAULT_LOG = open(os.path.join(ROOT, "gpurun_out", f"faulthandler.{os.getpid()}.log"), "w")
        faulthandler.enable(file=_FAULT_LOG, all_threads=True)
    except Exception:
        _FAULT_LOG = None


def pytest_unconfigure(config):
    global _FAULT_LOG
    if _FAULT_LOG is not None:
        try:
            import faulthandler
            faulthandler.disable()
            name = _FAULT_LOG.name
            _FAULT_LOG.close()
            if os.path.getsize(name) == 0:
                os.remove(name)
        except Exception:
            pass
        _FAULT_LOG = None


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no HIP device is visible")
    return torch.device("cuda:0")
