import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")

# torch ships its own copy of the HIP runtime.  When it is imported after librtc_hip.so has already
# initialised the system's runtime, the second runtime in the process finds no GPU ("No HIP GPUs are
# available"); imported first, the library binds to the runtime torch loaded and both share it.  Tests
# that use torch tensors as device buffers therefore need torch in the process before the first render
# call, whatever subset of the tests is selected.
try:
    import torch  # noqa: F401
except ImportError:
    pass
