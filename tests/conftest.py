import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# PyTorch ships its own copy of the HIP runtime.  A process that uses both torch and libmipt.so must load torch FIRST (then
# libmipt's libamdhip64 dependency resolves to the copy that is already there); the other way round the process holds two
# runtimes and the second one finds "No HIP GPUs".  Several GPU tests hand torch device buffers to the C-ABI, and which test
# module loads libmipt first depends on what was selected on the command line -- so settle the order here.
try:
    import torch  # noqa: F401
except ImportError:  # the CPU-only tests of the product and the oracle do not need it
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cube_obj(tmp_path_factory):
    from cuda_path_tracer_amd import generators as G
    return G.write_cube_obj(str(tmp_path_factory.mktemp("assets") / "cube.obj"))
