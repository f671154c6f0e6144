import os
import sys

import pytest
import torch  # noqa: F401  (first: torch bundles its own HIP runtime; loading libmgvcycle.so before it breaks torch.cuda)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mg():
    import multigrid_jl_amd
    return multigrid_jl_amd


@pytest.fixture(scope="session")
def built():
    """Build native pieces once if they are missing (HIP lib cross-compiles on CPU)."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "multigrid.jl_amd", "csrc", "libmgvcycle.so")
    orc = os.path.join(ROOT, "oracle", "liboracle_mg.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        g.build()
    return True
