import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(HERE, "golden")


@pytest.fixture(scope="session")
def vamp():
    """The product package (loads the C-ABI library; fails loudly if it is missing)."""
    lib = os.path.join(ROOT, "vamp_mvt_amd", "libvamp_mvt_amd.so")
    if not os.path.exists(lib):
        sys.path.insert(0, ROOT)
        import __graft_entry__

        __graft_entry__.build_library()
    import vamp_mvt_amd

    return vamp_mvt_amd
