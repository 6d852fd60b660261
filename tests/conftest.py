import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=["latent_split", "z_fold"])
def pair_variant(request):
    """Runs a test under both forward pair-kernel variants (enf_set_zfold, include/enf_hip.h)."""
    from enf_pde_amd import _lib
    lib = _lib.load()
    lib.enf_set_zfold(1 if request.param == "z_fold" else 0)
    yield request.param
    lib.enf_set_zfold(-1)


@pytest.fixture(params=["unfolded", "z_fold"])
def bwd_variant(request):
    """Runs a test under both backward pair-kernel variants (enf_set_zfold_bwd, include/enf_hip.h)."""
    from enf_pde_amd import _lib
    lib = _lib.load()
    lib.enf_set_zfold_bwd(1 if request.param == "z_fold" else 0)
    yield request.param
    lib.enf_set_zfold_bwd(-1)
