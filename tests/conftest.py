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


@pytest.fixture(params=["latent_split", "z_fold", "z_fold_zsplit"])
def pair_variant(request):
    """Runs a test under every forward pair-kernel variant (EnfDesc.pair_fwd_variant, include/enf_hip.h)."""
    from enf_pde_amd.enf.models import EquivariantCrossAttentionNeF as NeF
    prev = NeF.default_pair_variants
    NeF.default_pair_variants = (request.param, prev[1])
    yield request.param
    NeF.default_pair_variants = prev


@pytest.fixture(params=["unfolded", "z_fold"])
def bwd_variant(request):
    """Runs a test under both backward pair-kernel variants (EnfDesc.pair_bwd_variant, include/enf_hip.h)."""
    from enf_pde_amd.enf.models import EquivariantCrossAttentionNeF as NeF
    prev = NeF.default_pair_variants
    NeF.default_pair_variants = (prev[0], "z_fold" if request.param == "z_fold" else "latent_split")
    yield request.param
    NeF.default_pair_variants = prev
