"""MFMA fragment-layout self test: pack_panel + make_frags + gemm_stage against an exact matmul.
Integer-valued, asymmetric operands so a swapped row/col map or a wrong k permutation cannot hide
(cdna_hip_programming.md section 3, 'Always A=I-check with ASYMMETRIC B')."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bf16", [0, 1])
@pytest.mark.parametrize("M,K", [(64, 64), (64, 128), (64, 32), (16, 64)])
def test_panel_gemm_exact(cuda, bf16, M, K):
    from enf_pde_amd import _lib
    lib = _lib.load_test()          # the layout self-test entry points live in the test library only
    g = torch.Generator().manual_seed(M * 1000 + K + bf16)
    W = torch.randint(-4, 5, (K, M), generator=g).float().to(cuda)       # plain (in, out)
    X = torch.randint(-4, 5, (K, 16), generator=g).float().to(cuda)      # (in, cols)
    packed = torch.zeros(M * K * (2 if bf16 else 4), dtype=torch.uint8, device=cuda)
    Y = torch.zeros(M, 16, device=cuda)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.enf_debug_pack(p(packed), p(W), M, K, bf16, st) == 0
    assert lib.enf_debug_gemm(p(packed), p(X), p(Y), M, K, bf16, st) == 0
    torch.cuda.synchronize()
    ref = W.t() @ X
    assert torch.equal(Y, ref), f"max err {(Y - ref).abs().max().item()}"
