"""enf-pde_amd -- MI355X (gfx950) engine for the Equivariant Neural Field decoder of
david-knigge/enf-pde: fused HIP kernels behind a C-ABI (include/enf_hip.h), under a host-side
mirror of the reference's module interface.

    from enf_pde_amd.enf.models import EquivariantCrossAttentionNeF          # NEF:70-235
    from enf_pde_amd.enf.steerable_attention.invariant import get_ca_invariant
    from enf_pde_amd.fitting import get_model_pde, inner_loop, decode

There is no CPU or eager fallback: every compute call goes through libenf_hip.so and raises
if the library or a GPU is missing.
"""
__version__ = "0.1.0"
