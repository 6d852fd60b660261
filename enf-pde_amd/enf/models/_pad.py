"""Zero-padding of a narrow model (num_hidden 16 / 32 / any even width below a kernel width) to the width the HIP
kernels are built for (64 or 128).

Exact: a padded feature is 0 wherever it is produced by a Dense with zero-padded columns and bias (gelu(0) = 0,
relu(0) = 0), is multiplied by zero-padded rows wherever it is consumed, and the three places where the width itself
enters the arithmetic take the true width from EnfDesc.d_true: LayerNorm statistics (zeros add nothing to the sums),
the D^-1/2 logit scale, and the [sin | cos] split of the RFF encoding (padded per half; cos(0) = 1 on padded
coefficients meets zero rows).  The pads are ordinary differentiable tensor ops, so the training path runs in the
padded width and gradients flow back to the true tensors.
"""
import torch
import torch.nn.functional as Fnn

KERNEL_WIDTHS = (64, 128)


def padded_width(D):
    for w in KERNEL_WIDTHS:
        if D <= w:
            return w
    raise NotImplementedError(f"num_hidden={D} exceeds the widest kernel ({KERNEL_WIDTHS[-1]})")


def padded_heads(H):
    """num_heads = 3 runs as 4 heads, the fourth all zero (kernels exist for 1, 2 and -- 64-wide only -- 4 heads)."""
    return 4 if H == 3 else H


def _pad_axis(t, axis, kind, D, Dp, H, Hp):
    if kind is None or (D == Dp and (H == Hp or kind not in ("HD", "2HD"))):
        return t
    axis = axis % t.dim()
    shp = list(t.shape)

    def blocks(nb, width, new_width, groups=1, new_per_group=None):
        """axis = nb blocks of `width` -> blocks of `new_width`; with groups, each group of nb/groups blocks is
        extended to new_per_group blocks (zero heads)."""
        per = nb // groups
        npg = per if new_per_group is None else new_per_group
        v = t.reshape(shp[:axis] + [groups, per, width] + shp[axis + 1:])
        tail = [0, 0] * (v.dim() - axis - 3)
        v = Fnn.pad(v, tail + [0, new_width - width, 0, npg - per])
        return v.reshape(shp[:axis] + [groups * npg * new_width] + shp[axis + 1:])
    if kind == "D":
        return blocks(1, D, Dp)
    if kind == "Dh":
        return blocks(1, D // 2, Dp // 2)
    if kind == "R":                              # [sin (D/2) | cos (D/2)]
        return blocks(2, D // 2, Dp // 2)
    if kind == "HD":
        return blocks(H, D, Dp, 1, Hp)
    if kind == "2HD":                            # [gamma (H, D) | beta (H, D)]
        return blocks(2 * H, D, Dp, 2, Hp)
    raise ValueError(kind)


_RFF = [(None, "Dh"), ("R", "D"), ("D",), ("D", "D"), ("D",)]
_KB = lambda a, b: [(a, b), (b,)]
_FFN = lambda a, h, b: _KB(a, h) + [(h,), (h,)] + _KB(h, b)
# per ENF_W_* tensor, the kind of each axis
AXES = ([(None, "D"), ("D",), ("D",), ("D",)] + _RFF + _RFF + _KB("D", "HD") * 3 + _FFN("D", "D", "2HD") + _FFN("D", "D", "D") +
        _KB("HD", "HD") + _FFN("HD", "HD", "HD") + _KB("HD", "D") + _KB("D", "D") + [("D", None), (None,)])
assert len(AXES) == 46


# a latent self-attention block (NEF:137-167): the attention's 30 tensors as above (ENF_W_LNA_G .. ENF_W_MX_B1), then out_proj
# HD -> D (project_heads) and a D -> D -> D FFN
BLOCK_AXES = AXES[2:32] + _KB("HD", "D") + _FFN("D", "D", "D")
assert len(BLOCK_AXES) == 38


def _pad_list(tensors, axes, D, Dp, H, Hp):
    out = []
    for t, kinds in zip(tensors, axes):
        for ax, kind in enumerate(kinds):
            t = _pad_axis(t, ax, kind, D, Dp, H, Hp)
        out.append(t)
    return out


def pad_block_tensors(tensors, D, Dp, H, Hp=None):
    """The 38 tensors of one self-attention block of a (width D, H heads) model as a (width Dp, Hp heads) block."""
    Hp = H if Hp is None else Hp
    return list(tensors) if D == Dp and H == Hp else _pad_list(tensors, BLOCK_AXES, D, Dp, H, Hp)


def pad_tensors(tensors, D, Dp, H, Hp=None):
    """The 46 weight tensors (ENF_W_* order) of a (width D, H heads) model as a (width Dp, Hp heads) model."""
    Hp = H if Hp is None else Hp
    if D == Dp and H == Hp:
        return list(tensors)
    out = []
    for t, kinds in zip(tensors, AXES):
        for ax, kind in enumerate(kinds):
            t = _pad_axis(t, ax, kind, D, Dp, H, Hp)
        out.append(t)
    return out
