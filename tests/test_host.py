"""CPU tests of the boundary: the C-ABI library loads and exports every symbol of include/enf_hip.h
(no compute without a GPU), descriptor validation / error mapping, and the host-side mirror of the
reference interface (constructor keywords, parameter tree, latents, error behaviour)."""
import ctypes
import os
import re
import sys
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from enf_pde_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "enf_hip.h")).read()
    declared = set(re.findall(r"\b(enf_[a-z_]+)\s*\(", hdr))
    assert {"enf_forward", "enf_backward_latents", "enf_pack_weights", "enf_packed_weight_bytes", "enf_workspace_bytes",
            "enf_strerror", "enf_check_desc", "enf_forward_stages"} <= declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.enf_abi_version() == 2
    # the product library exports the header and nothing for tests: the layout self-tests, the K3 epilogue dump and the
    # removed process-wide setters live in libenf_hip_test.so / nowhere
    for name in ("enf_debug_gemm", "enf_debug_pack", "enf_test_read_wave_sums", "enf_set_zfold", "enf_set_zfold_bwd", "enf_set_relu_masks"):
        assert not hasattr(lib, name), name
    from enf_pde_amd import _lib
    tl = _lib.load_test()
    for name in list(declared) + ["enf_debug_gemm", "enf_debug_pack", "enf_test_read_wave_sums"]:
        assert hasattr(tl, name), name


def test_desc_struct_matches_header():
    from enf_pde_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "enf_hip.h")).read()
    # field by field against the header's struct (all int32_t but the trailing mask pointer)
    body = hdr[hdr.index("typedef struct EnfDesc {"):hdr.index("} EnfDesc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for kind, names in re.findall(r"\b(int32_t|void\*)\s+([^;]+);", body):
        fields += [(n.strip(), kind) for n in names.split(",")]
    assert [n for n, _ in fields] == [f[0] for f in _lib.EnfDesc._fields_]
    for (n, kind), (_, ctype) in zip(fields, _lib.EnfDesc._fields_):
        assert ctype is (ctypes.c_void_p if kind == "void*" else ctypes.c_int32), n
    assert ctypes.sizeof(_lib.EnfDesc) == 18 * 4 + 8 and _lib.EnfDesc.relu_masks.offset == 72
    # the option enums of the binding are the header's
    for table, prefix in ((_lib.VARIANT, "ENF_VARIANT_"), (_lib.PREC, "ENF_PREC_")):
        for name, val in table.items():
            m = re.search(prefix + name.upper() + r" = (\d)", hdr)
            assert m is None or int(m.group(1)) == val, name
    for name, val in _lib.MASK_MODE.items():
        assert re.search(r"#define ENF_MASK_" + name.upper() + r" (\d)", hdr).group(1) == str(val)
    assert "ENF_NUM_TENSORS" in hdr and _lib.ENF_NUM_TENSORS == 46
    # enum order of the invariants is the binding's id table
    ids = re.findall(r"ENF_INV_([A-Z_]+) = (\d)", hdr)
    for name, i in ids:
        if name != "COUNT":
            assert _lib.INVARIANT_IDS[name.lower()] == int(i)


def test_check_desc_and_error_mapping(lib):
    from enf_pde_amd import _lib
    ok = _lib.make_desc(2, 100, 64, 2, 128, 16, 1, 2, 0, 1, 1)
    assert lib.enf_check_desc(ctypes.byref(ok)) == 0
    assert lib.enf_packed_weight_bytes(ctypes.byref(ok)) > 531585 * 2
    assert lib.enf_workspace_bytes(ctypes.byref(ok)) > 0
    bad_inv = _lib.make_desc(2, 100, 64, 2, 128, 16, 1, 2, 10, 1, 1)
    with pytest.raises(ValueError, match="Unknown invariant"):
        _lib.check(lib.enf_check_desc(ctypes.byref(bad_inv)))
    bad_dim = _lib.make_desc(2, 100, 64, 2, 128, 16, 1, 3, 0, 1, 1)     # rel_pos_periodic needs num_in == 2
    with pytest.raises(AssertionError):
        _lib.check(lib.enf_check_desc(ctypes.byref(bad_dim)))
    unsupported = _lib.make_desc(2, 100, 64, 3, 32, 16, 1, 2, 0, 1, 1)
    with pytest.raises(NotImplementedError):
        _lib.check(lib.enf_check_desc(ctypes.byref(unsupported)))
    empty = _lib.make_desc(0, 100, 64, 2, 128, 16, 1, 2, 0, 1, 1)
    with pytest.raises(_lib.EnfError):
        _lib.check(lib.enf_check_desc(ctypes.byref(empty)))
    assert lib.enf_packed_weight_bytes(ctypes.byref(unsupported)) == 0
    assert lib.enf_invariant_dim(0, 2) == 4 and lib.enf_invariant_dim(2, 2) == 1 and lib.enf_invariant_dim(4, 3) == 3
    assert lib.enf_invariant_pose_dim(3, 2) == 3
    assert lib.enf_invariant_dim(7, 3) == 5 and lib.enf_invariant_dim(8, 3) == 6 and lib.enf_invariant_pose_dim(7, 3) == 4
    # Ponita2D (the self-attention invariant of ponita): three invariants, queries (pos_x, pos_y, theta)
    assert lib.enf_invariant_dim(9, 3) == 3 and lib.enf_invariant_pose_dim(9, 3) == 3
    _lib.check(lib.enf_check_desc(ctypes.byref(_lib.make_desc(2, 16, 16, 2, 64, 8, 1, 3, 9, 1, 0))))
    with pytest.raises(AssertionError):
        _lib.check(lib.enf_check_desc(ctypes.byref(_lib.make_desc(2, 16, 16, 2, 64, 8, 1, 2, 9, 1, 0))))
    # ball / ball_lat: 3-d coordinates, 64-wide kernels only
    _lib.check(lib.enf_check_desc(ctypes.byref(_lib.make_desc(2, 100, 25, 4, 64, 32, 1, 3, 7, 1, 1))))
    for bad in (_lib.make_desc(2, 100, 25, 2, 128, 32, 1, 3, 7, 1, 1), _lib.make_desc(2, 100, 25, 4, 64, 32, 1, 2, 8, 1, 1)):
        with pytest.raises((_lib.EnfError, NotImplementedError, AssertionError)):
            _lib.check(lib.enf_check_desc(ctypes.byref(bad)))
    # per-call options are validated too: unknown variant, masks requested without a buffer
    for kw in (dict(variants=(4, 0)), dict(variants=(0, 3)), dict(variants=(0, -1))):
        assert lib.enf_check_desc(ctypes.byref(_lib.make_desc(2, 100, 64, 2, 128, 16, 1, 2, 0, 1, 1, **kw))) == -1
    nomask = _lib.make_desc(2, 100, 64, 2, 128, 16, 1, 2, 0, 1, 1)
    nomask.mask_mode = 2
    assert lib.enf_check_desc(ctypes.byref(nomask)) == -1
    # the variant a descriptor resolves to, and the sizes that follow from it, depend on the descriptor alone
    small, large = _lib.make_desc(2, 100, 64, 2, 128, 16, 1, 2, 0, 1, 1), _lib.make_desc(16, 4096, 64, 2, 128, 16, 1, 2, 0, 1, 1)
    assert lib.enf_pair_variant(ctypes.byref(small), 0) == 1 and lib.enf_pair_variant(ctypes.byref(large), 0) == 2
    assert lib.enf_pair_variant(ctypes.byref(small), 1) == 1 and lib.enf_pair_variant(ctypes.byref(large), 1) == 2
    forced = _lib.make_desc(2, 100, 64, 2, 128, 16, 1, 2, 0, 1, 1, variants=(2, 2))
    assert lib.enf_pair_variant(ctypes.byref(forced), 0) == 2 and lib.enf_pair_variant(ctypes.byref(forced), 1) == 2
    assert lib.enf_pair_scratch_bytes(ctypes.byref(small)) == 0 < lib.enf_pair_scratch_bytes(ctypes.byref(forced))
    assert lib.enf_workspace_bytes(ctypes.byref(forced)) > lib.enf_workspace_bytes(ctypes.byref(small))
    assert lib.enf_pair_variant(ctypes.byref(small), 0) == 1                      # ... and nothing sticks between calls
    # fewer than 192 tiles of 128 queries and >= 128 latents: the z-fold kernel over equal runs of latent steps
    c3 = _lib.make_desc(4, 96 * 48, 128, 2, 128, 32, 3, 2, 1, 1, 1)
    assert lib.enf_pair_variant(ctypes.byref(c3), 0) == 3
    assert lib.enf_pair_variant(ctypes.byref(_lib.make_desc(4, 96 * 48, 64, 2, 128, 32, 3, 2, 1, 1, 1)), 0) == 1       # 64 latents: not split
    c3ls = _lib.make_desc(4, 96 * 48, 128, 2, 128, 32, 3, 2, 1, 1, 1, variants=(1, 0))
    assert lib.enf_workspace_bytes(ctypes.byref(c3)) > lib.enf_workspace_bytes(ctypes.byref(c3ls))
    # NULL buffers are rejected before anything touches the (absent) GPU
    assert lib.enf_forward(ctypes.byref(ok), None, 0, None, None, None, None, None, None, None, None, 0, None) == -1
    # ... by the one-call inner step and the one-call training backward too (every pointer NULL, every number zero)
    def zeros(fn):
        out = []
        for t in fn.argtypes[1:]:
            if t in (ctypes.c_float, ctypes.c_double):
                out.append(0.0)
            elif t in (ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint, ctypes.c_size_t, ctypes.c_longlong):
                out.append(0)
            else:
                out.append(None)
        return out
    assert lib.enf_fit_step(ctypes.byref(ok), *zeros(lib.enf_fit_step)) == -1
    assert lib.enf_backward_all(ctypes.byref(ok), *zeros(lib.enf_backward_all)) == -1
    assert lib.enf_backward_all_scratch_bytes(ctypes.byref(ok), 1) > 0
    bad = _lib.make_desc(2, 16, 16, 2, 64, 8, 1, 2, 9, 1, 0)              # an invalid descriptor is refused before the arguments are looked at
    assert lib.enf_fit_step(ctypes.byref(bad), *zeros(lib.enf_fit_step)) < 0
    assert lib.enf_backward_all_scratch_bytes(ctypes.byref(bad), 1) == 0


def _partition(desc):
    from enf_pde_amd import _lib
    lib = _lib.load()
    v = [ctypes.c_int32(-1) for _ in range(3)]
    rc = lib.enf_pair_partition(ctypes.byref(desc), *[ctypes.byref(x) for x in v])
    return rc, tuple(x.value for x in v)


@pytest.mark.parametrize("B,N,Z,forced", [(4, 96 * 48, 128, False), (8, 2048, 128, False), (1, 4096, 512, False), (2, 100, 64, True),
                                          (3, 300, 7, True), (1, 16, 2, True), (5, 1000, 33, True), (2, 128 * 70, 200, False)])
def test_split_z_fold_partition_covers_every_latent_step_once(B, N, Z, forced):
    """enf_pair_partition: the runs the split z-fold kernel's workgroups walk, replayed with the kernel's own arithmetic (enf_pair_fwd.hip:
    segment loop; enf_zsplit_merge_kernel: parts of a tile), cover every (signal, tile, latent) exactly once, every segment lands in a slot
    below `parts`, no two segments of a tile share a slot, and the merge reads exactly the slots that were written."""
    from enf_pde_amd import _lib
    d = _lib.make_desc(B, N, Z, 2, 128, 16, 1, 2, 0, 1, 1, variants=(3 if forced else 0, 0))
    rc, (run, wgs, parts) = _partition(d)
    assert rc == 1 and _lib.load().enf_pair_variant(ctypes.byref(d), 0) == 3
    tiles = (N + 127) // 128 * B
    total = tiles * Z
    assert wgs == -(-total // run) and (forced or (wgs <= 256 and run >= 32))
    seen = np.zeros((tiles, Z), dtype=np.int32)
    slots = [set() for _ in range(tiles)]
    for c in range(wgs):
        f0, f1 = c * run, min((c + 1) * run, total)
        assert f0 < f1
        while f0 < f1:
            tf, z_lo = divmod(f0, Z)
            it = min(Z - z_lo, f1 - f0)
            part = c - (tf * Z) // run
            assert 0 <= part < parts and part not in slots[tf]
            slots[tf].add(part)
            seen[tf, z_lo:z_lo + it] += 1
            f0 += it
    assert (seen == 1).all()
    for tf in range(tiles):
        n = (tf * Z + Z - 1) // run - (tf * Z) // run + 1
        assert slots[tf] == set(range(n))
    assert max(len(x) for x in slots) == parts
    # the workspace holds `parts` partial-sum slots
    ls = _lib.make_desc(B, N, Z, 2, 128, 16, 1, 2, 0, 1, 1, variants=(2, 0))
    extra = _lib.load().enf_pair_scratch_bytes(ctypes.byref(d)) - _lib.load().enf_pair_scratch_bytes(ctypes.byref(ls))
    assert extra == 4 * parts * (B * N * 256 + B * N * 2 * 3)


def test_partition_is_reported_only_for_the_split_variant():
    from enf_pde_amd import _lib
    for kw in (dict(), dict(variants=(1, 0)), dict(variants=(2, 0))):
        rc, vals = _partition(_lib.make_desc(16, 4096, 64, 2, 128, 16, 1, 2, 0, 1, 1, **kw))
        assert rc == 0 and vals == (-1, -1, -1)
    # config 3's decode: 144 tiles x 128 latents in 256 runs of 72, a tile in at most 3 parts
    assert _partition(_lib.make_desc(4, 96 * 48, 128, 2, 128, 32, 3, 2, 1, 1, 1)) == (1, (72, 256, 3))
    # config 4's fit shape (8 signals x 512 points x 128 latents): 128 runs of 32 would fill half the chip -- the latent-split kernel
    c4 = _lib.make_desc(8, 512, 128, 2, 128, 16, 1, 2, 0, 1, 1)
    assert _partition(c4)[0] == 0 and _lib.load().enf_pair_variant(ctypes.byref(c4), 0) == 1


def test_invariant_factory_mirrors_reference():
    from enf_pde_amd.enf.steerable_attention.invariant import get_ca_invariant, get_sa_invariant
    for name in ("rel_pos_periodic", "latitude_periodic", "polar_periodic", "ponita", "abs_pos", "rel_pos", "norm_rel_pos", "ball", "ball_lat"):
        inv = get_ca_invariant(NS(invariant_type=name, num_in=2))
        spec = R.invariant_spec(name, 2)
        assert (inv.dim, inv.num_x_pos_dims, inv.num_z_pos_dims, inv.num_z_ori_dims) == \
               (spec["dim"], spec["dx"], spec["z_pos"], spec["z_ori"]), name
    assert get_ca_invariant(NS(invariant_type="rel_pos", num_in=3)).dim == 3
    sa = get_sa_invariant(NS(invariant_type="ponita", num_in=2))                 # invariant/__init__.py:30-32
    assert type(sa).__name__ == "Ponita2D" and (sa.dim, sa.num_x_ori_dims, sa.num_z_ori_dims) == (3, 1, 1)
    with pytest.raises(ValueError, match="Unknown invariant type"):
        get_ca_invariant(NS(invariant_type="bogus", num_in=2))
    with pytest.raises(AssertionError):
        get_ca_invariant(NS(invariant_type="rel_pos_periodic", num_in=3))
    assert type(get_ca_invariant(NS(invariant_type="ball", num_in=3))).__name__ == "BallInvariant"
    assert type(get_ca_invariant(NS(invariant_type="ball_lat", num_in=3))).__name__ == "BallLatInvariant"


def _nef(**kw):
    from enf_pde_amd.enf.models import EquivariantCrossAttentionNeF
    from enf_pde_amd.enf.steerable_attention.invariant import get_ca_invariant
    inv = get_ca_invariant(NS(invariant_type=kw.pop("invariant", "rel_pos_periodic"), num_in=2))
    args = dict(num_hidden=128, num_heads=2, num_layers=0, num_out=1, latent_dim=16, cross_attn_invariant=inv,
                self_attn_invariant=inv, embedding_type="rff", embedding_freq_multiplier=(0.05, 0.1),
                condition_value_transform=True, use_gaussian_window=True)
    args.update(kw)
    return EquivariantCrossAttentionNeF(**args)


def test_model_constructor_errors():
    with pytest.raises(ValueError, match="Unknown embedding type"):
        _nef(embedding_type="siren")
    with pytest.raises(NotImplementedError):
        _nef(embedding_type="polynomial")
    assert len(_nef(num_layers=2).init(0, device="cpu")["params"]) == 5          # + self_attention_blocks_0/1 (NEF:137-167)
    assert _nef(num_layers=1, num_hidden=32)._Dp == 64                             # layers also run zero-padded (tests/test_gpu_layers.py)
    with pytest.raises(NotImplementedError):
        _nef(condition_value_transform=False)                                      # no shipped config; DESIGN.md 7
    with pytest.raises(AssertionError):
        _nef(num_hidden=63)


def test_param_tree_matches_oracle_tree_and_count():
    from enf_pde_amd.enf.models import TENSOR_PATHS
    nef = _nef()
    prm = nef.init(0, device="cpu")
    ts = nef.param_tensors(prm)
    assert sum(t.numel() for t in ts) == 531585
    assert [tuple(t.shape) for t in ts] == nef._expected_shapes()
    ref = R.init_params(0, dict(num_hidden=128, num_heads=2, latent_dim=16, num_out=1, invariant="rel_pos_periodic",
                                embedding_freq_multiplier=(0.05, 0.1)))
    for path in TENSOR_PATHS:           # same names, same shapes as the Flax tree restated by the oracle
        node = ref["params"]
        for k in path:
            node = node[k]
        t = prm["params"]
        for k in path:
            t = t[k]
        assert tuple(node.shape) == tuple(t.shape), path
    # initialiser statistics: RFF coefficients ~ N(0, std^2), LN scale 1, Dense bias 0
    q = prm["params"]["cross_attention_blocks_0"]["attn"]["invariant_embedding_query"]
    assert abs(q["encoding"]["coefficients"].std().item() - 0.05) < 0.01
    assert abs(q["layers_0"]["linear"]["kernel"].std().item() - (2 / 128) ** 0.5) < 0.01
    assert abs(q["linear_final"]["kernel"].abs().max().item() - (6 / 128) ** 0.5) < 0.01
    assert torch.all(prm["params"]["latent_stem"]["bias"] == 0)
    loaded = nef.load_params(ref, device="cpu")
    assert torch.allclose(loaded["params"]["out_proj"]["layers_4"]["kernel"].double(),
                          torch.tensor(ref["params"]["out_proj"]["layers_4"]["kernel"]), atol=1e-7)


def test_apply_without_gpu_fails_loudly():
    from enf_pde_amd import _lib
    nef = _nef()
    prm = nef.init(0, device="cpu")
    x, p, a, s = torch.zeros(1, 8, 2), torch.zeros(1, 4, 2), torch.ones(1, 4, 16), torch.ones(1, 4, 1)
    with pytest.raises(_lib.EnfError, match="no CPU path"):
        nef.apply(prm, x, p, a, s)


def test_latent_containers_match_reference_init():
    from enf_pde_amd.enf.latents.autodecoder_meta import PositionOrientationFeatureAutodecoderMeta
    from enf_pde_amd.enf.latents.autodecoder import PositionOrientationFeatureAutodecoder
    ad = PositionOrientationFeatureAutodecoderMeta(num_signals=1, num_latents=64, latent_dim=16, num_pos_dims=2,
                                                   num_ori_dims=0, gaussian_window_size=-1, coordinate_system="cartesian")
    P = ad.init(device="cpu")
    ref = R.init_latents(1, 64, 16, "rel_pos_periodic")
    assert set(P["params"]) == {"p_pos", "a", "gaussian_window"}
    for k in ref:
        assert np.allclose(P["params"][k].numpy(), ref[k], atol=1e-6), k
    p, a, w = ad.apply(P)
    assert p.shape == (1, 64, 2) and a.shape == (1, 64, 16) and w.shape == (1, 64, 1)
    pol = PositionOrientationFeatureAutodecoder(3, 128, 32, 2, 0, coordinate_system="polar").init(device="cpu")
    refp = R.init_latents(3, 128, 32, "latitude_periodic", coordinate_system="polar")
    assert np.allclose(pol["params"]["p_pos"].numpy(), refp["p_pos"], atol=1e-6)
    assert np.allclose(pol["params"]["gaussian_window"].numpy(), refp["gaussian_window"], atol=1e-6)
    ball = PositionOrientationFeatureAutodecoder(2, 25, 32, 4, 0, coordinate_system="ball").init(device="cpu")     # config_ihc.yaml
    refb = R.init_latents(2, 25, 32, "ball", coordinate_system="ball", num_in=3)
    assert ball["params"]["p_pos"].shape == (2, 25, 4)
    assert np.allclose(ball["params"]["p_pos"].numpy(), refb["p_pos"], atol=1e-5)
    assert np.allclose(ball["params"]["gaussian_window"].numpy(), 1.0)
    pon = PositionOrientationFeatureAutodecoderMeta(1, 16, 8, 2, 1, gaussian_window_size=-1).init(device="cpu")
    assert np.allclose(pon["params"]["p_ori"].numpy(), R.init_latents(1, 16, 8, "ponita")["p_ori"], atol=1e-6)
    pp, _, _ = PositionOrientationFeatureAutodecoderMeta(1, 16, 8, 2, 1, gaussian_window_size=-1).apply(pon)
    assert pp.shape == (1, 16, 3)
    nowin = PositionOrientationFeatureAutodecoderMeta(1, 16, 8, 2, 0, gaussian_window_size=None)
    assert nowin.apply(nowin.init(device="cpu"))[2] is None                      # ADM:21-24


def test_get_model_pde_and_fit_helpers():
    from enf_pde_amd.fitting import get_model_pde, make_masks, default_meta_sgd_lrs, shard_range
    cfg = NS(nef=NS(num_in=2, num_out=1, num_layers=0, num_hidden=128, num_heads=2, condition_value_transform=True,
                    latent_dim=16, num_latents=64, use_gaussian_window=True, embedding_type="rff",
                    embedding_freq_multiplier_invariant=0.05, embedding_freq_multiplier_value=0.1,
                    invariant_type="rel_pos_periodic"))
    nef, ode = get_model_pde(cfg)
    assert ode is None and nef.cross_attn_invariant.num_z_pos_dims == 2 and nef.cross_attn_invariant.num_z_ori_dims == 0
    m = make_masks(100, 30, 3, generator=torch.Generator().manual_seed(0), device="cpu")
    assert m.shape == (30, 4) and all(len(set(m[:, j].tolist())) == 30 for j in range(4))
    lrs = default_meta_sgd_lrs(16, device="cpu", with_ori=True)
    assert lrs["a"].shape == (16,) and lrs["p_pos"].shape == (1,) and set(lrs) == {"p_pos", "a", "gaussian_window", "p_ori"}
    for n, w in ((16, 8), (17, 4), (3, 8), (64, 1)):
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1


def test_checkpoint_roundtrip(tmp_path):
    """flat .npz of the Flax-named tree: save_tree / load_tree round-trip, and the 46 tensor paths survive."""
    import numpy as np
    import torch
    from oracle import enf_ref_np as R
    from tests.helpers import make_cfg
    from enf_pde_amd.checkpoint import save_tree, load_tree, flatten_tree
    from enf_pde_amd.enf.models import TENSOR_PATHS
    cfg = make_cfg("ponita", D=64, H=2, C=8, O=2)
    prm = R.init_params(0, cfg, jitter=0.1)
    path = tmp_path / "nef.npz"
    save_tree(path, prm)
    back = load_tree(path)
    flat_a, flat_b = flatten_tree(prm), flatten_tree(back)
    assert set(flat_a) == set(flat_b) and len(flat_a) == len(TENSOR_PATHS)
    for k in flat_a:
        np.testing.assert_allclose(np.asarray(flat_a[k], dtype=np.float32), flat_b[k].numpy(), rtol=0, atol=0)
    for p in TENSOR_PATHS:
        assert "params/" + "/".join(p) in flat_b
    lat = {"params": {"p_pos": torch.zeros(2, 4, 2), "a": torch.ones(2, 4, 8), "gaussian_window": torch.full((2, 4, 1), 0.5)}}
    save_tree(tmp_path / "lat.npz", lat)
    assert torch.equal(load_tree(tmp_path / "lat.npz")["params"]["a"], lat["params"]["a"])


def test_ode_host_pieces_on_cpu():
    """Device-agnostic parts of the latent ODE (invariants over (p, p), polynomial features, solvers) against the oracle;
    the fused convolution has no CPU path."""
    import torch
    from oracle import ode_ref_np as O
    from enf_pde_amd.enf.steerable_attention.invariant import get_sa_invariant
    from enf_pde_amd.fitting.ode_models import PolynomialFeatures, sep_gconv
    from enf_pde_amd.fitting.trainers.trainer_utils import solve_latent_ode
    from enf_pde_amd.fitting import get_model_pde
    from tests.test_ode_oracle import ode_cfg, ode_inputs
    for name in ("rel_pos_periodic", "ponita", "polar_periodic", "latitude_periodic", "rel_pos", "norm_rel_pos", "abs_pos",
                 "ball", "ball_lat"):
        cfg = ode_cfg(name)
        p = ode_inputs(cfg, 2, 5, 3, 1)[0]
        inv = get_sa_invariant(NS(invariant_type=name, num_in=cfg["num_in"]))
        pe = np.concatenate([p[..., :2], np.cos(p[..., 2:]), np.sin(p[..., 2:])], -1) if name == "ponita" else p
        got = inv(torch.tensor(pe), torch.tensor(pe)).numpy()
        assert got.shape[-1] == inv.dim and np.abs(got - O.sa_invariant(name, pe)).max() < 1e-12, name
    x = np.random.default_rng(0).standard_normal((3, 4, 4))
    poly = PolynomialFeatures(3)
    assert poly.num_features(4) == 340 and np.allclose(poly(torch.tensor(x)).numpy(), O.poly_features(x, 3))
    f = lambda z, t: (-z[0], 2.0 * z[1], torch.zeros_like(z[2]))
    x0 = (torch.ones(2, 3, 2), torch.ones(2, 3, 4), torch.full((2, 3, 1), 0.7))
    fn = lambda z, t: (-z[0], 2.0 * z[1], np.zeros_like(z[2]))
    for method in ("euler", "rk4"):
        got = solve_latent_ode(f, x0, 0, 4, 0.5, method=method)
        ref = O.solve_latent_ode(fn, tuple(v.numpy().astype(np.float64) for v in x0), 0, 4, 0.5, method=method)
        assert all(np.allclose(g.numpy(), r, rtol=1e-5) for g, r in zip(got, ref)) and got[0].shape == (2, 9, 3, 2)
    with pytest.raises(ValueError, match="Unknown method"):
        solve_latent_ode(f, x0, 0, 1, 0.5, method="heun")
    xg = x0[0].clone().requires_grad_(True)                       # stop_gradient: each step starts from a detached state
    solve_latent_ode(f, (xg, x0[1], x0[2]), 0, 2, 0.5, method="euler", stop_gradient=True)[0][:, -1].sum().backward()
    assert xg.grad is None or not xg.grad.any()
    with pytest.raises(RuntimeError, match="HIP"):
        sep_gconv(torch.zeros(1, 4, 16), torch.zeros(1, 4, 4, 16), torch.zeros(16, 16))
    cfg = NS(nef=NS(num_in=2, num_out=1, num_layers=0, num_hidden=128, num_heads=2, condition_value_transform=True,
                    latent_dim=16, num_latents=64, use_gaussian_window=True, embedding_type="rff",
                    embedding_freq_multiplier_invariant=0.05, embedding_freq_multiplier_value=0.1, invariant_type="rel_pos_periodic"),
             node=NS(name="ponita", num_layers=3, num_hidden=128, widening_factor=2, kernel_size="global", degree=3, basis_dim=64))
    nef, ode = get_model_pde(cfg)
    assert type(ode).__name__ == "PonitaODEGen" and ode.ponita.num_hidden == 128 and ode.ponita.basis_dim == 64
    cfg.node.name = "mlp"
    assert type(get_model_pde(cfg)[1]).__name__ == "MLPODE"
    cfg.node.name = "gru"
    with pytest.raises(ValueError, match="Unknown ODE model"):
        get_model_pde(cfg)


def test_trainer_phase_schedule():
    """_base_pde_trainer.py:280-299 with config_navier_stokes.yaml's epochs (nef 0..500, ode 500..2000)."""
    from enf_pde_amd.fitting.trainers import MetaSGDPDETrainer
    tr = MetaSGDPDETrainer.__new__(MetaSGDPDETrainer)
    tr.config = NS(training=NS(nef=NS(train_from_epoch=0, train_until_epoch=500), ode=NS(train_from_epoch=500, train_until_epoch=2000)))
    tr.ode_model = object()
    assert tr.select_train_step(2000) == tr.ode_train_step and tr.select_train_step(501) == tr.ode_train_step
    assert tr.select_train_step(1).__name__ == "<lambda>" and tr.select_train_step(500).__name__ == "<lambda>"    # nef phase
    with pytest.raises(ValueError, match="No training step set"):
        tr.select_train_step(0)                                    # the reference starts counting epochs at 1
    with pytest.raises(ValueError):
        tr.select_train_step(2001)
    tr.config.training.nef.train_until_epoch = 800
    assert tr.select_train_step(600) == tr.dual_train_step
    tr.ode_model = None
    assert tr.select_train_step(600).__name__ == "<lambda>"
    import torch
    traj = torch.arange(2 * 10 * 4 * 4, dtype=torch.float32).reshape(2, 10, 4, 4, 1)
    st = NS(rng=torch.Generator().manual_seed(0))
    assert torch.equal(tr._nef_frames(st, traj), traj[:, 0])                       # fit_on_num_steps absent -> 1
    tr.config.training.nef.fit_on_num_steps = 2                                    # config_ihc.yaml
    tr.config.dataset = NS(traj_len_train=6)
    fr = tr._nef_frames(st, traj)
    assert fr.shape == (4, 4, 4, 1)
    ids = {int(f[0, 0, 0]) // 16 % 10 for f in fr}
    assert len(ids) == 2 and max(ids) < 6                                          # two distinct training frames


def test_bench_refuses_a_world_that_is_not_gpus():
    """bench.py --gpus N must run N ranks or fail (it never reports an N-GPU line from another number of ranks)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, bench, "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and not r.stdout.strip()
    import torch
    if torch.cuda.device_count() < 2:        # no torchrun environment: it would spawn 2 ranks itself, but there are not 2 GPUs
        r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "GPU(s)" in r.stderr and not r.stdout.strip()


def test_bench_configs_cover_baseline_json():
    import json
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert sorted(bench.CONFIGS) == list(range(1, len(base["configs"]) + 1))
    c2 = bench.CONFIGS[bench.HEADLINE]
    assert (c2["Z"], c2["grid"], c2["D"], c2["H"]) == (64, (64, 64), 128, 2)          # the metric's "64 latents x 64^2 grid"
    assert bench.pair_flops(c2) == 494080 and bench.points_per_signal(c2) == 4 * 512 + 4096
    assert bench.points_per_signal(bench.CONFIGS[5]) == 4 * 512 + 41 * 256 * 256
    for c in bench.CONFIGS.values():
        lat = bench._Autodecoder(c).init(device="cpu")["params"]
        assert lat["p_pos"].shape == (1, c["Z"], 2) and lat["a"].shape == (1, c["Z"], c["C"])


def test_train_state_checkpoint_keeps_dtypes_and_rng(tmp_path):
    """checkpoint.save_train_state / load_train_state on a hand-made state (CPU): integer counters stay integers, tensors keep
    their dtype, the generator resumes its sequence, a mismatching template is refused."""
    from enf_pde_amd import checkpoint as ck
    from enf_pde_amd.fitting.trainers.pde_trainer import TrainState
    g = torch.Generator().manual_seed(5)
    mk = lambda *s: torch.randn(*s, generator=g)
    params = {"nef": {"params": {"w": mk(3, 4), "b": mk(4)}}, "meta_sgd_lrs": {"a": mk(8)},
              "autodecoder": {"params": {"a": mk(1, 5, 8)}}}
    opt = lambda ts: {"count": 11, "mu": [mk(*t.shape) for t in ts], "nu": [mk(*t.shape).abs() for t in ts]}
    st = TrainState(params=params, nef_opt_state=opt([params["nef"]["params"]["b"], params["nef"]["params"]["w"]]),
                    autodecoder_opt_state=opt([params["autodecoder"]["params"]["a"]]), meta_sgd_opt_state=opt([params["meta_sgd_lrs"]["a"]]),
                    ode_opt_state=None, step=11, rng=g)
    path = str(tmp_path / "state.npz")
    ck.save_train_state(path, st, config=NS(meta=NS(num_inner_steps=3), name="x"), epoch=4)
    nxt = torch.randn(6, generator=g)                                   # what the generator yields after the save point
    zero = lambda ts: {"count": 0, "mu": [torch.zeros_like(t) for t in ts], "nu": [torch.zeros_like(t) for t in ts]}
    tmpl = TrainState(params={k: ck.unflatten_tree({n: torch.zeros_like(v) for n, v in ck.flatten_tree(params[k]).items()}) for k in params},
                      nef_opt_state=zero(st.nef_opt_state["mu"]), autodecoder_opt_state=zero(st.autodecoder_opt_state["mu"]),
                      meta_sgd_opt_state=zero(st.meta_sgd_opt_state["mu"]), ode_opt_state=None)
    new, epoch, conf = ck.load_train_state(path, tmpl)
    assert epoch == 4 and new.step == 11 and conf == {"meta": {"num_inner_steps": 3}, "name": "x"}
    assert type(new.nef_opt_state["count"]) is int and new.nef_opt_state["count"] == 11
    for k, v in ck.flatten_tree(params).items():
        assert torch.equal(ck.flatten_tree(new.params)[k], v)
    assert all(torch.equal(a, b) for a, b in zip(new.meta_sgd_opt_state["nu"], st.meta_sgd_opt_state["nu"]))
    assert torch.equal(torch.randn(6, generator=new.rng), nxt)
    with np.load(path) as z:
        assert z["nef_opt_state/count"].dtype == np.int64 and z["rng_state"].dtype == np.uint8
    tmpl.params["nef"]["params"]["extra"] = torch.zeros(1)
    with pytest.raises(ValueError):
        ck.load_train_state(path, tmpl)
    # load_tree no longer casts integer entries to float
    ck.save_tree(str(tmp_path / "t.npz"), {"count": torch.tensor(7), "w": torch.ones(2, dtype=torch.float64)})
    tr = ck.load_tree(str(tmp_path / "t.npz"))
    assert tr["count"].dtype == torch.int64 and tr["w"].dtype == torch.float32


def test_emitted_kernels_carry_no_known_unsafe_instruction_form(tmp_path):
    """Emitted-code guards over EVERY kernel source of the library (the Makefile's SRCS), one `hipcc -S` each:
    (1) the instruction form behind the "K3 run-to-run deviations" (scripts/k3_race/README.md): `v_pk_add_f32` whose second operand
        is the HIGH register of a pair broadcast by `op_sel:[0,1]` and negated -- what hipcc's SLP vectoriser makes of a plain
        `x = (x - mu) * rstd` loop.  The LayerNorm applies go through ln_apply (scalar asm fmas) instead; a new loop of that shape, or
        a compiler that packs differently, would bring the form back without any parity test noticing (it misbehaved about once
        per 10^5 executions).  The other packed operand forms the kernels contain were run in the same context and did not deviate
        where the controls did (README, "Packed-operand forms"), so only this form is rejected;
    (2) no vector instruction inside an inline-asm block (invisible to hipcc's hazard recognizer) feeds a v_mfma operand with fewer
        than the two wait states gfx950 needs, and no asm transcendental feeds a VALU without one (scripts/check_mfma_hazards.py,
        rule 2; scripts/ubench/valu_mfma_hazard.hip, trans_hazard.hip) -- fp32 instantiations included, whose MFMA operands ARE the
        registers ln_apply writes."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "enf-pde_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^SRCS\s*=\s*(.+)$", mk, re.M).group(1).split()
    assert len(srcs) >= 11 and "enf_pair_bwd" in srcs and "enf_tail" in srcs
    procs = {}
    for f in srcs:
        out = str(tmp_path / (f + ".s"))
        procs[f] = (out, subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unknown-pragmas",
                                           "--cuda-device-only", "-S", os.path.join(csrc, f + ".hip"), "-o", out],
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    form = re.compile(r"v_pk_add_f32\b.*op_sel:\[0,1\].*neg_lo:\[0,1\]")
    for f, (out, pr) in procs.items():
        _, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, err.decode()[-2000:]
        text = open(out).read()
        has_kernels = "__global__" in open(os.path.join(csrc, f + ".hip")).read()
        assert ("s_endpgm" in text) == has_kernels                          # (really the device code; enf_api.hip is host-only)
        hits = [ln.strip() for ln in text.splitlines() if form.search(ln)]
        assert not hits, f"{f}: {len(hits)} x the failing form, e.g. {hits[0]}"
        chk = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_mfma_hazards.py"), out], capture_output=True, text=True)
        assert chk.returncode == 0, chk.stderr[-1000:]
        asm_hits = [ln for ln in chk.stdout.splitlines() if " asm " in ln]
        assert not asm_hits, f"{f}: {asm_hits[:3]}"
