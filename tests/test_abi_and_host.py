"""CPU (no GPU): the C-ABI library loads and exports every symbol include/dgvit_hip.h declares, size queries and
argument validation work without a device, and the host modules mirror the reference's nn.Module surface."""
import ctypes
import os
import re

import pytest
import torch

from helpers import O, ROOT

HEADER = os.path.join(ROOT, "include", "dgvit_hip.h")
DIAG_HEADER = os.path.join(ROOT, "include", "dgvit_hip_diag.h")


@pytest.fixture(scope="module")
def amd():
    import __graft_entry__
    __graft_entry__.build()          # hipcc cross-compiles gfx950 without a GPU; no-op when up to date
    import dgvit_amd
    return dgvit_amd


def _header_functions(path=HEADER):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(dgvit_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if n != "dgvit_config"))


def test_library_exports_every_header_symbol(amd):
    from dgvit_amd import _lib
    lib = amd.load_library()
    declared = _header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"libdgvit_hip.so does not export {name}"
        assert name in _lib.SIGNATURES, f"ctypes binding has no signature for {name}"
    assert sorted(_lib.SIGNATURES) == declared, "binding and header disagree"
    assert lib.dgvit_abi_version() == 7
    # the product library has no setters and none of the diagnostic entry points
    assert not [n for n in declared if n.startswith("dgvit_set_")]
    for name in _header_functions(DIAG_HEADER):
        assert not hasattr(lib, name), f"libdgvit_hip.so exports the diagnostic entry point {name}"


def test_diagnostic_library_exports_both_headers(amd):
    from dgvit_amd import _lib
    extra = _header_functions(DIAG_HEADER)
    assert sorted(_lib.DIAG_SIGNATURES) == extra and len(extra) >= 10, "diagnostic binding and header disagree"
    with amd.diagnostic_library() as dlib:
        for name in _header_functions() + extra:
            assert hasattr(dlib, name), f"libdgvit_hip_diag.so does not export {name}"
        assert _lib.load() is dlib
    assert _lib.load() is amd.load_library() and _lib.load() is not dlib


def test_product_kernels_do_not_spill(amd):
    """build.py's register audit: no product kernel uses scratch memory or spills VGPRs (the build fails otherwise; this re-reads
    the per-kernel figures the compiler reported)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dgvit_build", os.path.join(ROOT, "dgvit-depth-goal-guided-vision-transformer-_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    kernels, bad = b.audit()
    assert len(kernels) > 100 and not bad, [k["name"] for k in bad]
    assert not b.SPILL_ALLOW
    # the experiments that live in the diagnostic library only, by exact kernel name: the product's
    # attn_fwd_pipe_kernel is NOT one of them
    leaked = [k["name"] for k in kernels if "gemm_f32_pipe_kernel" in k["name"] or "frame_attn" in k["name"]
              or "frame_mlp" in k["name"] or "frame_final" in k["name"]]
    assert not leaked, f"experiments leaked into the product library: {leaked}"


def test_size_queries_and_validation_without_gpu(amd):
    from dgvit_amd._lib import dgvit_config
    lib = amd.load_library()
    cfg = dgvit_config(84, 84, 12, 12, 256, 6, 8, 64, 2048)   # (pool_mean and flags default to 0)
    train = lib.dgvit_got_workspace_floats(ctypes.byref(cfg), 512, 1)
    infer = lib.dgvit_got_workspace_floats(ctypes.byref(cfg), 512, 0)
    assert train > infer > 0
    # per layer: ln1, xmid, ln2, xout (T*D each) + qkv (3I) + ao (I) + h1, a1 (M each) + 4 stat rows
    T, D, I, M = 512 * 50, 256, 512, 2048
    per_layer = T * (4 * D + 4 * I + 2 * M + 4) + 512 * 8 * 50
    assert train >= 6 * per_layer
    assert lib.dgvit_got_backward_scratch_floats(ctypes.byref(cfg), 512) > T * M
    # reference assertion text for indivisible images (GoalFormer.py:131)
    bad = dgvit_config(84, 84, 16, 20, 256, 6, 8, 64, 2048)
    assert lib.dgvit_got_workspace_floats(ctypes.byref(bad), 4, 1) < 0
    assert b"divisible by the patch size" in lib.dgvit_last_error()
    bad = dgvit_config(84, 84, 12, 12, 256, 6, 8, 48, 2048)
    assert lib.dgvit_got_workspace_floats(ctypes.byref(bad), 4, 1) < 0
    assert b"dim_head" in lib.dgvit_last_error()
    bad = dgvit_config(224, 224, 8, 8, 256, 6, 8, 64, 2048)   # 785 tokens > fused-attention limit
    assert lib.dgvit_got_workspace_floats(ctypes.byref(bad), 4, 1) < 0
    assert lib.dgvit_linear_backward_scratch_floats(512, 128, 256) > 0
    assert lib.dgvit_gemm_scratch_floats(2, 1536, 256, 25600) >= 1536 * 256
    # forward form: scratch only when the shape takes the in-launch split-K path (few tiles / nearly empty last round)
    assert lib.dgvit_gemm_scratch_floats(0, 25600, 2048, 256) == 0           # 6400 tiles = 25.00 per CU: no split
    assert lib.dgvit_gemm_scratch_floats(0, 2080, 64, 2048) >= 33 * 64 * 64  # 33 tiles of 64 x 64, K = 2048: split
    assert lib.dgvit_gemm_scratch_floats(0, 25600, 256, 2048) >= 64 * 4 * 64 * 64   # 1600 tiles: the last 64 are cut in 4


def test_no_cpu_fallback(amd):
    m = amd.GoTPolicy(2, 2, 1, 2, 64)
    with pytest.raises(amd.DgvitError, match="no CPU fallback"):
        m([torch.zeros(1, 128, 160), torch.zeros(1, 2)])
    with pytest.raises(amd.DgvitError):
        amd.functional.linear(torch.zeros(2, 4), torch.zeros(3, 4))
    with pytest.raises(RuntimeError, match="not callable"):
        m.trans.transformer(torch.zeros(1))


@pytest.mark.parametrize("kind,heads", [("policy", 4), ("qnet", 4), ("detpolicy", 4), ("policy", 1), ("qnet", 1)])
def test_state_dict_abi_matches_reference_keys(amd, kind, heads):
    """Keys, order and shapes equal the reference's (pinned by make_golden.py's strict load into the reference); heads == 1 with
    l_f_size == 64 is the projection-less Attention (to_out = nn.Identity(): no to_out keys, GoalFormer.py:56,66-69)."""
    cfg = O.GoTConfig(dim=64, depth=4, heads=heads)
    assert cfg.project_out == (heads != 1)
    ctor = {"policy": amd.GoTPolicy, "qnet": amd.GoTQNetwork, "detpolicy": amd.DeterministicGoTPolicy}[kind]
    spec = {"policy": O.policy_param_spec, "qnet": O.qnet_param_spec, "detpolicy": O.detpolicy_param_spec}[kind](cfg)
    m = ctor(2, 2, cfg.depth, cfg.heads, cfg.dim)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, s) for k, s, _ in spec]
    assert [k for k, _ in m.named_parameters()] == [k for k, _, _ in spec]
    m.load_state_dict(O.make_params(spec, 0), strict=True)
    # survives deepcopy / pickling / .to(), like DRL.py:169 and attention_imitating.py:199 need
    import copy, io
    m2 = copy.deepcopy(m)
    buf = io.BytesIO()
    torch.save(m2, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))
    assert m.to("cpu") is m
    for attr in ("trans", "fc_embed", "fc1", "fc2"):
        assert hasattr(m, attr)
    table = m.trans.param_table()
    assert len(table) == 4 + 11 * cfg.depth and sum(t is None for t in table) == (0 if cfg.project_out else 2 * cfg.depth)


def test_init_follows_reference(amd):
    """Xavier-uniform(gain 1) on Linear weights (got_sac_network.py:30-33); RMSNorm gain 1; mean/log_std of the
    deterministic policy are created after the Xavier pass and keep nn.Linear's default init (:410-413)."""
    torch.manual_seed(0)
    m = amd.GoTPolicy(2, 2, 2, 2, 64)
    w = m.trans.transformer.layers[0][1].fn.net[0].weight
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5
    assert w.abs().max().item() <= bound + 1e-6 and w.abs().max().item() > 0.9 * bound
    assert torch.equal(m.trans.layer_norm.g, torch.ones(64))
    d = amd.DeterministicGoTPolicy(2, 2, 1, 2, 64)
    assert d.mean_linear.weight.abs().max().item() <= (1.0 / 32) ** 0.5 + 1e-6
    assert m.trans.pos_embedding.shape == (1, 65, 64) and m.trans.cls_token.shape == (1, 1, 64)
    m84 = amd.GoTPolicy(2, 2, 1, 2, 64, image_size=(84, 84), patch_size=(12, 12))
    assert m84.trans.pos_embedding.shape == (1, 50, 64) and m84.trans.to_patch_embedding[1].weight.shape == (64, 144)
    with pytest.raises(AssertionError, match="divisible by the patch size"):
        amd.GoT(image_size=(84, 84), patch_size=(16, 20), num_classes=2, dim=64, depth=1, heads=2, mlp_dim=64)


def test_drop_in_module_names(amd):
    from dgvit_amd.GoalFormer import GoT
    from dgvit_amd.got_sac_network import GoTPolicy, GoTQNetwork, DeterministicGoTPolicy, weights_init_
    assert GoT is amd.GoT and GoTPolicy is amd.GoTPolicy and GoTQNetwork is amd.GoTQNetwork
    assert DeterministicGoTPolicy is amd.DeterministicGoTPolicy and callable(weights_init_)


def test_bf16_size_queries_and_host_switch_without_gpu(amd):
    """bf16 configuration: arena / workspace / scratch sizes follow the documented layout; unsupported shapes are refused with a
    message; GoT.set_compute_dtype is pure host state and leaves parameters and state_dict untouched."""
    from dgvit_amd._lib import dgvit_config
    lib = amd.load_library()
    cfg = dgvit_config(224, 224, 16, 16, 768, 12, 12, 64, 3072)
    D, I, M, L, pd, B, N = 768, 768, 3072, 12, 256, 64, 197
    per_layer = 2 * (3 * I * D + D * I + M * D + D * M)          # the four GEMM weights and their transposes
    assert lib.dgvit_got_bf16_weight_elems(ctypes.byref(cfg)) == D * pd + L * per_layer
    T = B * N
    infer = lib.dgvit_got_bf16_workspace_bytes(ctypes.byref(cfg), B, 0)
    train = lib.dgvit_got_bf16_workspace_bytes(ctypes.byref(cfg), B, 1)
    assert infer >= T * (2 * D * 2 + 3 * I * 2 + I * 2 + M * 2 + 3 * D * 4)          # one layer's operands + residual buffers
    assert train >= L * T * (2 * D * 2 + 3 * I * 2 + I * 2 + 2 * M * 2 + 2 * D * 4)    # every layer keeps its activations
    assert train > infer
    assert lib.dgvit_got_bf16_backward_scratch_bytes(ctypes.byref(cfg), B) > T * (3 * I + M) * 2
    assert lib.dgvit_wgrad_bf16_scratch_floats(768, 3072, T) >= 768 * 3072
    bad = dgvit_config(84, 84, 12, 12, 256, 6, 8, 32, 2048)      # dim_head 32: fp32 path only
    assert lib.dgvit_got_bf16_workspace_bytes(ctypes.byref(bad), 4, 0) < 0
    assert b"dim_head" in lib.dgvit_last_error()
    bad = dgvit_config(84, 84, 12, 12, 100, 2, 2, 64, 2048)      # dim not a multiple of 8
    assert lib.dgvit_got_bf16_weight_elems(ctypes.byref(bad)) < 0
    m = amd.GoT(image_size=(84, 84), patch_size=(12, 12), num_classes=2, dim=64, depth=1, heads=2, mlp_dim=64, channels=1)
    keys = list(m.state_dict().keys())
    assert m.compute_dtype == torch.float32
    assert m.set_compute_dtype(torch.bfloat16) is m and m.compute_dtype == torch.bfloat16
    assert list(m.state_dict().keys()) == keys and all(p.dtype == torch.float32 for p in m.parameters())
    with pytest.raises(ValueError):
        m.set_compute_dtype(torch.float16)
    with pytest.raises(amd.DgvitError):                           # no CPU fallback in the bf16 configuration either
        m(torch.rand(2, 84, 84), torch.rand(2, 64))
    import copy
    assert copy.deepcopy(m).compute_dtype == torch.bfloat16


def test_integration_md_binding_snippet_matches_the_library(amd):
    """INTEGRATION.md's C-ABI snippet is what a host author copies: its dgvit_config must have the library's size (round 1
    documented 9 of the 10 fields, which makes the library read pool_mean past the end of the caller's struct), its
    argtypes must agree with the in-tree binding, and its example constructor call must fill every field."""
    import ctypes
    import re
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class dgvit_config\(ctypes\.Structure\):\n\s+_fields_ = \[\(n, ctypes\.c_int\) for n in \(([^)]*)\)\]", text)
    assert m, "INTEGRATION.md no longer holds the dgvit_config snippet"
    names = [n.strip().strip('"') for n in m.group(1).split(",") if n.strip()]
    from dgvit_amd import _lib
    assert names == [f[0] for f in _lib.dgvit_config._fields_]
    lib = amd.load_library()
    assert lib.dgvit_config_size() == 4 * len(names) == ctypes.sizeof(_lib.dgvit_config)
    call = re.search(r"cfg = dgvit_config\(([^)]*)\)", text)
    assert call and len(call.group(1).split(",")) == len(names), "the example dgvit_config(...) call must pass every field"
    assert f"dgvit_abi_version() == {_lib.ABI_VERSION}" in text
