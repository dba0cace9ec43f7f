"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol the header
declares; module structure / seeded init / checkpoint keys; CLI surface; loud failures."""
import ctypes
import json
import os
import re

import pytest
import torch

from discogan_modernized_amd import _lib, losses, model, ops
from discogan_modernized_amd import image_translation as it
from discogan_modernized_amd import distributed_image_translation as dit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "discogan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/discogan_hip.h but not exported"
    assert sorted(_lib.SIGNATURES.keys()) == syms, "ctypes SIGNATURES and the header disagree"
    L = _lib.load()
    assert L.dg_version() >= 100
    assert L.dg_loss_workspace_bytes() > 0
    assert L.dg_conv_workspace_bytes(0, 2, 8, 8, 512, 1024, 2, 1) > 0      # split-K plan, no GPU needed
    assert L.dg_set_option(b"nonsense", 1) != 0 and b"unknown option" in L.dg_last_error()


def test_grouped_entry_points_validate_their_arguments():
    """The *_g entry points reject bad groups / tables / precision before anything is launched (no GPU needed): negative status,
    message from dg_last_error, nothing thrown across the ABI."""
    import ctypes as C
    L = _lib.load()
    P = C.c_void_p
    one = (P * 1)(None)
    tab4 = (P * 4)(8, 8, 8, 8)                      # non-null dummies: validation fails before they are dereferenced
    # groups out of range
    assert L.dg_conv_fwd_g(0, tab4, tab4, tab4, 2, 8, 8, 64, 128, 2, 1, 0, 1, None, 0, tab4, 0, None) < 0 and b"groups" in L.dg_last_error()
    assert L.dg_conv_fwd_g(5, tab4, tab4, tab4, 2, 8, 8, 64, 128, 2, 1, 0, 1, None, 0, tab4, 0, None) < 0
    # precision out of range
    assert L.dg_conv_fwd_g(2, tab4, tab4, tab4, 2, 8, 8, 64, 128, 2, 1, 7, 1, None, 0, tab4, 0, None) < 0 and b"prec" in L.dg_last_error()
    # null problem pointer / null table
    assert L.dg_conv_fwd_g(1, one, tab4, tab4, 2, 8, 8, 64, 128, 2, 1, 0, 1, None, 0, tab4, 0, None) < 0 and b"null" in L.dg_last_error()
    assert L.dg_conv_dgrad_g(2, None, tab4, tab4, 2, 8, 8, 64, 128, 2, 1, 0, 1, None, 0, tab4, 0, None) < 0
    # share: only the weight gradient, groups % share == 0, members naming ONE output
    assert L.dg_conv_wgrad_g(3, 2, tab4, tab4, tab4, 2, 8, 8, 64, 128, 2, 1, 0, 1, 1, tab4, 0, None) < 0 and b"share" in L.dg_last_error()
    mixed = (P * 4)(8, 16, 8, 8)
    assert L.dg_conv_wgrad_g(4, 2, tab4, tab4, mixed, 2, 8, 8, 64, 128, 2, 1, 0, 1, 1, tab4, 0, None) < 0 and b"same dw" in L.dg_last_error()
    # plan_groups is 1 or groups
    assert L.dg_conv_fwd_g(2, tab4, tab4, tab4, 2, 8, 8, 64, 128, 2, 1, 0, 3, None, 0, tab4, 0, None) < 0 and b"plan_groups" in L.dg_last_error()
    # unsupported geometry is reported the same way
    assert L.dg_conv_fwd_g(2, tab4, tab4, tab4, 2, 9, 8, 64, 128, 2, 1, 0, 1, None, 0, tab4, 0, None) < 0 and b"powers of two" in L.dg_last_error()
    # BatchNorm / loss tables
    assert L.dg_bn_act_fwd_g(0, tab4, tab4, 16, 64, tab4, tab4, tab4, 1, 0.2, None) < 0
    assert L.dg_bn_train_stats_g(2, 1, None, 16, 64, 1e-5, 0.1, None, None, None, tab4, tab4, 0, None) < 0
    assert L.dg_mse_fwd_g(2, tab4, None, 16, tab4, tab4, 8192, None) < 0 and b"null pointer table" in L.dg_last_error()
    assert L.dg_bce_fwd_g(9, tab4, 4, None, tab4, None) < 0
    assert L.dg_build_flags() == 0                  # the product library: no experiments, no timing switches


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdiscogan_hip.so")
    with pytest.raises(_lib.DiscoganHipError, match="no CPU fallback"):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    with pytest.raises(_lib.DiscoganHipError, match="no CPU fallback"):
        ops.conv_fwd(torch.zeros(2, 64, 8, 8), torch.zeros(64, 64, 4, 4), 2, 1)
    g = model.Generator(image_size=16)
    with pytest.raises(_lib.DiscoganHipError):
        g(torch.rand(2, 3, 16, 16))
    from discogan_modernized_amd import optim
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        optim.Adam(g.parameters())


@pytest.mark.parametrize("S", [16, 64])
def test_seeded_init_matches_fixture(S):
    """Same RNG consumption order and values as the oracle/reference construction (host side only)."""
    fix = json.load(open(os.path.join(GOLD, f"oracle_s{S}_n4.json")))
    torch.manual_seed(1234)
    nets = dict(gen_A=model.Generator(True, image_size=S), gen_B=model.Generator(True, image_size=S),
                dis_A=model.Discriminator(image_size=S), dis_B=model.Discriminator(image_size=S))
    for name, net in nets.items():
        assert list(net.state_dict().keys()) == fix["meta"]["state_dict_keys"][name]
        for k, v in net.state_dict().items():
            if v.dtype.is_floating_point:
                f = v.reshape(-1)
                ref = fix["init"][name][k]
                assert list(v.shape) == ref["shape"]
                idx = [int((i * 2654435761) % f.numel()) for i in range(1, 9)]
                assert [float(f[i]) for i in idx] == ref["samples"], f"{name}.{k}"


def test_reference_512_state_dict_contract():
    """Appendix B: 86 / 38 entries, parameter order and logical shapes of the reference checkpoints."""
    fix = json.load(open(os.path.join(GOLD, "ref_s512_n2.json")))
    keys_g, keys_d = fix["meta"]["state_dict_keys"]["gen_A"], fix["meta"]["state_dict_keys"]["dis_A"]
    assert len(keys_g) == 86 and len(keys_d) == 38
    ch = model.stage_channels(512)
    assert ch == [64, 128, 256, 512, 1024, 2048, 2048]
    # build on the meta device: no 2.7 GB allocation
    with torch.device("meta"):
        g, d = model.Generator(True), model.Discriminator()
    assert list(g.state_dict().keys()) == keys_g and list(d.state_dict().keys()) == keys_d
    for k, v in g.state_dict().items():
        if k in fix["init"]["gen_A"]:
            assert list(v.shape) == fix["init"]["gen_A"][k]["shape"], k
    assert sum(p.numel() for p in g.parameters()) == 230192968
    assert sum(p.numel() for p in d.parameters()) == 111852288
    assert g.main is None and isinstance(g.encoder, torch.nn.Sequential) and isinstance(g.decoder, torch.nn.Sequential)


def test_weight_memory_layout():
    g = model.Generator(image_size=16)
    w = g.encoder[2].weight                      # Conv2d(64,128): logical [128,64,4,4], memory KRSC
    assert tuple(w.shape) == (128, 64, 4, 4) and ops.is_krsc(w)
    assert g.encoder[0].weight.is_contiguous()  # 3-channel edge weight stays logical
    wt = g.decoder[0].weight                     # ConvTranspose2d(100,128): logical [100,128,4,4]
    assert tuple(wt.shape) == (100, 128, 4, 4) and ops.is_krsc(wt)
    sd = g.state_dict()
    g2 = model.Generator(image_size=16)
    g2.load_state_dict({k: v.contiguous() for k, v in sd.items()})
    assert ops.is_krsc(g2.encoder[2].weight) and torch.equal(g2.encoder[2].weight, w)


def test_cli_surface_matches_reference_defaults():
    a = it.parse_args([])
    expect = dict(device="cuda", task_name="facescrub", results_dir="./results/", models_dir="./models/",
                  model_arch="discogan", epochs=100, batch_size=64, learning_rate=0.0002, beta1=0.5, beta2=0.999,
                  image_size=64, gan_curriculum=10000, starting_rate=0.01, default_rate=0.5, style_A=None,
                  style_B=None, constraint=None, constraint_type=None, n_test=200, update_interval=3,
                  log_interval=50, image_save_interval=1000, model_save_interval=10000)
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    b = dit.parse_args(["--task_name", "celebA", "--style_A", "Male", "--style_B", "Smiling", "--batch_size", "64"])
    assert b.task_name == "celebA" and b.style_A == "Male" and b.distributed is False and b.local_rank == 0
    for flag in ("load_gen_A", "load_gen_B", "load_dis_A", "load_dis_B"):
        assert getattr(b, flag) is None
    with pytest.raises(SystemExit):
        it.parse_args(["--model_arch", "nope"])


def test_loss_helper_signatures():
    import inspect
    assert list(inspect.signature(losses.get_fm_loss).parameters)[:4] == ["real_feats", "fake_feats", "criterion", "device"]
    assert list(inspect.signature(losses.get_gan_loss).parameters)[:4] == ["dis_real", "dis_fake", "criterion", "device"]


def test_conv_planner_dispatch_at_512px_batch32():
    """Host-side planning only (no kernel runs): which conv kernel every stride-2 layer of the 512 px nets gets on the two fast
    matrix paths at batch 32 -- the dispatch DESIGN.md section 3.1 describes.  op 0 forward, 1 input-grad, 2 weight-grad."""
    from discogan_modernized_amd import _lib
    from discogan_modernized_amd.model import stage_channels
    L = _lib.load()
    ch = stage_channels(512)                 # 64, 128, 256, 512, 1024, 2048, 2048
    layers = [(ch[i - 1], ch[i], 512 >> i) for i in range(1, len(ch))]           # (C, K, H of the layer input)
    try:
        # the planning queries name their arithmetic themselves (round 4: no process-wide "bf16" option involved)
        # f32x3: plane kernels (igemm_dma_x3.hip; narrow input-grads: igemm_dma_x3_dgw.hip)
        for C, K, H in layers:
            got = [L.dg_conv_x3_planes_ok(op, 32, H, H, C, K, 2, 1) for op in (0, 1, 2)]
            # every forward has a plane kernel (>= 192 columns: 256 x 256 tile = 1; <= 128: the window forward kernel = 3, which reads the
            # transposed weight planes), every input-grad (>= 192 columns: 1; <= 128: the window kernel = 2, which takes its gradient
            # operand in the quad-chunk layout) and every weight-grad
            # (3 exists in the experiments build only: the product library reports 0 = the register-staged tiles split the operands)
            want = [1 if K >= 192 else (3 if L.dg_build_flags() & 1 else 0), 2 if C <= 128 else 1, 1]
            assert got == want, (C, K, H, got, want)
        # bf16: 2 = LDS-DMA kernel (igemm_dma.hip) or the window kernel, 1 = register-staged tiles
        for C, K, H in layers:
            got = [L.dg_conv_bf16_operands_ok(op, 32, H, H, C, K, 2, 1) for op in (0, 1, 2)]
            want = [2 if K >= 192 else 1, 2, 2 if K >= 192 else 1]
            assert got == want, (C, K, H, got, want)
        _lib.set_option("dma_mfma", 1)       # A/B switch: no window kernels
        assert L.dg_conv_bf16_operands_ok(1, 32, 256, 256, 64, 128, 2, 1) == 1
        assert L.dg_conv_x3_planes_ok(1, 32, 256, 256, 64, 128, 2, 1) == 0
    finally:
        _lib.set_option("dma_mfma", 0)
    # precision as an argument of the planning queries: the three arithmetics plan different K-tiles / splits for the same shape
    ws = [L.dg_conv_workspace_bytes_p(0, 64, 8, 8, 256, 512, 2, 1, prec, 1) for prec in (0, 1, 2)]
    assert all(w > 0 for w in ws)
    assert L.dg_conv_plan_splits_p(0, 64, 8, 8, 256, 512, 2, 1, 0, 1) >= 2
    # a grouped launch planned as a whole needs fewer K-slabs per problem
    assert L.dg_conv_plan_splits_p(0, 64, 8, 8, 256, 512, 2, 1, 0, 4) < L.dg_conv_plan_splits_p(0, 64, 8, 8, 256, 512, 2, 1, 0, 1)
    # the timing switch that drops operand loads does not exist in the product library
    assert L.dg_set_option(b"dbg_zero", 1) != 0


def test_graft_entry_build():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(_lib.LIB_PATH)
