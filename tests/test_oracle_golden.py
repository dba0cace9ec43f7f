"""Pins the oracle (oracle/discogan_ref.py) to the committed golden vectors.  CPU only.

  * ref_s512_n2.json  : outputs of the TRUE reference (model.py + image_translation.py loop body)
  * oracle_s64/16_n4  : the oracle's own capture for the derived nets (drift detector)
"""
import json
import os

import pytest
import torch

from oracle import discogan_ref as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sample_idx(numel, k=8):
    return [int((i * 2654435761) % numel) for i in range(1, k + 1)]


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def run_against(fix, S, N, loss_rtol0, loss_rtol, grad_rtol):
    st = O.build_state(image_size=S, seed=1234)
    # state_dict keys and seeded weights: bit-exact
    for name, net in st.nets.items():
        assert list(net.state_dict().keys()) == fix["meta"]["state_dict_keys"][name]
        for k, v in net.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            ref = fix["init"][name][k]
            assert list(v.shape) == ref["shape"], f"{name}.{k}"
            f = v.reshape(-1)
            assert float(f.double().sum()) == ref["sum"], f"seeded init {name}.{k}"
            assert [float(f[i]) for i in sample_idx(f.numel())] == ref["samples"], f"seeded init {name}.{k}"
    A, B = O.synthetic_batch(N, S, seed=0)
    for it, rec in enumerate(fix["iters"]):
        out = O.train_iteration(st, A, B, it, do_step=False)
        got = O.losses_to_floats(out)
        rt = loss_rtol0 if it == 0 else loss_rtol
        for k, v in rec["losses"].items():
            assert abs(got[k] - v) <= rt * abs(v) + 1e-7, f"iter {it} {k}: {got[k]} vs {v}"
        assert ("D" if O.is_dis_step(it, st.args) else "G") == rec["step"]
        live = ("dis_A", "dis_B") if rec["step"] == "D" else ("gen_A", "gen_B")
        for name in live:
            for pn, p in st.nets[name].named_parameters():
                ref_norm = rec["grad_norms"][name][pn]
                gn = float(p.grad.double().norm())
                assert abs(gn - ref_norm) <= (grad_rtol if it == 0 else 20 * grad_rtol) * ref_norm + 1e-12, \
                    f"iter {it} {name}.{pn}: {gn} vs {ref_norm}"
        (st.optim_dis if rec["step"] == "D" else st.optim_gen).step()
        for name, net in st.nets.items():
            for bn_, b in net.named_buffers():
                if b.dtype == torch.int64:
                    assert int(b) == rec["buffers"][name][bn_]


@pytest.mark.parametrize("S", [16, 64])
def test_oracle_reproduces_its_fixtures(S):
    run_against(load(f"oracle_s{S}_n4.json"), S, 4, 1e-5, 1e-3, 1e-4)


@pytest.mark.slow
def test_oracle_matches_true_reference_at_512():
    """The restatement against outputs of /root/reference itself (the only size it can run)."""
    fix = load("ref_s512_n2.json")
    assert fix["meta"]["source"].startswith("reference /root/reference")
    run_against(fix, 512, 2, 1e-5, 2e-2, 1e-4)


def test_fixture_is_data_not_source():
    for name in ("ref_s512_n2.json", "oracle_s64_n4.json", "oracle_s16_n4.json"):
        fix = load(name)
        assert set(fix.keys()) == {"init", "iters", "meta"}
        assert len(fix["iters"]) == 3 and [r["step"] for r in fix["iters"]] == ["D", "G", "G"]


def test_depth_rule_and_counts():
    """parameter counts and key layout at 512 / 64."""
    assert O.stage_channels(512) == [64, 128, 256, 512, 1024, 2048, 2048]
    assert O.stage_channels(64) == [64, 128, 256, 512]
    g, d = O.Generator(True, image_size=64), O.Discriminator(image_size=64)
    assert sum(p.numel() for p in g.parameters()) == 7153480
    assert sum(p.numel() for p in d.parameters()) == 2765568
    x = torch.rand(2, 3, 64, 64)
    assert g(x).shape == (2, 3, 64, 64)
    p, feats = d(x)
    assert p.shape == (2, 1, 1, 1) and [tuple(f.shape[1:]) for f in feats] == [(128, 16, 16), (256, 8, 8), (512, 4, 4)]
    with pytest.raises(ValueError):
        O.stage_channels(100)


def test_log_line_format():
    st = O.build_state(image_size=16, seed=1234)
    A, B = O.synthetic_batch(4, 16)
    out = O.train_iteration(st, A, B, 0)
    line = O.format_log(0, 100, out)
    import re
    assert re.match(r"^Iter \[0/100\] GEN: \d+\.\d{4}/\d+\.\d{4}, FM: \d+\.\d{4}/\d+\.\d{4}, "
                    r"RECON: \d+\.\d{4}/\d+\.\d{4}, DIS: \d+\.\d{4}/\d+\.\d{4}$", line), line
