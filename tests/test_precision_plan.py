"""Host logic of the x3 precision rung (CPU; round 5): per-block conditioning figures, the sequence rule of plan_escalation, the
policy switch, the weight pair pack, and the rounding model oracle/precision_sim.py uses to predict what the rung achieves."""
import math

import pytest
import torch

from iggt_official_amd import precision
from iggt_official_amd.layers.blocks import Block, _x3_weight


def _blocks(n, qk_norm=True):
    torch.manual_seed(0)
    return [Block(dim=128, num_heads=2, qk_norm=qk_norm, init_values=1.0) for _ in range(n)]


def test_block_condition_figures():
    flat = torch.ones(1024)
    c = precision.block_condition(flat, flat, torch.ones(64), torch.ones(64), 0.125)
    assert c["pr_norm1"] == pytest.approx(1.0) and c["logit_rms"] == pytest.approx(1.0)      # 0.125 * sqrt(64)
    one_hot = torch.zeros(1024)
    one_hot[3] = 5.0
    assert precision.participation_ratio(one_hot) == pytest.approx(1.0 / 1024)
    g = torch.Generator().manual_seed(1)
    for sigma, expect in ((0.5, math.exp(-1.0)), (1.0, math.exp(-4.0))):                    # E[g^2]^2 / E[g^4] = exp(-4 sigma^2)
        pr = precision.participation_ratio(torch.exp(sigma * torch.randn(200_000, generator=g)))
        assert pr == pytest.approx(expect, rel=0.35), (sigma, pr)
    no_qk = precision.block_condition(flat, flat)
    assert no_qk["logit_rms"] == 0.0


def test_outlier_channels_do_not_trip_the_layernorm_criterion():
    """Round 6: a handful of very large LayerNorm scales ("outlier dimensions" of trained ViTs) is not a heavy tail.  The
    participation ratio over ALL channels cannot tell them apart (one gamma = 10 among 1 023 of ~1: 0.11 < the round-5 line of
    0.15); the trimmed figure the rule uses can."""
    one = torch.ones(1024)
    one[5] = 10.0
    assert precision.participation_ratio(one) < 0.15 < 0.9 < precision.participation_ratio(one, precision.ESC_PR_TRIM)
    three = torch.ones(1024)
    three[[5, 100, 700]] = torch.tensor([10.0, 20.0, 6.0])
    c = precision.block_condition(one, three, torch.ones(64), torch.ones(64), 0.125)
    assert c["pr_norm1_raw"] < 0.15 and c["pr_norm2_raw"] < 0.02 and min(c["pr_norm1"], c["pr_norm2"]) > 0.9
    assert not precision.should_escalate(c)
    # ... while a heavy tail over all channels still trips it: the synthetic doses (iggt_official_amd/synthetic.py)
    from iggt_official_amd import synthetic

    def trunk_min(mode):
        vals = []
        for blk in ("aggregator.patch_embed.blocks.3", "aggregator.frame_blocks.7", "aggregator.global_blocks.23"):
            n1 = synthetic.make_tensor(blk + ".norm1.weight", (1024,), 0, mode)
            n2 = synthetic.make_tensor(blk + ".norm2.weight", (1024,), 0, mode)
            vals.append(precision.block_condition(n1, n2))
        return vals

    for mode, esc in (("stress", False), ("trained_like(qk=0.5,norm=0.5)", False), ("trained_like(qk=0,norm=1)", True),
                      ("trained_like(qk=0,norm=0,tok=0,col=1,gauss=0,out=1,outmag=10)", False),
                      ("trained_like(qk=0,norm=0,tok=0,col=1,gauss=0,out=3,outmag=8,outshare=1,massive=8)", False)):
        conds = trunk_min(mode)
        assert all(precision.should_escalate(c) == esc for c in conds), (mode, conds)
    out = trunk_min("trained_like(qk=0,norm=0,tok=0,col=1,gauss=0,out=1,outmag=10)")
    assert all(min(c["pr_norm1_raw"], c["pr_norm2_raw"]) < 0.15 for c in out)       # round 5 would have escalated all of them


def test_plan_escalation_escalates_the_block_and_everything_upstream():
    blocks = _blocks(6)
    assert precision.plan_escalation(blocks) == [False] * 6 and all(b._x3_request is False for b in blocks)
    with torch.no_grad():
        blocks[3].norm2.weight.copy_(torch.exp(1.5 * torch.randn(128)))      # concentrated LayerNorm scales: PR << 0.15
    assert blocks[3].own_escalation() and not blocks[2].own_escalation()
    assert precision.plan_escalation(blocks) == [True, True, True, True, False, False]
    with torch.no_grad():
        blocks[5].attn.q_norm.weight.fill_(10.0)                             # sharp softmax: logit r.m.s. 0.125 * 10 * 8 = 10 > 7
    assert blocks[5].own_condition()["logit_rms"] == pytest.approx(10.0, rel=1e-5)
    assert precision.plan_escalation(blocks) == [True] * 6
    # the figures are cached per parameter version: an in-place change is seen, an unchanged block is not re-evaluated
    key = blocks[0]._cond_key
    blocks[0].own_condition()
    assert blocks[0]._cond_key is key


def test_policy_switch_and_operand_format():
    blocks = _blocks(3)
    try:
        precision.set_escalation("all")
        assert precision.plan_escalation(blocks) == [True] * 3
        precision.set_escalation("off")
        with torch.no_grad():
            blocks[1].norm1.weight.copy_(torch.exp(2.0 * torch.randn(128)))
        assert precision.plan_escalation(blocks) == [False] * 3
        precision.set_escalation("auto")
        assert precision.plan_escalation(blocks) == [True, True, False]
        old = precision.operand_dtype()
        try:
            precision.set_operand_dtype(torch.bfloat16)                      # the reference's autocast arithmetic: never escalated
            assert precision.escalation() == "off" and precision.plan_escalation(blocks) == [False] * 3
        finally:
            precision.set_operand_dtype(old)
        with pytest.raises(ValueError):
            precision.set_escalation("sometimes")
    finally:
        precision.set_escalation("auto")


def test_weight_pairs_reconstruct_to_22_bits():
    torch.manual_seed(2)
    w = torch.randn(96, 160) * 0.03
    w[0, 0], w[1, 1] = 3.0e4, 1.0e-6
    w3 = _x3_weight(w)
    assert w3.dtype == torch.float16 and w3.shape == (96, 480)
    assert torch.equal(w3[:, :160], w3[:, 160:320])                          # [W_hi | W_hi | W_lo]
    rec = w3[:, :160].double() + w3[:, 320:].double()
    tol = torch.maximum(w.double().abs() * 2.0 ** -21, torch.tensor(2.0 ** -24, dtype=torch.float64))
    assert bool(((rec - w.double()).abs() <= tol).all())


def test_simulated_pair_rounding_matches_the_device_split():
    """oracle/precision_sim.py `_round(t, "fp16x2")` is the model the rung was designed against: same hi / lo as the kernels'
    split (csrc/x3.hip split_pack), including subnormal lo parts."""
    from oracle import precision_sim

    t = torch.tensor([1.0, 1.0 + 2.0 ** -12, 3.14159265, 1.0e-3, 6.0e-5, 7.0e-8, -2.5e4])
    r = precision_sim._round(t, "fp16x2")
    hi = t.half().float()
    assert torch.equal(r, hi + (t - hi).half().float())
    tol = torch.maximum(t.abs() * 2.0 ** -21, torch.tensor(2.0 ** -24))      # 22 significant bits, or fp16's subnormal grid
    assert bool(((r - t).abs() <= tol).all())
