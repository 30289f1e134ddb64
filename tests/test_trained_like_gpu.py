"""Parity under weight statistics in which a trained checkpoint departs from bounded-uniform draws (GPU; round-3 review item 1b).

`iggt_official_amd/synthetic.py` mode "trained_like": log-normal scales on every q_norm / k_norm / norm1 / norm2 weight
(reference attention.py:43-44,54: learned affines), Gaussian Linear / Conv weights with ~0.1 % of the input columns at 8x,
camera / register tokens at 30x the norm of a patch token (reference aggregator.py:123-124).  Fixtures by the REFERENCE modules
on CPU fp32 (oracle/make_golden.py) at BASELINE.json configs[1]'s size (8 views @ 518^2) and on the demo7 photographs at the
loader's crop-518, in three doses (the dose-response table is profiles/r04_trained_like_sweep.txt, probes/trained_like_sweep.py):

  tlA  sigma 0.5 (q/k-norm) / 0.5 (norm1/2): the heaviest tails at which the reference's OWN bf16 autocast mode still tracks its
       fp32 path to < 1e-2 -- global-attention logits of std 2, max 8-10, mean top probability 0.3.  Gated at north_star's 1e-3.
  tlB  sigma 0.75 / 0.5: past that edge (the fp16-operand simulation WITHOUT mean compensation reads 1.2e-3).  Gated at 1e-3 (round 5; measured 4.6e-4), no block escalated.
  tlC  the review's literal recipe, sigma 1 / 1: logits of std 15, max 88 -- a near-argmax softmax.  The map is ill-conditioned as a
       function of its own weights: the reference's fp32 arithmetic sits 8e-4 from an fp64 evaluation of the same restatement, its
       bf16 mode 0.56.  No 16-bit operand path can meet 1e-3 against it; the numbers are reported, outputs must be finite, and the
       attention dispatcher's kernel choice (adaptive static / estimated / online-max vs online-max everywhere) must not move
       the tokens by more than the operand rounding itself does -- both are rounding differences the weights amplify alike.

Every case also reports what the static-bound attention did on these statistics: mode per block (norm bound / estimated shift /
online-max only), work handed to the online-max pass, and the HIP-event time of the global attention the model paid per block."""
import pytest
import torch

from conftest import load_golden, report
from helpers import build_gpu_model, errors

pytestmark = pytest.mark.gpu

KEYS = ("depth", "depth_conf", "world_points", "world_points_conf")


def _images(g, m):
    from oracle import weights

    if m.get("scene"):      # photographs: through the product's own loader, checked against the fixture's bytes elsewhere
        import os

        from conftest import GOLDEN
        from iggt.utils.load_fn import load_and_preprocess_images

        paths = [os.path.join(GOLDEN, "images", m["scene"], f) for f in m["files"]]
        tgt = m["resize_target_size"]
        return load_and_preprocess_images(paths, mode=m["loader_mode"], resize_target_size=None if tgt is None else tuple(tgt))
    return weights.make_images(m["S"], m["H"], m["W"], seed=m["image_seed"], device="cuda")


def _run(case, warm=3):
    """Forward of `case` on the HIP path; returns (errors per quantity, attention report).  `warm` extra forwards first, so that
    the adaptive switch of every block has settled: forward 1 runs the round-3 sequence (norm bound, flagged tiles redone), its
    guard snapshot turns the estimated-shift launches on for the blocks that flagged (models/aggregator.py), forward 2 runs them,
    forward 3 is the steady state."""
    from iggt_official_amd import profiling

    g = load_golden(case)
    m = g["meta"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    images = _images(g, m)
    cap = {}
    h = model.aggregator.register_forward_hook(lambda mod, i, o: cap.__setitem__("tokens", o[0]))
    for _ in range(warm):
        model(images)
        torch.cuda.synchronize()     # the guard snapshot of this forward has landed before the next one looks at it
    profiling.enable("global_attn")
    profiling.enable("global_attn_x3")
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    pred = model(images)
    t1.record()
    torch.cuda.synchronize()
    recs = profiling.summarize(profiling.disable("global_attn")) + profiling.summarize(profiling.disable("global_attn_x3"))
    h.remove()
    ss, ts, cs = m["spatial_stride"], m["token_stride"], m.get("channel_stride", 1)
    res = {f"tokens_{li}": errors(cap["tokens"][li][:, :, ::ts, ::cs], g[f"tokens_{li}"]) for li in (4, 11, 17, 23)}
    res["pose_enc"] = errors(torch.stack(pred["pose_enc"], 0), g["pose_enc"])
    for k in KEYS:
        res[k] = errors(pred[k][:, :, ::ss, ::ss], g[k])
    for k, v in pred.items():
        if torch.is_tensor(v):
            assert torch.isfinite(v).all(), (case, k)
    att = model.aggregator.static_softmax_stats()
    guards = [None if b.attn_guard() is None else b.attn_guard().tolist() for b in model.aggregator.global_blocks]
    # per global block: x = x3 precision rung (pairs, online maximum), o = online-max only, e = estimated shift, n = norm bound
    att["global_mode_per_block"] = "".join("x" if gd is None else "o" if gd[1] < 0 else ("e" if gd[4] == 1 else "n") for gd in guards)
    att["global_rows_redone"] = sum(max(gd[5], 0) for gd in guards if gd is not None)
    esc = model.aggregator.escalation_report()
    att["escalated_blocks"] = len(esc["x3"])
    att["escalation"] = dict(own_verdict=len(esc["own_verdict"]), bf16_fallback=esc["bf16_fallback"],
                             min_participation_ratio=esc["min_participation_ratio"], max_logit_rms=esc["max_logit_rms"])
    att["global_attn_ms_per_block"] = [round(r[0], 4) for r in recs]
    att["global_attn_ms_mean"] = sum(r[0] for r in recs) / max(len(recs), 1)
    att["tokens"] = m["S"] * (5 + (m["H"] // 14) * (m["W"] // 14))
    att["forward_ms"] = t0.elapsed_time(t1)
    return res, att, cap["tokens"]


def _report(case, res, att, extra=None):
    report(f"trained_like/{case}", dict(errors={k: dict(max=v[0], l2=v[1], l2_centered=v[2]) for k, v in res.items()},
                                        attention=att, **(extra or {})))


@pytest.mark.parametrize("case", ["full_s8_518_tlA", "real_demo7_s4_crop518_tlA"])
def test_trained_like_dose_a_meets_the_tolerance(case):
    """Heavy-tailed but well-conditioned: every gate of the bounded-uniform fixtures applies (1e-3 l2, 1.5e-3 of the range)."""
    from iggt_official_amd import precision

    res, att, _ = _run(case)
    # the same forward without the mean-input compensation: what the compensation is worth on these statistics
    precision.set_mean_compensation(False)
    try:
        res_nc, _, _ = _run(case, warm=0)
    finally:
        precision.set_mean_compensation(True)
    _report(case, res, att, dict(tokens_23_l2_without_compensation=res_nc["tokens_23"][1],
                                 world_points_l2_without_compensation=res_nc["world_points"][1]))
    for k, v in res.items():
        assert v[1] < 1e-3, (case, k, v)
        if not k.startswith("tokens"):
            assert v[0] < 2e-3, (case, k, v)     # max over range: outputs are exp(.) of the head maps, range 80-280 here


def test_trained_like_dose_b_reported():
    """sigma 0.75 on the q/k-norm scales: past the edge at which the reference's bf16 mode leaves 1e-2."""
    res, att, _ = _run("full_s8_518_tlB")
    _report("full_s8_518_tlB", res, att)
    # the host-side snapshot turned the estimated-shift launches on where the norm bound had flagged tiles, and they stuck
    assert att["global_mode_per_block"].count("e") >= 12, att["global_mode_per_block"]
    assert att["escalated_blocks"] == 0, att          # passes on single fp16 operands and must not pay for the x3 rung
    for k, v in res.items():
        assert v[1] < 1e-3, (k, v)


# Round 5 (review item 1): the doses between tlB and tlC.  The reference's fp32 is well-conditioned there (<= 3e-5 from an fp64
# evaluation, profiles/r04_trained_like_sweep.txt) while single fp16 operands are predicted 3e-3 .. 2e-2 off -- and leave-one-in
# ablation shows no single rounding site carrying it (profiles/r05_precision_groups.txt).  The x3 precision rung (csrc/x3.hip,
# precision.plan_escalation) runs every block that is ill-conditioned by its own LayerNorm / q-k-norm scales, and every block in
# front of it, on fp16 hi + lo operand pairs (three MFMA passes per product).
#   tlD  sigma 1 / 0.5: global-attention logits of std 15 (sharp softmax), LayerNorm scales as in tlA
#   tlE  sigma 0.75 / 0.75
#   tlF  sigma 0 / 1: ordinary logits; ~19 of 1 024 channels carry the normalised signal
@pytest.mark.parametrize("case", ["full_s8_518_tlD", "full_s8_518_tlE", "full_s8_518_tlF", "real_demo7_s4_crop518_tlD",
                                  "real_demo7_s4_crop518_tlE", "real_demo7_s4_crop518_tlF"])
def test_heavy_tailed_well_conditioned_doses_meet_the_tolerance_on_the_x3_rung(case):
    """Gated at north_star's 1e-3 like every bounded-uniform fixture; the same forward with the rung switched off is run and
    REPORTED (what single fp16 operands give at this dose, and what the rung costs)."""
    from iggt_official_amd import precision

    prev = precision.escalation_policy()
    precision.set_escalation("off")
    try:
        res_off, att_off, _ = _run(case, warm=2)
    finally:
        precision.set_escalation(prev)
    res, att, _ = _run(case, warm=1)
    _report(case, res, att, dict(single_fp16_operands=dict(
        errors={k: dict(max=v[0], l2=v[1]) for k, v in res_off.items()}, forward_ms=att_off["forward_ms"],
        global_mode_per_block=att_off["global_mode_per_block"])))
    assert att["escalated_blocks"] > 0, att
    for k, v in res.items():
        assert v[1] < 1e-3, (case, k, v)
        if not k.startswith("tokens"):
            assert v[0] < 2e-3, (case, k, v)


@pytest.mark.parametrize("case", ["full_s8_518_tlC", "real_demo7_s4_crop518_tlC"])
def test_trained_like_literal_recipe_on_the_x3_rung(case):
    """The round-3 review's literal recipe (sigma 1 / 1, tokens 30x, 8x columns): the map is ill-conditioned as a function of its own
    weights -- the reference's fp32 arithmetic itself sits 7.8e-4 from an fp64 evaluation (its bf16 mode 0.56), so 1e-3 against
    the fp32 fixture is the noise floor of the REFERENCE, not a property of this path.  Single fp16 operands measured 8.3e-2
    (8 views) / 3.6e-2 (photographs) in round 4; on the x3 rung every block runs on operand pairs and the distance is gated at
    TLC_GATE -- a few times the reference's own distance from fp64 (two fp32-grade evaluations in different summation orders).
    The single-fp16 forward is still run and reported; with the 8-view case also the dispatcher's kernel choice (adaptive static /
    estimated / online-max vs online-max everywhere), which must not move the tokens more than the operand rounding does."""
    from iggt_official_amd import precision

    res, att, _ = _run(case, warm=1)
    prev = precision.escalation_policy()
    precision.set_escalation("off")
    try:
        res_off, att_off, tok = _run(case)
        extra = dict(single_fp16_operands=dict(errors={k: dict(max=v[0], l2=v[1]) for k, v in res_off.items()},
                                               global_mode_per_block=att_off["global_mode_per_block"],
                                               forward_ms=att_off["forward_ms"]))
        if case.startswith("full"):
            tok23 = tok[23].clone()
            precision.set_static_softmax(False)
            try:
                res_on, _, tok_on = _run(case, warm=0)
            finally:
                precision.set_static_softmax(True)
            kernel_choice = errors(tok23, tok_on[23])
            extra["single_fp16_operands"].update(tokens_23_l2_online_max_only=res_on["tokens_23"][1],
                                                 tokens_23_l2_between_dispatch_modes=kernel_choice[1])
            assert kernel_choice[1] < 2.0 * res_off["tokens_23"][1] + 1e-3, (kernel_choice, res_off["tokens_23"])
    finally:
        precision.set_escalation(prev)
    _report(case, res, att, extra)
    assert att["escalated_blocks"] == 72, att
    assert res_off["tokens_23"][1] < 0.8, res_off["tokens_23"]
    for k, v in res.items():
        assert v[1] < TLC_GATE, (case, k, v)


TLC_GATE = 1e-3     # measured: tokens 2.2e-4 (8 views) / 8.6e-5 (photographs), outputs <= 3.6e-4 (profiles/r05_parity_report.json)


def test_graph_capture_takes_the_estimated_shift_decision():
    """under enable_graphs() the host-side decision to issue the estimated-shift launches used to miss the capture -- the
    two warm-up forwards ran back to back, the guard snapshot of the first had not landed when the second looked, and the captured
    graph stayed on the round-3 sequence (flagged tiles redone by the online-max kernel).  graphs.GraphCache.run now synchronises
    between warm-ups and repeats them while decisions change: on the sigma 0.75 / 0.5 checkpoint the REPLAYED forward runs at
    least half of its global blocks in estimated mode, hands nothing to whole-tile redo, and meets the tolerance."""
    from iggt_official_amd import precision

    g = load_golden("full_s8_518_tlB")
    m = g["meta"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    precision.reset_guards(model)
    images = _images(g, m)
    model.enable_graphs(True)
    try:
        model(images)                 # warm-ups + capture + first replay
        pred = model(images)          # replay
        torch.cuda.synchronize()
        guards = [b.attn_guard().tolist() for b in model.aggregator.global_blocks]
        modes = "".join("o" if gd[1] < 0 else ("e" if gd[4] == 1 else "n") for gd in guards)
        report("trained_like/full_s8_518_tlB_graphs", dict(global_mode_per_block=modes,
                                                           flagged_tiles=[gd[1] for gd in guards]))
        assert modes.count("e") >= 12, modes
        # estimated mode hands over single rows (a handful per block: measured 1 row in 1 of 24 blocks), never whole tiles
        assert sum(max(gd[5], 0) for gd in guards) <= 24 * 16 and all(gd[1] <= 2 for gd in guards), guards
        ss = m["spatial_stride"]
        for k in KEYS:
            e = errors(pred[k][:, :, ::ss, ::ss], g[k])
            assert e[1] < 1e-3, (k, e)
    finally:
        model.enable_graphs(False)
        precision.reset_guards(model)


def test_graph_replay_on_the_x3_rung():
    """The escalated forward as hipGraph segments: the plan (which blocks run on operand pairs) is taken from cached figures, so
    capture involves no device read-back; the replayed outputs meet the fixture's gates like the eager ones."""
    g = load_golden("full_s8_518_tlD")
    m = g["meta"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    images = _images(g, m)
    model.enable_graphs(True)
    try:
        model(images)                 # warm-ups + capture + first replay
        pred = model(images)          # replay
        torch.cuda.synchronize()
        assert len(model.aggregator.escalation_report()["x3"]) == 71
        ss = m["spatial_stride"]
        for k in KEYS:
            e = errors(pred[k][:, :, ::ss, ::ss], g[k])
            assert e[1] < 1e-3, (k, e)
        e = errors(torch.stack(pred["pose_enc"], 0), g["pose_enc"])
        assert e[1] < 1e-3, e
    finally:
        model.enable_graphs(False)


# Round 6 (review item 2 / ADVICE r5): OUTLIER CHANNELS.  Trained ViTs -- DINOv2 included -- carry a few LayerNorm scales that are
# an order of magnitude above the rest ("outlier dimensions"), and massive activations on a handful of residual-stream channels.
# That is not the heavy tail over ALL channels the doses above model, but the round-5 rule could not tell them apart: one
# gamma = 10 among 1 023 of ~1 has a participation ratio of 0.11 < 0.15, so such a checkpoint would have put that block and
# everything upstream on the x3 rung (2.6x the 32-view forward).  Fixtures by the reference modules, 8 views @ 518^2:
#   tlG  one gamma = 10 in every norm1 / norm2 of the model (each LayerNorm its own channel), everything else as "stress"
#   tlH  "DINOv2-like": the same 3 channels in every block carry gamma x 8 and rows x 8 of every mlp.fc2 (massive activations)
# Both must meet 1e-3 on SINGLE fp16 operands, and the trimmed participation ratio (precision.block_condition) must leave every
# block un-escalated; the rung is run as well and reported (what it would have cost, what it would have bought).
@pytest.mark.parametrize("case", ["full_s8_518_tlG", "full_s8_518_tlH"])
def test_outlier_channels_pass_on_single_operands_and_are_not_escalated(case):
    from iggt_official_amd import precision

    res, att, _ = _run(case, warm=2)
    prev = precision.escalation_policy()
    precision.set_escalation("all")
    try:
        res_x3, att_x3, _ = _run(case, warm=1)
    finally:
        precision.set_escalation(prev)
    esc = build_gpu_model(load_golden(case)["meta"]["mode"], 0).aggregator.escalation_report()
    _report(case, res, att, dict(min_participation_ratio_untrimmed=esc["min_participation_ratio_untrimmed"],
                                 on_the_x3_rung=dict(errors={k: dict(max=v[0], l2=v[1]) for k, v in res_x3.items()},
                                                     forward_ms=att_x3["forward_ms"])))
    assert att["escalated_blocks"] == 0 and att["escalation"]["own_verdict"] == 0, att
    assert esc["min_participation_ratio_untrimmed"] < 0.15 < precision.ESC_PR_MIN < esc["min_participation_ratio"]
    for k, v in res.items():
        assert v[1] < 1e-3, (case, k, v)
        if not k.startswith("tokens"):
            assert v[0] < 2e-3, (case, k, v)
