"""Parity under weight statistics in which a trained checkpoint departs from bounded-uniform draws (GPU; round-3 review item 1b).

`iggt_official_amd/synthetic.py` mode "trained_like": log-normal scales on every q_norm / k_norm / norm1 / norm2 weight
(reference attention.py:43-44,54: learned affines), Gaussian Linear / Conv weights with ~0.1 % of the input columns at 8x,
camera / register tokens at 30x the norm of a patch token (reference aggregator.py:123-124).  Fixtures by the REFERENCE modules
on CPU fp32 (oracle/make_golden.py) at BASELINE.json configs[1]'s size (8 views @ 518^2) and on the demo7 photographs at the
loader's crop-518, in three doses (the dose-response table is profiles/r04_trained_like_sweep.txt, probes/trained_like_sweep.py):

  tlA  sigma 0.5 (q/k-norm) / 0.5 (norm1/2): the heaviest tails at which the reference's OWN bf16 autocast mode still tracks its
       fp32 path to < 1e-2 -- global-attention logits of std 2, max 8-10, mean top probability 0.3.  Gated at north_star's 1e-3.
  tlB  sigma 0.75 / 0.5: past that edge (the fp16-operand simulation WITHOUT mean compensation reads 1.2e-3).  Gated at 3e-3, reported.
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
    pred = model(images)
    torch.cuda.synchronize()
    recs = profiling.summarize(profiling.disable("global_attn"))
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
    guards = [b.attn_guard().tolist() for b in model.aggregator.global_blocks]
    att["global_mode_per_block"] = "".join("o" if gd[1] < 0 else ("e" if gd[4] == 1 else "n") for gd in guards)
    att["global_rows_redone"] = sum(max(gd[5], 0) for gd in guards)
    att["global_attn_ms_per_block"] = [round(r[0], 4) for r in recs]
    att["global_attn_ms_mean"] = sum(r[0] for r in recs) / max(len(recs), 1)
    att["tokens"] = m["S"] * (5 + (m["H"] // 14) * (m["W"] // 14))
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
    for k, v in res.items():
        assert v[1] < 3e-3, (k, v)


@pytest.mark.parametrize("case", ["full_s8_518_tlC", "real_demo7_s4_crop518_tlC"])
def test_trained_like_literal_recipe_is_reported_not_gated(case):
    """The review's literal recipe (sigma 1 / 1, tokens 30x, 8x columns).  Reported; see the module docstring for why 1e-3 is
    not a meaningful gate here.  What must hold: finite outputs, tokens that are not garbage (the fp16-rounding simulation on the
    CPU predicts 0.3-0.4), and a kernel choice inside the attention dispatcher (adaptive static / estimated / online-max vs
    online-max for every block) that moves the tokens no more than the distance to the reference -- two rounding differences
    amplified by the same weights; a kernel fault would show as a much larger one."""
    from iggt_official_amd import precision

    res, att, tok = _run(case)
    tok23 = tok[23].clone()
    precision.set_static_softmax(False)
    try:
        res_on, _, tok_on = _run(case, warm=0)
    finally:
        precision.set_static_softmax(True)
    kernel_choice = errors(tok23, tok_on[23])
    _report(case, res, att, dict(tokens_23_l2_online_max_only=res_on["tokens_23"][1],
                                 tokens_23_l2_between_dispatch_modes=kernel_choice[1]))
    assert res["tokens_23"][1] < 0.8, res["tokens_23"]
    assert kernel_choice[1] < 2.0 * res["tokens_23"][1] + 1e-3, (kernel_choice, res["tokens_23"])
