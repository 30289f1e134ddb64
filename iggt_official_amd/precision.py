"""16-bit operand format of the trunk GEMM / attention kernels (include/iggt_hip.h).

The reference's GPU mode is autocast(bfloat16) (demo.py:190-193); its documented output, and the parity
target of this repository, is the fp32 CPU path.  bf16 operands (8 significant bits) put the aggregated tokens
6.5e-3 (relative l2) away from fp32; IEEE half (11 bits) at the same MFMA rate brings that to 8e-4
(oracle/precision_sim.py), inside the 1e-3 target, so fp16 is the default.  Everything that is not an MFMA operand --
accumulation, LayerNorm, q/k-norm, RoPE, softmax statistics, LayerScale, the residual stream, the dense heads -- is
fp32 in both modes.  fp16 stores saturate at +-65504 (csrc/common.h pack_h2); LayerNorm outputs, q/k/v, attention
outputs and GELU activations of a ViT-L are orders of magnitude below that.

Of the eight rounding sites of a block (oracle/precision_sim.py --ablate) the WEIGHTS dominate the fp16 error: their
rounding is the same for every token, so it does not average out (tokens 8.6e-4 -> 3.8e-4 with exact weights; the
seven activation sites together matter less than the weights of any one GEMM).  Almost all of that is the response
to the MEAN input vector: x W^T = x Wh^T + mu dW^T + (x - mu) dW^T with Wh = round16(W), dW = W - Wh, mu = mean over
tokens of the GEMM input.  "Mean-input compensation" restores the middle term exactly: a column mean over ~1024 evenly
spaced rows of the input and one small matrix-vector product give b' = b + dW mu, which replaces the bias of the GEMM
(csrc/elementwise.hip colmean / bias_correct; layers/blocks.py compensated_bias).  Simulated token error 8.6e-4 ->
4.3e-4 for two tiny kernels per GEMM (~1.5 % of the step).  On with fp16 operands, off in bf16 mode, which keeps the
reference's autocast arithmetic; IGGT_MEAN_COMP=0 turns it off.  With view sharding each rank uses the mean of its
own rows (no collective): results then differ between shardings by O(1e-4) relative, inside the tolerance.

    IGGT_OPERAND_DTYPE=bf16|f16      environment override (read at import)
    set_operand_dtype(torch.bfloat16)  programmatic; affects modules' next forward (packed weights are re-made)
"""
import os

import torch

_NAMES = {"f16": torch.float16, "fp16": torch.float16, "half": torch.float16, "bf16": torch.bfloat16,
          "bfloat16": torch.bfloat16}
_operand = _NAMES[os.environ.get("IGGT_OPERAND_DTYPE", "f16").lower()]


def operand_dtype() -> torch.dtype:
    return _operand


def set_operand_dtype(dt) -> None:
    global _operand
    if isinstance(dt, str):
        dt = _NAMES[dt.lower()]
    if dt not in (torch.float16, torch.bfloat16):
        raise ValueError("operand dtype must be torch.float16 or torch.bfloat16")
    _operand = dt


_mean_comp = os.environ.get("IGGT_MEAN_COMP", "1") != "0"


def mean_compensation() -> bool:
    """Mean-input compensation of the weight rounding: fp16 operands only (see the module docstring)."""
    return _mean_comp and _operand == torch.float16


def set_mean_compensation(on: bool) -> None:
    global _mean_comp
    _mean_comp = bool(on)


# Which of a block's four GEMMs get the compensation (each costs two small kernels per call): IGGT_MEAN_COMP_SITES, a comma
# list out of qkv,proj,fc1,fc2 (default: all).  Part of the pack key of every block (layers/blocks.py).
_comp_sites = frozenset(s for s in os.environ.get("IGGT_MEAN_COMP_SITES", "qkv,proj,fc1,fc2").split(",") if s)


def mean_compensation_sites() -> frozenset:
    return _comp_sites if mean_compensation() else frozenset()


def set_mean_compensation_sites(sites) -> None:
    global _comp_sites
    sites = frozenset(sites)
    if not sites <= {"qkv", "proj", "fc1", "fc2"}:
        raise ValueError("sites out of qkv, proj, fc1, fc2")
    _comp_sites = sites


def operand_name() -> str:
    return "f16" if _operand == torch.float16 else "bf16"


# Static-bound softmax of the aggregator blocks (csrc/attention_v3.hip): numerators 2^(s - |q|max |k|max) with a shift known
# before the first key tile; rows that would underflow are recomputed by the online-max kernel.  IGGT_STATIC_SOFTMAX=0
# selects the online-max kernel for every attention.
_static_softmax = os.environ.get("IGGT_STATIC_SOFTMAX", "1") != "0"


def static_softmax() -> bool:
    return _static_softmax


def set_static_softmax(on: bool) -> None:
    global _static_softmax
    _static_softmax = bool(on)


# Adaptive switch of the static-bound attention: a block whose query tiles keep failing the acceptance test (more than 1/8
# flagged) goes straight to the online-max kernel for the next 16 calls (include/iggt_hip.h, `guard`).  IGGT_STATIC_GUARD=0
# always tries the static kernel first.
_static_guard = os.environ.get("IGGT_STATIC_GUARD", "1") != "0"


def static_guard() -> bool:
    return _static_guard


def set_static_guard(on: bool) -> None:
    global _static_guard
    _static_guard = bool(on)


# Estimated-shift mode of the static-bound attention + row-granular hand-over to the online-max pass (csrc/attention_est.hip,
# round 4): where the norm bound is loose (trained-like q/k-norm affines, sink keys, register tokens of outlying norm) the
# adaptive switch moves a call site to a shift estimated from each row's exact maximum over a key sample instead of giving the
# call to the online-max kernel.  One-pass launches only (single GPU and the gather-first form).  IGGT_ATTN_EST=0: round-3
# behaviour (norm bound, whole 256-row tiles flagged); needs the adaptive switch (IGGT_STATIC_GUARD).
_attn_est = os.environ.get("IGGT_ATTN_EST", "1") != "0"


def attn_estimated_shift() -> bool:
    return _attn_est and _static_guard and _static_softmax


def set_attn_estimated_shift(on: bool) -> None:
    global _attn_est
    _attn_est = bool(on)


# Multi-GPU: hide the K/V all-gather behind the attention over a rank's own keys (layers/blocks.py _attend_overlapped; needs
# the static softmax).  IGGT_GATHER_OVERLAP=0 restores gather -> one attention launch.
_gather_overlap = os.environ.get("IGGT_GATHER_OVERLAP", "1") != "0"


def gather_overlap() -> bool:
    return _gather_overlap


def set_gather_overlap(on: bool) -> None:
    global _gather_overlap
    _gather_overlap = bool(on)


# fp16 range telemetry (debug): IGGT_DEBUG_SATURATION=1 makes the block engine count, after every kernel that stores 16-bit
# activations (LayerNorm output, qkv, attention output, MLP hidden), the entries that were clamped to +-65504 or are not
# finite (csrc/elementwise.hip count_saturated_kernel).  fp16 stores never produce inf -- they clamp -- so a silent
# saturation is exactly what this counter exposes; a non-zero count means: run that checkpoint with bf16 operands.
_debug_sat = os.environ.get("IGGT_DEBUG_SATURATION", "0") == "1"
_sat_counters = {}


def debug_saturation() -> bool:
    return _debug_sat


def set_debug_saturation(on: bool) -> None:
    global _debug_sat
    _debug_sat = bool(on)
    _sat_counters.clear()


def count_saturation(site: str, x) -> None:
    """(block engine hook) accumulate the saturated-entry count of the 16-bit matrix x under `site`."""
    from . import _C

    c = _sat_counters.get((site, str(x.device)))
    if c is None:
        c = _sat_counters[(site, str(x.device))] = torch.zeros(1, dtype=torch.int64, device=x.device)
    _C.count_saturated(x, c)


def saturation_report() -> dict:
    """{site: clamped / non-finite 16-bit stores since set_debug_saturation(True)} (synchronises)."""
    out = {}
    for (site, _dev), c in _sat_counters.items():
        out[site] = out.get(site, 0) + int(c.item())
    return out


# Range folding of the trunk blocks' weights at pack time (layers/blocks.py fold_ranges): exact power-of-two rescaling of
# outlying weight columns / rows against the LayerNorm / LayerScale / V-projection partner that absorbs the factor, so that a
# checkpoint with weights beyond +-65504 keeps fp16 operands; a block that still does not fit runs on bf16 alone.
# IGGT_RANGE_FOLD=0: round-3 behaviour (such a checkpoint is rejected at pack time / sent to bf16 by load_checkpoint).
_range_fold = os.environ.get("IGGT_RANGE_FOLD", "1") != "0"


def range_folding() -> bool:
    return _range_fold


def set_range_folding(on: bool) -> None:
    global _range_fold
    _range_fold = bool(on)


# Per-block precision rung "x3" (round 5; csrc/x3.hip, layers/blocks.py Block._forward_x3): a block whose own weights make the map
# ill-conditioned for 11-bit operands runs every MFMA product on fp16 hi + lo PAIRS in three passes (22 significant bits, 3x the
# MFMA work of that block).  Decided per block at pack time from the block's own parameters -- never globally:
#   * participation ratio of the LayerNorm scales, PR(g) = (sum g^2)^2 / (C sum g^4) in (0, 1]: the share of the 1 024 channels
#     that carries the normalised signal (1 for flat scales; 0.37 / 0.11 / 0.018 for log-normal scales of sigma 0.5 / 0.75 / 1).
#     Below ESC_PR_MIN the reference's fp32 is still 1e-5 from fp64 while single fp16 operands leave 3e-3 .. 2e-2
#     (profiles/r04_trained_like_sweep.txt, r05_precision_groups.txt);
#   * predicted r.m.s. of the attention logits from the q/k-norm scales, scale * |g_q * g_k|_2 (1.0 for flat scales; measured
#     2 / 5.5 / 15 at sigma 0.5 / 0.75 / 1): above ESC_LOGIT_RMS_MAX the softmax is sharp enough that operand rounding of q and k
#     moves the probabilities by per cent.
# Thresholds, from the dose fixtures measured on the MI355X at 8 x 518^2 / on the demo7 photographs (tests/test_trained_like_gpu.py,
# profiles/r05_parity_report.json; single fp16 operands -> x3 rung, aggregated tokens, relative l2 against the reference's fp32):
#     sigma_qk / sigma_n   blocks' PR (min .. max)   logit r.m.s. (median, max)   single fp16         x3
#     0.5  / 0.5           0.27 .. 0.46              1.5, 2.2                     2.8e-4 / 4.8e-4     -- (not escalated)
#     0.75 / 0.5           0.27 .. 0.46              2.5, 5.5                     4.0e-4              -- (not escalated)
#     1    / 0.5           0.27 .. 0.46              5.0, 15.4                    1.0e-3 / 1.5e-3     4.3e-5 / 4.4e-5
#     0.75 / 0.75          0.055 .. 0.24             2.5, 5.5                     5.0e-4 / 8.0e-4     3.2e-6 / 3.3e-6
#     0    / 1             0.016 .. 0.125            1.0, 1.05                    1.1e-2 / 6.1e-3     2.8e-5 / 1.5e-5
#     1    / 1             0.016 .. 0.125            5.0, 15.4                    8.5e-2 / 3.7e-2     2.2e-4 / 8.6e-5
# Round 5 put the line at PR < 0.15 over ALL channels (a factor 1.8 from the doses that pass on single operands; trips on most blocks
# of sigma_n = 0.75, marginal on single operands: 8e-4 on photographs); ESC_LOGIT_RMS_MAX = 7 leaves 28 % to the largest block of
# sigma_qk = 0.75.
# An ill-conditioned block AMPLIFIES the rounding of everything upstream of it, so the owner of a block sequence escalates a
# block when it OR ANY LATER block trips a criterion (plan_escalation below; layers/blocks.py Block._x3_request).
# IGGT_ESCALATE = auto (default) | off | all;  fp16 operands only (bf16 mode keeps the reference's autocast arithmetic).
# Round 6 (review item 2): the participation ratio is taken over the scales WITHOUT their ESC_PR_TRIM largest (per 1 024
# channels).  A heavy tail over ALL channels (the doses above) survives the trimming; a few OUTLIER channels do not -- and those are
# what trained ViTs, DINOv2 included, are known for: one gamma = 10 among 1 023 of ~1 has a raw PR of 0.11 and used to put that
# block AND EVERYTHING UPSTREAM on the rung (2.6x the forward at 32 views) although nothing about it is ill-conditioned.  Measured on
# the MI355X with single fp16 operands against reference fixtures (tests/test_trained_like_gpu.py, profiles/r06_parity_report.json):
# "one gamma = 10 in every LayerNorm" (tlG) and "the same 3 channels x 8 in every block + massive activations on them" (tlH) pass
# 1e-3 like the bounded-uniform checkpoint.  Trimmed figures of the doses (min .. max over the 144 LayerNorms of the trunk):
#     sigma_n 0.5: 0.39 .. 0.50   |   0.75: 0.155 .. 0.29   |   1: 0.05 .. 0.19   |   tlG, tlH, stress: 0.96
# ESC_PR_MIN = 0.25 keeps every decision of round 5 on those doses (sigma_n = 0.75: 89 % of the LayerNorms trip, so with the
# sequence rule all 72 blocks escalate) and leaves a factor 1.6 to sigma_n = 0.5.
ESC_PR_MIN = 0.25
ESC_PR_TRIM = 4            # channels dropped per 1 024 before the participation ratio is taken
ESC_LOGIT_RMS_MAX = 7.0
_escalate = os.environ.get("IGGT_ESCALATE", "auto").lower()
if _escalate not in ("auto", "off", "all"):
    raise ValueError("IGGT_ESCALATE must be auto, off or all")


def escalation() -> str:
    return _escalate if _operand == torch.float16 else "off"


def escalation_policy() -> str:
    """The policy as set (IGGT_ESCALATE / set_escalation), whatever the operand format: what a caller that switches the rung off
    for a comparison forward has to restore."""
    return _escalate


def set_escalation(mode: str) -> None:
    global _escalate
    if mode not in ("auto", "off", "all"):
        raise ValueError("escalation mode must be 'auto', 'off' or 'all'")
    _escalate = mode


def participation_ratio(g: torch.Tensor, trim: int = 0) -> float:
    """(sum g^2)^2 / (n sum g^4) over the scales without their `trim` largest magnitudes (n = what is left)."""
    g2 = g.detach().double().flatten() ** 2
    if trim > 0 and trim < g2.numel():
        g2 = g2.sort().values[:g2.numel() - trim]
    return float(g2.sum() ** 2 / (g2 * g2).sum().clamp_min(1e-300) / g2.numel())


def block_condition(n1w, n2w, qw=None, kw=None, scale: float = 0.125) -> dict:
    """Conditioning figures of one transformer block from its own parameters (see above); qw / kw: q_norm / k_norm scales of a
    q/k-norm block, None for the DINOv2 blocks (their logits are not bounded by a norm; only the LayerNorm criterion applies).
    pr_norm1 / pr_norm2: the TRIMMED participation ratios the rule uses; pr_*_raw: over all channels (round 5's figure)."""
    t1, t2 = n1w.numel() * ESC_PR_TRIM // 1024, n2w.numel() * ESC_PR_TRIM // 1024
    c = dict(pr_norm1=participation_ratio(n1w, t1), pr_norm2=participation_ratio(n2w, t2),
             pr_norm1_raw=participation_ratio(n1w), pr_norm2_raw=participation_ratio(n2w), logit_rms=0.0)
    if qw is not None and kw is not None:
        c["logit_rms"] = float(scale * (qw.detach().double() * kw.detach().double()).norm())
    return c


def should_escalate(cond: dict) -> bool:
    mode = escalation()
    if mode != "auto":
        return mode == "all"
    return min(cond["pr_norm1"], cond["pr_norm2"]) < ESC_PR_MIN or cond["logit_rms"] > ESC_LOGIT_RMS_MAX


def reset_guards(model) -> None:
    """Reset the adaptive attention switch of every aggregator inside `model` (models/aggregator.py Aggregator.reset_guards):
    the outputs of the next forwards no longer depend on the inputs the model has seen before."""
    for m in model.modules():
        if m.__class__.__name__ == "Aggregator" and hasattr(m, "reset_guards"):
            m.reset_guards()


def plan_escalation(blocks) -> list:
    """blocks: layers.blocks.Block modules in EXECUTION order.  Sets every block's `_x3_request` to "this block or a later one is
    ill-conditioned by its own figures" and returns the list of booleans.  Cheap after the first call (the figures are cached
    per parameter version)."""
    own = [b.own_escalation() for b in blocks]
    need, plan = False, [False] * len(blocks)
    for i in range(len(blocks) - 1, -1, -1):
        need = need or own[i]
        plan[i] = need
        blocks[i]._x3_request = need
    return plan


def check_operand_range(name: str, w: torch.Tensor, dt: torch.dtype) -> None:
    """Weights are converted with a plain cast (no clamp): a value beyond the fp16 range would become inf and poison every
    output.  Called once per pack (blocks.py); the comparison runs on the device, the verdict is read back once."""
    if dt != torch.float16:
        return
    m = float(w.detach().abs().max())
    if not (m <= 65504.0):
        raise ValueError(f"{name}: max |w| = {m:.4g} does not fit fp16 operands; select bf16 operands "
                         "(IGGT_OPERAND_DTYPE=bf16 or precision.set_operand_dtype(torch.bfloat16))")
