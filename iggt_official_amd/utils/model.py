"""Checkpoint ingestion -- reference utils/model.py:27-55 (`align_and_update_state_dicts`, caller demo.py:112-116).

`align_and_update_state_dicts` keeps the reference's contract: of the checkpoint's tensors exactly those whose key exists
in the model with the same shape are returned; shape mismatches, model keys the checkpoint lacks and checkpoint keys the
model lacks are reported through `logger`.  `load_checkpoint` is the whole demo.py:112-119 sequence plus what the MI355X
operand format needs: the official weights were trained under bf16 autocast, the trunk runs on fp16 MFMA operands by
default (iggt_official_amd/precision.py), so every GEMM weight is checked against the fp16 range (|w| <= 65504): weights
beyond it are range-folded exactly at pack time (layers/blocks.py fold_ranges, round 4), a block that still does not fit runs
on bf16 operands alone, and with IGGT_RANGE_FOLD=0 the whole trunk is switched to bf16 operands -- loudly -- as in round 3."""
import logging

import torch

FP16_MAX = 65504.0


def align_and_update_state_dicts(logger, model_state_dict, ckpt_state_dict):
    log = logger or logging.getLogger(__name__)
    result, unloaded, unmatched = {}, [], []
    unused = dict(ckpt_state_dict)
    for key in sorted(model_state_dict.keys()):
        want = model_state_dict[key]
        if key not in ckpt_state_dict:
            unloaded.append(f"*UNLOADED* {key}, Model Shape: {tuple(want.shape)}")
            continue
        have = ckpt_state_dict[key]
        if tuple(have.shape) == tuple(want.shape):
            result[key] = have
            unused.pop(key, None)
        else:
            unmatched.append(f"*UNMATCHED* {key}, Model Shape: {tuple(want.shape)} <-> Ckpt Shape: {tuple(have.shape)}")
    for msg in unloaded:
        log.warning(msg)
    for key in sorted(unused):
        log.warning(f"$UNUSED$ {key}, Ckpt Shape: {tuple(unused[key].shape)}")
    for msg in unmatched:
        log.warning(msg)
    return result


def operand_range_report(state_dict):
    """Largest |w| over the 2-D (GEMM) weights of the trunk and the keys that do not fit fp16."""
    worst, bad = 0.0, []
    for k, v in state_dict.items():
        if not torch.is_tensor(v) or not v.dtype.is_floating_point or v.dim() < 2:
            continue
        if not k.startswith("aggregator."):
            continue
        m = float(v.detach().abs().max())
        if not (m <= FP16_MAX):   # also catches NaN / inf
            bad.append(k)
        worst = max(worst, m) if m == m else float("inf")
    return worst, bad


def load_checkpoint(model, path_or_state_dict, logger=None, map_location="cpu"):
    """demo.py:112-119: read the checkpoint, strip the DDP "module." prefix, align with the model's schema, load (strict=False:
    track_head.* is out of scope, relative-position index buffers are rebuilt), then validate the operand range."""
    from .. import precision

    log = logger or logging.getLogger(__name__)
    sd = torch.load(path_or_state_dict, map_location=map_location) if isinstance(path_or_state_dict, str) \
        else path_or_state_dict
    if "model" in sd and isinstance(sd["model"], dict) and not any(torch.is_tensor(v) for v in sd.values()):
        sd = sd["model"]
    sd = {k.replace("module.", "", 1): v for k, v in sd.items()}
    aligned = align_and_update_state_dicts(log, model.state_dict(), sd)
    missing, unexpected = model.load_state_dict(aligned, strict=False)
    worst, bad = operand_range_report(aligned)
    if bad and precision.operand_dtype() == torch.float16:
        if precision.range_folding():
            # round 4: exact power-of-two range folding at pack time (layers/blocks.py fold_ranges) keeps fp16 operands for
            # weights that are beyond the range only through the checkpoint's parametrisation; a block with entries no partner
            # can absorb runs on bf16 alone
            log.warning(f"{len(bad)} trunk weight tensors exceed the fp16 range (first: {bad[0]}): they are range-folded at "
                        "pack time; blocks that still do not fit run on bf16 operands individually")
        else:
            log.warning(f"{len(bad)} trunk weight tensors exceed the fp16 range (first: {bad[0]}); switching the trunk to "
                        "bf16 operands (precision.set_operand_dtype): outputs then follow the reference's autocast(bf16) mode")
            precision.set_operand_dtype(torch.bfloat16)
    report = {"loaded": len(aligned), "missing": list(missing), "unexpected": list(unexpected), "max_abs_weight": worst,
              "beyond_fp16": bad}
    # round 5: which blocks leave the default arithmetic on THIS checkpoint -- the x3 precision rung (precision.py: blocks that are
    # ill-conditioned by their own LayerNorm / q-k-norm scales and everything upstream of them run on fp16 operand pairs at ~3x
    # their MFMA work) and the blocks that fell to bf16 operands (un-foldable weights).  On a GPU-resident model the packs are
    # built here, so the lists are final; on a CPU-resident one only the escalation plan is (packing happens on the device).
    agg = getattr(model, "aggregator", None)
    if agg is not None and hasattr(agg, "plan_escalation"):
        agg.plan_escalation()
        if next(agg.parameters()).is_cuda:
            for blk in agg.execution_order():
                blk.packed()
        rep = agg.escalation_report()
        report.update(escalated_blocks=rep["x3"], ill_conditioned_blocks=rep["own_verdict"], bf16_blocks=rep["bf16_fallback"],
                      min_participation_ratio=rep["min_participation_ratio"], max_logit_rms=rep["max_logit_rms"])
        if rep["x3"]:
            log.warning(f"{len(rep['x3'])} of {rep['blocks']} transformer blocks run on the x3 precision rung (fp16 operand pairs, "
                        f"~3x their MFMA work): {len(rep['own_verdict'])} are ill-conditioned by their own norm scales "
                        f"(min participation ratio {rep['min_participation_ratio']:.3f}, max predicted logit r.m.s. "
                        f"{rep['max_logit_rms']:.1f}); IGGT_ESCALATE=off keeps single fp16 operands")
        if rep["bf16_fallback"]:
            log.warning(f"blocks on bf16 operands (weights beyond the fp16 range that no partner can absorb): {rep['bf16_fallback']}")
    return report
