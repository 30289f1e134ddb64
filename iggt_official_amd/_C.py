"""ctypes binding of libiggt_hip.so (the C ABI declared in include/iggt_hip.h).

The product path has NO fallback: if the library cannot be loaded, or a tensor is not on a ROCm
device, every op raises.  (The CPU restatement under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes
import os

import torch

# IGGT_HIP_LIB: developer override (A/B builds of the same ABI, probes/); the product always loads the in-tree library
_LIB_PATH = os.environ.get("IGGT_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                           "libiggt_hip.so")
_lib = None

ABI_VERSION = 26

_c_void_p, _c_int, _c_long, _c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> argtypes, mirrors include/iggt_hip.h one to one
_SIGNATURES = {
    "iggt_hip_abi_version": [],
    "iggt_gemm_bf16": [_c_void_p, _c_long, _c_void_p, _c_long, _c_int, _c_int, _c_int,
                       _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_int,
                       _c_int, _c_int, _c_int, _c_void_p],
    "iggt_gemm_f16": [_c_void_p, _c_long, _c_void_p, _c_long, _c_int, _c_int, _c_int,
                      _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_int,
                      _c_int, _c_int, _c_int, _c_void_p],
    "iggt_flash_attn_bf16_d64": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                 _c_long, _c_long, _c_long, _c_long, _c_long, _c_long, _c_long, _c_long,
                                 _c_float, _c_int, _c_void_p],
    "iggt_flash_attn_f16_d64": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                _c_long, _c_long, _c_long, _c_long, _c_long, _c_long, _c_long, _c_long,
                                _c_float, _c_int, _c_void_p],
    "iggt_flash_attn_static_bf16_d64": [_c_void_p] * 4 + [_c_int] * 4 + [_c_long] * 8
                                      + [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p]
                                      + [_c_void_p, _c_long, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_flash_attn_static_f16_d64": [_c_void_p] * 4 + [_c_int] * 4 + [_c_long] * 8
                                      + [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p]
                                      + [_c_void_p, _c_long, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_flash_attn_static_ws_bytes": [_c_int, _c_int, _c_int, _c_int],
    "iggt_flash_attn_static_est_ws_bytes": [_c_int, _c_int, _c_int, _c_int],
    "iggt_flash_attn_static_ksplit": [_c_int, _c_int, _c_int, _c_int],
    "iggt_flash_attn_static_partial_bf16_d64": [_c_void_p] * 3 + [_c_int] * 4 + [_c_long] * 6
                                              + [_c_void_p] * 4 + [_c_int] * 4 + [_c_void_p, _c_int] + [_c_void_p] * 3,
    "iggt_flash_attn_static_partial_f16_d64": [_c_void_p] * 3 + [_c_int] * 4 + [_c_long] * 6
                                              + [_c_void_p] * 4 + [_c_int] * 4 + [_c_void_p, _c_int] + [_c_void_p] * 3,
    "iggt_flash_attn_static_combine_bf16_d64": [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 4 + [_c_int] * 4 + [_c_long] * 8
                                              + [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "iggt_flash_attn_static_combine_f16_d64": [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 4 + [_c_int] * 4 + [_c_long] * 8
                                              + [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "iggt_k_rownorm_max_bf16": [_c_void_p, _c_long, _c_int, _c_void_p, _c_void_p],
    "iggt_k_rownorm_max_f16": [_c_void_p, _c_long, _c_int, _c_void_p, _c_void_p],
    "iggt_flash_attn_d64_kernel_name": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_char_p,
                                        _c_int],
    "iggt_layernorm_f32": [_c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long,
                           _c_int, _c_int, _c_int, _c_float, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_qknorm_rope_bf16": [_c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_long,
                              _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                              _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_long, _c_long, _c_float, _c_void_p,
                              _c_void_p],
    "iggt_qknorm_rope_f16": [_c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_long,
                             _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                             _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_long, _c_long, _c_float, _c_void_p,
                             _c_void_p],
    "iggt_im2row_patch14": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_qkv_split_f16": [_c_void_p, _c_long, _c_void_p, _c_long, _c_long, _c_void_p, _c_long, _c_long, _c_void_p, _c_long,
                           _c_long, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                           _c_int, _c_float, _c_float, _c_void_p],
    "iggt_split3_f16": [_c_void_p, _c_long, _c_void_p, _c_long, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_flash_attn_x3_f16_d64": [_c_void_p] * 7 + [_c_long] + [_c_int] * 4 + [_c_long] * 8 + [_c_void_p],
    "iggt_colmean_h16": [_c_void_p, _c_long, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p],
    "iggt_bias_correct_h16": [_c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p],
    "iggt_head_tail_f32": [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_int,
                           _c_int, _c_void_p],
    "iggt_window_attn_f32": [_c_void_p, _c_long, _c_int, _c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_long,
                             _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_void_p],
    "iggt_dpt_tail_f32": [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                          _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_conv2d_nhwc_f32": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p,
                             _c_int]
                            + [_c_int] * 24 + [_c_void_p],
    "iggt_conv2d_nhwc_f32_ws": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p,
                                _c_int]
                               + [_c_int] * 24 + [_c_void_p, _c_long, _c_void_p],
    "iggt_bilinear_ac_nhwc_f32": [_c_void_p, _c_int, _c_void_p, _c_int] + [_c_int] * 6 + [_c_void_p] * 3,
    "iggt_linear_f32": [_c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p,
                        _c_long, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_linear_f32_ws": [_c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p,
                           _c_long, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_long, _c_void_p],
    "iggt_linear_f32_ws_bytes": [],
    "iggt_attn_f32": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                      _c_long, _c_long, _c_long, _c_long, _c_long, _c_long, _c_long, _c_long, _c_float, _c_void_p],
    "iggt_adaln_modulate_f32": [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p, _c_long,
                                _c_int, _c_int, _c_float, _c_void_p],
    "iggt_pose_update_f32": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p],
    "iggt_conv1x1_c32_nchw_f32": [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_long, _c_int, _c_void_p],
    "iggt_pose_to_extri_intri_f32": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_unproject_depth_f32": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_resize_bicubic_u8": [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,
                               _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p],
    "iggt_u8hwc_to_f32chw": [_c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                             _c_int, _c_float, _c_void_p],
    "iggt_knn_morton_codes": [_c_void_p, _c_long, _c_float, _c_float, _c_float, _c_float, _c_void_p, _c_void_p],
    "iggt_knn_search": [_c_void_p, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "iggt_knn_mean_features_f32": [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p],
    "iggt_moments_f32": [_c_void_p, _c_long, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p],
    "iggt_moments_width": [_c_int],
    "iggt_project3_f32": [_c_void_p, _c_long, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "iggt_stretch3_f32": [_c_void_p, _c_long, _c_void_p, _c_void_p],
    "iggt_nn1_label_f32": [_c_void_p, _c_long, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "iggt_nn1_search_split_f32": [_c_void_p, _c_long, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "iggt_nn1_label_tiled_f32": [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_int,
                                 _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "iggt_count_saturated_h16": [_c_void_p, _c_long, _c_int, _c_int, _c_int, _c_void_p, _c_void_p],
    "iggt_layernorm_rows_f32": [_c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                _c_int, _c_float, _c_void_p],
    "iggt_avgpool2_nhwc_f32": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    "iggt_sample_points_nhwc_f32": [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_long, _c_void_p, _c_long, _c_int,
                                    _c_void_p],
    "iggt_track_corr_f32": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int,
                            _c_void_p, _c_long, _c_void_p],
    "iggt_track_posemb_f32": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_long, _c_void_p, _c_long, _c_int,
                              _c_void_p],
    "iggt_track_tokens_f32": [_c_void_p, _c_void_p, _c_long, _c_int, _c_void_p, _c_long, _c_int, _c_void_p, _c_long,
                              _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_int, _c_float, _c_void_p],
    "iggt_track_update_f32": [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_int, _c_int, _c_float, _c_void_p],
    "iggt_hdbscan_core_dist_f32": [_c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "iggt_hdbscan_nearest_foreign_f32": [_c_void_p] * 8 + [_c_long, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p],
    "iggt_hdbscan_labels_from_mst": [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, ctypes.c_double, _c_int, _c_void_p],
    "iggt_write_special_tokens": [_c_void_p, _c_long, _c_long, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                  _c_int, _c_int, _c_void_p],
}


_LONG_RETURN = {"iggt_flash_attn_static_ws_bytes", "iggt_flash_attn_static_est_ws_bytes", "iggt_linear_f32_ws_bytes"}


class HipExtensionError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def exported_symbols():
    return list(_SIGNATURES)


def load():
    """Load the shared library (once).  Raises HipExtensionError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipExtensionError(
            f"{_LIB_PATH} not found: build it with `python -m iggt_official_amd.build_ext` "
            "(there is no CPU / eager fallback for the IGGT hot path)")
    lib = ctypes.CDLL(_LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _c_long if name in _LONG_RETURN else _c_int
    v = lib.iggt_hip_abi_version()
    if v != ABI_VERSION:
        raise HipExtensionError(f"libiggt_hip ABI {v} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def _check(rc, name):
    if rc != 0:
        raise HipExtensionError(f"{name} failed with code {rc}"
                                + (" (argument contract)" if rc < 0 else " (hipError_t)"))


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise HipExtensionError("IGGT HIP ops need tensors on a ROCm device (no CPU fallback)")


# ------------------------------------------------------------------------------------------------
# thin typed wrappers.  The 16-bit operand format (bf16 or fp16, include/iggt_hip.h) follows the tensors' dtype.
# ------------------------------------------------------------------------------------------------
H16 = (torch.bfloat16, torch.float16)


def _h16(*ts):
    """Common 16-bit dtype of the given tensors -> symbol suffix ('bf16' | 'f16')."""
    dt = ts[0].dtype
    if dt not in H16 or any(t.dtype != dt for t in ts):
        raise HipExtensionError(f"expected matching bf16 or fp16 operands, got {[str(t.dtype) for t in ts]}")
    return "f16" if dt == torch.float16 else "bf16"


def gemm_h16(a, w, out, *, bias=None, gamma=None, add_table=None, accumulate=False, act=0,
              rows_in=0, rows_out=0, row_off=0, M=None):
    """out[row(m)] (=|+=) act(a @ w.T + bias) * gamma (+ add_table).  a [M,K] and w [N,K] bf16 or fp16 (row
    stride ok), out fp32 or the operands' 16-bit type, 2-D with unit column stride."""
    _dev(a, w, out, bias, gamma, add_table)
    sfx = _h16(a, w) if out.dtype == torch.float32 else _h16(a, w, out)
    assert a.stride(-1) == 1 and w.stride(-1) == 1 and out.stride(-1) == 1
    M = a.shape[0] if M is None else M
    N, K = w.shape
    assert a.shape[1] == K
    for t in (bias, gamma, add_table):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    fn = getattr(load(), "iggt_gemm_" + sfx)
    rc = fn(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, N, K,
            _ptr(bias), _ptr(gamma), _ptr(add_table), out.data_ptr(), out.stride(0),
            int(out.dtype == torch.float32), int(accumulate), act, rows_in, rows_out, row_off, _stream())
    _check(rc, "iggt_gemm_" + sfx)
    return out


gemm_bf16 = gemm_h16  # historical name (operands may be bf16 or fp16)


def flash_attn_d64(q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale,
                   q_rows_per_wg=0):
    """Token-major attention; q/k/v/o are bf16 or fp16 tensors (any view) whose data_ptr is element (0,0,0,0)."""
    _dev(q, k, v, o)
    sfx = _h16(q, k, v, o)
    fn = getattr(load(), f"iggt_flash_attn_{sfx}_d64")
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Nq, Nk,
            q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, float(scale), q_rows_per_wg, _stream())
    _check(rc, f"iggt_flash_attn_{sfx}_d64")
    return o


LOG2E = 1.4426950408889634
QKMAX_NUMEL = 32 + 32 * 4096   # iggt_qknorm_rope_*: 32 per-head norm maxima + scratch for the per-block partial maxima


GUARD_WORDS = 8


def new_attn_guard(device):
    """Persistent adaptive-switch state of one static-bound attention call site (include/iggt_hip.h): int32 [8] =
    {state (-1: never measured), redone work items of the last launch (-1: static kernel skipped), work items, calls,
     mode of the static kernel (0: norm bound, 1: estimated shift), rows redone one by one (-1: skipped), 0, 0}."""
    return torch.tensor([-1, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=device)


def _guard_ok(g):
    assert g is None or (g.dtype == torch.int32 and g.numel() >= GUARD_WORDS and g.is_contiguous() and g.is_cuda)


def flash_attn_d64_static(q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, qkmax, flags,
                          q_rows_per_wg=0, part_ws=None, guard=None, guard_prev=None, est_ws=None, key_period=0,
                          key_nspecial=0, est_mode=0):
    """Static-bound attention (include/iggt_hip.h): q carries scale * log2(e), qkmax fp32 [>= 32]: entries 16..31 = per-head
    norm bound of k as written by qknorm_rope(..., q_scale, qkmax) or k_rownorm_max; flags int32 scratch; part_ws: optional
    byte scratch (static_attn_ws_bytes) that lets small grids split the keys into ranges; guard / guard_prev: new_attn_guard
    tensors of this call site / of the same launch one layer earlier (None: always try the static kernel); est_ws: optional
    byte scratch (static_attn_est_ws_bytes) that turns on the row-granular hand-over and the estimated-shift mode, whose key
    sample includes the first key_nspecial keys of every key_period keys; est_mode: the mode when guard is None."""
    _dev(q, k, v, o, qkmax, flags, part_ws, guard, guard_prev, est_ws)
    sfx = _h16(q, k, v, o)
    assert qkmax.dtype == torch.float32 and qkmax.numel() >= 32 and qkmax.is_contiguous()
    assert flags.dtype == torch.int32 and flags.is_contiguous()
    assert part_ws is None or (part_ws.dtype == torch.uint8 and part_ws.is_contiguous())
    assert est_ws is None or (est_ws.dtype == torch.uint8 and est_ws.is_contiguous())
    _guard_ok(guard), _guard_ok(guard_prev)
    fn = getattr(load(), f"iggt_flash_attn_static_{sfx}_d64")
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Nq, Nk,
            q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, qkmax.data_ptr(), flags.data_ptr(), flags.numel(),
            _ptr(part_ws), 0 if part_ws is None else part_ws.numel(), q_rows_per_wg, _ptr(guard), _ptr(guard_prev),
            _ptr(est_ws), 0 if est_ws is None else est_ws.numel(), int(key_period), int(key_nspecial), int(est_mode), _stream())
    _check(rc, f"iggt_flash_attn_static_{sfx}_d64")
    return o


def static_attn_est_ws_bytes(B, H, Nq, Nk):
    """Bytes of scratch for flash_attn_d64_static(..., est_ws=...) at this shape."""
    return int(load().iggt_flash_attn_static_est_ws_bytes(B, H, Nq, Nk))


EST_HI_CAP, EST_KS2, EST_BIAS = 1024, 16, 1048576.0   # csrc/attention_common.h


def static_attn_est_views(est_ws, B, H, Nq, Nk=None):
    """Typed views into an est_ws buffer (csrc/attention_common.h est_view_at; reports, tests and probes): rowshift fp32
    [BH, Nq] (stored + EST_BIAS), rowlist int32 [BH, Nq], rowcount / hicount int32 [BH], hilist int32 [BH, EST_HI_CAP],
    rowflag uint8 [BH, NqP]."""
    Nk = Nq if Nk is None else Nk
    BH, NqP = B * H, (Nq + 15) // 16 * 16
    NqL = ((Nq + 7) // 8 + 127) // 128 * 128
    nWG = (Nk + 31) // 32
    off = [0]

    def take(nbytes):
        a = off[0]
        off[0] += nbytes
        return est_ws[a:a + nbytes]

    v = dict(rowshift=take(BH * Nq * 4).view(torch.float32).view(BH, Nq), rowlist=take(BH * Nq * 4).view(torch.int32).view(BH, Nq),
             rowcount=take(BH * 4).view(torch.int32), hicount=take(BH * 4).view(torch.int32), dense=take(BH * 4).view(torch.int32),
             hilist=take(BH * EST_HI_CAP * 4).view(torch.int32).view(BH, EST_HI_CAP))
    take(BH * nWG * 4), take(BH * nWG * 16)
    NqS = (Nq + 255) // 256 * 256
    v["slotrow"] = take(BH * NqS * 4).view(torch.int32).view(BH, NqS)
    take(EST_KS2 * BH * NqL * 4), take(EST_KS2 * BH * NqL * 4)
    off[0] = (off[0] + 15) // 16 * 16
    take(EST_KS2 * BH * NqL * 128)
    v["rowflag"] = take(BH * NqP).view(BH, NqP)
    v["NqL"] = NqL
    v["bytes"] = off[0]
    return v


def static_attn_ws_bytes(B, H, Nq, Nk):
    """Bytes of partial workspace with which flash_attn_d64_static may split the keys of this shape (0: it never would)."""
    return int(load().iggt_flash_attn_static_ws_bytes(B, H, Nq, Nk))


def static_attn_ksplit(B, H, Nq, Nk):
    """Number of key ranges the static-bound dispatcher would cut this shape into (1: one pass)."""
    return int(load().iggt_flash_attn_static_ksplit(B, H, Nq, Nk))


def flash_attn_d64_static_partial(q, k, v, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, qkmax, o_part, l_part, c_part,
                                  slot0, ksplit, q_rows_per_wg=0, guard=None, guard_prev=None, seg_len=0, skip_seg=-1,
                                  seg_kmax=None):
    """Keys (k, v, Nk) -> partial slots [slot0, slot0 + ksplit) as `ksplit` equal key ranges, or (seg_len > 0) as the key
    segments of seg_len rows, leaving segment skip_seg out (include/iggt_hip.h)."""
    _dev(q, k, v, qkmax, o_part, l_part, c_part, guard, guard_prev, seg_kmax)
    assert seg_kmax is None or (seg_kmax.dtype == torch.float32 and seg_kmax.is_contiguous() and seg_kmax.numel() >= 32 * ksplit)
    sfx = _h16(q, k, v, o_part)
    assert l_part.dtype == torch.float32 and c_part.dtype == torch.float32 and l_part.shape == c_part.shape
    assert o_part.is_contiguous() and l_part.is_contiguous() and c_part.is_contiguous()
    _guard_ok(guard), _guard_ok(guard_prev)
    fn = getattr(load(), f"iggt_flash_attn_static_partial_{sfx}_d64")
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs,
            _ptr(qkmax), o_part.data_ptr(), l_part.data_ptr(), c_part.data_ptr(), slot0, ksplit, seg_len, skip_seg,
            _ptr(seg_kmax), q_rows_per_wg, _ptr(guard), _ptr(guard_prev), _stream())
    _check(rc, f"iggt_flash_attn_static_partial_{sfx}_d64")


def flash_attn_d64_static_combine(o_part, l_part, c_part, nslots, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs,
                                  v_rs, o_bs, o_rs, flags, q_rows_per_wg=0, guard=None, guard_prev=None):
    """Fold nslots partial slots into o + flagged-tile fallback over the full key set (include/iggt_hip.h)."""
    _dev(o_part, l_part, c_part, q, k, v, o, flags, guard, guard_prev)
    sfx = _h16(q, k, v, o, o_part)
    _guard_ok(guard), _guard_ok(guard_prev)
    fn = getattr(load(), f"iggt_flash_attn_static_combine_{sfx}_d64")
    rc = fn(o_part.data_ptr(), l_part.data_ptr(), c_part.data_ptr(), nslots, q.data_ptr(), k.data_ptr(), v.data_ptr(),
            o.data_ptr(), B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, flags.data_ptr(), flags.numel(),
            q_rows_per_wg, _ptr(guard), _ptr(guard_prev), _stream())
    _check(rc, f"iggt_flash_attn_static_combine_{sfx}_d64")
    return o


def k_rownorm_max(k, qkmax):
    """qkmax[16 + h] = max row norm of head h over the rows of k (16-bit [rows, 1024], row stride ok)."""
    _dev(k, qkmax)
    sfx = _h16(k)
    assert k.dim() == 2 and k.shape[1] == 1024 and k.stride(1) == 1
    assert qkmax.dtype == torch.float32 and qkmax.numel() >= QKMAX_NUMEL and qkmax.is_contiguous()
    fn = getattr(load(), "iggt_k_rownorm_max_" + sfx)
    _check(fn(k.data_ptr(), k.stride(0), k.shape[0], qkmax.data_ptr(), _stream()), "iggt_k_rownorm_max_" + sfx)
    return qkmax


def attn_kernel_label(B, H, Nq, Nk, operand_name, static_bound=False, q_rows_per_wg=0, with_part_ws=False):
    """Name of the attention kernel instantiation the dispatcher launches for this shape (for reports)."""
    buf = ctypes.create_string_buffer(160)
    rc = load().iggt_flash_attn_d64_kernel_name(B, H, Nq, Nk, int(operand_name in ("f16", "fp16")), int(static_bound),
                                                int(with_part_ws), q_rows_per_wg, buf, 160)
    _check(rc, "iggt_flash_attn_d64_kernel_name")
    return buf.value.decode()


def layernorm(x0, w, b, out, eps, *, x1=None, rows=None, rows_in=0, rows_stride=0, row_off=0,
              orows_stride=0, orow_off=0, ldx=None, ldo=None, split3=False):
    """LayerNorm over the last dim of x0 (or of concat(x0, x1)); fp32 in, bf16 / fp16 / fp32 out [rows, C].
    split3: out is fp16 [rows, 3 C] = [hi | lo | hi] (the x3 precision rung, include/iggt_hip.h)."""
    _dev(x0, x1, w, b, out)
    assert x0.dtype == torch.float32 and x0.stride(-1) == 1 and out.stride(-1) == 1
    C = x0.shape[-1] * (2 if x1 is not None else 1)
    if split3:
        assert out.dtype == torch.float16 and out.shape[-1] == 3 * C
    rows = out.shape[0] if rows is None else rows
    ld0 = x0.stride(-2) if ldx is None else ldx
    ld1 = 0 if x1 is None else (x1.stride(-2) if ldx is None else ldx)
    rc = load().iggt_layernorm_f32(x0.data_ptr(), ld0, _ptr(x1), ld1,
                                   w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                   out.stride(-2) if ldo is None else ldo,
                                   3 if split3 else {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}[out.dtype], rows, C,
                                   float(eps),
                                   rows_in, rows_stride, row_off, orows_stride, orow_off, _stream())
    _check(rc, "iggt_layernorm_f32")
    return out


def qknorm_rope(qkv, q_out, k_out, v_out, qw, qb, kw, kb, cos_t, sin_t, T, P, gw, patch_start, eps,
                heads_per_group=0, k_group_stride=0, v_group_stride=0, q_scale=1.0, qkmax=None):
    _dev(qkv, q_out, k_out, v_out, qw, cos_t, qkmax)
    assert qkmax is None or (qkmax.dtype == torch.float32 and qkmax.numel() >= QKMAX_NUMEL and qkmax.is_contiguous())
    sfx = _h16(qkv, q_out, k_out) if v_out is None else _h16(qkv, q_out, k_out, v_out)
    assert qkv.shape[-1] == 3072
    fn = getattr(load(), "iggt_qknorm_rope_" + sfx)
    rc = fn(qkv.data_ptr(), qkv.stride(0), q_out.data_ptr(), q_out.stride(0),
            k_out.data_ptr(), k_out.stride(0), _ptr(v_out), 0 if v_out is None else v_out.stride(0),
            qw.data_ptr(), qb.data_ptr(), kw.data_ptr(), kb.data_ptr(),
            cos_t.data_ptr(), sin_t.data_ptr(), T, P, gw, patch_start, float(eps),
            heads_per_group, k_group_stride, v_group_stride, float(q_scale), _ptr(qkmax), _stream())
    _check(rc, "iggt_qknorm_rope_" + sfx)


def im2row_patch14(img, out, S, H, W, Kpad, split3=False):
    """split3: out fp16 [rows, 3 Kpad] = [hi | lo | hi] (x3 precision rung)."""
    _dev(img, out)
    assert img.dtype == torch.float32 and img.is_contiguous() and out.dtype in H16 and out.is_contiguous()
    assert not split3 or (out.dtype == torch.float16 and out.shape[-1] == 3 * Kpad)
    rc = load().iggt_im2row_patch14(img.data_ptr(), out.data_ptr(), 2 if split3 else int(out.dtype == torch.float16), S, H, W,
                                    Kpad, _stream())
    _check(rc, "iggt_im2row_patch14")
    return out


# ---- x3 precision rung (include/iggt_hip.h, csrc/x3.hip): fp16 hi + lo operand pairs, three MFMA passes per product ------------
def qkv_split(qkv32, q_out, q_lo, k_out, k_lo, v_out, v_lo, *, qw=None, qb=None, kw=None, kb=None, cos_t=None, sin_t=None,
              P=1, gw=1, patch_start=0, eps=1e-5, q_scale=1.0):
    """qkv32 fp32 [T, 3072] -> optional q/k LayerNorm(64) + RoPE -> q * q_scale, k, v as fp16 pairs: hi into q_out / k_out / v_out
    ([T, 1024] views, row stride ok), lo the given number of ELEMENTS behind each."""
    _dev(qkv32, q_out, k_out, v_out, qw, cos_t)
    assert qkv32.dtype == torch.float32 and qkv32.shape[-1] == 3072 and qkv32.stride(-1) == 1
    for t in (q_out, k_out, v_out):
        assert t.dtype == torch.float16 and t.stride(-1) == 1
    for t in (qw, qb, kw, kb, cos_t, sin_t):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    rc = load().iggt_qkv_split_f16(qkv32.data_ptr(), qkv32.stride(0), q_out.data_ptr(), q_out.stride(0), int(q_lo),
                                   k_out.data_ptr(), k_out.stride(0), int(k_lo), v_out.data_ptr(), v_out.stride(0), int(v_lo),
                                   _ptr(qw), _ptr(qb), _ptr(kw), _ptr(kb), _ptr(cos_t), _ptr(sin_t), qkv32.shape[0], int(P),
                                   int(gw), int(patch_start), float(eps), float(q_scale), _stream())
    _check(rc, "iggt_qkv_split_f16")


def split3(x, out, act=0):
    """x fp32 [rows, N] -> (act = 1: exact GELU) -> out fp16 [rows, 3 N] = [hi | lo | hi]."""
    _dev(x, out)
    assert x.dtype == torch.float32 and out.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1
    assert out.shape == (x.shape[0], 3 * x.shape[1])
    _check(load().iggt_split3_f16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], int(act),
                                  _stream()), "iggt_split3_f16")
    return out


def flash_attn_x3(q, q_lo, k, k_lo, v, v_lo, o, o_seg, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs):
    """Attention on fp16 hi / lo pairs (q carries scale * log2 e); o gets [hi | lo | hi] segments o_seg elements apart (0: hi only)."""
    _dev(q, q_lo, k, k_lo, v, v_lo, o)
    for t in (q, q_lo, k, k_lo, v, v_lo, o):
        assert t.dtype == torch.float16
    rc = load().iggt_flash_attn_x3_f16_d64(q.data_ptr(), q_lo.data_ptr(), k.data_ptr(), k_lo.data_ptr(), v.data_ptr(),
                                           v_lo.data_ptr(), o.data_ptr(), int(o_seg), B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs,
                                           v_bs, v_rs, o_bs, o_rs, _stream())
    _check(rc, "iggt_flash_attn_x3_f16_d64")
    return o


def colmean(x, mu, row_step=1):
    """mu[k] = mean of x[::row_step, k]; x 16-bit [rows, K] (row stride ok), mu fp32 [K]."""
    _dev(x, mu)
    assert x.dtype in H16 and x.stride(-1) == 1 and mu.dtype == torch.float32 and mu.is_contiguous()
    rc = load().iggt_colmean_h16(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], row_step,
                                 int(x.dtype == torch.float16), mu.data_ptr(), _stream())
    _check(rc, "iggt_colmean_h16")
    return mu


def bias_correct(dw, mu, bias, out):
    """out = (bias or 0) + dw @ mu; dw 16-bit [N, K], mu fp32 [K], bias / out fp32 [N]."""
    _dev(dw, mu, bias, out)
    assert dw.dtype in H16 and dw.stride(-1) == 1 and mu.dtype == torch.float32 and out.dtype == torch.float32
    rc = load().iggt_bias_correct_h16(dw.data_ptr(), dw.stride(0), dw.shape[0], dw.shape[1], mu.data_ptr(),
                                      _ptr(bias), out.data_ptr(), int(dw.dtype == torch.float16), _stream())
    _check(rc, "iggt_bias_correct_h16")
    return out


HEAD_ACT = {"linear": 0, "exp": 1, "relu": 2, "inv_log": 3, "sigmoid": 4, "norm": 5}
CONF_ACT = {"expp1": 0, "expp0": 1, "sigmoid": 2}


def head_tail(x, w, b, activation, conf_activation):
    """x NHWC fp32 [..., 32] -> (pts [..., Cout-1], conf [...]): 1x1 conv + activate_head (include/iggt_hip.h)."""
    _dev(x, w, b)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] >= 32
    Cout = w.shape[0]
    assert w.shape[1] == 32 and w.is_contiguous() and b.is_contiguous() and w.dtype == torch.float32
    lead = x.shape[:-1]
    npix = x.numel() // x.shape[-1]
    pts = torch.empty(*lead, Cout - 1, dtype=torch.float32, device=x.device)
    conf = torch.empty(*lead, dtype=torch.float32, device=x.device)
    rc = load().iggt_head_tail_f32(x.data_ptr(), x.shape[-1], w.data_ptr(), b.data_ptr(), pts.data_ptr(),
                                   conf.data_ptr(), npix, Cout, HEAD_ACT[activation], CONF_ACT[conf_activation],
                                   _stream())
    _check(rc, "iggt_head_tail_f32")
    return pts, conf


def window_attn(q, k, v, out, heads, head_dim, scale, *, q_windows=False, ow=8, pad=0, bias=None):
    """Part-head window attention (include/iggt_hip.h).  k, v, out: NHWC fp32 views [b,h,w,>=head_dim*heads] (channel
    slices allowed, unit channel stride); q: the same kind of map, or window-major [nW,64,C] when q_windows."""
    _dev(q, k, v, out, bias)
    for t in (q, k, v, out):
        assert t.dtype == torch.float32 and t.stride(-1) == 1
    b, h, w = k.shape[:3]

    def ld(t):   # NHWC view of a contiguous [b,h,w,ld] buffer: pixel stride
        assert t.stride(1) == t.stride(2) * t.shape[2] and t.stride(0) == t.stride(1) * t.shape[1]
        return t.stride(2)

    if q_windows:
        assert q.dim() == 3 and q.shape[1] == 64 and q.is_contiguous() and q.shape[0] == b * (h // 8) * (w // 8)
        q_ld = q.stride(1)
    else:
        q_ld = ld(q)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.shape == (heads, ow * ow, 64)
    rc = load().iggt_window_attn_f32(q.data_ptr(), q_ld, int(q_windows), k.data_ptr(), ld(k), v.data_ptr(), ld(v),
                                     out.data_ptr(), ld(out), _ptr(bias), b, h, w, heads, head_dim, ow, pad,
                                     float(scale), _stream())
    _check(rc, "iggt_window_attn_f32")
    return out


def dpt_tail(x, size, xpart, ypart, w_hi, w_lo, b1, w2, b2, activation, conf_activation, nchw=False):
    """x NHWC fp32 [N,Hi,Wi,128] -> (pts [N,Ho,Wo,Cout-1], conf [N,Ho,Wo]): upsample + position map + conv3x3 + ReLU +
    conv1x1 + activate_head in one kernel (include/iggt_hip.h).  nchw=True: the part head's tail -- all Cout channels
    un-activated as [N,Cout,Ho,Wo]; returns that single tensor."""
    _dev(x, xpart, ypart, w_hi, w_lo, b1, w2, b2)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == 128
    N, Hi, Wi, _ = x.shape
    Ho, Wo = size
    Cout = w2.shape[0]
    assert w_hi.dtype == torch.bfloat16 and w_hi.shape == (32, 9 * 128) and w_hi.is_contiguous() and w_lo.is_contiguous()
    assert w2.shape == (Cout, 32) and w2.dtype == torch.float32 and w2.is_contiguous()
    if xpart is not None:
        assert xpart.shape == (Wo, 64) and ypart.shape == (Ho, 64) and xpart.is_contiguous() and ypart.is_contiguous()
    if nchw:
        out = torch.empty(N, Cout, Ho, Wo, dtype=torch.float32, device=x.device)
        rc = load().iggt_dpt_tail_f32(x.data_ptr(), N, Hi, Wi, Ho, Wo, _ptr(xpart), _ptr(ypart), w_hi.data_ptr(),
                                      w_lo.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(), 0, Cout, 0, 0, 1,
                                      _stream())
        _check(rc, "iggt_dpt_tail_f32")
        return out
    pts = torch.empty(N, Ho, Wo, Cout - 1, dtype=torch.float32, device=x.device)
    conf = torch.empty(N, Ho, Wo, dtype=torch.float32, device=x.device)
    rc = load().iggt_dpt_tail_f32(x.data_ptr(), N, Hi, Wi, Ho, Wo, _ptr(xpart), _ptr(ypart), w_hi.data_ptr(),
                                  w_lo.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), pts.data_ptr(),
                                  conf.data_ptr(), Cout, HEAD_ACT[activation], CONF_ACT[conf_activation], 0, _stream())
    _check(rc, "iggt_dpt_tail_f32")
    return pts, conf


def write_special_tokens(dst, src0, src1, S, nrows, row_off, first_view_is_zero):
    """dst: fp32 [S, P, C] contiguous rows."""
    _dev(dst, src0, src1)
    assert dst.dtype == torch.float32 and dst.dim() == 3 and dst.stride(2) == 1
    rc = load().iggt_write_special_tokens(dst.data_ptr(), dst.stride(0), dst.stride(1), src0.data_ptr(),
                                          src1.data_ptr(), S, nrows, row_off, dst.shape[2],
                                          int(first_view_is_zero), _stream())
    _check(rc, "iggt_write_special_tokens")


def conv2d_nhwc(x, w_hi, w_lo, bias, y, *, KH, KW, stride=1, pad_y=0, pad_x=0, Ho=None, Wo=None, res=None, res2=None,
                relu_in=False, relu_res=False, act=0, prec=3, Cin=None, Cout=None, cout_phys=None, ps=1,
                osy=1, osx=1, ooy=0, oox=0):
    """x [N,Hi,Wi,ldx] fp32 NHWC, y [N,Hout,Wout,ldy] fp32 NHWC (see include/iggt_hip.h)."""
    _dev(x, w_hi, w_lo, bias, y, res, res2)
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.stride(3) == 1 and y.stride(3) == 1
    assert x.is_contiguous() and y.is_contiguous() and (res is None or res.is_contiguous())
    N, Hi, Wi, ldx = x.shape
    _, Hout, Wout, ldy = y.shape
    Cin = ldx if Cin is None else Cin
    Cout = w_hi.shape[0] if Cout is None else Cout
    cout_phys = Cout if cout_phys is None else cout_phys
    Ho = Hout if Ho is None else Ho
    Wo = Wout if Wo is None else Wo
    assert w_hi.dtype == (torch.float16 if prec == 2 else torch.bfloat16) and w_hi.is_contiguous()
    assert w_hi.shape[1] == KH * KW * Cin and (w_lo is None or (w_lo.dtype == torch.bfloat16 and w_lo.shape == w_hi.shape))
    ws = _conv_ws(x.device)
    rc = load().iggt_conv2d_nhwc_f32_ws(x.data_ptr(), ldx, w_hi.data_ptr(), _ptr(w_lo), _ptr(bias), _ptr(res),
                                        _ptr(res2), 0 if res is None else res.shape[3], y.data_ptr(), ldy, N, Hi, Wi, Cin, Ho,
                                        Wo, Cout, KH, KW, stride, pad_y, pad_x, Hout, Wout, osy, osx, ooy, oox, cout_phys,
                                        ps, int(relu_in), int(relu_res), act, prec, ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "iggt_conv2d_nhwc_f32_ws")
    return y


_CONV_WS = {}
CONV_WS_BYTES = 64 << 20


def _conv_ws(device):
    """Split-K scratch of iggt_conv2d_nhwc_f32_ws: one buffer per (device, stream).  Launches on one stream are ordered, so
    they can share the partial-sum buffer; a second stream (another model, an eager forward beside a graph replay, a hipGraph
    capture -- every capture runs on its own stream, graphs.py) gets its own.  Never freed: captured graphs hold the address."""
    key = (device, _stream())
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.empty(CONV_WS_BYTES, dtype=torch.uint8, device=device)
    return ws


def bilinear_ac_nhwc(x, y, xpart=None, ypart=None):
    """align_corners=True bilinear resize x [N,Hi,Wi,C] -> y [N,Ho,Wo,C] (fp32 NHWC), optional position map."""
    _dev(x, y, xpart, ypart)
    assert x.is_contiguous() and y.is_contiguous() and x.dtype == torch.float32
    N, Hi, Wi, C = x.shape
    _, Ho, Wo, _ = y.shape
    rc = load().iggt_bilinear_ac_nhwc_f32(x.data_ptr(), C, y.data_ptr(), C, N, Hi, Wi, Ho, Wo, C, _ptr(xpart),
                                          _ptr(ypart), _stream())
    _check(rc, "iggt_bilinear_ac_nhwc_f32")
    return y


# ------------------------------------------------------------------------------------------------
# fp32 small operators of the heads (csrc/smallops.hip)
# ------------------------------------------------------------------------------------------------
LIN_ACT = {None: 0, "none": 0, "gelu": 1, "relu": 2, "silu": 3, "sigmoid": 4}


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise HipExtensionError(f"{name} must be a contiguous fp32 tensor")
    return t


def linear_f32(x, weight, bias=None, *, act=None, gamma=None, res=None, out=None):
    """act(x @ weight.T + bias) * gamma (+ res), exact fp32 (skinny problems: M = x rows is small).  x [M, K] (unit
    column stride, any row stride), weight [N, K] contiguous, res / out [M, N]; res may be `out` itself."""
    _dev(x, weight, bias, gamma, res, out)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    weight = _f32c(weight, "weight")
    M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    if res is not None:
        assert res.dtype == torch.float32 and res.shape == (M, N) and res.stride(1) == 1
    ws = _linear_ws(x.device)
    if os.environ.get("IGGT_LINEAR_SPLITK", "1") == "0":    # A/B runs: one workgroup per output tile
        ws = torch.empty(0, dtype=torch.uint8, device=x.device)
    rc = load().iggt_linear_f32_ws(x.data_ptr(), x.stride(0), weight.data_ptr(), K, _ptr(_f32c(bias, "bias")),
                                   _ptr(_f32c(gamma, "gamma")), _ptr(res), 0 if res is None else res.stride(0),
                                   out.data_ptr(), out.stride(0), M, N, K, LIN_ACT[act], ws.data_ptr() if ws.numel() else 0, ws.numel(),
                                   _stream())
    _check(rc, "iggt_linear_f32_ws")
    return out


_LINEAR_WS = {}


def _linear_ws(device):
    """Split-K scratch of iggt_linear_f32_ws: one zero-filled buffer per (device, stream), see _conv_ws."""
    key = (device, _stream())
    ws = _LINEAR_WS.get(key)
    if ws is None:
        ws = _LINEAR_WS[key] = torch.zeros(load().iggt_linear_f32_ws_bytes(), dtype=torch.uint8, device=device)
    return ws


def attn_f32(q, k, v, o, B, H, Nq, Nk, head_dim, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale):
    """fp32 attention, token-major strided operands (see include/iggt_hip.h)."""
    _dev(q, k, v, o)
    for t in (q, k, v, o):
        assert t.dtype == torch.float32
    rc = load().iggt_attn_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Nq, Nk, head_dim,
                              q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, float(scale), _stream())
    _check(rc, "iggt_attn_f32")
    return o


def adaln_modulate(x, shift, scale, gate, eps, out=None):
    """gate * (LN_noaffine(x) * (1 + scale) + shift) + x; shift / scale / gate: column slices of one [rows, 3C] matrix."""
    _dev(x, shift, scale, gate)
    rows, C = x.shape
    assert x.stride(1) == 1 and shift.stride(0) == scale.stride(0) == gate.stride(0) and shift.stride(1) == 1
    if out is None:
        out = torch.empty(rows, C, dtype=torch.float32, device=x.device)
    rc = load().iggt_adaln_modulate_f32(x.data_ptr(), x.stride(0), shift.data_ptr(), scale.data_ptr(), gate.data_ptr(),
                                        shift.stride(0), out.data_ptr(), out.stride(0), rows, C, float(eps), _stream())
    _check(rc, "iggt_adaln_modulate_f32")
    return out


def pose_update(delta, pred, out, first):
    _dev(delta, pred, out)
    n = delta.shape[0]
    assert delta.shape == (n, 9) and delta.is_contiguous() and pred.is_contiguous() and out.is_contiguous()
    rc = load().iggt_pose_update_f32(delta.data_ptr(), pred.data_ptr(), out.data_ptr(), n, int(first), _stream())
    _check(rc, "iggt_pose_update_f32")
    return out


def conv1x1_c32_nchw(x, w, b):
    """x NHWC fp32 [N,H,W,>=32] contiguous -> [N,Cout,H,W] (Cout <= 8)."""
    _dev(x, w, b)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, H, W, ld = x.shape
    Cout = w.shape[0]
    w = _f32c(w.reshape(Cout, -1), "w")
    assert w.shape[1] == 32
    y = torch.empty(N, Cout, H, W, dtype=torch.float32, device=x.device)
    rc = load().iggt_conv1x1_c32_nchw_f32(x.data_ptr(), ld, w.data_ptr(), _ptr(_f32c(b, "b")), y.data_ptr(), H * W,
                                          N * H * W, Cout, _stream())
    _check(rc, "iggt_conv1x1_c32_nchw_f32")
    return y


def pose_to_extri_intri(pose, H, W, build_intrinsics=True):
    """pose [..., 9] fp32 -> (extrinsics [..., 3, 4], intrinsics [..., 3, 3] or None)."""
    _dev(pose)
    p2 = pose.reshape(-1, 9).float().contiguous()
    n = p2.shape[0]
    extri = torch.empty(n, 3, 4, dtype=torch.float32, device=pose.device)
    intri = torch.empty(n, 3, 3, dtype=torch.float32, device=pose.device) if build_intrinsics else None
    rc = load().iggt_pose_to_extri_intri_f32(p2.data_ptr(), extri.data_ptr(), _ptr(intri), n, int(H), int(W), _stream())
    _check(rc, "iggt_pose_to_extri_intri_f32")
    lead = pose.shape[:-1]
    return extri.view(*lead, 3, 4), (None if intri is None else intri.view(*lead, 3, 3))


def unproject_depth(depth, extri, intri):
    """depth [S,H,W] fp32, extri [S,3,4], intri [S,3,3] -> world points [S,H,W,3] fp32."""
    _dev(depth, extri, intri)
    depth, extri, intri = _f32c(depth, "depth"), _f32c(extri, "extri"), _f32c(intri, "intri")
    S, H, W = depth.shape
    out = torch.empty(S, H, W, 3, dtype=torch.float32, device=depth.device)
    rc = load().iggt_unproject_depth_f32(depth.data_ptr(), extri.data_ptr(), intri.data_ptr(), out.data_ptr(), S, H, W,
                                         _stream())
    _check(rc, "iggt_unproject_depth_f32")
    return out


def resize_bicubic_u8(img, hbounds, hkk, vbounds, vkk, Ho, Wo):
    """img uint8 [Hi,Wi,3] (device) -> uint8 [Ho,Wo,3], Pillow-exact bicubic; tables: int32 device tensors."""
    _dev(img, hbounds, hkk, vbounds, vkk)
    assert img.dtype == torch.uint8 and img.is_contiguous() and img.dim() == 3 and img.shape[2] == 3
    for t in (hbounds, hkk, vbounds, vkk):
        assert t.dtype == torch.int32 and t.is_contiguous()
    Hi, Wi, _ = img.shape
    tmp = torch.empty(Hi, Wo, 3, dtype=torch.uint8, device=img.device)
    out = torch.empty(Ho, Wo, 3, dtype=torch.uint8, device=img.device)
    rc = load().iggt_resize_bicubic_u8(img.data_ptr(), Hi, Wi, hbounds.data_ptr(), hkk.data_ptr(), hkk.shape[1],
                                       vbounds.data_ptr(), vkk.data_ptr(), vkk.shape[1], tmp.data_ptr(), out.data_ptr(),
                                       Ho, Wo, _stream())
    _check(rc, "iggt_resize_bicubic_u8")
    return out


def u8hwc_to_f32chw(src, dst, crop_y, crop_x, pad_y, pad_x, h, w, pad_value=1.0):
    """dst fp32 [3,Hd,Wd] <- src uint8 [Hs,Ws,3] window (crop, h x w) at (pad_y, pad_x), / 255, rest = pad_value."""
    _dev(src, dst)
    assert src.dtype == torch.uint8 and src.is_contiguous() and dst.dtype == torch.float32 and dst.is_contiguous()
    rc = load().iggt_u8hwc_to_f32chw(src.data_ptr(), src.shape[0], src.shape[1], dst.data_ptr(), dst.shape[1], dst.shape[2],
                                     crop_y, crop_x, pad_y, pad_x, h, w, float(pad_value), _stream())
    _check(rc, "iggt_u8hwc_to_f32chw")
    return dst


def count_saturated(x, counter):
    """counter (int64 [1], device) += number of saturated (fp16: |x| = 65504) or non-finite entries of the 16-bit matrix x."""
    _dev(x, counter)
    assert x.dtype in H16 and x.dim() == 2 and x.stride(1) == 1 and counter.dtype == torch.int64
    rc = load().iggt_count_saturated_h16(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], int(x.dtype == torch.float16),
                                         counter.data_ptr(), _stream())
    _check(rc, "iggt_count_saturated_h16")


# ---- post-processing behind the forward path (csrc/postprocess.hip) -----------------------------------------------------
def _f32_all(*ts):
    for t in ts:
        assert t.dtype == torch.float32 and t.is_contiguous(), "fp32 contiguous tensors expected"


def knn_morton_codes(points, center, inv_cell):
    """points fp32 [M,3] (device) -> int32 [M] Morton codes on a 1024^3 grid around `center` (3 floats), cell 1/inv_cell."""
    _dev(points)
    _f32_all(points)
    M = points.shape[0]
    codes = torch.empty(M, dtype=torch.int32, device=points.device)
    rc = load().iggt_knn_morton_codes(points.data_ptr(), M, float(center[0]), float(center[1]), float(center[2]),
                                      float(inv_cell), codes.data_ptr(), _stream())
    _check(rc, "iggt_knn_morton_codes")
    return codes


def knn_search(points, order, k, want_dist=False):
    """Exact kNN (self excluded) of points fp32 [M,3] given order = argsort of their Morton codes (int64 [M]).
    -> idx int32 [M,k] (ascending distance, -1 = no such neighbour) and, on request, squared distances fp32 [M,k]."""
    _dev(points, order)
    _f32_all(points)
    assert order.dtype == torch.int64 and order.is_contiguous() and order.shape[0] == points.shape[0]
    M = points.shape[0]
    nt = (M + 255) // 256
    ws = torch.empty(nt * 256 * 4, dtype=torch.float32, device=points.device)
    boxes = torch.empty(nt * 6, dtype=torch.float32, device=points.device)
    idx = torch.empty(M, k, dtype=torch.int32, device=points.device)
    d2 = torch.empty(M, k, dtype=torch.float32, device=points.device) if want_dist else None
    rc = load().iggt_knn_search(points.data_ptr(), order.data_ptr(), M, int(k), ws.data_ptr(), boxes.data_ptr(),
                                idx.data_ptr(), _ptr(d2), _stream())
    _check(rc, "iggt_knn_search")
    return (idx, d2) if want_dist else idx


def knn_mean_features(feat, idx):
    """feat fp32 [M,F], idx int32 [M,k] -> fp32 [M,F]: mean of the valid neighbours' features (0 where none)."""
    _dev(feat, idx)
    _f32_all(feat)
    assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.shape[0] == feat.shape[0]
    out = torch.empty_like(feat)
    rc = load().iggt_knn_mean_features_f32(feat.data_ptr(), idx.data_ptr(), feat.shape[0], idx.shape[1], feat.shape[1],
                                           out.data_ptr(), _stream())
    _check(rc, "iggt_knn_mean_features_f32")
    return out


def moments(x, shift, nblocks=512):
    """x fp32 [M,C] (C <= 16), shift fp32 [C] -> (sum(x - shift) fp64 [C], sum((x-shift)(x-shift)^T) fp64 [C,C])."""
    _dev(x, shift)
    _f32_all(x, shift)
    M, C = x.shape
    width = load().iggt_moments_width(C)
    if width < 0:
        raise HipExtensionError(f"iggt_moments_f32 supports at most 16 channels (got {C})")
    nblocks = max(1, min(nblocks, (M + 255) // 256))
    part = torch.empty(nblocks, width, dtype=torch.float32, device=x.device)
    rc = load().iggt_moments_f32(x.data_ptr(), M, C, shift.data_ptr(), part.data_ptr(), nblocks, _stream())
    _check(rc, "iggt_moments_f32")
    tot = part.double().sum(0)
    ct = 4 if C <= 4 else (8 if C <= 8 else 16)
    s1 = tot[:C].clone()
    iu = torch.triu_indices(ct, ct, device=x.device)
    g = torch.zeros(ct, ct, dtype=torch.float64, device=x.device)
    g[iu[0], iu[1]] = tot[ct:]
    g = g + g.T - torch.diag(torch.diagonal(g))
    return s1, g[:C, :C].contiguous()


def project3(x, v):
    """x fp32 [M,C] @ v fp32 [C,3] -> fp32 [M,3]."""
    _dev(x, v)
    _f32_all(x, v)
    assert v.shape == (x.shape[1], 3)
    out = torch.empty(x.shape[0], 3, dtype=torch.float32, device=x.device)
    rc = load().iggt_project3_f32(x.data_ptr(), x.shape[0], x.shape[1], v.data_ptr(), out.data_ptr(), _stream())
    _check(rc, "iggt_project3_f32")
    return out


def stretch3(img, lohi):
    """In place: img fp32 [M,3] channel j -> clamp((v - lohi[j]) / (lohi[3+j] - lohi[j]), 0, 1) (0.5 if degenerate)."""
    _dev(img, lohi)
    _f32_all(img, lohi)
    assert img.shape[1] == 3 and lohi.numel() == 6
    rc = load().iggt_stretch3_f32(img.data_ptr(), img.shape[0], lohi.data_ptr(), _stream())
    _check(rc, "iggt_stretch3_f32")
    return img


def nn1_label(query, ref, ref_labels):
    """labels int32 [Mq] of the nearest (first minimum of the squared distance) row of ref fp32 [Mr,C] for every row of query."""
    _dev(query, ref, ref_labels)
    _f32_all(query, ref)
    assert ref_labels.dtype == torch.int32 and ref_labels.is_contiguous() and query.shape[1] == ref.shape[1]
    Mq, Mr = query.shape[0], ref.shape[0]
    qtiles, rtiles = (Mq + 255) // 256, (Mr + 255) // 256
    nsplit = min(rtiles, max(1, 2048 // qtiles))      # enough workgroups to fill the chip when the queries alone do not
    if nsplit > 1:
        d2 = torch.empty(nsplit, Mq, dtype=torch.float32, device=query.device)
        bi = torch.empty(nsplit, Mq, dtype=torch.int32, device=query.device)
        rc = load().iggt_nn1_search_split_f32(query.data_ptr(), Mq, ref.data_ptr(), Mr, ref.shape[1], nsplit, d2.data_ptr(),
                                              bi.data_ptr(), _stream())
        _check(rc, "iggt_nn1_search_split_f32")
        idx = fold_nn1_planes(d2, bi)
        return torch.where(idx >= 0, ref_labels[idx.clamp(min=0)], torch.full((), -1, dtype=torch.int32, device=query.device))
    out = torch.empty(Mq, dtype=torch.int32, device=query.device)
    rc = load().iggt_nn1_label_f32(query.data_ptr(), Mq, ref.data_ptr(), Mr, ref.shape[1], ref_labels.data_ptr(), out.data_ptr(),
                                   _stream())
    _check(rc, "iggt_nn1_label_f32")
    return out


def fold_nn1_planes(d2, bi):
    """Planes of iggt_nn1_search_split_f32 (d2 fp32 [S, Mq] best squared distance of sample range s, bi int32 [S, Mq] its row, -1 =
    empty range) -> int64 [Mq] row of the FIRST minimum over all samples (-1: none).  The planes cover increasing index ranges and
    each holds the first minimum of its range (strict < in the kernel), so the first plane that attains the smallest distance
    holds it.  Pure torch (CPU test: tests/test_hdbscan.py)."""
    dmin = d2.amin(0, keepdim=True)
    hit = (d2 == dmin) & (bi >= 0)
    first = torch.where(hit.any(0), hit.int().argmax(0), torch.zeros((), dtype=torch.int64, device=d2.device))
    return bi.gather(0, first[None])[0].long()


def nn1_label_tiled(query, ref, ref_idx, ref_labels):
    """nn1_label on spatially sorted inputs (include/iggt_hip.h): query fp32 [Mq,C] and ref fp32 [Mr,C] ordered along one
    space-filling curve, ref_idx int32 [Mr] = original position of every ref row (tie-break), ref_labels int32 [Mr] in the sorted
    order -> labels int32 [Mq] in the sorted query order."""
    _dev(query, ref, ref_idx, ref_labels)
    _f32_all(query, ref)
    assert query.shape[1] == ref.shape[1]
    for t in (ref_idx, ref_labels):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() == ref.shape[0]
    qlo, qhi = hdbscan_tile_boxes(query)
    rlo, rhi = hdbscan_tile_boxes(ref)
    out = torch.empty(query.shape[0], dtype=torch.int32, device=query.device)
    rc = load().iggt_nn1_label_tiled_f32(query.data_ptr(), query.shape[0], qlo.data_ptr(), qhi.data_ptr(), ref.data_ptr(),
                                         ref.shape[0], rlo.data_ptr(), rhi.data_ptr(), ref.shape[1], ref_idx.data_ptr(),
                                         ref_labels.data_ptr(), out.data_ptr(), _stream())
    _check(rc, "iggt_nn1_label_tiled_f32")
    return out


# ------------------------------------------------------------------------------------------------
# track head (csrc/track.hip)
def layernorm_rows(x, w, b, eps, out=None, add=None):
    """LayerNorm over the last dim of x (+ add) [M, C] fp32 (unit column stride, any row stride / width)."""
    _dev(x, w, b, out, add)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    M, C = x.shape
    if add is not None:
        assert add.dtype == torch.float32 and add.shape == (M, C) and add.stride(1) == 1
    if out is None:
        out = torch.empty(M, C, dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.shape == (M, C) and out.stride(1) == 1
    rc = load().iggt_layernorm_rows_f32(x.data_ptr(), x.stride(0), _ptr(add), 0 if add is None else add.stride(0),
                                        _f32c(w, "weight").data_ptr(), _f32c(b, "bias").data_ptr(), out.data_ptr(),
                                        out.stride(0), M, C, float(eps), _stream())
    _check(rc, "iggt_layernorm_rows_f32")
    return out


def avgpool2_nhwc(x):
    """x [N, H, W, C] fp32 contiguous -> [N, H // 2, W // 2, C]."""
    _dev(x)
    assert x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    N, H, W, C = x.shape
    y = torch.empty(N, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    _check(load().iggt_avgpool2_nhwc_f32(x.data_ptr(), y.data_ptr(), N, H, W, C, _stream()), "iggt_avgpool2_nhwc_f32")
    return y


def sample_points_nhwc(feat, xy):
    """feat [H, W, C] fp32 contiguous, xy [N, 2] pixel coordinates -> [N, C] (bilinear, border padding)."""
    _dev(feat, xy)
    assert feat.dtype == torch.float32 and feat.dim() == 3 and feat.is_contiguous()
    assert xy.dtype == torch.float32 and xy.dim() == 2 and xy.shape[1] == 2 and xy.stride(1) == 1
    H, W, C = feat.shape
    N = xy.shape[0]
    out = torch.empty(N, C, dtype=torch.float32, device=feat.device)
    rc = load().iggt_sample_points_nhwc_f32(feat.data_ptr(), H, W, C, xy.data_ptr(), xy.stride(0), out.data_ptr(), C, N,
                                            _stream())
    _check(rc, "iggt_sample_points_nhwc_f32")
    return out


def track_corr(pyramid, feats, coords, radius, out):
    """pyramid: list of [S, H_l, W_l, C] fp32 contiguous maps; feats [N, S, C], coords [N, S, 2] contiguous;
    out [N * S, ld] with ld >= levels * (2 radius + 1)^2 (the padding columns are zeroed)."""
    _dev(feats, coords, out, *pyramid)
    L = len(pyramid)
    S, _, _, C = pyramid[0].shape
    N = feats.shape[0]
    for t in list(pyramid) + [feats, coords, out]:
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert feats.shape == (N, S, C) and coords.shape == (N, S, 2) and out.shape[0] == N * S
    ptrs = (ctypes.c_void_p * L)(*[m.data_ptr() for m in pyramid])
    hs = (ctypes.c_int * L)(*[m.shape[1] for m in pyramid])
    ws = (ctypes.c_int * L)(*[m.shape[2] for m in pyramid])
    rc = load().iggt_track_corr_f32(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(hs, ctypes.c_void_p),
                                    ctypes.cast(ws, ctypes.c_void_p), L, S, C, feats.data_ptr(), coords.data_ptr(), N,
                                    radius, out.data_ptr(), out.stride(0), _stream())
    _check(rc, "iggt_track_corr_f32")
    return out


def track_posemb(tabx, taby, xy):
    """tabx [W, Ch], taby [H, Ch] fp32; xy [N, 2] -> [N, 2 Ch]."""
    _dev(tabx, taby, xy)
    for t in (tabx, taby):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert xy.dtype == torch.float32 and xy.stride(1) == 1
    W, Ch = tabx.shape
    H = taby.shape[0]
    N = xy.shape[0]
    out = torch.empty(N, 2 * Ch, dtype=torch.float32, device=xy.device)
    rc = load().iggt_track_posemb_f32(tabx.data_ptr(), taby.data_ptr(), H, W, Ch, xy.data_ptr(), xy.stride(0),
                                      out.data_ptr(), 2 * Ch, N, _stream())
    _check(rc, "iggt_track_posemb_f32")
    return out


def track_tokens(coords, corr, feats, pos, ref, E, max_scale, out=None):
    """coords [N, S, 2], corr [N * S, Cc], feats [N * S, Cf], pos [N, D], ref [2, D] -> [N * S, D], D = 2E + 4 + Cc + Cf."""
    _dev(coords, corr, feats, pos, ref, out)
    N, S, _ = coords.shape
    Cc, Cf = corr.shape[1], feats.shape[1]
    D = 2 * E + 4 + Cc + Cf
    for t in (coords, corr, feats, pos, ref):
        assert t.dtype == torch.float32 and t.stride(-1) == 1
    assert coords.is_contiguous() and ref.is_contiguous() and pos.shape == (N, D) and ref.shape == (2, D)
    if out is None:
        out = torch.empty(N * S, D, dtype=torch.float32, device=coords.device)
    rc = load().iggt_track_tokens_f32(coords.data_ptr(), corr.data_ptr(), corr.stride(0), Cc, feats.data_ptr(),
                                      feats.stride(0), Cf, pos.data_ptr(), pos.stride(0), ref.data_ptr(), out.data_ptr(),
                                      out.stride(0), N, S, E, float(max_scale), _stream())
    _check(rc, "iggt_track_tokens_f32")
    return out


def track_update(coords, delta, pred, stride):
    """coords [N, S, 2] (updated in place), delta [N * S, >= 2], pred [S, N, 2] (written)."""
    _dev(coords, delta, pred)
    N, S, _ = coords.shape
    assert coords.is_contiguous() and pred.is_contiguous() and pred.shape == (S, N, 2) and delta.stride(1) == 1
    rc = load().iggt_track_update_f32(coords.data_ptr(), delta.data_ptr(), delta.stride(0), pred.data_ptr(), N, S,
                                      float(stride), _stream())
    _check(rc, "iggt_track_update_f32")
    return pred


# ------------------------------------------------------------------------------------------------
# HDBSCAN (csrc/hdbscan.hip, csrc/hdbscan_tree.hip)
def hdbscan_tile_boxes(x):
    """x fp32 [M, C] -> (box_lo, box_hi) fp32 [ceil(M / 256), C]: per-channel extent of every tile of 256 consecutive rows."""
    M, C = x.shape
    nt = (M + 255) // 256
    pad = torch.cat([x, x[-1:].expand(nt * 256 - M, C)]).view(nt, 256, C)
    return pad.amin(1).contiguous(), pad.amax(1).contiguous()


def hdbscan_core_dist(x, k, boxes=None):
    """x fp32 [M, C] (device, C in {3, 8, 16}) -> core distances fp32 [M]: distance to the k-th nearest row, itself counted.
    boxes: hdbscan_tile_boxes(x) if already known."""
    _dev(x)
    _f32_all(x)
    M, C = x.shape
    lo, hi = boxes if boxes is not None else hdbscan_tile_boxes(x)
    _f32_all(lo, hi)
    core = torch.empty(M, dtype=torch.float32, device=x.device)
    _check(load().iggt_hdbscan_core_dist_f32(x.data_ptr(), M, C, int(k), lo.data_ptr(), hi.data_ptr(), core.data_ptr(), _stream()),
           "iggt_hdbscan_core_dist_f32")
    return core


def hdbscan_nearest_foreign(x, core2, comp, idx, tile_lo, tile_hi, boxes=None, component_bound=True, nsplit=None):
    """One Boruvka round (include/iggt_hip.h): arrays ordered by component -> (best_w2 fp32 [M], best_p int32 [M]).
    component_bound: let workgroups inside one component share the component's best weight so far (points that cannot hold
    the component's cheapest outgoing edge then report (inf, -1)); False = every point's own cheapest foreign edge.
    nsplit: workgroups per block of 512 queries (None: by size).  A few blocks hold an outlier whose bound covers nearly every
    tile; split over the tiles of the walk they stop being the round's critical path.  The planes are folded here under the
    kernel's own total order (weight, min original index, max original index)."""
    _dev(x, core2, comp, idx, tile_lo, tile_hi)
    _f32_all(x, core2)
    M, C = x.shape
    for t in (comp, idx, tile_lo, tile_hi):
        assert t.dtype == torch.int32 and t.is_contiguous()
    assert tile_lo.numel() == (M + 255) // 256 == tile_hi.numel()
    lo, hi = boxes if boxes is not None else hdbscan_tile_boxes(x)
    _f32_all(lo, hi)
    ntiles = (M + 255) // 256
    G = max(1, min(32, ntiles // 32)) if nsplit is None else int(nsplit)
    w2 = torch.empty(G, M, dtype=torch.float32, device=x.device)
    bp = torch.empty(G, M, dtype=torch.int32, device=x.device)
    cbound = None
    if component_bound:   # component ids are member indices: < M.  0x7f800000 = +inf
        cbound = torch.full((M,), 0x7f800000, dtype=torch.int32, device=x.device)
    rc = load().iggt_hdbscan_nearest_foreign_f32(x.data_ptr(), core2.data_ptr(), comp.data_ptr(), idx.data_ptr(), tile_lo.data_ptr(),
                                                 tile_hi.data_ptr(), lo.data_ptr(), hi.data_ptr(), M, C, w2.data_ptr(), bp.data_ptr(),
                                                 0 if cbound is None else cbound.data_ptr(), G, _stream())
    _check(rc, "iggt_hdbscan_nearest_foreign_f32")
    return fold_foreign_planes(w2, bp, idx)


def fold_foreign_planes(w2, bp, idx):
    """Planes of one Boruvka round (iggt_hdbscan_nearest_foreign_f32 with nsplit > 1: w2 fp32 [G, M] squared weights, bp int32
    [G, M] partner positions, -1 = none) -> (w2 [M], bp [M]): per position the smallest weight, then the smallest (min, max) pair of
    ORIGINAL indices (idx int [M]: original index of every position) -- the kernel's own total order, exact (equal floats compare
    equal).  Pure torch: tests/test_hdbscan.py checks it on the CPU against a per-position scan."""
    G, M = w2.shape
    if G == 1:
        return w2[0], bp[0]
    wmin = w2.amin(0)
    me = idx.long()[None]
    other = idx.long()[bp.clamp(min=0).long()]
    key = torch.where((w2 == wmin[None]) & (bp >= 0), torch.minimum(me, other) * M + torch.maximum(me, other),
                      torch.full((1, 1), 1 << 62, dtype=torch.int64, device=w2.device))
    sel = key.argmin(0, keepdim=True)
    return wmin, bp.gather(0, sel)[0].contiguous()


def hdbscan_labels_from_mst(eu, ev, ew, n_points, min_cluster_size, eps=0.0, allow_single_cluster=False):
    """HOST walk over the spanning tree (numpy int32 / int32 / float32 arrays of n_points - 1 edges) -> labels int32 [n_points]."""
    import numpy as np

    eu = np.ascontiguousarray(eu, dtype=np.int32)
    ev = np.ascontiguousarray(ev, dtype=np.int32)
    ew = np.ascontiguousarray(ew, dtype=np.float32)
    assert eu.shape == ev.shape == ew.shape == (max(int(n_points) - 1, 0),)
    labels = np.empty(int(n_points), dtype=np.int32)
    rc = load().iggt_hdbscan_labels_from_mst(eu.ctypes.data, ev.ctypes.data, ew.ctypes.data, int(n_points), int(min_cluster_size),
                                             float(eps or 0.0), int(bool(allow_single_cluster)), labels.ctypes.data)
    _check(rc, "iggt_hdbscan_labels_from_mst")
    return labels
