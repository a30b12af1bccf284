"""ctypes binding of libptamd.so (include/ptamd.h).  There is NO fallback: if the
HIP library is missing or a call fails, this raises."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTAMD_LIB_TAG selects an ablation build made with PTAMD_BUILD_TAG (profiles/tools); unset = the product library
_TAG = os.environ.get("PTAMD_LIB_TAG", "")
LIB_PATH = os.path.join(_HERE, "csrc", f"libptamd_{_TAG}.so" if _TAG else "libptamd.so")

OK = 0
ERRORS = {-1: "bad shape", -2: "sequence too long for the NeRF LDS staging", -3: "workspace missing or too small",
          -4: "HIP runtime error", -5: "pointer not 16-byte aligned"}
ST_BAD_RESIDUE, ST_TOO_SHORT, ST_BAD_THETA, ST_NONFINITE = 1, 2, 4, 8

_p, _i, _i64, _f, _sz, _u64, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint64, C.c_uint32


class GemmArgs(C.Structure):
    _fields_ = [("M", _i), ("N", _i), ("K", _i),
                ("A", _p), ("lda", _i), ("a_kmajor", _i),
                ("B", _p), ("ldb", _i), ("b_kmajor", _i),
                ("C", _p), ("ldc", _i),
                ("bias", _p),
                ("residual", _p), ("ldr", _i),
                ("flags", _i),
                ("dropout_p", _f), ("seed", _u64), ("stream_id", _u32),
                ("split_k", _i),
                ("workspace", _p), ("workspace_bytes", _sz),
                ("colsum", _p),
                ("gate_scale", _f),
                ("arith", _i),
                ("reserved_cus", _i),
                ("a_scale", _p), ("a_scale_stride", _i),
                ("b_scale", _p), ("b_scale_stride", _i),
                ("gate_mask", _p)]


class WScaleJob(C.Structure):
    _fields_ = [("w", _p), ("rows", _i), ("cols", _i), ("ld", _i), ("row_scale", _p), ("col_scale", _p), ("stats", _p),
                ("rows_only", _i)]


class BoundJob(C.Structure):
    _fields_ = [("ln_gamma_stats", _p), ("ln_beta_stats", _p), ("w_stats", _p), ("w_stat_index", _i), ("bias_stats", _p),
                ("sqrt_d", _f), ("post_scale", _f), ("out_scale", _p), ("out_value", _p)]


class LnReduceJob(C.Structure):
    _fields_ = [("partials", _p), ("D", _i), ("dgamma", _p), ("dbeta", _p)]


class FillJob(C.Structure):
    _fields_ = [("dst", _p), ("n", _i64), ("value", _u32)]


class HpSplitJob(C.Structure):
    _fields_ = [("x", _p), ("ld", _i), ("rows", _i), ("K", _i), ("planes", _p), ("scale", _p)]


class GemmHpArgs(C.Structure):
    _fields_ = [("M", _i), ("N", _i), ("K", _i),
                ("A", _p), ("A_scale", _p),
                ("B", _p), ("B_scale", _p),
                ("C", _p), ("ldc", _i),
                ("bias", _p),
                ("residual", _p), ("ldr", _i),
                ("flags", _i),
                ("dropout_p", _f), ("seed", _u64), ("stream_id", _u32),
                ("split_k", _i),
                ("workspace", _p), ("workspace_bytes", _sz),
                ("gate_scale", _f),
                ("reserved_cus", _i),
                ("gate_mask", _p), ("gate_mask_out", _p),
                ("kv_planes", _p), ("kv_inv", _p), ("kv_col0", _i), ("kv_heads", _i)]


class WprepSeg(C.Structure):
    _fields_ = [("offset", _i64), ("rows", C.c_int32), ("cols", C.c_int32), ("stats_row0", C.c_int32), ("stats_index", C.c_int32),
                ("row_scale_index", C.c_int32), ("col_scale_index", C.c_int32), ("colmax_index", C.c_int32),
                ("colsq_index", C.c_int32), ("row_planes", _u64), ("col_planes", _u64), ("ld", C.c_int32), ("rowmax_index", C.c_int32)]


class WprepBound(C.Structure):
    _fields_ = [("ln_gamma_stats", C.c_int32), ("ln_beta_stats", C.c_int32), ("w_stats", C.c_int32), ("w_stat_index", C.c_int32),
                ("bias_stats", C.c_int32), ("sqrt_d", _f), ("post_scale", _f), ("out_scale", C.c_int32), ("out_value", C.c_int32)]


class WprepPlan(C.Structure):
    _fields_ = [("segs", _p), ("nsegs", C.c_int32),
                ("blocks_a", _p), ("nblocks_a", C.c_int32), ("nblocks_a_matrices", C.c_int32),
                ("plain", _p), ("nplain", C.c_int32),
                ("blocks_b", _p), ("nblocks_b", C.c_int32),
                ("bounds", _p), ("nbounds", C.c_int32),
                ("groups", _p), ("ngroups", C.c_int32),
                ("colnorm_segs", _p),
                ("scales", _p), ("values", _p),
                ("colmax", _p), ("ncolmax", C.c_int32),
                ("colsq", _p),
                ("stats", _p), ("nstats", C.c_int32),
                ("numel", _i64), ("with_planes", C.c_int32)]


# name -> (restype, argtypes); mirrors include/ptamd.h one to one
SIGNATURES = {
    "ptamd_version": (C.c_char_p, []),
    "ptamd_last_hip_error": (C.c_char_p, []),
    "ptamd_sidechain_atoms": (_i, [_i]),
    "ptamd_angles_fwd": (_i, [_p, _p, _i64, _p]),
    "ptamd_angles_bwd": (_i, [_p, _p, _p, _i64, _p]),
    "ptamd_nerf_workspace_bytes": (_sz, [_i, _i]),
    "ptamd_nerf_place": (_i, [_p, _p, _p, _p, _p, _p, _i64, _p, _p, _p]),
    "ptamd_pairwise_dist": (_i, [_p, _i, _i, _p, _p]),
    "ptamd_nerf_fwd": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "ptamd_nerf_bwd": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _sz, _p]),
    "ptamd_drmsd_workspace_bytes": (_sz, [_i, _i]),
    "ptamd_drmsd_fwd_bwd": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _sz, _p]),
    "ptamd_drmsd_workspace_bytes_budget": (_sz, [_i, _i, _sz]),
    "ptamd_drmsd_fwd_bwd_budget": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _sz, _sz, _p]),
    "ptamd_kabsch_rmsd": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "ptamd_mse_angles_workspace_bytes": (_sz, []),
    "ptamd_mse_angles_fwd": (_i, [_p, _p, _i64, _p, _p, _sz, _p]),
    "ptamd_mse_angles_bwd": (_i, [_p, _p, _i64, _p, _f, _i, _p, _p]),
    "ptamd_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "ptamd_gate_mask_bytes": (_sz, [_i, _i]),
    "ptamd_gemm": (_i, [C.POINTER(GemmArgs), _p]),
    "ptamd_gemm_products": (_i, [C.POINTER(GemmArgs)]),
    "ptamd_gemm_group": (_i, [C.POINTER(GemmArgs), _i, _p]),
    "ptamd_hp_bytes": (_sz, [_i, _i]),
    "ptamd_hp_padded_rows": (_i, [_i]),
    "ptamd_hp_split": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "ptamd_hp_split_rows": (_i, [C.POINTER(HpSplitJob), _i, _p]),
    "ptamd_hp_split_cols": (_i, [C.POINTER(HpSplitJob), _i, _p]),
    "ptamd_gemm_hp_workspace_bytes": (_sz, [_i, _i, _i]),
    "ptamd_gemm_hp": (_i, [C.POINTER(GemmHpArgs), _p]),
    "ptamd_weight_scales": (_i, [C.POINTER(WScaleJob), _i, _p]),
    "ptamd_bound_scales": (_i, [C.POINTER(BoundJob), _i, _p]),
    "ptamd_layernorm_fwd": (_i, [_p, _p, _p, _i64, _i, _p, _p, _p, _p, _p, _p]),
    "ptamd_layernorm_fwd_sum": (_i, [_p, _i, _i64, _p, _p, _f, _u64, _u32, _p, _p, _p, _i64, _i, _p, _p, _p, _p, _p, _p]),
    "ptamd_layernorm_bwd_dropout": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _u64, _u32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _p, _sz, _p]),
    "ptamd_layernorm_bwd_workspace_bytes": (_sz, [_i]),
    "ptamd_layernorm_bwd_reduce": (_i, [C.POINTER(LnReduceJob), _i, _p]),
    "ptamd_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _p, _p, _p, _p, _sz, _p]),
    "ptamd_embed_fwd": (_i, [_p, _p, _p, _i, _i, _i, _f, _u64, _p, _p]),
    "ptamd_embed_bwd_workspace_bytes": (_sz, [_i]),
    "ptamd_embed_bwd": (_i, [_p, _p, _i, _i, _i, _f, _u64, _p, _p, _sz, _p]),
    "ptamd_attention_fwd": (_i, [_p, _p, _i, _i, _i, _i, _f, _u64, _u32, _i, _p, _p, _p, _p, _p, _p]),
    "ptamd_attention_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u64, _u32, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "ptamd_attention_kv_bytes": (_sz, [_i, _i]),
    "ptamd_attention_kv_inv_floats": (_sz, [_i, _i]),
    "ptamd_attention_reads_kv_planes": (_i, [_i, _i, _i, _i, _i]),
    "ptamd_attention_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ptamd_attention_keep_bits_bytes": (_sz, [_i, _i, _i]),
    "ptamd_attention_bwd_reads_keep_bits": (_i, [_i, _i, _i, _i, _i]),
    "ptamd_colsum_workspace_bytes": (_sz, [_i]),
    "ptamd_colsum": (_i, [_p, _i64, _i, _i, _i, _p, _p, _sz, _p]),
    "ptamd_relu_dropout_bwd": (_i, [_p, _p, _i64, _f, _p, _p]),
    "ptamd_tanh_bwd": (_i, [_p, _p, _i64, _p, _p]),
    "ptamd_dropout_bwd": (_i, [_p, _i64, _i, _f, _u64, _u32, _p, _p]),
    "ptamd_im2col1d": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "ptamd_col2im1d": (_i, [_p, _i, _i, _i, _i, _p, _i, _p]),
    "ptamd_conv_weight_pack": (_i, [_p, _i, _i, _i, _p, _p]),
    "ptamd_conv_weight_unpack_add": (_i, [_p, _i, _i, _i, _p, _p]),
    "ptamd_onehot": (_i, [_p, _i64, _i, _p, _p]),
    "ptamd_posenc_add_fwd": (_i, [_p, _p, _i, _i, _i, _f, _u64, _p, _p]),
    "ptamd_posenc_add_bwd": (_i, [_p, _i64, _f, _u64, _p, _p]),
    "ptamd_grad_sqnorm_workspace_bytes": (_sz, []),
    "ptamd_grad_sqnorm": (_i, [_p, _i64, _p, _p, _sz, _p]),
    "ptamd_wprep_rows_per_block": (_i, []),
    "ptamd_wprep_plain_floats_per_block": (_i, []),
    "ptamd_weights_prep": (_i, [C.POINTER(WprepPlan), _p, _i, _p]),
    "ptamd_sgd_step_prep": (_i, [C.POINTER(WprepPlan), _i, _p, _p, _i64, _p, _f, _f, _f, _i, _p]),
    "ptamd_adam_step_prep": (_i, [C.POINTER(WprepPlan), _i, _p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _f, _f, _i, _i, _p]),
    "ptamd_sgd_step": (_i, [_p, _p, _i64, _p, _f, _f, _f, _i, _p]),
    "ptamd_adam_step": (_i, [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _f, _f, _i, _i, _p]),
    "ptamd_fill_u32": (_i, [C.POINTER(FillJob), _i, _p]),
}

_lib = None
MISSING = set()


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m protein_transformer_amd.build` "
                "(the MI355X path has no CPU or PyTorch fallback)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                MISSING.add(name)       # tests/test_abi.py requires this set to be empty
                continue
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous where the ABI says so."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _raw_stream(device_index=None):
    """The current HIP stream of a device as an integer handle - one C call (torch.cuda.current_stream() builds a Stream
    object through three Python layers: 40 of them per step were 15 % of the host time of a launch-bound step)."""
    if device_index is None:
        device_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(device_index)


def stream():
    return C.c_void_p(_raw_stream())


def check(rc, what):
    if rc != OK:
        detail = ERRORS.get(rc, f"code {rc}")
        if rc == -4:
            detail += ": " + lib().ptamd_last_hip_error().decode()
        raise RuntimeError(f"libptamd {what}: {detail}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libptamd operates on device tensors only (no CPU path); got a CPU tensor")


_workspaces = {}


def workspace(tag, nbytes, device):
    """Scratch buffer from the PyTorch caching allocator, grown on demand.  One per (tag, device, STREAM): kernels
    enqueued on one stream run in order, so a buffer is never shared by two launches that could overlap - two models
    driven from two streams of one process get separate workspaces."""
    key = (tag, device, _raw_stream(device.index))
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
