"""ctypes binding of libzsg.so (C ABI in include/zsg.h).

The HIP library is the product: there is NO CPU / eager-PyTorch fallback.  If the shared object is missing or a
symbol does not resolve, importing the compute modules raises immediately ("fail loudly", task ③).
`import torch` happens first so that exactly one HIP runtime (libamdhip64.so.7) is resident.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: single libamdhip64 in the process)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZSG_LIB_PATH") or os.path.join(_HERE, "libzsg.so")      # (override: A/B two builds inside one GPU allocation)
ZSG_MAX_SEG = 8


class Taps(C.Structure):
    _fields_ = [("n", C.c_int32), ("w0", C.c_int32), ("wstep", C.c_int32), ("d0", C.c_int32), ("dstep", C.c_int32)]


class Seg(C.Structure):
    _fields_ = [("rows_y", C.c_int32), ("rows_x", C.c_int32), ("src_H", C.c_int32), ("src_W", C.c_int32),
                ("sy", C.c_int32), ("sx", C.c_int32), ("out_W", C.c_int32), ("osy", C.c_int32), ("osx", C.c_int32),
                ("opy", C.c_int32), ("opx", C.c_int32), ("reserved", C.c_int32),
                ("src_off", C.c_int64), ("src_bstride", C.c_int64), ("out_off", C.c_int64), ("out_bstride", C.c_int64),
                ("ty", Taps), ("tx", Taps)]


class ConvDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("N", C.c_int32), ("src_ld", C.c_int32), ("out_ld", C.c_int32),
                ("wR", C.c_int32), ("wS", C.c_int32), ("wC", C.c_int32), ("wc0", C.c_int32), ("wt_ld", C.c_int32),
                ("relu", C.c_int32), ("merge_x", C.c_int32), ("nseg", C.c_int32), ("tile_hint", C.c_int32),
                ("epi_flags", C.c_int32), ("seg", Seg * ZSG_MAX_SEG)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


P = C.c_void_p
I32, I64, F32, SZ = C.c_int32, C.c_int64, C.c_float, C.c_size_t
DP = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/zsg.h declares
SIGNATURES = {
    "zsg_version": (I32, []),
    "zsg_last_error": (C.c_char_p, []),
    "zsg_source_stamp": (C.c_char_p, []),
    "zsg_set_deterministic": (I32, [I32]),
    "zsg_conv_igemm": (I32, [DP, P, P, P, P, P, P, P, P]),
    "zsg_comm_unique_id": (I32, [P]),
    "zsg_comm_init": (I32, [C.POINTER(P), P, I32, I32]),
    "zsg_comm_allreduce_bucket": (I32, [P, P, I64, P]),
    "zsg_comm_broadcast": (I32, [P, P, I64, I32, P]),
    "zsg_comm_wait": (I32, [P, P]),
    "zsg_comm_destroy": (I32, [P]),
    "zsg_conv_wino": (I32, [DP, P, P, P, P, P, P, P, P]),
    "zsg_bn_relu_maxpool_fwd": (I32, [P, I32, I32, I32, I32, P, P, P, P, I32, I32, I32, I32, I32, P, P, P]),
    "zsg_bn_relu_maxpool_bwd": (I32, [P, P, P, I32, I32, I32, I32, P, P, P, P, I32, I32, I32, I32, I32, P, P, P, I32, P, SZ, P]),
    "zsg_conv_igemm_bnb": (I32, [DP, P, P, P, P, P, P, P, P, P, P]),
    "zsg_conv_igemm_partial_rows": (I32, [DP]),
    "zsg_conv_bn_tail_tickets": (I32, [DP, I32]),
    "zsg_conv_igemm_bnpre": (I32, [DP, P, P, P, P, P, P, P, P, P, F32, F32, P, P, P, P, P, P, P, P]),
    "zsg_conv_igemm_bnstat": (I32, [DP, P, P, P, P, P, P, P, P, P, F32, F32, P]),
    "zsg_conv_wino_bnstat": (I32, [DP, P, P, P, P, P, P, P, P, P, F32, F32, P]),
    "zsg_conv_igemm_bnb_tail": (I32, [DP, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, P]),
    "zsg_conv_wino_bnb_tail": (I32, [DP, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, P]),
    "zsg_bn_bwd_apply": (I32, [P, P, P, I64, I32, P, P, P, P, P, P, P]),
    "zsg_conv_wino_bnb": (I32, [DP, P, P, P, P, P, P, P, P, P, P]),
    "zsg_bn_backward_from_partials": (I32, [P, P, P, I64, I32, P, P, P, P, P, P, P, I32, P, I32, P, SZ, P]),
    "zsg_wino_u_elems": (I64, [I32, I32]),
    "zsg_wino_weights": (I32, [P, I32, I32, P]),
    "zsg_conv_wgrad_workspace_bytes": (SZ, [DP]),
    "zsg_conv_wgrad": (I32, [DP, P, P, P, I32, P, SZ, P]),
    "zsg_conv_wgrad_wino_workspace_bytes": (SZ, [DP]),
    "zsg_conv_wgrad_wino": (I32, [DP, P, P, P, I32, P, SZ, P]),
    "zsg_conv_wgrad_wino_batched": (I32, [DP, I32, P, P, P, I32, P, SZ, P]),
    "zsg_transpose_w": (I32, [P, P, I32, I32, I32, I32, P]),
    "zsg_transpose_w_batched": (I32, [P, P, P, I32, I32, P]),
    "zsg_pad_rows": (I32, [P, I64, I32, I32, P, I32, P]),
    "zsg_colsum": (I32, [P, I32, I64, I32, I32, I32, I32, P, I32, P]),
    "zsg_bn_workspace_bytes": (SZ, [I64, I32]),
    "zsg_bn_stats": (I32, [P, I64, I32, P, P, P, P, F32, F32, P, SZ, P]),
    "zsg_bn_stats_from_partials": (I32, [P, I32, I64, I32, P, P, P, P, F32, F32, P]),
    "zsg_bn_eval_stats": (I32, [P, P, I32, F32, P, P, P]),
    "zsg_bn_fold": (I32, [P, P, P, F32, P, I32, I32, P, P]),
    "zsg_bn_inline_max_chunks": (I32, []),
    "zsg_bn_apply_from_partials": (I32, [P, I64, I32, P, I32, P, P, P, I32, P, P, P, P, P, P, F32, F32, P]),
    "zsg_bn_apply": (I32, [P, I64, I32, P, P, P, P, P, I32, P, P, P]),
    "zsg_bn_backward": (I32, [P, P, P, P, I64, I32, P, P, P, P, P, P, P, I32, P, SZ, P]),
    "zsg_maxpool_fwd": (I32, [P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P, P, P]),
    "zsg_maxpool_bwd": (I32, [P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P, P]),
    "zsg_upsample_add_fwd": (I32, [P, P, I32, I32, I32, I32, I32, I32, P, P]),
    "zsg_upsample_add_bwd": (I32, [P, I32, I32, I32, I32, I32, I32, P, I32, P]),
    "zsg_relu_fwd": (I32, [P, I64, P, P]),
    "zsg_relu_bwd": (I32, [P, P, I64, P, I32, P]),
    "zsg_avgpool_fwd": (I32, [P, I32, I32, I32, P, P]),
    "zsg_avgpool_bwd": (I32, [P, I32, I32, I32, P, I32, P]),
    "zsg_l2norm_fwd": (I32, [P, I64, I32, P, P, P]),
    "zsg_l2norm_bwd": (I32, [P, P, P, I64, I32, P, P]),
    "zsg_nchw_to_nhwc4": (I32, [P, I32, I32, I32, I32, P, P]),
    "zsg_u8hwc_to_nhwc4": (I32, [P, I64, P, P]),
    "zsg_resize_u8": (I32, [P, I32, I32, I32, P, P, I32, P, P, I32, I32, I32, P, P, P]),
    "zsg_resize_u8_batched": (I32, [P, I32, I32, I32, I32, I32, I32, P]),
    "zsg_interleave": (I32, [P, I64, I32, I32, P, I32, I32, I32, P]),
    "zsg_head_lang_map": (I32, [P, P, I32, I32, I32, I32, P, P]),
    "zsg_head_lang_map_packed": (I32, [P, P, I32, I32, P, I32, P, P]),
    "zsg_stage_inputs": (I32, [P, I32, I32, I32, I32, P, P, I32, P, P, I32, P, P, I32, P]),
    "zsg_head_border_sums": (I32, [P, I32, I32, I32, I32, P, P]),
    "zsg_head_border_finalize": (I32, [P, I32, I32, P, P, P, P]),
    "zsg_batch_sum": (I32, [P, I32, I64, P, P]),
    "zsg_lstm_gather_last": (I32, [P, P, I32, I32, I32, P, P]),
    "zsg_lstm_fwd": (I32, [P, P, P, P, P, P, P, I32, I32, I32, P, P, P, P, I32, I32, P]),
    "zsg_lstm_bwd": (I32, [P, I32, I32, P, P, P, P, P, P, I32, I32, I32, P, P]),
    "zsg_loss_workspace_bytes": (SZ, [I32, I32]),
    "zsg_loss_fwd_bwd": (I32, [P, P, P, I32, I32, F32, F32, F32, F32, I32, F32, P, P, P, P, P, SZ, P]),
    "zsg_eval_workspace_bytes": (SZ, [I32]),
    "zsg_eval": (I32, [P, P, P, P, I32, I32, F32, P, P, P, P, P, P, P]),
    "zsg_iou": (I32, [P, P, I32, I32, P, P]),
    "zsg_adam_step": (I32, [P, P, P, P, I64, F32, F32, F32, F32, F32, F32, P, P]),
    "zsg_adam_step_range": (I32, [P, P, P, P, I64, F32, F32, F32, F32, F32, F32, P, I32, P]),
    "zsg_memset_f32": (I32, [P, I64, F32, P]),
    "zsg_set_stream_workspace": (I32, [P, P, SZ]),
    "zsg_set_main_priority": (I32, [I32]),
    "zsg_get_main_priority": (I32, []),
    "zsg_event_create": (P, []),
    "zsg_event_destroy": (I32, [P]),
    "zsg_set_completion_event": (I32, [P]),
    "zsg_event_record": (I32, [P, P]),
    "zsg_stream_wait_event": (I32, [P, P]),
    "zsg_prof_enable": (I32, [I32]),
    "zsg_prof_collect": (I32, [C.POINTER(ProfEntry), I32]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libzsg.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C zsgnet-pytorch_amd/csrc`).  There is no CPU fallback for the ZSGNet hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.zsg_version() != 100:
        raise RuntimeError(f"libzsg.so version {lib.zsg_version()} does not match the Python binding (100)")
    return lib


lib = _load()
lib.zsg_set_deterministic(1 if os.environ.get("ZSG_DETERMINISTIC", "0") == "1" else 0)


class ZsgError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise ZsgError(f"{what} failed ({rc}): {lib.zsg_last_error().decode()}")


def stream_ptr():
    """hipStream_t of torch's current stream (kernels are launched on it)."""
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("zsgnet-pytorch_amd: no MI355X visible (torch.cuda.is_available() is False); "
                           "the hot path has no CPU fallback")
