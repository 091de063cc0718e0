"""ctypes binding of libttsdec.so (include/ttsdec.h).  No torch types cross this
boundary: only raw device pointers, sizes and the stream handle."""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
# (TTSDEC_LIB: another build of the same library, for same-box A/B measurements of two source states - tools/ only)
LIB_PATH = os.environ.get("TTSDEC_LIB") or os.path.join(HERE, "lib", "libttsdec.so")

# error codes / enums (mirrors include/ttsdec.h)
OK = 0
ERR_INVALID_ARG, ERR_DIMS, ERR_HIP, ERR_NOT_BOUND, ERR_WORKSPACE, ERR_DEVICE = -1, -2, -3, -4, -5, -6
DROPOUT_OFF, DROPOUT_MASKS, DROPOUT_PHILOX = 0, 1, 2
POSTNET_F32, POSTNET_BF16, POSTNET_SPLIT_F16 = 0, 1, 2
PREC_F32, PREC_SPLIT_F16 = 0, 1
ABI_VERSION = 2  # include/ttsdec.h TTSDEC_VERSION these bindings were written for
CELL_TACO2PROD, CELL_TACO2 = 0, 1
POSTNET_TYPE_MEL, POSTNET_TYPE_MEL2 = 0, 1
W_DECODER_COUNT = 21
W_POSTNET_PER_LAYER = 5
W_POSTNET2_PER_LAYER = 11


def option_ids():
    """name -> TTSDEC_OPT_* of the tuning / measurement options (include/ttsdec.h), as the library itself names them."""
    lib, out, i = load(), {}, 0
    while True:
        n = lib.ttsdec_option_name(i)
        if n is None:
            return out
        out[n.decode()] = i
        i += 1

# every symbol include/ttsdec.h declares
SYMBOLS = (
    "ttsdec_version",
    "ttsdec_strerror",
    "ttsdec_last_hip_error",
    "ttsdec_create",
    "ttsdec_destroy",
    "ttsdec_set_precision",
    "ttsdec_get_precision",
    "ttsdec_set_option",
    "ttsdec_get_option",
    "ttsdec_option_name",
    "ttsdec_num_weight_tensors",
    "ttsdec_packed_bytes",
    "ttsdec_pack_weights",
    "ttsdec_bind_weights",
    "ttsdec_workspace_bytes",
    "ttsdec_decode",
    "ttsdec_postnet_workspace_bytes",
    "ttsdec_postnet",
    "ttsdec_cell_step",
    "ttsdec_profile_step",
    "ttsdec_profile_loop",
    "ttsenc_create",
    "ttsenc_destroy",
    "ttsenc_last_hip_error",
    "ttsenc_num_weight_tensors",
    "ttsenc_packed_bytes",
    "ttsenc_pack_weights",
    "ttsenc_bind_weights",
    "ttsenc_workspace_bytes",
    "ttsenc_forward",
    "ttsenc_set_precision",
    "ttsenc_get_precision",
    "ttsvits_create",
    "ttsvits_destroy",
    "ttsvits_set_precision",
    "ttsvits_get_precision",
    "ttsvits_last_hip_error",
    "ttsvits_num_weight_tensors",
    "ttsvits_packed_bytes",
    "ttsvits_pack_weights",
    "ttsvits_bind_weights",
    "ttsvits_text_encoder_workspace_bytes",
    "ttsvits_text_encoder",
    "ttsvits_flow_workspace_bytes",
    "ttsvits_flow_reverse",
)
ENC_W_COUNT = 20


class Dims(C.Structure):
    _fields_ = [
        ("d_mel", C.c_int32),
        ("r", C.c_int32),
        ("d_pre", C.c_int32),
        ("d_ctx", C.c_int32),
        ("h_att", C.c_int32),
        ("h_dec", C.c_int32),
        ("p_zoneout", C.c_float),
        ("p_dropout", C.c_float),
        ("postnet_layers", C.c_int32),
        ("postnet_hidden", C.c_int32),
        ("postnet_kernel", C.c_int32),
        ("bn_eps", C.c_float),
        ("cell_type", C.c_int32),
        ("d_pre_hidden", C.c_int32),
        ("postnet_type", C.c_int32),
    ]


class EncDims(C.Structure):
    _fields_ = [
        ("alphabet_size", C.c_int32),
        ("d_emb", C.c_int32),
        ("d_out", C.c_int32),
        ("conv_kernel", C.c_int32),
        ("bn_eps", C.c_float),
    ]


class VitsDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_vocab", "inter_channels", "hidden_channels", "filter_channels", "n_heads", "n_layers", "kernel_size", "window_size",
        "flow_hidden", "flow_kernel", "flow_wn_layers", "n_flows", "flow_tf_layers", "flow_tf_heads", "flow_tf_kernel", "gin_channels", "cond_layer_idx")]


class TtsdecError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str = ""):
        self.code = code
        msg = f"{what}: {_strerror(code)} (code {code})"
        if detail:
            msg += f" [{detail}]"
        super().__init__(msg)


_lib = None
_lock = threading.Lock()


def _strerror(code: int) -> str:
    try:
        return load().ttsdec_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "?"


def load() -> C.CDLL:
    """Loads libttsdec.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python torch-tts_amd/build.py, or __graft_entry__.build()). "
                "There is no CPU fallback for this path."
            )
        lib = C.CDLL(LIB_PATH)
        vp, i32, u64, sz, f32 = C.c_void_p, C.c_int, C.c_uint64, C.c_size_t, C.c_float
        lib.ttsdec_version.restype = i32
        lib.ttsdec_version.argtypes = []
        got = lib.ttsdec_version()
        if got != ABI_VERSION:  # (an older or newer build, e.g. through TTSDEC_LIB: its entry points take other argument lists)
            raise RuntimeError(f"{LIB_PATH} has ABI version {got}, these bindings are for version {ABI_VERSION}: rebuild it "
                               "(python torch-tts_amd/build.py --force)")
        lib.ttsdec_strerror.restype = C.c_char_p
        lib.ttsdec_strerror.argtypes = [i32]
        lib.ttsdec_last_hip_error.restype = C.c_char_p
        lib.ttsdec_last_hip_error.argtypes = [vp]
        lib.ttsdec_create.restype = i32
        lib.ttsdec_create.argtypes = [C.POINTER(Dims), C.POINTER(vp)]
        lib.ttsdec_destroy.restype = i32
        lib.ttsdec_destroy.argtypes = [vp]
        lib.ttsdec_set_precision.restype = i32
        lib.ttsdec_set_precision.argtypes = [vp, i32]
        lib.ttsdec_get_precision.restype = i32
        lib.ttsdec_get_precision.argtypes = [vp]
        lib.ttsdec_set_option.restype = i32
        lib.ttsdec_set_option.argtypes = [vp, i32, i32]
        lib.ttsdec_get_option.restype = i32
        lib.ttsdec_get_option.argtypes = [vp, i32, C.POINTER(i32)]
        lib.ttsdec_option_name.restype = C.c_char_p
        lib.ttsdec_option_name.argtypes = [i32]
        lib.ttsdec_num_weight_tensors.restype = i32
        lib.ttsdec_num_weight_tensors.argtypes = [vp]
        lib.ttsdec_packed_bytes.restype = sz
        lib.ttsdec_packed_bytes.argtypes = [vp]
        lib.ttsdec_pack_weights.restype = i32
        lib.ttsdec_pack_weights.argtypes = [vp, C.POINTER(vp), i32, vp, vp]
        lib.ttsdec_bind_weights.restype = i32
        lib.ttsdec_bind_weights.argtypes = [vp, vp]
        lib.ttsdec_workspace_bytes.restype = sz
        lib.ttsdec_workspace_bytes.argtypes = [vp, i32, i32]
        lib.ttsdec_decode.restype = i32
        lib.ttsdec_decode.argtypes = [
            vp, vp, i32, i32, i32, i32, i32,  # h, memory, B, L, t_begin, n_steps, t_stride
            f32, i32, i32, vp, u64,           # stop_threshold, check_stop, dropout_mode, masks, seed
            vp, i32, vp,                      # teacher, teacher_T, teacher_flags
            vp, vp, vp, vp,                   # y, s, w, T_out
            vp, sz, vp,                       # workspace, workspace_bytes, stream
        ]
        lib.ttsdec_postnet_workspace_bytes.restype = sz
        lib.ttsdec_postnet_workspace_bytes.argtypes = [vp, i32, i32]
        lib.ttsdec_postnet.restype = i32
        lib.ttsdec_postnet.argtypes = [vp, vp, i32, i32, i32, vp, vp, sz, vp]
        lib.ttsdec_cell_step.restype = i32
        lib.ttsdec_cell_step.argtypes = [
            vp, vp, vp, i32, i32,             # h, x, memory, B, L
            vp, vp, vp, vp, vp, vp,           # w, ctx, h_att, c_att, h_dec, c_dec
            i32, vp, u64, i32,                # dropout_mode, masks, seed, step
            vp, vp, sz, vp,                   # x_dec, workspace, workspace_bytes, stream
        ]
        lib.ttsdec_profile_step.restype = i32
        lib.ttsdec_profile_step.argtypes = [
            vp, vp, i32, i32, i32, i32, vp, u64,  # h, memory, B, L, iters, dropout_mode, masks, seed
            vp, vp, vp, vp, sz, vp,               # y, s, w, workspace, workspace_bytes, stream
            C.POINTER(f32), C.POINTER(C.c_char_p), i32, C.POINTER(i32),
        ]
        lib.ttsdec_profile_loop.restype = i32
        lib.ttsdec_profile_loop.argtypes = [
            vp, vp, i32, i32, i32, i32, vp, u64,       # h, memory, B, L, n_steps, dropout_mode, masks, seed
            vp, vp, vp, vp, vp, sz, vp,                 # y, s, w, T_out, workspace, workspace_bytes, stream
            C.POINTER(f32), C.POINTER(C.c_char_p), i32, C.POINTER(i32), C.POINTER(f32),
        ]
        lib.ttsenc_create.restype = i32
        lib.ttsenc_create.argtypes = [C.POINTER(EncDims), C.POINTER(vp)]
        lib.ttsenc_destroy.restype = i32
        lib.ttsenc_destroy.argtypes = [vp]
        lib.ttsenc_last_hip_error.restype = C.c_char_p
        lib.ttsenc_last_hip_error.argtypes = [vp]
        lib.ttsenc_num_weight_tensors.restype = i32
        lib.ttsenc_num_weight_tensors.argtypes = [vp]
        lib.ttsenc_packed_bytes.restype = sz
        lib.ttsenc_packed_bytes.argtypes = [vp]
        lib.ttsenc_pack_weights.restype = i32
        lib.ttsenc_pack_weights.argtypes = [vp, C.POINTER(vp), i32, vp, vp]
        lib.ttsenc_bind_weights.restype = i32
        lib.ttsenc_bind_weights.argtypes = [vp, vp]
        lib.ttsenc_workspace_bytes.restype = sz
        lib.ttsenc_workspace_bytes.argtypes = [vp, i32, i32]
        lib.ttsenc_forward.restype = i32
        lib.ttsenc_forward.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, sz, vp, vp]
        lib.ttsvits_create.restype = i32
        lib.ttsvits_create.argtypes = [C.POINTER(VitsDims), C.POINTER(vp)]
        lib.ttsvits_destroy.restype = i32
        lib.ttsvits_destroy.argtypes = [vp]
        lib.ttsenc_set_precision.restype = i32
        lib.ttsenc_set_precision.argtypes = [vp, i32]
        lib.ttsenc_get_precision.restype = i32
        lib.ttsenc_get_precision.argtypes = [vp]
        lib.ttsvits_set_precision.restype = i32
        lib.ttsvits_set_precision.argtypes = [vp, i32]
        lib.ttsvits_get_precision.restype = i32
        lib.ttsvits_get_precision.argtypes = [vp]
        lib.ttsvits_last_hip_error.restype = C.c_char_p
        lib.ttsvits_last_hip_error.argtypes = [vp]
        lib.ttsvits_num_weight_tensors.restype = i32
        lib.ttsvits_num_weight_tensors.argtypes = [vp]
        lib.ttsvits_packed_bytes.restype = sz
        lib.ttsvits_packed_bytes.argtypes = [vp]
        lib.ttsvits_pack_weights.restype = i32
        lib.ttsvits_pack_weights.argtypes = [vp, C.POINTER(vp), i32, vp, vp]
        lib.ttsvits_bind_weights.restype = i32
        lib.ttsvits_bind_weights.argtypes = [vp, vp]
        lib.ttsvits_text_encoder_workspace_bytes.restype = sz
        lib.ttsvits_text_encoder_workspace_bytes.argtypes = [vp, i32, i32]
        lib.ttsvits_text_encoder.restype = i32
        lib.ttsvits_text_encoder.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, sz, vp, vp]
        lib.ttsvits_flow_workspace_bytes.restype = sz
        lib.ttsvits_flow_workspace_bytes.argtypes = [vp, i32, i32]
        lib.ttsvits_flow_reverse.restype = i32
        lib.ttsvits_flow_reverse.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, sz, vp]
        _lib = lib
        return _lib


def check(code: int, what: str, handle=None) -> None:
    if code != OK:
        detail = ""
        if handle is not None and code == ERR_HIP:
            detail = load().ttsdec_last_hip_error(handle).decode()
        raise TtsdecError(code, what, detail)
