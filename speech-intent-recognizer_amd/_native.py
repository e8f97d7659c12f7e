"""ctypes binding of libsir_hip.so (C ABI: include/sir_hip.h).

There is no fallback: ``lib()`` raises if the library is missing (build it with
``python __graft_entry__.py`` or ``make -C speech-intent-recognizer_amd/csrc``), and every op that
computes raises if its tensors are not on a HIP device.
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsir_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

SIR_OK = 0
SIR_ETIMEOUT = -5
WAVE_F32, WAVE_I16 = 0, 1
BWD_ALL, BWD_HEAD_GRU, BWD_CNN = 0, 1, 2


class FeatureConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("n_fft", C.c_int), ("hop_length", C.c_int), ("n_mels", C.c_int),
                ("f_min", C.c_float), ("f_max", C.c_float), ("window", C.c_void_p), ("mel_fb", C.c_void_p)]


class Augment(C.Structure):
    _fields_ = [("shift", C.c_void_p), ("noise_sigma", C.c_void_p), ("noise_seed", C.c_uint64),
                ("time_mask", C.c_void_p), ("freq_mask", C.c_void_p)]


class ModelWeights(C.Structure):
    _fields_ = [("conv_w", C.c_void_p * 3), ("bn_w", C.c_void_p * 3), ("bn_b", C.c_void_p * 3),
                ("bn_mean", C.c_void_p * 3), ("bn_var", C.c_void_p * 3),
                ("gru_w_ih", C.c_void_p * 4), ("gru_w_hh", C.c_void_p * 4),
                ("gru_b_ih", C.c_void_p * 4), ("gru_b_hh", C.c_void_p * 4),
                ("attn_w", C.c_void_p), ("attn_b", C.c_void_p), ("fc_w", C.c_void_p), ("fc_b", C.c_void_p),
                ("num_classes", C.c_int)]


class ModelGrads(C.Structure):
    _fields_ = [("conv_w", C.c_void_p * 3), ("bn_w", C.c_void_p * 3), ("bn_b", C.c_void_p * 3),
                ("gru_w_ih", C.c_void_p * 4), ("gru_w_hh", C.c_void_p * 4),
                ("gru_b_ih", C.c_void_p * 4), ("gru_b_hh", C.c_void_p * 4),
                ("attn_w", C.c_void_p), ("attn_b", C.c_void_p), ("fc_w", C.c_void_p), ("fc_b", C.c_void_p)]


# name -> (restype, argtypes); must list every function declared in include/sir_hip.h
SIGNATURES = {
    "sir_abi_version": (C.c_int, []),
    "sir_last_error": (C.c_char_p, []),
    "sir_create": (C.c_int, [C.POINTER(FeatureConfig), C.POINTER(C.c_void_p)]),
    "sir_destroy": (C.c_int, [C.c_void_p]),
    "sir_features_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "sir_features_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Augment),
                                   C.c_void_p]),
    "sir_mix_to_mono": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.c_int64, C.c_void_p]),
    "sir_resample_out_len": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "sir_resample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "sir_gather_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "sir_model_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "sir_model_workspace_offsets": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_int]),
    "sir_model_set_weights_version": (C.c_int, [C.c_void_p, C.c_uint64]),
    "sir_model_infer": (C.c_int, [C.c_void_p, C.POINTER(ModelWeights), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sir_check_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sir_pipeline_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "sir_pipeline_destroy": (C.c_int, [C.c_void_p]),
    "sir_pipeline_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "sir_pipeline_end": (C.c_int, [C.c_void_p, C.c_int]),
    "sir_pipeline_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sir_model_train_fwd": (C.c_int, [C.c_void_p, C.POINTER(ModelWeights), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                      C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "sir_ce_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                              C.c_void_p]),
    "sir_model_train_bwd": (C.c_int, [C.c_void_p, C.POINTER(ModelWeights), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_float, C.c_uint64, C.POINTER(ModelGrads), C.c_void_p, C.c_size_t, C.c_void_p]),
    "sir_model_train_bwd_part": (C.c_int, [C.c_void_p, C.POINTER(ModelWeights), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_float, C.c_uint64, C.POINTER(ModelGrads), C.c_void_p, C.c_size_t, C.c_int,
                                           C.c_void_p]),
    "sir_model_train_workspace_offsets": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_int]),
    "sir_adam_step": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_void_p]),
    "sir_profile_kernel_count": (C.c_int, []),
    "sir_profile_kernel_name": (C.c_char_p, [C.c_int]),
    "sir_profile_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "sir_profile_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
}

_lib = None
_lock = threading.Lock()


class SirError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into lib/libsir_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j4"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-8000:])
    if res.returncode != 0:
        raise SirError("building libsir_hip.so failed")
    return LIB_PATH


def lib():
    """The loaded library; raises SirError if it has not been built (no CPU fallback exists)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SirError(f"{LIB_PATH} is missing: the HIP extension is required, there is no CPU "
                               "fallback (run `python __graft_entry__.py` to build it)")
            # torch ships its own libamdhip64; import it FIRST so that libsir_hip.so binds to the
            # HIP runtime torch uses (two runtimes in one process do not share devices/streams)
            import torch  # noqa: F401
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def check(rc, what=""):
    if rc != SIR_OK:
        msg = lib().sir_last_error().decode(errors="replace")
        raise SirError(f"{what} failed with code {rc}: {msg}")


def require_hip(*tensors):
    import torch
    if not torch.cuda.is_available():
        raise SirError("no HIP device visible: this path runs on MI355X only (no CPU fallback)")
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SirError("tensor is not on a HIP device: this path runs on MI355X only (no CPU fallback)")


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
