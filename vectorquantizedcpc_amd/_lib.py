"""ctypes binding of ``libvqcpc_hip.so`` (C ABI declared in ``include/vqcpc.h``).

There is no CPU fallback: if the shared library is missing this module raises, and every
compute entry point fails on a machine without a gfx950 device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvqcpc_hip.so")

# every symbol include/vqcpc.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "vqcpc_abi_version", "vqcpc_last_error", "vqcpc_device_count",
    "vqcpc_encoder_create", "vqcpc_encoder_destroy", "vqcpc_encoder_encode",
    "vqcpc_encoder_forward_stats", "vqcpc_encoder_context", "vqcpc_encoder_stage", "vqcpc_encoder_vq_encode", "vqcpc_encoder_set_option",
    "vqcpc_encoder_check", "vqcpc_vocoder_check", "vqcpc_vocoder_last_path", "vqcpc_vocoder_last_slots", "vqcpc_vocoder_workspace_bytes", "vqcpc_vocoder_plan",
    "vqcpc_vocoder_create", "vqcpc_vocoder_destroy", "vqcpc_vocoder_generate",
    "vqcpc_vocoder_logits", "vqcpc_vocoder_condition", "vqcpc_vocoder_glue", "vqcpc_vocoder_set_option",
    "vqcpc_vocoder_last_timing", "vqcpc_vocoder_kernel_times",
    "vqcpc_melfront_create", "vqcpc_melfront_destroy", "vqcpc_melfront_frames", "vqcpc_melfront_run",
    "vqcpc_loudness_create", "vqcpc_loudness_destroy", "vqcpc_loudness_blocks", "vqcpc_loudness_integrated",
    "vqcpc_loudness_normalize",
    "vqcpc_resampler_create", "vqcpc_resampler_destroy", "vqcpc_resampler_out_len", "vqcpc_resampler_run",
]


class EncoderWeights(C.Structure):
    _fields_ = [("conv_weight", C.c_void_p), ("ln_weight", C.c_void_p * 5), ("ln_bias", C.c_void_p * 5),
                ("fc_weight", C.c_void_p * 4), ("out_weight", C.c_void_p), ("out_bias", C.c_void_p),
                ("codebook", C.c_void_p), ("rnn_w_ih", C.c_void_p), ("rnn_w_hh", C.c_void_p),
                ("rnn_b_ih", C.c_void_p), ("rnn_b_hh", C.c_void_p),
                ("in_channels", C.c_int), ("channels", C.c_int), ("n_embeddings", C.c_int),
                ("z_dim", C.c_int), ("c_dim", C.c_int)]


class VocoderWeights(C.Structure):
    _fields_ = [("code_embedding", C.c_void_p), ("speaker_embedding", C.c_void_p),
                ("prenet_w_ih", (C.c_void_p * 2) * 2), ("prenet_w_hh", (C.c_void_p * 2) * 2),
                ("prenet_b_ih", (C.c_void_p * 2) * 2), ("prenet_b_hh", (C.c_void_p * 2) * 2),
                ("ar_embedding", C.c_void_p), ("ar_w_ih", C.c_void_p), ("ar_w_hh", C.c_void_p),
                ("ar_b_ih", C.c_void_p), ("ar_b_hh", C.c_void_p),
                ("fc1_weight", C.c_void_p), ("fc1_bias", C.c_void_p), ("fc2_weight", C.c_void_p), ("fc2_bias", C.c_void_p),
                ("n_codes", C.c_int), ("dz", C.c_int), ("n_speakers", C.c_int), ("ds", C.c_int), ("Hp", C.c_int),
                ("de", C.c_int), ("Hr", C.c_int), ("Hf", C.c_int), ("n_cls", C.c_int),
                ("upsample_t", C.c_int), ("bits_mu_law", C.c_int)]


_lib = None


def load():
    """Load the HIP library (raises if it has not been built: ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C vectorquantizedcpc_amd/csrc` "
            "(or __graft_entry__.build()).  vectorquantizedcpc_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64p = C.c_void_p, C.c_int, C.c_void_p
    lib.vqcpc_abi_version.restype = i32
    lib.vqcpc_last_error.restype = C.c_char_p
    lib.vqcpc_device_count.restype = i32
    lib.vqcpc_encoder_create.argtypes = [C.POINTER(EncoderWeights), C.POINTER(vp)]
    lib.vqcpc_encoder_destroy.argtypes = [vp]
    lib.vqcpc_encoder_destroy.restype = None
    lib.vqcpc_encoder_encode.argtypes = [vp, vp, i32, i32, i32, vp, vp, i64p, vp, vp]
    lib.vqcpc_encoder_forward_stats.argtypes = [vp, vp, vp, i64p, i32, vp, vp, vp, vp]
    lib.vqcpc_encoder_context.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.vqcpc_encoder_stage.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.vqcpc_encoder_vq_encode.argtypes = [vp, vp, i32, vp, i64p, vp]
    lib.vqcpc_encoder_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.vqcpc_encoder_check.argtypes = [vp]
    lib.vqcpc_vocoder_check.argtypes = [vp]
    lib.vqcpc_vocoder_last_path.argtypes = [vp]
    lib.vqcpc_vocoder_last_slots.argtypes = [vp]
    lib.vqcpc_vocoder_workspace_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.vqcpc_vocoder_plan.argtypes = [i32] * 7 + [C.POINTER(C.c_int), i32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    lib.vqcpc_vocoder_create.argtypes = [C.POINTER(VocoderWeights), C.POINTER(vp)]
    lib.vqcpc_vocoder_destroy.argtypes = [vp]
    lib.vqcpc_vocoder_destroy.restype = None
    lib.vqcpc_vocoder_generate.argtypes = [vp, i64p, i64p, i32, i32, C.POINTER(C.c_int), C.c_uint64, C.c_uint32,
                                           C.POINTER(C.c_uint32), vp, i64p, i32, vp]
    lib.vqcpc_vocoder_logits.argtypes = [vp, i64p, i64p, i64p, i32, i32, i32, vp, vp]
    lib.vqcpc_vocoder_condition.argtypes = [vp, i64p, i64p, i32, i32, vp, vp]
    lib.vqcpc_vocoder_glue.argtypes = [vp, i64p, i64p, i32, i32, vp, vp]
    lib.vqcpc_vocoder_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.vqcpc_vocoder_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.vqcpc_vocoder_kernel_times.argtypes = [vp, i32, C.POINTER(C.c_float), vp]
    lib.vqcpc_melfront_create.argtypes = [i32, i32, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.POINTER(vp)]
    lib.vqcpc_melfront_destroy.argtypes = [vp]
    lib.vqcpc_melfront_destroy.restype = None
    lib.vqcpc_melfront_frames.argtypes = [vp, i32]
    lib.vqcpc_melfront_run.argtypes = [vp, vp, C.POINTER(C.c_int), i32, i32, vp, vp]
    lib.vqcpc_loudness_create.argtypes = [i32, C.POINTER(vp)]
    lib.vqcpc_loudness_destroy.argtypes = [vp]
    lib.vqcpc_loudness_destroy.restype = None
    lib.vqcpc_loudness_blocks.argtypes = [vp, i32]
    lib.vqcpc_loudness_integrated.argtypes = [vp, vp, C.POINTER(C.c_int), i32, i32, vp, vp, vp]
    lib.vqcpc_loudness_normalize.argtypes = [vp, vp, C.POINTER(C.c_int), i32, i32, vp, vp, vp]
    lib.vqcpc_resampler_create.argtypes = [i32, i32, C.POINTER(vp)]
    lib.vqcpc_resampler_destroy.argtypes = [vp]
    lib.vqcpc_resampler_destroy.restype = None
    lib.vqcpc_resampler_out_len.argtypes = [vp, i32]
    lib.vqcpc_resampler_run.argtypes = [vp, vp, C.POINTER(C.c_int), i32, i32, vp, i32, vp]
    _lib = lib
    return lib


def check(rc: int):
    """Non-zero status -> RuntimeError(vqcpc_last_error()), like torch raising from an operator."""
    if rc != 0:
        raise RuntimeError(f"libvqcpc_hip: {load().vqcpc_last_error().decode()} (status {rc})")


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


class WeightSlots:
    """The tensors behind a fixed list of ``state_dict()`` names, re-read per call WITHOUT ``state_dict()`` (which
    costs ~70 us on these modules -- more than a small kernel call): each name is resolved once to the
    ``_parameters`` / ``_buffers`` dict that holds it, so a later ``.to()``, ``load_state_dict`` or assignment is
    still seen.  ``key()`` = ((data_ptr, _version), ...), what the native handles are cached on."""

    def __init__(self, module, names):
        self.names = list(names)
        self.module = module
        self._resolve()

    def _resolve(self):
        self.slots, self.owners = [], []
        for name in self.names:
            *path, leaf = name.split(".")
            mod, chain = self.module, []
            for part in path:
                chain.append((mod._modules, part, mod._modules[part]))
                mod = mod._modules[part]
            self.slots.append((mod._parameters if leaf in mod._parameters else mod._buffers, leaf))
            self.owners.append(chain)

    def tensors(self):
        # a submodule replaced after the first call (enc.rnn = nn.LSTM(...)) leaves the resolved dicts pointing at the old
        # one: one identity check per path element (~1 us for the whole list) catches it
        for chain in self.owners:
            for mods, part, seen in chain:
                if mods.get(part) is not seen:
                    self._resolve()
                    return [d[k] for d, k in self.slots]
        return [d[k] for d, k in self.slots]

    @staticmethod
    def key(tensors):
        dev = tensors[0].device
        return (dev.type, dev.index) + tuple([(t.data_ptr(), t._version) for t in tensors])


def device_guard(dev):
    """``torch.cuda.device(dev)`` only when ``dev`` is not already current (the context manager costs ~5 us)."""
    import contextlib
    import torch
    return contextlib.nullcontext() if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def require_same_device(t, ref, what: str):
    """torch's 'expected all tensors to be on the same device' for an input next to the module's weights."""
    if t.device != ref.device:
        raise RuntimeError(f"Expected all tensors to be on the same device, but {what} is on {t.device} and the "
                           f"module's parameters are on {ref.device}")


def require_cuda(t, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} is on {t.device}: vectorquantizedcpc_amd runs on MI355X only "
                           "(move the module and its inputs with .to('cuda')); there is no CPU fallback")
