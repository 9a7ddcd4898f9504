"""Drop-in ``Encoder`` / ``VQEmbeddingEMA`` for the reference's ``model.py`` (inference path).

Same constructor (``Encoder(conf: ConfEncoder)``, ``model.py:17-34``), same
``state_dict()`` key set (``model.py:43-57``), same ``encode`` / ``forward`` signatures
(``model.py:59-86``).  The torch sub-modules below only HOLD the parameters (so
``load_state_dict(checkpoint["encoder"])``, ``.to(device)``, ``.eval()`` and the
``encoder.encoder[-1]`` forward hook of ``encode.py:34-40`` keep working); all arithmetic
runs in ``libvqcpc_hip.so`` through the C ABI of ``include/vqcpc.h``.
"""
import ctypes as C
import weakref
from dataclasses import dataclass
from itertools import chain
from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib

MISSING = "???"   # stands where the reference uses omegaconf.MISSING (model.py:6)


@dataclass
class ConfEncoder:
    """``model.py:17-31``."""
    in_channels: int = MISSING
    channels: int = MISSING
    n_embeddings: int = MISSING
    z_dim: int = MISSING
    c_dim: int = MISSING


class VQEmbeddingEMA(nn.Module):
    """Codebook buffers (``model.py:89-101``) + the reference's public ``encode`` / eval-mode ``forward``
    (``model.py:103-155``) as thin wrappers over the owning ``Encoder``'s native handle (the same VQ kernel
    ``Encoder.encode`` runs).  The EMA update of training mode (``model.py:136-145``) is outside the
    inference path.
    """

    def __init__(self, n_embeddings, embedding_dim, commitment_cost=0.25, decay=0.999, epsilon=1e-5):
        super().__init__()
        self.commitment_cost, self.decay, self.epsilon = commitment_cost, decay, epsilon
        init_bound = 1 / 512
        embedding = torch.empty(n_embeddings, embedding_dim).uniform_(-init_bound, init_bound)
        self.register_buffer("embedding", embedding)
        self.register_buffer("ema_count", torch.zeros(n_embeddings))
        self.register_buffer("ema_weight", self.embedding.clone())
        self._owner = None                              # weakref to the Encoder that holds the native handle

    def __getstate__(self):                             # copy.deepcopy / pickle: the owner re-links itself
        d = self.__dict__.copy()
        d["_owner"] = None
        return d

    def _rows(self, x: Tensor):
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("VQEmbeddingEMA: the MI355X path serves the codebook through its owning Encoder "
                               "(construct it as Encoder(conf).codebook)")
        _lib.require_cuda(x, "x")
        _lib.require_same_device(x, self.embedding, "x")
        D = self.embedding.size(1)
        if x.dim() < 2 or x.size(-1) != D:
            raise RuntimeError(f"expected x of shape (Batch, Time, {D}), got {tuple(x.shape)}")
        xf = x.detach().to(torch.float32).reshape(-1, D).contiguous()
        q = torch.empty_like(xf)
        idx = torch.empty(xf.size(0), dtype=torch.int64, device=x.device)
        h = owner._native()
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().vqcpc_encoder_vq_encode(h, xf.data_ptr(), xf.size(0), q.data_ptr(), idx.data_ptr(),
                                                           _lib.current_stream()))
        return owner, h, xf, q, idx

    @torch.no_grad()
    def encode(self, x: Tensor):
        """``model.py:103-115``: (quantized, indices (Batch, Time))."""
        _, _, _, q, idx = self._rows(x)
        return q.view_as(x), idx.view(x.size(0), x.size(1))

    def forward(self, x: Tensor):
        """``model.py:117-155`` in eval mode: (x + (q - x), 0.25 * mse, perplexity)."""
        if self.training:
            raise NotImplementedError("VQEmbeddingEMA: the EMA update of training mode (model.py:136-145) is outside "
                                      "the inference path; call .eval()")
        with torch.no_grad():
            _, h, xf, q, idx = self._rows(x)
            z_st = torch.empty_like(xf)
            stats = torch.empty(2, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().vqcpc_encoder_forward_stats(h, xf.data_ptr(), q.data_ptr(), idx.data_ptr(), xf.size(0),
                                                                   z_st.data_ptr(), stats[0:].data_ptr(), stats[1:].data_ptr(),
                                                                   _lib.current_stream()))
        return z_st.view_as(x), stats[0], stats[1]


class Encoder(nn.Module):
    """Spec-Conv1d/k4s2-LN-ReLU-[FC-LN-ReLU]x4-FC-VQ + LSTM (``model.py:33-86``) on MI355X."""

    def __init__(self, conf: ConfEncoder):
        super().__init__()
        self.conf = conf
        self.conv = nn.Conv1d(conf.in_channels, conf.channels, 4, 2, 1, bias=False)
        block = lambda: [nn.Linear(conf.channels, conf.channels, bias=False), nn.LayerNorm(conf.channels), nn.ReLU(True)]
        self.encoder = nn.Sequential(nn.LayerNorm(conf.channels), nn.ReLU(True),
                                     *chain.from_iterable(block() for _ in range(4)),
                                     nn.Linear(conf.channels, conf.z_dim))
        self.codebook = VQEmbeddingEMA(conf.n_embeddings, conf.z_dim)
        self.codebook._owner = weakref.ref(self)
        self.rnn = nn.LSTM(conf.z_dim, conf.c_dim, batch_first=True)
        self._handle = None
        self._handle_key = None

    # ------------------------------------------------------------------ native handle
    _WEIGHT_NAMES = (["conv.weight"] + [f"encoder.{n}.{k}" for n in (0, 3, 6, 9, 12) for k in ("weight", "bias")] +
                     [f"encoder.{n}.weight" for n in (2, 5, 8, 11)] + ["encoder.14.weight", "encoder.14.bias",
                      "codebook.embedding", "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0"])

    def _native(self):
        slots = self.__dict__.get("_slots")
        if slots is None:
            slots = self.__dict__["_slots"] = _lib.WeightSlots(self, self._WEIGHT_NAMES)
        ws = slots.tensors()
        key = _lib.WeightSlots.key(ws)
        if self._handle is not None and key == self._handle_key:
            return self._handle
        for w in ws:
            _lib.require_cuda(w, "Encoder parameter")
            if w.dtype != torch.float32:
                raise RuntimeError("Encoder: parameters must be float32")
            _lib.require_same_device(w, ws[0], "a parameter")
        self._release()
        sd = dict(zip(slots.names, ws))
        keep = []

        def p(name):
            t = sd[name].detach().contiguous()
            keep.append(t)
            return t.data_ptr()

        w = _lib.EncoderWeights()
        w.conv_weight = p("conv.weight")
        for i, n in enumerate((0, 3, 6, 9, 12)):
            w.ln_weight[i], w.ln_bias[i] = p(f"encoder.{n}.weight"), p(f"encoder.{n}.bias")
        for i, n in enumerate((2, 5, 8, 11)):
            w.fc_weight[i] = p(f"encoder.{n}.weight")
        w.out_weight, w.out_bias = p("encoder.14.weight"), p("encoder.14.bias")
        w.codebook = p("codebook.embedding")
        w.rnn_w_ih, w.rnn_w_hh = p("rnn.weight_ih_l0"), p("rnn.weight_hh_l0")
        w.rnn_b_ih, w.rnn_b_hh = p("rnn.bias_ih_l0"), p("rnn.bias_hh_l0")
        c = self.conf
        w.in_channels, w.channels, w.n_embeddings, w.z_dim, w.c_dim = (
            c.in_channels, c.channels, c.n_embeddings, c.z_dim, c.c_dim)
        h = C.c_void_p()
        with torch.cuda.device(ws[0].device):
            torch.cuda.current_stream().synchronize()
            _lib.check(_lib.load().vqcpc_encoder_create(C.byref(w), C.byref(h)))
        self._handle, self._handle_key = h, key
        for name, value in self.__dict__.get("_options", {}).items():     # options survive a rebuild of the handle
            _lib.check(_lib.load().vqcpc_encoder_set_option(h, name.encode(), value))
        return h

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.load().vqcpc_encoder_destroy(self._handle)
            self._handle = None

    def __getstate__(self):                             # the native handle is per object: a copy builds its own
        d = self.__dict__.copy()
        d["_handle"], d["_handle_key"] = None, None
        d.pop("_slots", None)
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        self.codebook._owner = weakref.ref(self)

    def set_option(self, name: str, value: int):
        """``vqcpc_encoder_set_option`` (``fused``: -1 auto, 0 layered kernels, 1 fused front end).  Options are kept on
        the Python object and re-applied when the native handle is rebuilt (``.to()``, ``load_state_dict``)."""
        _lib.check(_lib.load().vqcpc_encoder_set_option(self._native(), name.encode(), int(value)))
        self.__dict__.setdefault("_options", {})[name] = int(value)

    def check(self):
        """Synchronise the current stream and raise ``RuntimeError`` if the resident context scan of the last ``encode``
        gave up on an in-kernel exchange (``vqcpc_encoder_check``): that call's ``c`` is incomplete, the handle has fallen
        back to one launch per time step, and repeating the call gives the right result."""
        if self._handle is None:
            return
        torch.cuda.current_stream().synchronize()
        _lib.check(_lib.load().vqcpc_encoder_check(self._handle))

    def refresh(self):
        """Drop the native handle so that the next call re-reads the parameters.  The handle holds re-laid
        COPIES of the weights and is rebuilt automatically when a parameter's storage or ``_version`` changes
        (``load_state_dict``, ``.to``, optimizer steps); a write through ``.data`` bumps neither -- call this
        after one."""
        self._release()
        self._handle_key = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference surface
    def _encode_native(self, mel: Tensor, want_c: bool, conv_mode: int = 0, want_pre: bool = False):
        _lib.require_cuda(mel, "mel")
        if mel.dim() != 3 or mel.size(1) != self.conf.in_channels:
            raise RuntimeError(f"expected mel of shape (B, {self.conf.in_channels}, T), got {tuple(mel.shape)}")
        if mel.size(2) < 2:
            raise RuntimeError("Conv1d(k=4, s=2, p=1) needs at least 2 mel frames")
        _lib.require_same_device(mel, self.conv.weight, "mel")
        if mel.dtype != torch.float32 or not mel.is_contiguous() or mel.requires_grad:
            mel = mel.detach().to(torch.float32).contiguous()
        B, _, T = mel.shape
        To = (T - 2) // 2 + 1                        # nn.Conv1d(k4, s2, p1) output length (model.py:43)
        h = self._native()
        dev = mel.device
        last = self.encoder._modules["14"]
        hooks = last._forward_hooks
        z = torch.empty(B, To, self.conf.z_dim, device=dev)
        z_pre = torch.empty_like(z) if (want_pre or hooks) else None       # pre-VQ rows only when somebody reads them
        idx = torch.empty(B, To, dtype=torch.int64, device=dev)
        c = torch.empty(B, To, self.conf.c_dim, device=dev) if want_c else None
        with _lib.device_guard(dev):
            _lib.check(_lib.load().vqcpc_encoder_encode(
                h, mel.data_ptr(), B, T, conv_mode, z.data_ptr(), c.data_ptr() if want_c else None,
                idx.data_ptr(), z_pre.data_ptr() if z_pre is not None else None, _lib.current_stream()))
        for hook in list(hooks.values()):                   # encode.py:34-40 captures the pre-VQ activations
            hook(last, (None,), z_pre)
        return z, c, idx, z_pre

    @torch.no_grad()
    def encode(self, mel: Tensor, conv_mode: int = 0) -> Tuple[Tensor, Tensor, Tensor]:
        """``model.py:59-70``: (z, c, indices).  ``conv_mode``: see ``vqcpc.h`` (0 = as the reference)."""
        z, c, idx, _ = self._encode_native(mel, want_c=True, conv_mode=conv_mode)
        return z, c, idx

    @torch.no_grad()
    def encode_indices(self, mel: Tensor, conv_mode: int = 0) -> Tensor:
        """What ``convert.py:76`` keeps of ``encode``: the code indices (LSTM skipped)."""
        return self._encode_native(mel, want_c=False, conv_mode=conv_mode)[2]

    @torch.no_grad()
    def stage(self, mel: Tensor, stage: int, conv_mode: int = 0) -> Tensor:
        """Activations after one front-end stage (``vqcpc_encoder_stage``), rows (B, T/2, F)."""
        _lib.require_cuda(mel, "mel")
        _lib.require_same_device(mel, self.conv.weight, "mel")
        mel = mel.detach().to(torch.float32).contiguous()
        B, _, T = mel.shape
        F = self.conf.z_dim if stage == 10 else self.conf.channels
        out = torch.empty(B, (T - 2) // 2 + 1, F, device=mel.device)
        with torch.cuda.device(mel.device):
            _lib.check(_lib.load().vqcpc_encoder_stage(self._native(), mel.data_ptr(), B, T, conv_mode, stage,
                                                       out.data_ptr(), _lib.current_stream()))
        return out

    def forward(self, mels: Tensor):
        """``model.py:72-86`` in eval mode: (z, c, vq_loss, perplexity)."""
        if self.training:
            raise NotImplementedError("vectorquantizedcpc_amd.Encoder implements the inference path; "
                                      "call .eval() (the EMA/straight-through training branch, model.py:136-145, is out of scope)")
        with torch.no_grad():
            zq, _, idx, z_pre = self._encode_native(mels, want_c=False, want_pre=True)
            B, Tz, D = zq.shape
            z_st = torch.empty_like(zq)
            stats = torch.empty(2, device=zq.device)
            c = torch.empty(B, Tz, self.conf.c_dim, device=zq.device)
            lib, h, s = _lib.load(), self._native(), _lib.current_stream()
            with torch.cuda.device(zq.device):
                _lib.check(lib.vqcpc_encoder_forward_stats(h, z_pre.data_ptr(), zq.data_ptr(), idx.data_ptr(), B * Tz,
                                                           z_st.data_ptr(), stats[0:].data_ptr(), stats[1:].data_ptr(), s))
                _lib.check(lib.vqcpc_encoder_context(h, z_st.data_ptr(), B, Tz, c.data_ptr(), s))
        return z_st, c, stats[0], stats[1]
