"""Drop-in ``Vocoder`` for the reference's ``network_vocoder.py`` (inference path).

``Vocoder(conf: ConfVocoder)``, ``generate(z, speaker)`` and ``forward(x, z, speaker)`` keep
the reference signatures (``network_vocoder.py:31``, ``:41``, ``:69``).  The RNN_MS core the
reference imports from the third-party ``rnnms`` package (``network_vocoder.py:8``) is not
available offline; ``RNNMSVocoder`` below is this project's statement of it (shapes from
``config.py:67-77``) -- parity with ``rnnms`` itself is unpinned.  Sub-modules only
hold parameters; all arithmetic runs in ``libvqcpc_hip.so``.
"""
import ctypes as C
from dataclasses import dataclass, field

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib

MISSING = "???"


@dataclass
class ConfPreNet:
    num_layers: int = 2           # config.py:72
    bidirectional: bool = True    # config.py:73


@dataclass
class ConfWaveAR:
    size_i_embed_ar: int = 256    # config.py:75
    size_h_rnn: int = 896         # config.py:76
    size_h_fc: int = 256          # config.py:77


@dataclass
class ConfRNNMSVocoder:
    """``config.py:67-77`` (+ ``dim_i_feature`` set at ``config.py:198-199``)."""
    dim_i_feature: int = 128
    dim_voc_latent: int = 256
    bits_mu_law: int = 8
    upsampling_t: int = 160
    prenet: ConfPreNet = field(default_factory=ConfPreNet)
    wave_ar: ConfWaveAR = field(default_factory=ConfWaveAR)


@dataclass
class ConfVocoder:
    """``network_vocoder.py:11-24``; defaults from ``config.py:62-66``."""
    size_i_codebook: int = 512
    dim_i_embedding: int = 64
    n_speakers: int = 102
    dim_speaker_embedding: int = 64
    rnnms: ConfRNNMSVocoder = field(default_factory=ConfRNNMSVocoder)


class _WaveAR(nn.Module):
    def __init__(self, size_i_cnd, conf: ConfWaveAR, size_o):
        super().__init__()
        self.embedding = nn.Embedding(size_o, conf.size_i_embed_ar)
        self.rnn = nn.GRU(conf.size_i_embed_ar + size_i_cnd, conf.size_h_rnn, batch_first=True)
        self.fc1 = nn.Linear(conf.size_h_rnn, conf.size_h_fc)
        self.fc2 = nn.Linear(conf.size_h_fc, size_o)


class RNNMSVocoder(nn.Module):
    """Parameter container of the RNN_MS core: bi-GRU PreNet + embedding-AR GRU + 2 FC."""

    def __init__(self, conf: ConfRNNMSVocoder):
        super().__init__()
        if conf.prenet.num_layers != 2 or not conf.prenet.bidirectional:
            raise ValueError("RNNMSVocoder: the MI355X path implements the 2-layer bidirectional PreNet of config.py:71-73")
        self.conf = conf
        self.prenet = nn.GRU(conf.dim_i_feature, conf.dim_voc_latent // 2, num_layers=2, batch_first=True,
                             bidirectional=True)
        self.ar = _WaveAR(conf.dim_voc_latent, conf.wave_ar, 2 ** conf.bits_mu_law)


class Vocoder(nn.Module):
    """bidirectional_PreNet + WaveRNN (=RNN_MS) conditioned on VQ-CPC codes (``network_vocoder.py:26-78``)."""

    def __init__(self, conf: ConfVocoder):
        super().__init__()
        self.conf = conf
        self.code_embedding = nn.Embedding(conf.size_i_codebook, conf.dim_i_embedding)
        self.speaker_embedding = nn.Embedding(conf.n_speakers, conf.dim_speaker_embedding)
        if conf.rnnms.dim_i_feature != conf.dim_i_embedding + conf.dim_speaker_embedding:
            raise ValueError("rnnms.dim_i_feature must equal dim_i_embedding + dim_speaker_embedding (config.py:198-199)")
        self.rnnms = RNNMSVocoder(conf.rnnms)
        self._handle = None
        self._handle_key = None
        self._utterances_done = 0

    # ------------------------------------------------------------------ native handle
    def _native(self):
        slots = self.__dict__.get("_slots")
        if slots is None:                       # resolved once: state_dict() costs more than a short decode call's launch
            slots = self.__dict__["_slots"] = _lib.WeightSlots(self, list(self.state_dict().keys()))
        ws = slots.tensors()
        key = _lib.WeightSlots.key(ws)
        if self._handle is not None and key == self._handle_key:
            return self._handle
        for w in ws:
            _lib.require_cuda(w, "Vocoder parameter")
            if w.dtype != torch.float32:
                raise RuntimeError("Vocoder: parameters must be float32")
            _lib.require_same_device(w, ws[0], "a parameter")
        self._release()
        sd = dict(zip(slots.names, ws))
        keep = []

        def p(name):
            t = sd[name].detach().contiguous()
            keep.append(t)
            return t.data_ptr()

        w = _lib.VocoderWeights()
        w.code_embedding, w.speaker_embedding = p("code_embedding.weight"), p("speaker_embedding.weight")
        for layer in range(2):
            for d, suf in enumerate(("", "_reverse")):
                w.prenet_w_ih[layer][d] = p(f"rnnms.prenet.weight_ih_l{layer}{suf}")
                w.prenet_w_hh[layer][d] = p(f"rnnms.prenet.weight_hh_l{layer}{suf}")
                w.prenet_b_ih[layer][d] = p(f"rnnms.prenet.bias_ih_l{layer}{suf}")
                w.prenet_b_hh[layer][d] = p(f"rnnms.prenet.bias_hh_l{layer}{suf}")
        w.ar_embedding = p("rnnms.ar.embedding.weight")
        w.ar_w_ih, w.ar_w_hh = p("rnnms.ar.rnn.weight_ih_l0"), p("rnnms.ar.rnn.weight_hh_l0")
        w.ar_b_ih, w.ar_b_hh = p("rnnms.ar.rnn.bias_ih_l0"), p("rnnms.ar.rnn.bias_hh_l0")
        w.fc1_weight, w.fc1_bias = p("rnnms.ar.fc1.weight"), p("rnnms.ar.fc1.bias")
        w.fc2_weight, w.fc2_bias = p("rnnms.ar.fc2.weight"), p("rnnms.ar.fc2.bias")
        c, r = self.conf, self.conf.rnnms
        w.n_codes, w.dz, w.n_speakers, w.ds = c.size_i_codebook, c.dim_i_embedding, c.n_speakers, c.dim_speaker_embedding
        w.Hp, w.de, w.Hr, w.Hf = r.dim_voc_latent // 2, r.wave_ar.size_i_embed_ar, r.wave_ar.size_h_rnn, r.wave_ar.size_h_fc
        w.n_cls, w.upsample_t, w.bits_mu_law = 2 ** r.bits_mu_law, r.upsampling_t, r.bits_mu_law
        h = C.c_void_p()
        with torch.cuda.device(ws[0].device):
            torch.cuda.current_stream().synchronize()
            _lib.check(_lib.load().vqcpc_vocoder_create(C.byref(w), C.byref(h)))
        self._handle, self._handle_key = h, key
        for name, value in self.__dict__.get("_options", {}).items():     # options survive a rebuild of the handle
            _lib.check(_lib.load().vqcpc_vocoder_set_option(h, name.encode(), value))
        return h

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.load().vqcpc_vocoder_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def __getstate__(self):                             # the native handle is per object: a copy builds its own
        d = self.__dict__.copy()
        d["_handle"], d["_handle_key"] = None, None
        d.pop("_slots", None)
        return d

    def refresh(self):
        """Drop the native handle (re-laid COPIES of the weights) so the next call re-reads the parameters: needed
        after an in-place write through ``.data``, which changes neither a parameter's storage nor its ``_version``."""
        self._release()
        self._handle_key = None

    def set_option(self, name: str, value: int):
        """Decode-loop options of ``vqcpc_vocoder_set_option`` (``use_graph``, ``steps_per_graph``, ``xcd``, ...).  They are
        kept on the Python object and re-applied when the native handle is rebuilt (``.to()``, ``load_state_dict``)."""
        _lib.check(_lib.load().vqcpc_vocoder_set_option(self._native(), name.encode(), int(value)))
        self.__dict__.setdefault("_options", {})[name] = int(value)

    def check(self):
        """Synchronise the current stream and raise if a call since the last check went wrong in a way only the device
        knows (``vqcpc_vocoder_check``): ``IndexError`` for a code index or speaker id outside its embedding table (what
        ``nn.Embedding`` raises at ``network_vocoder.py:73,75``), ``RuntimeError`` if an in-kernel hand-off timed out or the
        resident decoders' workgroups were not dealt 32 per XCD -- that call's waveform is incomplete (or was not written),
        and repeating the call gives the samples the fast path would have produced."""
        if self._handle is None:
            return
        torch.cuda.current_stream().synchronize()
        rc = _lib.load().vqcpc_vocoder_check(self._handle)
        if rc != 0:
            msg = _lib.load().vqcpc_last_error().decode()
            if "index out of range in self" in msg:
                raise IndexError("index out of range in self")
            raise RuntimeError(f"libvqcpc_hip: {msg} (status {rc})")

    def last_timing(self):
        """(milliseconds, samples) of the last decode loop, from HIP events on its stream."""
        ms, n = C.c_float(), C.c_int()
        torch.cuda.current_stream().synchronize()
        _lib.check(_lib.load().vqcpc_vocoder_last_timing(self._native(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_path(self) -> int:
        """Decode loop of the last call (``vqcpc_vocoder_last_path``): 2 the per-XCD resident decoders (up to 68 utterances in
        flight), 3 their matrix-core form (69 .. 511), 0 the launch-per-step kernels -- which also take every call whose
        dimensions are not the reference's (size_h_rnn 896, size_h_fc 256, 8-bit mu-law: ``config.py:69,76-77``), at about
        half the speed; 1 = nothing has run yet."""
        return int(_lib.load().vqcpc_vocoder_last_path(self._native()))

    def last_slots(self) -> int:
        """Decode slots the last call's loop ran through (``vqcpc_vocoder_last_slots``)."""
        return int(_lib.load().vqcpc_vocoder_last_slots(self._native()))

    def workspace_bytes(self) -> int:
        """Device bytes of the native handle's grow-only work buffers (conditioning rows, schedules, exchange areas)."""
        n = C.c_uint64()
        _lib.check(_lib.load().vqcpc_vocoder_workspace_bytes(self._native(), C.byref(n)))
        return int(n.value)

    def kernel_times(self, reps: int = 1000):
        """(us GRU step, us fc1, us fc2 + draw, decode slots per launch, GRU kernel kind): ``vqcpc_vocoder_kernel_times``."""
        out = (C.c_float * 5)()
        _lib.check(_lib.load().vqcpc_vocoder_kernel_times(self._native(), int(reps), out, _lib.current_stream()))
        return tuple(float(v) for v in out)

    # ------------------------------------------------------------------ reference surface
    def _prep(self, z: Tensor, speaker: Tensor):
        _lib.require_cuda(z, "z")
        _lib.require_same_device(z, self.code_embedding.weight, "z")
        if z.dim() != 2 or speaker.dim() != 1 or speaker.size(0) != z.size(0):
            raise RuntimeError(f"expected z (B, T') and speaker (B), got {tuple(z.shape)} and {tuple(speaker.shape)}")
        if z.is_floating_point() or speaker.is_floating_point():
            raise RuntimeError("z and speaker must be integer tensors (nn.Embedding indices, network_vocoder.py:73,75)")
        z = z.detach().to(torch.int64).contiguous()
        speaker = speaker.detach().to(device=z.device, dtype=torch.int64).contiguous()
        # nn.Embedding raises on an out-of-range index (network_vocoder.py:73,75).  The glue kernel checks every index it reads
        # and reports through the handle's status word, which check() turns into the same IndexError: no reduce kernels and
        # no device synchronisation here (round 3 read min / max back on every call).
        return z, speaker

    @torch.no_grad()
    def generate(self, z: Tensor, speaker: Tensor, *, n_codes=None, seed=None, utt_base=None, utt_ids=None,
                 return_mulaw: bool = False, max_steps: int = 0, async_: bool = False):
        """``network_vocoder.py:69-78``: waveform (B, 2*upsampling_t*T') from code indices and speaker ids.

        Keyword extras (not in the reference): ``n_codes`` per-utterance valid code counts of a
        padded batch; ``seed`` / ``utt_base`` / ``utt_ids`` of the sampling protocol (default: torch's
        seed and the number of utterances this module has generated so far; ``utt_ids`` gives every
        row its own stream id, so results do not depend on batching); ``return_mulaw`` also returns
        the int64 mu-law classes.

        The reference's caller takes the result to the host right away (``convert.py:77-83``), so by default this call
        synchronises its stream once and checks the handle's status word (``check()``): an index outside its table raises
        ``IndexError`` like ``nn.Embedding``; if an in-kernel hand-off of the resident decoders gave up (a shared GPU), the
        call is repeated ONCE -- same sampling streams, so the same samples -- with a warning.  ``async_=True`` only enqueues
        the work (no synchronisation: pipelined callers); such callers call ``check()`` themselves before they use the
        waveform.
        """
        if not async_:
            kw = dict(n_codes=n_codes, seed=seed, utt_base=utt_base, utt_ids=utt_ids, return_mulaw=return_mulaw, max_steps=max_steps)
            if utt_base is None:
                kw["utt_base"] = self._utterances_done             # a repeat must draw from the same streams
            out = self.generate(z, speaker, async_=True, **kw)
            try:
                self.check()
            except RuntimeError as e:
                import warnings
                warnings.warn(f"Vocoder.generate: decode repeated ({e})")
                out = self.generate(z, speaker, async_=True, **kw)
                self.check()
            if utt_base is None and utt_ids is None:
                self._utterances_done += int(z.size(0))
            return out
        z, speaker = self._prep(z, speaker)
        B, Tc = z.shape
        h = self._native()
        L = 2 * self.conf.rnnms.upsampling_t * Tc
        wav = torch.empty(B, L, device=z.device)
        mulaw = torch.empty(B, L, dtype=torch.int64, device=z.device) if return_mulaw else None
        seed = (torch.initial_seed() if seed is None else int(seed)) & 0xFFFFFFFFFFFFFFFF
        if utt_base is None:
            utt_base = self._utterances_done
            if utt_ids is None:
                self._utterances_done += B
        ids = None
        if utt_ids is not None:
            ids = (C.c_uint32 * B)(*[int(v) & 0xFFFFFFFF for v in utt_ids])
        nc = None
        if n_codes is not None:
            nc = (C.c_int * B)(*[int(v) for v in n_codes])
        with _lib.device_guard(z.device):
            _lib.check(_lib.load().vqcpc_vocoder_generate(
                h, z.data_ptr(), speaker.data_ptr(), B, Tc, nc, seed, int(utt_base) & 0xFFFFFFFF, ids, wav.data_ptr(),
                mulaw.data_ptr() if return_mulaw else None, int(max_steps), _lib.current_stream()))
        return (wav, mulaw) if return_mulaw else wav

    @torch.no_grad()
    def forward(self, x: Tensor, z: Tensor, speaker: Tensor) -> Tensor:
        """``network_vocoder.py:41-67``: teacher-forced energies (B, T_s, 2**bits)."""
        z, speaker = self._prep(z, speaker)
        x = x.detach().to(device=z.device, dtype=torch.int64).contiguous()
        B, Tc = z.shape
        if x.dim() != 2 or x.size(0) != B:
            raise RuntimeError(f"expected x (B, T_s), got {tuple(x.shape)}")
        Ts = x.size(1)
        if Ts > 2 * self.conf.rnnms.upsampling_t * Tc:
            raise RuntimeError("x is longer than the conditioning series covers")
        if x.min().item() < 0 or x.max().item() >= 2 ** self.conf.rnnms.bits_mu_law:
            raise IndexError("index out of range in self")
        logits = torch.empty(B, Ts, 2 ** self.conf.rnnms.bits_mu_law, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(_lib.load().vqcpc_vocoder_logits(self._native(), x.data_ptr(), z.data_ptr(), speaker.data_ptr(),
                                                        B, Tc, Ts, logits.data_ptr(), _lib.current_stream()))
        return logits

    @torch.no_grad()
    def glue(self, z: Tensor, speaker: Tensor) -> Tensor:
        """What ``network_vocoder.py:69-77`` hands to ``rnnms``: (B, 2T', dim_i_embedding + dim_speaker_embedding)."""
        z, speaker = self._prep(z, speaker)
        B, Tc = z.shape
        out = torch.empty(B, 2 * Tc, self.conf.dim_i_embedding + self.conf.dim_speaker_embedding, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(_lib.load().vqcpc_vocoder_glue(self._native(), z.data_ptr(), speaker.data_ptr(), B, Tc, out.data_ptr(),
                                                      _lib.current_stream()))
        return out

    @torch.no_grad()
    def condition(self, z: Tensor, speaker: Tensor) -> Tensor:
        """PreNet output (B, 2T', dim_voc_latent) -- stage-level checks."""
        z, speaker = self._prep(z, speaker)
        B, Tc = z.shape
        out = torch.empty(B, 2 * Tc, self.conf.rnnms.dim_voc_latent, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(_lib.load().vqcpc_vocoder_condition(self._native(), z.data_ptr(), speaker.data_ptr(), B, Tc,
                                                           out.data_ptr(), _lib.current_stream()))
        return out
