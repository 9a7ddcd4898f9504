"""PyTorch-CPU restatement of the path with library ops (TEST INFRASTRUCTURE, like the rest
of oracle/): what the reference executes on its CPU path, used (a) as a second, independent
check of the C oracle's vocoder spec and (b) as the timed ``cpu_baseline`` ("port") of
``bench.py`` -- MKL/oneDNN multi-threaded, batched over utterances.

Encoder: the reference's own operator sequence (``model.py:59-70``: conv1d, LayerNorm, ReLU,
Linear, addmm + argmin + embedding, LSTM).  Vocoder: nn.GRU prenet + GRUCell loop + two
Linear + the project's Philox-keyed Gumbel-max draw (``rnnms`` is absent: parity unpinned).
"""
import numpy as np
import torch
import torch.nn.functional as F

from vectorquantizedcpc_amd import synth


@torch.no_grad()
def encoder_encode(sd, mel, want_c=True):
    z = F.conv1d(mel, sd["conv.weight"], None, 2, 1).transpose(1, 2)
    z = F.relu(F.layer_norm(z, (z.shape[-1],), sd["encoder.0.weight"], sd["encoder.0.bias"]))
    for lin, ln in ((2, 3), (5, 6), (8, 9), (11, 12)):
        z = F.linear(z, sd[f"encoder.{lin}.weight"])
        z = F.relu(F.layer_norm(z, (z.shape[-1],), sd[f"encoder.{ln}.weight"], sd[f"encoder.{ln}.bias"]))
    z = F.linear(z, sd["encoder.14.weight"], sd["encoder.14.bias"])
    E = sd["codebook.embedding"]
    xf = z.reshape(-1, E.shape[1])
    d = torch.addmm(torch.sum(E ** 2, dim=1) + torch.sum(xf ** 2, dim=1, keepdim=True), xf, E.t(), alpha=-2.0, beta=1.0)
    idx = torch.argmin(d, dim=-1)
    q = F.embedding(idx, E).view_as(z)
    c = None
    if want_c:
        H = sd["rnn.weight_hh_l0"].shape[1]
        rnn = torch.nn.LSTM(E.shape[1], H, batch_first=True)
        rnn.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("rnn.")})
        c, _ = rnn(q)
    return q, c, idx.view(z.shape[0], z.shape[1]), z


def noise_from_words(w):
    """uint32 Philox words -> Gumbel noise, the protocol's conversion: 23 bits + 1/2 is exact in fp32 and
    strictly inside (0, 1) (a 24-bit form rounds to 1.0 at the top word: +inf noise)."""
    uni = torch.from_numpy(((w >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0))
    return -torch.log(-torch.log(uni))


def make_noise(B, n_steps, seed, utt_base=0, chunk=256):
    """Gumbel noise (B, n_steps, 256) of the sampling protocol:
    Philox(counter=(t, utt, k>>2, 0), key=seed)[k&3] -> ((w>>9)+0.5)*2^-23 -> -log(-log(u))."""
    out = torch.empty(B, n_steps, 256)
    for t0 in range(0, n_steps, chunk):
        n = min(chunk, n_steps - t0)
        ctr = np.zeros((B * n * 64, 4), np.uint32)
        ctr[:, 0] = np.tile(np.repeat(np.arange(t0, t0 + n, dtype=np.uint32), 64), B)
        ctr[:, 1] = np.repeat(np.arange(utt_base, utt_base + B, dtype=np.uint32), n * 64)
        ctr[:, 2] = np.tile(np.arange(64, dtype=np.uint32), B * n)
        w = synth.philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(B, n, 256)
        out[:, t0:t0 + n] = noise_from_words(w)
    return out


class TorchVocoder:
    def __init__(self, sd, upsample=160):
        self.sd = sd
        self.up = upsample
        hp = sd["rnnms.prenet.weight_hh_l0"].shape[1]
        self.prenet = torch.nn.GRU(sd["rnnms.prenet.weight_ih_l0"].shape[1], hp, num_layers=2, batch_first=True,
                                   bidirectional=True)
        self.prenet.load_state_dict({k[len("rnnms.prenet."):]: v for k, v in sd.items() if k.startswith("rnnms.prenet.")})
        hr = sd["rnnms.ar.rnn.weight_hh_l0"].shape[1]
        self.cell = torch.nn.GRUCell(sd["rnnms.ar.rnn.weight_ih_l0"].shape[1], hr)
        self.cell.load_state_dict({"weight_ih": sd["rnnms.ar.rnn.weight_ih_l0"], "weight_hh": sd["rnnms.ar.rnn.weight_hh_l0"],
                                   "bias_ih": sd["rnnms.ar.rnn.bias_ih_l0"], "bias_hh": sd["rnnms.ar.rnn.bias_hh_l0"]})
        self.hr = hr

    @torch.no_grad()
    def condition(self, z, spk):
        ze = F.embedding(z, self.sd["code_embedding.weight"])
        zu = F.interpolate(ze.transpose(1, 2), scale_factor=2).transpose(1, 2)        # network_vocoder.py:74
        se = F.embedding(spk, self.sd["speaker_embedding.weight"]).unsqueeze(1).expand(-1, zu.size(1), -1)
        cond, _ = self.prenet(torch.cat((zu, se), dim=-1))
        return cond

    @torch.no_grad()
    def generate(self, z, spk, seed, utt_base=0, n_steps=None, inputs=None, want_logits=False, noise=None, cond=None):
        sd = self.sd
        B = z.shape[0]
        if cond is None:                                   # bench.py's timed chunks pass the prenet output in
            cond = self.condition(z, spk)
        total = self.up * cond.shape[1]
        n_steps = total if n_steps is None else min(n_steps, total)
        if noise is None:
            noise = make_noise(B, n_steps, seed, utt_base)
        h = torch.zeros(B, self.hr)
        x = torch.full((B,), 128, dtype=torch.long)
        samples = torch.empty(B, n_steps, dtype=torch.long)
        logits_all = torch.empty(B, n_steps, 256) if want_logits else None
        emb = sd["rnnms.ar.embedding.weight"]
        for t in range(n_steps):
            if inputs is not None:
                x = inputs[:, t]
            inp = torch.cat((emb[x], cond[:, t // self.up]), dim=1)
            h = self.cell(inp, h)
            o = F.linear(F.relu(F.linear(h, sd["rnnms.ar.fc1.weight"], sd["rnnms.ar.fc1.bias"])),
                         sd["rnnms.ar.fc2.weight"], sd["rnnms.ar.fc2.bias"])
            x = torch.argmax(o + noise[:, t], dim=1)       # exponential race, as ATen's Categorical.sample
            samples[:, t] = x
            if want_logits:
                logits_all[:, t] = o
        y = 2.0 * samples.double() / 255.0 - 1.0
        wav = (torch.sign(y) / 255.0 * ((256.0) ** y.abs() - 1.0)).float()            # preprocess.py:30-35
        return samples, wav, logits_all
