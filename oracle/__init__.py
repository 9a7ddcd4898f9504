"""CPU oracle for the VQ-CPC inference hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``vectorquantizedcpc_amd``) never does.

* Encoder / VQ half: bit-exact restatement of ``/root/reference/model.py`` on its
  PyTorch-CPU path, pinned by ``tests/golden/`` (made from the reference itself by
  ``tools/gen_golden.py``).
* Vocoder half: the project's own spec of the absent third-party ``rnnms`` package
  -- **parity unpinned** (see ``vqcpc_oracle.c`` header and DESIGN.md).
"""
from .ref import (  # noqa: F401
    build, lib, encoder_encode, vq_encode, vq_forward_stats, conv1d_k4s2, layernorm,
    linear, lstm, sumsq64, philox4x32_10, sample_noise, noise_from_word, mulaw_decode,
    vocoder_generate, vocoder_condition, sample_from_logits,
)
