"""MI355X-native VQ-CPC inference path (drop-in for the reference's ``model.py`` /
``network_vocoder.py`` method surface).  See DESIGN.md and include/vqcpc.h."""
from .model import ConfEncoder, Encoder, VQEmbeddingEMA          # noqa: F401
from .network_vocoder import (ConfPreNet, ConfRNNMSVocoder, ConfVocoder, ConfWaveAR,  # noqa: F401
                              RNNMSVocoder, Vocoder)
