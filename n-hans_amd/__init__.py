"""nhans_amd -- MI355X-native N-HANS per-frame inference hot path (STFT -> conditioned residual
mask network -> overlap-add iSTFT) behind the reference's apply_* / nhans_* API surface."""
from . import spec  # noqa: F401

__all__ = ["spec"]
