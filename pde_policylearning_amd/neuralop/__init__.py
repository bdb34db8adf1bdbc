"""Host-side mirror of the reference's `neuralop` operator API for the hot path
(reference neuralop/__init__.py:1-9): same class names, constructor arguments,
parameter names and forward semantics; the arithmetic runs in the HIP engine."""
from .models import FNO, FNO2d, FNO3d, SpectralConv, FactorizedSpectralConv  # noqa: F401
