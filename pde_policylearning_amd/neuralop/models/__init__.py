from .spectral_convolution import SpectralConv, FactorizedSpectralConv  # noqa: F401
from .fno_block import FNOBlocks  # noqa: F401
from .tfno import FNO, FNO2d, FNO3d, Lifting, Projection  # noqa: F401
