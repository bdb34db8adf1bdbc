"""The names `neuralop.models` exports in the reference (neuralop/models/__init__.py:1-7).  FNO / FNO2d / FNO3d, RNO2d and
SpectralRegressor run on the HIP engine; FNO1d, TFNO*, SFNO and UNO keep their names and signatures and raise when
constructed (outside the accelerated path)."""
from .tfno import TFNO, TFNO1d, TFNO2d, TFNO3d  # noqa: F401
from .tfno import FNO, FNO1d, FNO2d, FNO3d, Lifting, Projection  # noqa: F401
from .tfno import SFNO  # noqa: F401
from .uno import UNO  # noqa: F401
from .rno import RNO2d  # noqa: F401
from .spectral_regressor import SpectralRegressor  # noqa: F401
from .model_dispatcher import get_model, available_models  # noqa: F401
from .spectral_convolution import SpectralConv, FactorizedSpectralConv  # noqa: F401
from .fno_block import FNOBlocks  # noqa: F401
