from .basics import SpectralConv2d, SpectralConv3d  # noqa: F401
from .pinobserver import MultiplicativeNet, PINObserver2d, PINObserverFullField, PlanePredHead  # noqa: F401
