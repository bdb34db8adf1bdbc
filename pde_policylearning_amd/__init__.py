"""fnoengine: MI355X-native FNO spectral-convolution engine (hand-written HIP for gfx950
behind a C ABI) with the neuralop / pde-policylearning operator API on top."""
__version__ = "0.1.0"
