"""Deterministic, platform-independent tensor fill used by the golden fixtures.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/, by
oracle/make_golden.py and by __graft_entry__.smoke(); never by the product path.

The golden fixtures under tests/golden/ do not store model weights (they would
be tens of MB).  Instead both the generator (which runs the reference in the
build container) and the tests (which run anywhere) rebuild every parameter as

    scale[name] * unit_fill(shape, seed=crc32(name))

where ``unit_fill`` is a pure-integer splitmix64 hash of the flat element index
mapped to [-1, 1), so it is bit-identical on every machine / numpy version.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def unit_fill(shape, seed):
    """float64 array of `shape`, values in [-1, 1), deterministic in (index, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        s = _splitmix64(np.uint64(seed & 0xFFFFFFFF) + np.uint64(0x1234567))
        h = _splitmix64(idx ^ s)
    # top 53 bits -> [0,1) -> [-1,1)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (2.0 * u - 1.0).reshape(shape)


def name_seed(name):
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def fill_named(name, shape, scale, dtype=np.float32, complex_=False):
    """Parameter value for `name`: scale * unit_fill.  complex_ => independent re/im."""
    if complex_:
        re = unit_fill(shape, name_seed(name + "#re"))
        im = unit_fill(shape, name_seed(name + "#im"))
        out = (re + 1j * im) * scale
        return np.asarray(out, dtype=np.complex64 if dtype == np.float32 else np.complex128).reshape(shape)
    return np.asarray(unit_fill(shape, name_seed(name)) * scale, dtype=dtype).reshape(shape)
