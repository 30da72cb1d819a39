"""MI355X-native wavefront path tracer behind a C ABI (librtx_hip.so).

`raytracer_amd.capi` binds the C ABI declared in include/rtx_capi.h with ctypes;
`raytracer_amd.scenes` holds the synthetic scene builders used by the tests and
the benchmark.  The library is HIP-only: importing `capi` works anywhere, but
creating a `Renderer` without librtx_hip.so or without a gfx950 GPU raises.
"""
from .capi import Config, Renderer, RtxError, load_library  # noqa: F401
from . import scenes  # noqa: F401
