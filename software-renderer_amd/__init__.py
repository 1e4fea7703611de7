"""software-renderer_amd — MI355X-native triangle rasterizer behind the reference's draw call.

The product is the C-ABI shared library `lib/libswr_hip.so` (HIP kernels for gfx950 +
include/swr.h).  This Python package is plumbing only: a ctypes binding of that C-ABI for the
tests and bench.py, plus synthetic scene generators.  It never imports anything from oracle/
and has no CPU fallback: if the library is missing or there is no HIP device, calls raise.

Import with `importlib.import_module("software-renderer_amd")` (the directory name has a
hyphen) or through the `swr_amd` alias module at the repo root.
"""
from . import scenes  # noqa: F401
from .binding import (  # noqa: F401
    FLAG_DEPTH_TEST,
    FLAG_METAL_RULES,
    FLAG_NO_COLOR,
    FLAG_REAL_LINES,
    Context,
    HostImage,
    SwrError,
    band_rows,
    build,
    device_count,
    host_register,
    host_unregister,
    library_path,
    load_library,
    render,
    tile_shape,
)
