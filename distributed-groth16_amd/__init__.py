"""distributed-groth16_amd -- MI355X (gfx950) Groth16 hot path behind the dist-primitives surface of
zkHubHQ/distributed-groth16.  The compute lives in csrc/ (hand-written HIP, C ABI in
include/dg16.h -> libdg16.so); this package is the thin Python host layer used by tests, bench.py
and the multi-GPU driver.  There is no CPU path: importing `lib` without libdg16.so, or creating a
context without a GPU, raises.

The directory name is not an importable identifier; load it through the `dg16_amd` shim at the repo
root (`import dg16_amd`).
"""

from .lib import (  # noqa: F401
    Context,
    Dg16Error,
    CURVES,
    lib_path,
    F_SCALARS_MONT,
    F_DEVICE_PTRS,
    F_OUT_AFFINE,
)
