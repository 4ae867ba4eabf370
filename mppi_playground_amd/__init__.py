"""mppi_playground_amd — MI355X-native MPPI.forward() hot path behind the pi_mpc.mppi.MPPI surface.

Layout
  csrc/      hand-written HIP (gfx950) kernels + the C ABI declared in include/mppi_hip.h
  _capi.py   ctypes binding of that ABI (fails loudly when the extension is missing)
  pi_mpc/    host-side mirror of the reference's `pi_mpc` package (MPPI class)
  envs/      the shipped model plugins (dynamics / cost callables carrying a native spec)

`from pi_mpc.mppi import MPPI` works after `import mppi_playground_amd` (which puts this directory
on sys.path exactly like the reference's `src/`), or use `from mppi_playground_amd.pi_mpc.mppi import MPPI`.
"""
import os as _os
import sys as _sys

__version__ = "0.1.0"

_HERE = _os.path.dirname(_os.path.abspath(__file__))
if _HERE not in _sys.path:
    _sys.path.append(_HERE)
