from pi_mpc.mppi import MPPI  # noqa: F401  (same export as the reference's src/pi_mpc/__init__.py:1-4)

__version__ = "0.1.0"
__all__ = ["MPPI"]
