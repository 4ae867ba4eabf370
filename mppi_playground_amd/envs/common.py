"""Shared model math in torch (used by env.step with batch 1 and by any torch caller of the plugins)."""
import torch


def angle_normalize(x: torch.Tensor) -> torch.Tensor:
    """Wrap to [-pi, pi): ((x + pi) mod 2 pi) - pi with Python-style modulo."""
    return torch.remainder(x + torch.pi, 2 * torch.pi) - torch.pi
