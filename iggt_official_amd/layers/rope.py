"""2-D RoPE host side: the cos/sin table consumed by the fused q/k-norm + RoPE HIP kernel.

Mirrors reference iggt/layers/rope.py:24-59 (PositionGetter) and :62-188
(RotaryPositionEmbedding2D).  The rotation itself runs inside `iggt_qknorm_rope_bf16`
(csrc/elementwise.hip); this module only owns the frequency table, built with the same fp32
arithmetic as rope.py:100-117, and never synchronises with the device (the reference's
`int(positions.max())` at rope.py:177 is replaced by the grid size).
"""
from typing import Dict, Tuple

import torch
import torch.nn as nn


class PositionGetter:
    """(y, x) integer grid positions, reference rope.py:24-59 (kept for API parity; the HIP kernel
    derives positions from the token index, so this is only used by callers that want them)."""

    def __init__(self):
        self.position_cache: Dict[Tuple[int, int], torch.Tensor] = {}

    def __call__(self, batch_size: int, height: int, width: int, device) -> torch.Tensor:
        if (height, width) not in self.position_cache:
            ys = torch.arange(height, device=device)
            xs = torch.arange(width, device=device)
            self.position_cache[height, width] = torch.cartesian_prod(ys, xs)
        return self.position_cache[height, width].view(1, height * width, 2).expand(batch_size, -1, -1).clone()


class RotaryPositionEmbedding2D(nn.Module):
    def __init__(self, frequency: float = 100.0, scaling_factor: float = 1.0):
        super().__init__()
        self.base_frequency = frequency
        self.scaling_factor = scaling_factor
        self._tables: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}

    def tables(self, head_dim: int, max_pos: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos, sin fp32 [max_pos + 1, head_dim // 4] (one column per distinct angle; the reference
        duplicates them to head_dim // 2, rope.py:112)."""
        half = head_dim // 2
        key = (half, max_pos, str(device))
        if key not in self._tables:
            # computed on the host with the reference's exact op sequence (rope.py:103-114)
            exponents = torch.arange(0, half, 2).float() / half
            inv_freq = 1.0 / (self.base_frequency ** exponents)
            positions = torch.arange(max_pos + 1, dtype=inv_freq.dtype)
            angles = torch.einsum("i,j->ij", positions, inv_freq)
            self._tables[key] = (angles.cos().contiguous().to(device), angles.sin().contiguous().to(device))
        return self._tables[key]
