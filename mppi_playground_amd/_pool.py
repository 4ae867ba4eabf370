"""Output tensors of per-tick calls: rows of one allocation, handed out one by one (host-side plumbing, no arithmetic).

The reference's example loops at their own sizes (racing: T = 25, N = 4000; example/racing.py:221-266) are bound by the HOST
once every call of the tick is one launch: seven `torch.empty` per tick at ~1.3 us each (scripts/example_tick_host.py,
profiles/r05_experiments.md).  A RowPool allocates a block of rows at once and splits it with ONE `unbind` (~0.2 us per row).
Every row is handed out exactly once — a fresh tensor like torch.empty's: nobody else holds it, contents undefined,
contiguous, 256-byte aligned — and a block goes back to torch's caching allocator when the last of its rows is dropped.
The visible difference to torch.empty: a row is a view of its block (`torch.save` of one writes the block; `.clone()` first).
Under stream capture (torch.cuda.graph: the generic path's `graph_callables`, or a caller's own graph) `take` is a plain
torch.empty from the graph's private pool, like the code it replaces.
What a retained row pins: its block — at most 256 rows / 1 MiB for small rows, and at most 16 rows once a row is
larger than 4 KiB (a held `state_seq` of a long horizon no longer keeps a megabyte alive).  `MPPI_ROW_POOL=0` in the environment
switches the pooling off altogether: every `take` is a plain torch.empty, exactly the reference's behaviour (a logging caller
that keeps or saves every returned tensor)."""
from __future__ import annotations

import math
import os

import torch

_ALIGN = 256  # bytes between rows (vector stores of the kernels need 16)
_LARGE_ROW_BYTES = 4096   # rows beyond this size come in blocks of at most ...
_LARGE_ROW_MAX = 16       # ... this many
_ENABLED = os.environ.get("MPPI_ROW_POOL", "1") not in ("0", "false", "off")
_capturing = torch._C._cuda_isCurrentStreamCapturing if hasattr(torch._C, "_cuda_isCurrentStreamCapturing") else (lambda: False)


def capturing() -> bool:
    """Is torch's current stream of the current GPU being captured into a graph?  (Callers on a GPU path only.)"""
    return _capturing()


class RowPool:
    __slots__ = ("_shape", "_device", "_dtype", "_pitch", "_per_block", "_strides", "_block", "_on_gpu")

    def __init__(self, shape, device, dtype, block_bytes: int = 1 << 20, max_rows: int = 256):
        self._shape, self._device, self._dtype = tuple(int(v) for v in shape), device, dtype
        self._on_gpu = torch.device(device).type == "cuda"
        item = torch.empty((), dtype=dtype).element_size()
        numel = math.prod(self._shape)
        self._pitch = max(1, -(-(numel * item) // _ALIGN)) * _ALIGN // item  # elements from row to row
        self._per_block = max(1, min(max_rows, block_bytes // (self._pitch * item)))
        if self._pitch * item > _LARGE_ROW_BYTES:
            self._per_block = min(self._per_block, _LARGE_ROW_MAX)
        if not _ENABLED:
            self._per_block = 0  # take() -> torch.empty
        strides, acc = [], 1
        for s in reversed(self._shape):
            strides.append(acc)
            acc *= s
        self._strides = tuple(reversed(strides))
        self._block = (None, [])  # (raw stream the rows were allocated under, rows not handed out yet): replaced as ONE object

    def take(self, stream: int, captured: bool = None) -> torch.Tensor:
        """A fresh row for work on raw stream `stream` of the pool's device (the block was allocated under torch's current
        stream, like a torch.empty at this point would be: a change of stream starts a new block).  Safe to call from several
        threads: `list.pop` is atomic, and a thread never pops from a block of another stream."""
        # (a row of ordinary memory must not be baked into a graph: it is freed when its block's rows are dropped.  A caller that
        # takes several rows passes `captured=capturing()` so that the query — 0.29 us — is made once per call)
        if not self._per_block or ((self._on_gpu and _capturing()) if captured is None else captured):
            return torch.empty(self._shape, device=self._device, dtype=self._dtype)
        held, rows = self._block
        if held == stream:
            try:
                return rows.pop()
            except IndexError:
                pass
        n = self._per_block
        block = torch.empty(n * self._pitch, device=self._device, dtype=self._dtype)
        rows = list(block.as_strided((n, *self._shape), (self._pitch, *self._strides)).unbind(0))
        row = rows.pop()
        self._block = (stream, rows)
        return row

    @property
    def rows_left(self) -> int:
        return len(self._block[1])
