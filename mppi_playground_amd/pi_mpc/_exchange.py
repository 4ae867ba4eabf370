"""Set-up of a sharded solver's per-solve exchange (pi_mpc.mppi.MPPI mix-in): torch.distributed
all_gather, the library's own RCCL communicator (mppi_comm_*) or its peer-to-peer buffers (mppi_p2p_*), each with its start-up
self-test."""
from __future__ import annotations

import ctypes as C

import torch

from mppi_playground_amd import _capi
from pi_mpc._lazy import _ptr


class ExchangeMixin:
    def _setup_exchange(self) -> None:
        """Pick the per-solve exchange of a sharded solver (environment variable MPPI_EXCHANGE):
          "nccl" (default)  one all_gather per solve through torch.distributed (RCCL on ProcessGroupNCCL's stream);
          "rccl"            the library's own communicator (mppi_comm_*): ncclAllGather issued by mppi_weights_reduce on
                            the solve's stream — no process-group stream, no events, the sharded solve is ONE library
                            call like the unsharded one;
          "p2p"             the library's peer-to-peer buffers (mppi_p2p_*: xGMI stores + polling, no collective launch);
          "auto"            "rccl" when it passes its start-up self-test on every rank, else "nccl".
        "rccl" / "p2p" raise when their set-up or self-test fails; every decision is agreed on by all ranks (all_reduce),
        so the ranks always take the same path."""
        import os

        import torch.distributed as dist

        mode = os.environ.get("MPPI_EXCHANGE", "nccl").lower()
        if mode not in ("nccl", "rccl", "p2p", "auto"):
            raise ValueError("MPPI_EXCHANGE must be nccl, rccl, p2p or auto")
        if mode == "nccl":
            return
        W, r, length = self._world, self._rank, int(self._summary.numel())

        def all_ok(ok: bool) -> bool:  # agreement point: every rank takes the same branch afterwards
            if W == 1:
                return ok
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self._pg)
            return bool(int(t.item()))

        def pattern_ok(entry: str) -> bool:  # rank w sends 1000*w + j + round; everybody must see everybody's
            base = torch.arange(length, device=self._device, dtype=torch.float32)
            got = torch.empty(W, length, device=self._device, dtype=torch.float32)
            ok = True
            for rnd in range(3):
                self._h.call(entry, _ptr(base + (1000.0 * r + rnd)), _ptr(got), self._stream())
                want = base[None, :] + (1000.0 * torch.arange(W, device=self._device)[:, None] + rnd)
                ok = ok and bool(torch.equal(got, want))
            return ok

        why = ""
        if mode in ("rccl", "auto"):
            ident = torch.zeros(128, dtype=torch.uint8, device=self._device)
            ok = True
            if r == 0:
                buf = (C.c_ubyte * 128)()
                ok = self._h.lib.mppi_comm_unique_id(buf) == 0
                ident = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=self._device)
                why = "" if ok else "librccl.so.1 is not loadable"
            if all_ok(ok):
                if W > 1:
                    dist.broadcast(ident, src=dist.get_global_rank(self._pg, 0) if self._pg is not None else 0,
                                   group=self._pg)
                blob = (C.c_ubyte * 128)(*ident.cpu().tolist())
                try:
                    self._h.call("mppi_comm_init", W, r, blob)
                    ok = pattern_ok("mppi_comm_exchange")
                except _capi.MppiError as e:
                    ok, why = False, str(e)
                if all_ok(ok):
                    self._h.call("mppi_set_option", b"exchange_comm", 1)
                    self._comm = True
                    return
            if mode == "rccl":
                raise _capi.MppiError("MPPI_EXCHANGE=rccl: the in-library collective is not usable here: "
                                      + (why or "self-test mismatch"))
            return  # auto: fall back to the torch.distributed all_gather, on every rank

        handle = (C.c_ubyte * 64)()
        ok = True
        try:
            self._h.call("mppi_p2p_alloc", W, r, handle)
        except _capi.MppiError as e:
            ok, why = False, str(e)
        if all_ok(ok):
            # 64 handle bytes + this rank's device ordinal
            mine = torch.tensor(list(bytes(handle)) + [int(self._device.index)], dtype=torch.uint8, device=self._device)
            allh = torch.empty(W * 65, dtype=torch.uint8, device=self._device)
            dist.all_gather_into_tensor(allh, mine, group=self._pg)
            rows = allh.cpu().numpy().reshape(W, 65)
            blob = (C.c_ubyte * (64 * W)).from_buffer_copy(rows[:, :64].tobytes())
            devs = (C.c_int32 * W)(*[int(v) for v in rows[:, 64]])
            try:
                self._h.call("mppi_p2p_connect", blob, devs)
            except _capi.MppiError as e:
                ok, why = False, str(e)
            if all_ok(ok):
                try:
                    ok = pattern_ok("mppi_p2p_exchange")
                except _capi.MppiError as e:
                    ok, why = False, str(e)
                if all_ok(ok):
                    self._h.call("mppi_set_option", b"exchange_p2p", 1)
                    self._p2p = True
                    return
        raise _capi.MppiError("MPPI_EXCHANGE=p2p: the peer-to-peer exchange is not usable here: " + (why or "self-test mismatch"))
