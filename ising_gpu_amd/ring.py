"""Slab ring: one slab per rank, periodic 1-row halo exchange per colour half-sweep, overlapped with the
interior update.

This replaces the reference's multi-GPU mechanism (one managed allocation, peer loads of the two rows outside
each slab and a cudaDeviceSynchronize on every device after each colour -- optimized/main.cu:1599-1658,
loadTile :413-428, barriers :1779-1784/:1800-1805) with the MI355X-native form: one process per GPU, explicit
RCCL send/recv of one colour row (X/4 bytes) to each ring neighbour over xGMI, issued right after the two
edge rows of a colour are updated and hidden behind the interior rows' kernel:

    for colour in (black, white):                      # it = sweep index + 1, as the reference
        wait for the halo rows of the OTHER colour      (posted one half-sweep ago)
        update edge rows 0 and Y-1 of `colour`          (one tiny launch; they read the other colour's halo rows)
        post send/recv of `colour` rows 0 and Y-1       (torch.distributed P2P = RCCL; runs on its own stream)
        update interior rows 1..Y-2 of `colour`         (overlaps with the exchange)

Results do not depend on the decomposition: the Philox stream id uses the global row (optimized/main.cu:514).

A backend that keeps ghost rows G deep (a library-owned slab on the ballot layout: ising_ghost_ptrs) is driven the way the
library's own ring drives it instead: G rows of both colours travel every G/2 sweeps, one fused launch runs in between
(SlabRing._sweep_deep; csrc/ising_ring.cpp: sweep_deep).

The orchestration only needs a *slab backend* with the small interface below, so the same code runs on GPUs
(IsingSlab + NCCL) and, in tests, on CPU tensors with gloo.
"""
from __future__ import annotations

from typing import List, Optional, Protocol

import torch
import torch.distributed as dist

from ._lib import BLACK, WHITE, HAM_BLACK, LAYOUT_AUTO, IsingError


class SlabBackend(Protocol):
    def init(self) -> None: ...
    def update_all(self, it: int, color: int) -> None: ...
    def update_edges(self, it: int, color: int) -> None: ...      # rows 0 and Y-1
    def update_interior(self, it: int, color: int) -> None: ...   # rows 1 .. Y-2
    def halo_tensors(self, color: int):
        """-> (send_top, send_bot, recv_top, recv_bot): 1-D uint8 tensors of one colour row each."""
    def count_up_down(self): ...
    def bond_equal(self) -> int: ...
    # optional -- the deep exchange (ising_ghost_ptrs): a backend that keeps G > 1 ghost rows on either side
    #   ghost_depth() -> G;  ghost_tensors(color) -> (send_top, send_bot, recv_top, recv_bot), G rows each;
    #   ghost_delivered(color);  sweep_ghost(first_it, nsweeps <= G/2): the sweeps as one launch, ghost rows included
    # optional -- ring_of_one: True when a lone slab's rows -1 / Y are halo rows (it sends its edge rows to itself)


class _DevMem:
    """Zero-copy view of device memory owned by libising_hip.so as a torch tensor (__cuda_array_interface__)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class HipSlabBackend:
    """IsingSlab (libising_hip.so) as a ring backend; kernels run on torch's current stream.

    `HipSlabBackend.create(...)` lets torch own the slab's device buffers (the C-ABI takes them as plain pointers),
    so the edge and halo rows RCCL touches are slices of ordinary torch tensors.  Wrapping an existing IsingSlab
    instead exposes the library-owned rows zero-copy through __cuda_array_interface__."""

    def __init__(self, slab, buffers=None):
        self.slab = slab
        self.device = torch.device("cuda", slab.cfg.device)
        slab.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._buffers = buffers or {}
        self._halo = {}   # (colour, pointers, bytes) -> tensors: a library-owned slab moves its rows when it changes layout
        self.use_J = bool(getattr(slab, "use_J", False))
        self.ring_of_one = slab.nslabs == 1 and bool(slab.cfg.ring_halo)

    def _tensors(self, color, ptrs, nb):
        key = (color, tuple(ptrs), nb)
        if key not in self._halo:
            buf = self._buffers.get("coupling" if color == HAM_BLACK else "lattice")
            if buf is not None:
                base = buf.data_ptr()
                self._halo[key] = tuple(buf[p - base:p - base + nb] for p in ptrs)
            else:
                self._halo[key] = tuple(torch.as_tensor(_DevMem(p, nb), device=self.device) for p in ptrs)
        return self._halo[key]

    @classmethod
    def create(cls, X, Y, device=0, J_prob=None, layout=LAYOUT_AUTO, **kw):
        from .lattice import IsingSlab, required_bytes
        dev = torch.device("cuda", device)
        bufs = {"lattice": torch.empty(required_bytes(X, Y, layout), dtype=torch.uint8, device=dev)}
        if J_prob is not None:
            bufs["coupling"] = torch.empty(required_bytes(X, Y), dtype=torch.uint8, device=dev)
        slab = IsingSlab(X, Y, device=device, J_prob=J_prob, layout=layout, lattice_mem=bufs["lattice"].data_ptr(),
                         lattice_mem_bytes=bufs["lattice"].numel(),
                         coupling_mem=bufs["coupling"].data_ptr() if "coupling" in bufs else 0,
                         coupling_mem_bytes=bufs["coupling"].numel() if "coupling" in bufs else 0, **kw)
        return cls(slab, bufs)

    def layout_id(self):
        return self.slab.current_layout()

    def init(self):
        self.slab.init()

    def init_couplings_black(self):
        self.slab.init_couplings_black()

    def init_couplings_white(self):
        self.slab.init_couplings_white()

    def update_all(self, it, color):
        self.slab.update_color(it, color)

    def update_edges(self, it, color):
        self.slab.update_edges(it, color)

    def update_interior(self, it, color):
        self.slab.update_color(it, color, 1, self.slab.Y - 1)

    def halo_tensors(self, color):
        ptrs, nb = self.slab.halo_ptrs(color)
        return self._tensors(color, ptrs, nb)

    def ghost_depth(self):
        return self.slab.ghost_ptrs(BLACK)[0]

    def ghost_tensors(self, color):
        _, ptrs, nb = self.slab.ghost_ptrs(color)
        return self._tensors(color, ptrs, nb)

    def ghost_delivered(self, color):
        self.slab.ghost_delivered(color)

    def sweep_ghost(self, first_it, nsweeps):
        self.slab.it = first_it - 1
        self.slab.sweep_ghost(nsweeps)

    def count_up_down(self):
        return self.slab.count()

    def bond_equal(self):
        return self.slab.bond_equal()


class SlabRing:
    """Drives one slab of a ring of `world` slabs (world = torch.distributed world size)."""

    def __init__(self, backend: SlabBackend, group: Optional[dist.ProcessGroup] = None, exchange: Optional[str] = None):
        """exchange: "p2p" (default; one send/recv pair per neighbour = one xGMI link each) or "allgather" (every rank
        contributes its two edge rows to one all-gather and picks its neighbours' rows out of the result: 2*world rows
        instead of 2, still a few KiB -- the same schedule on the most travelled collective, kept as a fallback).
        The environment variable ISING_RING_EXCHANGE overrides the default."""
        import os
        self.b = backend
        self.group = group
        self.exchange = exchange or os.environ.get("ISING_RING_EXCHANGE", "p2p")
        if self.exchange not in ("p2p", "allgather"):
            raise ValueError(f"unknown ring exchange {self.exchange!r}")
        self._gather = {}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.prev = (self.rank - 1) % self.world
        self.next = (self.rank + 1) % self.world
        self.it = 0
        self._pending: List[Optional[list]] = [None, None, None]  # per plane (black, white, black couplings)
        # a lone slab whose rows -1 / Y are halo rows is a ring of one: it exchanges with itself
        self.ringed = self.world > 1 or bool(getattr(backend, "ring_of_one", False))
        # Deep exchange (backends with ghost rows G deep, p2p only): G rows of both colours every G/2 sweeps and ONE launch
        # in between, the schedule of the library's own ring (csrc/ising_ring.cpp: sweep_deep).  _deep_posted: the ghost
        # rows hold (or are about to receive) the neighbours' current rows.
        self._pending_deep: List[Optional[list]] = [None, None]
        self._deep_posted = False

    # -- halo exchange -----------------------------------------------------------------------------------
    def _post_allgather(self, color: int):
        send_top, send_bot, recv_top, recv_bot = self.b.halo_tensors(color)
        nb = send_top.numel()
        if color not in self._gather:  # [world][2][row] staging, allocated once per plane on the rows' device
            self._gather[color] = (torch.empty(2 * nb, dtype=send_top.dtype, device=send_top.device),
                                   torch.empty(self.world * 2 * nb, dtype=send_top.dtype, device=send_top.device))
        mine, everyone = self._gather[color]
        mine[:nb].copy_(send_top)
        mine[nb:].copy_(send_bot)
        work = dist.all_gather_into_tensor(everyone, mine, group=self.group, async_op=True)
        self._pending[color] = [work, (everyone, nb, recv_top, recv_bot)]

    def _post(self, color: int):
        if self.exchange == "allgather":
            return self._post_allgather(color)
        send_top, send_bot, recv_top, recv_bot = self.b.halo_tensors(color)
        # Order matters when prev == next (world == 2): the peer's first receive (its recv_top, "from prev")
        # must match our LAST row, so the bottom row is sent first.
        ops = [
            dist.P2POp(dist.isend, send_bot, self.next, self.group),
            dist.P2POp(dist.isend, send_top, self.prev, self.group),
            dist.P2POp(dist.irecv, recv_top, self.prev, self.group),
            dist.P2POp(dist.irecv, recv_bot, self.next, self.group),
        ]
        self._pending[color] = dist.batch_isend_irecv(ops)

    def _depth(self) -> int:
        g = getattr(self.b, "ghost_depth", None)
        return int(g()) if (g is not None and self.ringed and self.exchange == "p2p") else 1

    @property
    def ghost_rows(self) -> int:
        """Rows per colour and neighbour of one exchange: G > 1 = the deep schedule (one exchange per G/2 sweeps), 1 = one row
        per colour half-sweep."""
        return self._depth()

    def _post_deep(self):
        for color in (BLACK, WHITE):
            send_top, send_bot, recv_top, recv_bot = self.b.ghost_tensors(color)
            ops = [  # (same order as _post: with two ranks the peer's first receive must match our LAST rows)
                dist.P2POp(dist.isend, send_bot, self.next, self.group),
                dist.P2POp(dist.isend, send_top, self.prev, self.group),
                dist.P2POp(dist.irecv, recv_top, self.prev, self.group),
                dist.P2POp(dist.irecv, recv_bot, self.next, self.group),
            ]
            self._pending_deep[color] = dist.batch_isend_irecv(ops)
        self._deep_posted = True

    def _wait_deep(self):
        for color in (BLACK, WHITE):
            works = self._pending_deep[color]
            if works:
                for w in works:
                    w.wait()
                self.b.ghost_delivered(color)
            self._pending_deep[color] = None

    def _sweep_deep(self, nsweeps: int, G: int):
        self._wait(BLACK)  # (one-row exchanges of an earlier phase: consumed before anything else touches those rows)
        self._wait(WHITE)
        left = nsweeps
        while left > 0:
            ns = min(left, G // 2)
            if not self._deep_posted:
                self._post_deep()
            self._wait_deep()
            self.b.sweep_ghost(self.it + 1, ns)
            self.it += ns
            left -= ns
            self._post_deep()  # rows -1 / Y current again: observables and the next call find them in place

    def _wait(self, color: int):
        """The rows of `color` posted last -- one row or the deep exchange -- are in place for what the current stream does next."""
        if color in (BLACK, WHITE) and self._pending_deep[color]:
            self._wait_deep()
        self._wait_rows(color)

    def _wait_rows(self, color: int):
        works = self._pending[color]
        if works:
            if self.exchange == "allgather":
                work, (everyone, nb, recv_top, recv_bot) = works
                work.wait()
                rows = everyone.view(self.world, 2, nb)
                recv_top.copy_(rows[self.prev, 1])  # previous slab's last row
                recv_bot.copy_(rows[self.next, 0])  # next slab's first row
            else:
                for w in works:
                    w.wait()  # NCCL: the current stream waits; gloo: host blocks
        self._pending[color] = None

    # -- driver steps ------------------------------------------------------------------------------------
    def _check_layouts(self):
        """Ballot and dense rows have the same size but another bit order: every rank must hold the same layout."""
        lay = getattr(self.b, "layout_id", None)
        if lay is None or self.world == 1:
            return
        t = torch.tensor([lay()], dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if int(lo[0]) != int(hi[0]):
            raise IsingError(f"ring slabs differ in device layout (this rank: {int(t[0])}, ring: {int(lo[0])}..{int(hi[0])})")

    def init(self):
        self._check_layouts()
        self.quiesce()
        self.b.init()
        self.it = 0
        self._deep_posted = False
        if self.ringed:
            if self._depth() > 1:
                self._post_deep()
            else:
                self._post(BLACK)
                self._post(WHITE)
        if getattr(self.b, "use_J", False):
            # -J: the white couplings are assembled from the black ones, including the neighbours' edge rows
            self.b.init_couplings_black()
            if self.ringed:
                self._post(HAM_BLACK)
                self._wait(HAM_BLACK)
            self.b.init_couplings_white()
        return self

    def _half_sweep(self, it: int, color: int):
        if not self.ringed:
            self.b.update_all(it, color)
            return
        self._wait(1 - color)
        self.b.update_edges(it, color)
        self._post(color)
        self.b.update_interior(it, color)

    def sweep(self, nsweeps: int = 1):
        G = self._depth()
        if G > 1 and nsweeps > 0:
            self._sweep_deep(nsweeps, G)
            return self
        if self._deep_posted and self.ringed and nsweeps > 0:
            # the backend stopped sweeping through ghost rows (e.g. a temperature without integer thresholds): back to one
            # row per colour half-sweep, starting from rows delivered the one-row way
            self._wait_deep()
            self._deep_posted = False
            self._post(BLACK)
            self._post(WHITE)
        for _ in range(nsweeps):
            self.it += 1
            self._half_sweep(self.it, BLACK)
            self._half_sweep(self.it, WHITE)
        return self

    def quiesce(self):
        """Make sure every posted exchange has been consumed by the current stream / host."""
        self._wait(BLACK)
        self._wait(WHITE)
        self._wait_deep()

    def count(self):
        """Global (up, down) over all slabs (the host-side sum of countSpins, optimized/main.cu:860-866)."""
        up, down = self.b.count_up_down()
        if self.world > 1:
            t = torch.tensor([up, down], dtype=torch.int64)
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, group=self.group)
            up, down = int(t[0]), int(t[1])
        return up, down

    def bond_equal(self) -> int:
        if self.ringed:
            self._wait(WHITE)  # black sites read the white halo rows
            self._pending[WHITE] = None
        a = self.b.bond_equal()
        if self.world > 1:
            t = torch.tensor([a], dtype=torch.int64)
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, group=self.group)
            a = int(t[0])
        return a


class LocalRing:
    """All slabs of a ring in ONE process (slabs may share a device): halo rows move by device-to-device
    copies instead of RCCL.  Same boundary/interior launch order as SlabRing; used by the CLI-style
    single-process driver and by the single-GPU tests of the nslabs > 1 kernel path."""

    def __init__(self, backends):
        self.b = list(backends)
        self.n = len(self.b)
        self.it = 0

    def _exchange(self, color: int):
        h = [b.halo_tensors(color) for b in self.b]
        for k in range(self.n):
            send_top, send_bot, _, _ = h[k]
            h[(k + 1) % self.n][2].copy_(send_bot, non_blocking=True)  # next slab's recv_top <- my last row
            h[(k - 1) % self.n][3].copy_(send_top, non_blocking=True)  # prev slab's recv_bot <- my first row

    def init(self):
        for b in self.b:
            b.init()
        self.it = 0
        if self.n > 1:
            self._exchange(BLACK)
            self._exchange(WHITE)
        if getattr(self.b[0], "use_J", False):
            for b in self.b:
                b.init_couplings_black()
            if self.n > 1:
                self._exchange(HAM_BLACK)
            for b in self.b:
                b.init_couplings_white()
        return self

    def sweep(self, nsweeps: int = 1):
        for _ in range(nsweeps):
            self.it += 1
            for color in (BLACK, WHITE):
                if self.n == 1:
                    self.b[0].update_all(self.it, color)
                    continue
                for b in self.b:
                    b.update_edges(self.it, color)
                self._exchange(color)
                for b in self.b:
                    b.update_interior(self.it, color)
        return self

    def count(self):
        ups, downs = zip(*(b.count_up_down() for b in self.b))
        return sum(ups), sum(downs)

    def bond_equal(self) -> int:
        return sum(b.bond_equal() for b in self.b)


class NativeRing:
    """One slab per process, the whole half-sweep schedule inside libising_hip.so (ising_rank_*): the library owns a
    second HIP stream per slab and moves the rows itself; torch.distributed only carries the attachment data once.
    transport "rccl": an RCCL communicator per slab (the 128-byte id travels through torch.distributed);
    transport "ipc":  no RCCL -- every rank maps its neighbours' rows through hipIpcMemHandle and pushes its edge rows
                      into them (ising_ipc_export / ising_ipc_attach; the blobs travel through torch.distributed).  Ranks
                      may share a device, which RCCL refuses.
    Same driver surface as SlabRing (init / sweep / count / bond_equal / quiesce)."""

    def __init__(self, slab, group: Optional[dist.ProcessGroup] = None, probe_timeout_ms: int = 60000, transport: str = "rccl"):
        self.slab = slab
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if slab.nslabs != self.world or slab.slab != self.rank:
            raise ValueError(f"slab {slab.slab} of {slab.nslabs} on rank {self.rank} of {self.world}")
        if transport not in ("rccl", "ipc"):
            raise ValueError(f"unknown native transport {transport!r}")
        self.transport = transport
        self.exchange = "rccl-native" if transport == "rccl" else "ipc-native"
        self.probe_timeout_ms = probe_timeout_ms
        self.it = 0
        self._attach()

    def _attach(self):
        from .lattice import rccl_unique_id
        if self.transport == "ipc":
            # a rank whose export fails (shm_open, hipHostRegister, hipIpcGetMemHandle) still takes part in the gather, with
            # None, so that every rank leaves this function the same way -- an early raise would leave the others in the
            # collective and the caller's agreement protocol (open_native_ring) a collective out of step
            try:
                mine, err = self.slab.ipc_export(), None
            except IsingError as e:
                mine, err = None, e
            if self.world == 1:
                blobs = [mine]
            else:
                blobs = [None] * self.world
                dist.all_gather_object(blobs, mine, group=self.group)
            if err is not None:
                raise err
            failed = [r for r, b in enumerate(blobs) if b is None]
            if failed:
                raise IsingError(f"IPC export failed on rank(s) {failed}")
            self.slab.ipc_attach(blobs)
            return
        if self.world == 1:
            uid = rccl_unique_id()
        else:
            nccl = dist.get_backend(self.group) == "nccl"
            t = torch.zeros(128, dtype=torch.uint8, device="cuda" if nccl else "cpu")
            if self.rank == 0:
                t.copy_(torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8))
            dist.broadcast(t, src=0, group=self.group)
            uid = bytes(t.cpu().numpy().tobytes())
        self.slab.rank_attach(uid)

    def init(self):
        self.slab.init()
        self.it = 0
        self.slab.rank_exchange(BLACK)
        self.slab.rank_exchange(WHITE)
        # the first exchange also builds the transport's connections: bounded wait, so that a transport that cannot come up
        # surfaces as an error the caller can fall back from instead of a hang
        self.slab.rank_wait(self.probe_timeout_ms)
        if self.slab.use_J:
            self.slab.rank_init_couplings()
        return self

    def sweep(self, nsweeps: int = 1):
        self.slab.it = self.it
        self.slab.rank_sweep(nsweeps)
        self.it += nsweeps
        return self

    def sweep_counted(self, nsweeps: int, every: int, energy: bool = False):
        """sweep with the whole lattice's (up, down[, bond_equal]) after every iteration that is a multiple of `every`, counted inside the deep launches."""
        self.slab.it = self.it
        out = self.slab.rank_sweep_counted(nsweeps, every, energy)
        self.it += nsweeps
        return out

    def quiesce(self):
        self.slab.rank_wait(-1)

    def count(self):
        return self.slab.rank_count()

    def bond_equal(self) -> int:
        return self.slab.rank_bond_equal()

    def checkpoint_save(self, path):
        """Collective: the decomposition-independent checkpoint file (ising_rank_checkpoint_save), every rank its own rows."""
        self.slab.it = self.it
        self.slab.rank_checkpoint_save(path)

    def checkpoint_load(self, path, apply_temperature: bool = True):
        """Collective: spins from the file, the halo / ghost rows exchanged again; continues at the checkpoint's temperature."""
        from .lattice import checkpoint_info
        self.it = self.slab.rank_checkpoint_load(path)
        if apply_temperature:
            self.slab.set_temperature(checkpoint_info(path)["temp"])
        self.slab.rank_exchange(BLACK)
        self.slab.rank_exchange(WHITE)
        return self

    def close(self, abort: bool = False):
        self.slab.rank_detach(abort)


def _agree(ok: bool) -> bool:
    """True when `ok` holds on every rank (one small all-reduce on whatever backend carries the control plane)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return ok
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t[0]))


def open_native_ring(slab, log=None, transports=("ipc", "rccl"), attempts=None, cross_check=None, check_sweeps: int = 40):
    """The library's own ring on a slab that owns its buffer (ring slabs on the ballot layout then keep ghost rows 64 deep
    and exchange every 32 sweeps, csrc/ising_ring.cpp: sweep_deep).  Primary transport (round 6): the peer transport over
    hipIpcMemHandle -- direct stores into the neighbours' rows, the reference's own mechanism across processes (optimized/main.cu:1496-1537,
    :1637-1642), the one whose exchange runs INSIDE the launch and whose launches carry several exchange epochs; RCCL send/recv is the
    fallback (its kernel runs when the launch's workgroups retire).  Ranks sharing a device can only use the first -- RCCL refuses them.  Every
    rank takes the same decision: the outcome of each attempt is agreed on before anyone moves on.  None when no transport
    comes up on every rank (the caller then builds a torch-owned slab and calls open_ring for the torch.distributed rings).
    `cross_check` (default: whenever more than one transport is listed): the first transport that comes up is not taken at its word -- the next one
    runs the same `check_sweeps` sweeps from the same start (they cross an exchange of ghost rows), and the whole lattice's counts must agree; if they
    do not, the LATER transport of the list is kept (RCCL, the one with the mileage between distinct devices) and the attempt record says so.  A
    transport that has never run between two devices should not be able to cost a scaling run its parity.
    `attempts` (a list, optional) receives one record per transport tried on THIS rank: {"transport", "ok", "error", "all_ranks_ok", "seconds"[, "cross_check"]}."""
    import os
    import time
    log = log or (lambda *a: None)
    if cross_check is None:
        cross_check = len(transports) > 1

    def bring_up(tr, sweeps):
        """(ring or None, counts or None): the transport attached, the lattice initialised, `sweeps` sweeps through it, the whole lattice counted"""
        ring, err, counts, t0 = None, None, None, time.perf_counter()
        try:
            ring = NativeRing(slab, transport=tr)
            ring.init()
            ring.sweep(sweeps)  # real sweeps through the transport before it is trusted
            ring.slab.rank_wait(ring.probe_timeout_ms)  # (bounded, like the first exchange: a transport that stalls is an error to fall back from, not a hang)
            torch.cuda.synchronize()
            counts = ring.count() if sweeps > 1 else None
            ok = True
        except Exception as e:  # noqa: BLE001 -- any transport failure means: the caller tries the next one
            log(f"ring transport {tr}-native failed on this rank: {e}")
            ok, err = False, f"{type(e).__name__}: {e}"
        agreed = _agree(ok)
        rec = {"transport": f"{tr}-native", "ok": ok, "error": err, "all_ranks_ok": agreed, "seconds": round(time.perf_counter() - t0, 3)}
        if attempts is not None:
            attempts.append(rec)
        if agreed:
            return ring, counts, rec
        if ring is not None:
            try:
                ring.close(abort=True)
            except Exception:  # noqa: BLE001
                pass
        elif slab is not None:
            try:
                slab.rank_detach(True)  # (an attachment that half came up)
            except Exception:  # noqa: BLE001
                pass
        return None, None, rec

    todo = list(transports)
    while todo:
        tr = todo.pop(0)
        ring, counts, rec = bring_up(tr, check_sweeps if (cross_check and todo) else 1)
        if ring is None:
            continue
        if not (cross_check and todo):
            return ring
        if os.environ.get("ISING_TEST_RING_CROSSCHECK_PERTURB"):  # (test aid: the first transport's counts off by one)
            counts = (counts[0] + 1, counts[1] - 1)
        # the next transport over the same sweeps
        ring.close()
        other = None
        while todo and other is None:
            tr2 = todo.pop(0)
            other, counts2, rec2 = bring_up(tr2, check_sweeps)
        if other is None:  # nothing to hold it against: the first transport it is
            rec["cross_check"] = "no second transport came up"
            again, _, _ = bring_up(tr, 1)
            return again
        if counts2 == counts:
            rec["cross_check"] = rec2["cross_check"] = f"{tr} and {tr2} agree after {check_sweeps} sweeps: {counts}"
            other.close()
            again, _, _ = bring_up(tr, 1)
            if again is not None:
                return again
            other, _, _ = bring_up(tr2, 1)  # (the first one did not come up a second time)
            return other
        rec["cross_check"] = rec2["cross_check"] = f"{tr} {counts} and {tr2} {counts2} DISAGREE after {check_sweeps} sweeps: {tr2} kept"
        log(f"ring transports disagree after {check_sweeps} sweeps ({tr}: {counts}, {tr2}: {counts2}): keeping {tr2}")
        return other
    return None


def open_ring(backend: "HipSlabBackend", prefer: str = "native", exchange: Optional[str] = None, log=None, attempts_out=None):
    """The ring a multi-process driver (bench.py) should use: the library's own RCCL ring when it comes up, otherwise
    the torch.distributed one (p2p, then all-gather).  Every rank takes the same decision: the outcome of each attempt is
    agreed on with an all-reduce before anyone moves on.  Returns (ring, name)."""
    log = log or (lambda *a: None)
    world = dist.get_world_size() if dist.is_initialized() else 1

    def agreed(ok: bool) -> bool:
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))

    attempts = []
    if prefer == "native" and exchange is None:
        attempts.append("rccl-native")
    if world > 1:
        attempts += [exchange] if exchange else ["p2p", "allgather"]
    elif getattr(backend, "ring_of_one", False):  # a lone slab with halo rows sends its edge rows to itself
        attempts += [exchange or "p2p"]
    last = None
    for name in attempts:
        ring = None
        try:
            if name == "rccl-native":
                ring = NativeRing(backend.slab)
            else:
                ring = SlabRing(backend, exchange=name)
            ring.init()
            ring.sweep(1)  # one real sweep through the transport before it is trusted
            ring.quiesce()
            torch.cuda.synchronize()
            ok = True
        except Exception as e:  # noqa: BLE001 -- any transport failure means: try the next one
            last = e
            ok = False
            log(f"ring transport {name} failed on this rank: {e}")
        all_ok = agreed(ok)
        if attempts_out is not None:
            attempts_out.append({"transport": name if name == "rccl-native" else f"torch-{name}", "ok": ok, "error": None if ok else f"{type(last).__name__}: {last}", "all_ranks_ok": all_ok})
        if all_ok:
            return ring, name
        if ring is not None and name == "rccl-native":
            try:
                ring.close(abort=True)
            except Exception:  # noqa: BLE001
                pass
    raise IsingError(f"no ring transport came up (last error: {last})")
