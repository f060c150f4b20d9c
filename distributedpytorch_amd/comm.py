"""Communication layer: one process per GPU, ``torch.distributed`` over RCCL (xGMI).

Reference: the scripts call ``dist.init_process_group`` with a TCP url
(multinode-DP/multinodeDP.py:159-162, multinode-MP/multinodeMP.py:172-178), use one
blocking ``all_reduce`` per parameter (DP.py:126-129) and blocking CPU-staged
``send``/``recv`` (MP.py:140,144,153,167).  Here:

* backend ``rccl`` (an alias of c10d's ``nccl``, which *is* RCCL on ROCm) moves
  device buffers directly - no host staging; ``gloo`` is the CPU loopback backend
  used for tests and for BASELINE config 1.
* collectives are issued asynchronously on RCCL's own stream; ordering against the
  compute stream is done by c10d's event record/wait, so the caller overlaps them
  with backward and only waits on the handle before the optimizer.
* sub-groups (DP group per pipeline stage, PP group per replica) for the hybrid.
* a debug mode (``DPA_SYNC_DEBUG=1``) that synchronises after every op so stream
  ordering bugs become deterministic failures.
"""
from __future__ import annotations

import datetime
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

_SYNC_DEBUG = os.environ.get("DPA_SYNC_DEBUG", "0") == "1"
# DPA_FORCE_COLLECTIVES=1: issue the real collective even at world size 1 (tests: the whole RCCL call path on one GPU)
_FORCE_COLLECTIVES = os.environ.get("DPA_FORCE_COLLECTIVES", "0") == "1"
_GLOO_NATIVE_DEVICE = os.environ.get("DPA_GLOO_DEVICE", "staged").lower() == "native"
# Which layer carries device collectives under the RCCL backend: "native" = csrc/rccl_comm.cpp (own C++ communicator,
# own comm stream, ncclCommSplit sub-groups, optional direct / bf16-wire all-reduce; the default), "c10d" =
# torch.distributed's ProcessGroupNCCL.  The native layer is checked by a self-test at start-up (a small and a
# bucket-sized all-reduce with a host watchdog, result verified); if any rank fails it, EVERY rank falls back to c10d.
_COMM_LAYER = os.environ.get("DPA_COMM", "native").lower()
_SELFTEST_TIMEOUT_S = float(os.environ.get("DPA_COMM_SELFTEST_TIMEOUT", "60"))
# With the native layer, torch.distributed is only the CONTROL PLANE (rendezvous, object broadcast, flag agreement, host
# barriers) and runs on gloo: RCCL is then initialised exactly once per process, by csrc/rccl_comm.cpp (c10d's own
# ProcessGroupNCCL would be a second world communicator: double channel buffers and proxy threads for one flag
# all-reduce).  DPA_CONTROL_PLANE=nccl restores the round-2 arrangement (c10d on RCCL beside the native communicator).
_CONTROL_PLANE = os.environ.get("DPA_CONTROL_PLANE", "gloo").lower()
# blocking native send() / recv(): host watchdog (seconds) instead of a silent hang on an unmatched transfer
_P2P_TIMEOUT_S = float(os.environ.get("DPA_P2P_TIMEOUT", "120"))
# All-reduce algorithm for fp32 sums on GPU tensors: "ring" = RCCL's ncclAllReduce, "direct" = one-shot reduce-scatter +
# all-gather as grouped RCCL send/recv (rccl_comm.cpp), "ipc" = ONE kernel over hipIpc-mapped peer memory
# (csrc/peer_allreduce.hip; needs every rank of the group on one xGMI node, at most 8; self-tested at start-up, and every
# rank falls back to the RCCL path when any rank fails it).  DPA_PEER_CHUNK_MB: staging bytes per launch (default 64),
# DPA_PEER_BLOCKS: workgroups per launch (default 32, the CUs the reduction may take from the backward pass).
_ALLREDUCE = os.environ.get("DPA_ALLREDUCE", "ring").lower()
_PEER_CHUNK_MB = int(os.environ.get("DPA_PEER_CHUNK_MB", "64"))
_PEER_BLOCKS = int(os.environ.get("DPA_PEER_BLOCKS", "32"))
_PEER_TIMEOUT_S = float(os.environ.get("DPA_PEER_TIMEOUT", "30"))


def set_sync_debug(flag: bool) -> None:
    global _SYNC_DEBUG
    _SYNC_DEBUG = bool(flag)


def resolve_backend(backend: str, device: Optional[torch.device] = None) -> str:
    b = (backend or "auto").lower()
    if b == "auto":
        use_gpu = torch.cuda.is_available() if device is None else device.type == "cuda"
        return "nccl" if use_gpu else "gloo"
    if b in ("rccl", "nccl"):
        return "nccl"
    if b == "gloo":
        return "gloo"
    raise ValueError(f"unknown backend {backend!r} (use auto|rccl|nccl|gloo)")


def rccl_env_defaults() -> None:
    """Environment for RCCL over the 8-GPU xGMI mesh.  Only sets defaults.

    RCCL's own channel / algorithm tuning for gfx950 is left alone: nothing here could be measured on
    more than one GPU, and an unmeasured override (e.g. NCCL_MIN_NCHANNELS) takes CUs from the
    overlapped backward kernels.
    """
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (required by the host driver)
    # compute + weight-gradient side stream + RCCL's stream must not share hardware queues (HIP maps a process's
    # streams onto GPU_MAX_HW_QUEUES queues, default 4: -9 % on ResNet-50 when two of them alias).  Only effective
    # when this runs before the first HIP call of the process; bench.py / launch.py also set it at start-up.
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")


def init_process_group(backend: str = "auto", init_method: Optional[str] = None, rank: Optional[int] = None,
                       world_size: Optional[int] = None, timeout_s: float = 300.0,
                       device: Optional[torch.device] = None) -> "Communicator":
    """Rendezvous + communicator creation.

    ``init_method`` is the reference's ``tcp://MASTER:PORT`` url; when omitted the
    ``MASTER_ADDR``/``MASTER_PORT``/``RANK``/``WORLD_SIZE`` environment is used
    (that is what torchrun / bench.py export).
    """
    be = resolve_backend(backend, device)
    if be == "nccl":
        rccl_env_defaults()
    on_gpu = device is not None and device.type == "cuda"
    native_wanted = _COMM_LAYER == "native" and be == "nccl" and on_gpu
    # native layer: c10d is the control plane only, on gloo - ONE RCCL world communicator per process (the native one)
    c10d_be = "gloo" if (native_wanted and _CONTROL_PLANE == "gloo") else be
    kwargs = dict(backend=c10d_be, timeout=datetime.timedelta(seconds=timeout_s))
    if init_method is not None and init_method != "env://":
        kwargs.update(init_method=init_method, rank=rank, world_size=world_size)
        # keep env in step so libraries that read it agree (MP.py:175-176 does the same)
        hostport = init_method.split("://", 1)[-1]
        if ":" in hostport:
            host, port = hostport.rsplit(":", 1)
            os.environ.setdefault("MASTER_ADDR", host)
            os.environ.setdefault("MASTER_PORT", port)
    else:
        if rank is not None:
            os.environ.setdefault("RANK", str(rank))
        if world_size is not None:
            os.environ.setdefault("WORLD_SIZE", str(world_size))
    if c10d_be == "nccl" and on_gpu:
        kwargs["device_id"] = device
    if c10d_be == "gloo":
        _pin_gloo_to_loopback()
    if not dist.is_initialized():
        dist.init_process_group(**kwargs)
    comm = Communicator(group=None, device=device)
    comm._timeout_s = timeout_s
    if native_wanted and (comm.world_size > 1 or _FORCE_COLLECTIVES):
        comm.attach_native()
    elif os.environ.get("DPA_FAKE_NATIVE", "0") == "1" and comm._c10d_backend() == "gloo" and comm.world_size > 1:
        # rehearsal of the native layer's Python side at world size > 1 without several GPUs (utils/fake_native.py):
        # the same self-test, the same call paths, the data carried by gloo
        from .utils.fake_native import GlooNative
        comm.native = GlooNative(None)
        comm._selftest_native(strict=False)
    if native_wanted and comm.native is None and comm._c10d_backend() != "nccl" and \
            (comm.world_size > 1 or _FORCE_COLLECTIVES):
        # the native layer is not there (never created, or it failed its self-test on some rank): device traffic
        # goes through c10d's ProcessGroupNCCL after all - created now, collectively, and only now
        comm._adopt_c10d_nccl()
    if _ALLREDUCE == "ipc" and on_gpu and comm.world_size > 1:
        comm.attach_peer()
    return comm


def _pin_gloo_to_loopback() -> None:
    """A single-node job (rendezvous at 127.0.0.1 / localhost) keeps gloo on the loopback interface: gloo otherwise looks
    its interface up through the HOSTNAME, which need not resolve inside a container (it then falls back to loopback with a
    warning on current builds, and failed outright on older ones).  Only when the user has not chosen an interface, the job
    is local and ``lo`` exists; ``--ifname`` / GLOO_SOCKET_IFNAME always win."""
    if os.environ.get("GLOO_SOCKET_IFNAME") or os.environ.get("MASTER_ADDR", "") not in ("127.0.0.1", "localhost", "::1"):
        return
    try:
        import socket
        if any(name == "lo" for _, name in socket.if_nameindex()):
            os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    except OSError:
        pass


def init_native_group(init_method: str, rank: int, world_size: int, device: torch.device,
                      timeout_s: float = 300.0) -> "Communicator":
    """Rendezvous WITHOUT torch.distributed: rank 0 serves a small TCP key-value store at ``tcp://host:port``
    (distributedpytorch_amd/store.py), the 128-byte RCCL unique id travels through it, and every collective of the
    returned communicator runs on the native RCCL layer (csrc/rccl_comm.cpp).  Same contract as the reference's
    ``dist.init_process_group(init_method="tcp://...")`` (DP.py:161, MP.py:177)."""
    from .ops import _ext
    from .store import TCPKVStore, parse_tcp_url
    rccl_env_defaults()
    host, port = parse_tcp_url(init_method)
    store = TCPKVStore(host, port, is_master=(rank == 0), timeout_s=timeout_s)
    C = _ext.load()
    if rank == 0:
        store.set("rccl/unique_id/world", C.RcclComm.unique_id())
    uid = store.get("rccl/unique_id/world", timeout_s)
    index = device.index if device.index is not None else torch.cuda.current_device()
    comm = Communicator(group=None, device=device, ranks=None, static=(rank, world_size), store=store)
    comm.native = C.RcclComm(uid, rank, world_size, index)
    comm._selftest_native(strict=True)
    if _ALLREDUCE == "ipc" and world_size > 1:
        comm.attach_peer()
    return comm


def destroy_process_group() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()


def get_rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


class _NativeWork:
    """Handle of an operation on the native RCCL communicator's stream (csrc/rccl_comm.cpp)."""

    def __init__(self, native, ticket: int):
        self._native, self._ticket = native, ticket    # the buffers are recorded on the comm stream by the C++ side

    def wait(self):
        self._native.wait(self._ticket)      # the current compute stream waits on the event; no host block
        return True

    def is_completed(self):
        return bool(self._native.query(self._ticket))


class _PeerWork:
    """Handle of an all-reduce on the peer-mapped layer's stream (csrc/peer_comm.cpp)."""

    def __init__(self, peer, ticket: int):
        self._peer, self._ticket = peer, ticket

    def wait(self):
        self._peer.wait(self._ticket)        # the current compute stream waits on the event; no host block
        return True

    def is_completed(self):
        return bool(self._peer.query(self._ticket))


class _DoneWork:
    """Stand-in handle for world_size == 1 / already-finished ops."""
    def wait(self):
        return True

    def is_completed(self):
        return True


class _StagedWork:
    """Handle of a device-tensor transfer staged through host memory (gloo): ``wait()`` completes the host transfer
    and, for a receive, copies into the device tensor."""

    def __init__(self, work, host: torch.Tensor, target: Optional[torch.Tensor]):
        self.work, self.host, self.target = work, host, target

    def wait(self):
        self.work.wait()
        if self.target is not None:
            self.target.copy_(self.host)
            self.target = None
        return True

    def is_completed(self) -> bool:
        return self.work.is_completed()


class Communicator:
    """Thin, group-scoped view of the process group.

    All tensor arguments are used in place.  With the RCCL backend they are device
    buffers and nothing is staged through the host.
    """

    def __init__(self, group=None, device: Optional[torch.device] = None, ranks: Optional[Sequence[int]] = None,
                 static: Optional[tuple] = None, store=None):
        self.group = group
        self.device = device
        self._ranks = list(ranks) if ranks is not None else None
        self._static = static                 # (rank, world_size) of a communicator that lives without c10d
        self.store = store                    # store.TCPKVStore of such a communicator (objects, bootstrap)
        self.initialized = dist.is_initialized() and static is None
        self.native = None                    # csrc/rccl_comm.cpp communicator over the same ranks
        self.peer = None                      # csrc/peer_comm.cpp: all-reduce over hipIpc-mapped peer memory (DPA_ALLREDUCE=ipc)
        # all-reduce algorithm of the native layer: "ring" = RCCL's own ncclAllReduce, "direct" = one-shot
        # reduce-scatter + all-gather over the 7 xGMI links (rccl_comm.cpp); wire format of the direct form
        self._direct = _ALLREDUCE == "direct"
        self.grad_wire = os.environ.get("DPA_GRAD_WIRE", "fp32").lower()
        self._scratch = {}
        self._obj_seq = 0
        self._timeout_s = 300.0
        self._c10d_sub_backend = None         # backend argument of dist.new_group for sub-groups (None = the default group's)

    # ---- native RCCL communicator: own comm stream, event hand-offs, direct / bf16-wire all-reduce ----
    def attach_native(self, parent: Optional["Communicator"] = None, member: bool = True) -> None:
        """Collective over this communicator's ranks (over the PARENT's ranks when ``parent`` is given: every rank
        of the parent calls it, members and non-members alike - ncclCommSplit is collective over the parent).

        World communicator: bootstrap a native ``RcclComm`` (the 128-byte unique id travels through the existing
        process group).  Sub-communicator: ``ncclCommSplit`` of the parent's native communicator."""
        from .ops import _ext
        C = _ext.load()
        host_ok = parent is not None and getattr(parent.native, "accepts_host", False)     # utils/fake_native.py (tests)
        if (self.device is None or self.device.type != "cuda") and not host_ok:
            raise RuntimeError("the native RCCL communicator needs a GPU device")
        if parent is not None:
            if parent.native is None:
                return
            color = self._ranks[0] if member else -1             # one colour per sub-group: its first global rank
            self.native = parent.native.split(color, self.rank if member else 0, self.rank if member else 0,
                                              self.world_size if member else 1)
            return
        box = [C.RcclComm.unique_id() if self.rank == 0 else None]
        if self.world_size > 1:
            dist.broadcast_object_list(box, src=self.to_global(0), group=self.group)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        try:
            self.native = C.RcclComm(box[0], self.rank, self.world_size, index)
        except Exception as e:  # noqa: BLE001 - e.g. ncclCommInitRank refusing the topology: c10d's group still works
            import sys
            print(f"[comm] rank {self.global_rank}: native RCCL communicator could not be created ({type(e).__name__}: {e})",
                  file=sys.stderr, flush=True)
            self.native = None
        self._selftest_native(strict=False)

    def _watch(self, ticket: int, what: str) -> None:
        """Host watchdog around one native operation of the self-test / of a blocking point-to-point call."""
        import time
        limit = _SELFTEST_TIMEOUT_S
        t0 = time.monotonic()
        while not self.native.query(ticket):
            if time.monotonic() - t0 > limit:
                raise TimeoutError(f"native {what} did not complete in {limit:.0f} s")
            time.sleep(0.0005)
        self.native.wait(ticket)

    def _selftest_native(self, strict: bool) -> None:
        """First contact of the native layer with the machine, every operation under a host watchdog and checked
        against known values: a 4 KB and a 32 MB all-reduce; at world size > 1 also broadcast, reduce, reduce-scatter /
        all-gather, a neighbour RING EXCHANGE through ``batch_p2p`` (ncclSend + ncclRecv in one group: what the
        pipeline engine issues at every stage boundary) and an ``ncclCommSplit`` into two halves with an all-reduce
        inside each.  ``strict``: raise on failure (no c10d to fall back to); else every rank drops to c10d when ANY
        rank failed (decided by an all-reduce on the control plane, so the ranks cannot disagree)."""
        W, r = self.world_size, self.rank
        dev = self.device

        def spin(comm, ticket, what):
            import time
            t0 = time.monotonic()
            while not comm.query(ticket):
                if time.monotonic() - t0 > _SELFTEST_TIMEOUT_S:
                    raise TimeoutError(f"{what} did not complete in {_SELFTEST_TIMEOUT_S:.0f} s")
                time.sleep(0.0005)
            comm.wait(ticket)

        def phase_allreduce():
            if self.native is None:
                raise RuntimeError("no native communicator on this rank")
            if _SELFTEST_TIMEOUT_S <= 0:
                raise TimeoutError("self-test disabled by DPA_COMM_SELFTEST_TIMEOUT=0 (fallback rehearsal)")
            for n in (1024, 8 * 2 ** 20):
                t = torch.full((n,), float(r + 1), dtype=torch.float32, device=dev)
                self._watch(self.native.all_reduce(t, 0), f"all-reduce of {n} floats")
                want = W * (W + 1) / 2.0
                got = t[:: max(1, n // 64)].float().cpu()
                if not bool((got == want).all()):
                    raise RuntimeError(f"native all-reduce returned {float(got[0])}, expected {want}")

        def phase_collectives_p2p():
            n = 4096
            t = torch.full((n,), float(r + 1), dtype=torch.float32, device=dev)
            self._watch(self.native.broadcast(t, W - 1), "broadcast")
            if float(t[-1]) != float(W):
                raise RuntimeError(f"native broadcast returned {float(t[-1])}, expected {float(W)}")
            t.fill_(float(r + 1))
            self._watch(self.native.reduce(t, 0), "reduce")
            if r == 0 and float(t[0]) != W * (W + 1) / 2.0:
                raise RuntimeError(f"native reduce returned {float(t[0])}")
            src = torch.arange(W * 8, dtype=torch.float32, device=dev) + 1000.0 * r
            piece = torch.empty(8, dtype=torch.float32, device=dev)
            self._watch(self.native.reduce_scatter(piece, src), "reduce-scatter")
            want = torch.arange(8, dtype=torch.float32) + 8.0 * r
            want = want * W + 1000.0 * (W * (W - 1) / 2.0)
            if not torch.equal(piece.cpu(), want):
                raise RuntimeError("native reduce-scatter returned wrong values")
            full = torch.empty(W * 8, dtype=torch.float32, device=dev)
            self._watch(self.native.all_gather(full, piece), "all-gather")
            q = (r + 1) % W                    # the neighbour's piece: (arange(8) + 8 q) * W + 1000 * W (W - 1) / 2
            if float(full[8 * q]) != 8.0 * q * W + 1000.0 * (W * (W - 1) / 2.0):
                raise RuntimeError("native all-gather returned wrong values")
            # neighbour ring through grouped send/recv: rank r -> r + 1, r - 1 -> r, one group per rank
            out = torch.full((n,), float(r), dtype=torch.bfloat16, device=dev)
            inp = torch.full((n,), -1.0, dtype=torch.bfloat16, device=dev)
            self._watch(self.native.batch_p2p([(True, out, (r + 1) % W), (False, inp, (r - 1) % W)]), "grouped send/recv ring")
            if float(inp[0]) != float((r - 1) % W) or float(inp[-1]) != float((r - 1) % W):
                raise RuntimeError(f"native send/recv ring delivered {float(inp[0])}, expected {float((r - 1) % W)}")

        def phase_split():
            # sub-communicators (what build_grid makes for DP x PP): [0, W/2) and [W/2, W) ...
            half = W // 2
            lo = r < half
            sub = self.native.split(0 if lo else 1, r, r if lo else r - half, half if lo else W - half)
            t = torch.ones(256, dtype=torch.float32, device=dev)
            sub_w = half if lo else W - half
            spin(sub, sub.all_reduce(t, 0), "all-reduce on an ncclCommSplit sub-communicator")
            if float(t[0]) != float(sub_w):
                raise RuntimeError(f"sub-communicator all-reduce returned {float(t[0])}, expected {sub_w}")
            sub.synchronize()
            del sub
            if W > 2:
                # ... and a split that leaves the last rank out (NCCL_SPLIT_NOCOLOR: build_grid's idle ranks)
                idle = r == W - 1
                sub = self.native.split(-1 if idle else 0, 0 if idle else r, 0 if idle else r, 1 if idle else W - 1)
                if idle != (sub is None):
                    raise RuntimeError("ncclCommSplit with NOCOLOR returned a communicator to the wrong ranks")
                if sub is not None:
                    t = torch.ones(256, dtype=torch.float32, device=dev)
                    spin(sub, sub.all_reduce(t, 0), "all-reduce on the NOCOLOR-split sub-communicator")
                    if float(t[0]) != float(W - 1):
                        raise RuntimeError(f"NOCOLOR-split all-reduce returned {float(t[0])}, expected {W - 1}")
                    sub.synchronize()
                    del sub

        def everyone(ok: bool) -> bool:
            """Agreement on the control plane (never on the layer under test): the c10d group - host tensor under gloo,
            device tensor when c10d itself runs on RCCL - or the TCP store of a c10d-free communicator."""
            if W <= 1:
                return ok
            if self._static is not None:
                return self._agree(ok)
            on_dev = self._c10d_backend() == "nccl"
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if on_dev else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return bool(int(flag.item()) == 1)

        # Phase by phase, with an agreement after each: a rank that failed a check must not leave the others alone in
        # the next COLLECTIVE of the layer under test (ncclCommSplit blocks on the host: no watchdog could end it).
        ok, why, all_ok = True, "", True
        phases = [phase_allreduce] + ([phase_collectives_p2p, phase_split] if W > 1 else [])
        for phase in phases:
            try:
                phase()
            except Exception as e:  # noqa: BLE001
                ok, why = False, f"{type(e).__name__}: {e}"
            all_ok = everyone(ok)
            if not all_ok:
                break
        if strict:
            if not all_ok:
                raise RuntimeError(f"native RCCL communicator failed its self-test: {why or 'on another rank'}")
            return
        if not all_ok:
            import sys
            print(f"[comm] rank {self.global_rank}: native RCCL layer failed its self-test"
                  f"{' here (' + why + ')' if not ok else ' on another rank'}; falling back to c10d", file=sys.stderr, flush=True)
            try:
                if self.native is not None:
                    self.native.abort()
            except Exception:  # noqa: BLE001
                pass
            self.native = None

    # ---- peer-mapped all-reduce: one kernel over hipIpc-mapped staging buffers ----
    def _exchange(self, obj):
        """All-gather of a small picklable object over the control plane (c10d group, or the TCP store of a c10d-free
        communicator).  Collective over this communicator."""
        if self._static is not None:
            key = self._obj_key("xchg")
            self.store.set_object(f"{key}/{self.rank}", obj, readers=self.world_size)
            return [self.store.get_object(f"{key}/{r}") for r in range(self.world_size)]
        out = [None] * self.world_size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def _agree(self, ok: bool) -> bool:
        """True when EVERY rank of this communicator passed ``ok`` (decided on the control plane: the ranks cannot
        end up on different paths)."""
        if self.world_size <= 1:
            return ok
        return all(self._exchange(bool(ok)))

    def attach_peer(self) -> None:
        """Collective over this communicator: allocate staging + signal memory, exchange the hipIpc handles over the
        control plane, map every peer, self-test.  Leaves ``self.peer`` None (all ranks alike) when anything fails -
        all-reduces then stay on the RCCL / c10d path."""
        import sys
        from .ops import _ext
        peer, ok, why = None, True, ""
        try:
            if self.device is None or self.device.type != "cuda":
                raise RuntimeError("needs a GPU device")
            C = _ext.load()
            if self.world_size > C.PEER_MAX_RANKS:
                raise RuntimeError(f"{self.world_size} ranks > {C.PEER_MAX_RANKS} (one xGMI node)")
            index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            peer = C.PeerAllReduce(self.rank, self.world_size, index, _PEER_CHUNK_MB << 20, _PEER_BLOCKS, _PEER_TIMEOUT_S)
            import secrets
            # third entry: a random epoch base for the sequence numbers; every rank adopts rank 0's (csrc/peer_comm.cpp set_base)
            mine = tuple(peer.handles()) + ((1 << 20) + secrets.randbits(30),)
        except Exception as e:  # noqa: BLE001
            ok, why, mine = False, f"{type(e).__name__}: {e}", None
        everyone = self._exchange(mine)
        if ok and any(h is None for h in everyone):
            ok, why = False, "another rank could not allocate / export its buffers"
        if ok:
            try:
                peer.open([(bytes(h[0]), bytes(h[1])) for h in everyone])
                peer.set_base(int(everyone[0][2]))
            except Exception as e:  # noqa: BLE001 - e.g. GPUs without peer access, a handle from another node
                ok, why = False, f"{type(e).__name__}: {e}"
        if not self._agree(ok):
            if not ok:
                print(f"[comm] rank {self.global_rank}: peer-mapped all-reduce unavailable ({why}); staying on RCCL",
                      file=sys.stderr, flush=True)
            return
        self.peer = peer
        ok, why = self._selftest_peer()
        if not self._agree(ok):
            print(f"[comm] rank {self.global_rank}: peer-mapped all-reduce failed its self-test"
                  f"{' here (' + why + ')' if not ok else ' on another rank'}; staying on RCCL", file=sys.stderr, flush=True)
            self.peer = None

    def _selftest_peer(self):
        """A ragged small tensor, one larger than a staging chunk, and the bf16 wire format, each under a host watchdog
        and checked element by element; then the kernel's own time-out word."""
        import time
        W, r, dev = self.world_size, self.rank, self.device
        cases, ok, why = [], True, ""
        try:
            chunk = int(self.peer.chunk_elems)
            for n, bf16 in ((1003, False), (chunk + 4099, False), (70001, True)):
                base = (torch.arange(n, dtype=torch.float32, device=dev) % 61.0) * 0.25       # exact in bf16
                cases.append((n, bf16, base, base + float(r + 1)))
            torch.cuda.synchronize(dev)
        except Exception as e:  # noqa: BLE001
            ok, why = False, f"{type(e).__name__}: {e}"
        # The kernel's polls are bounded, so the ranks must ENTER it within that bound of one another.  In steady state
        # the previous collective keeps them within a step's imbalance; at first contact they can be tens of seconds
        # apart (a cold process loads its code objects on its first device operation): finish every device-side
        # preparation, meet on the control plane, then launch.
        if not self._agree(ok):
            return False, why or "another rank could not prepare the self-test"
        try:
            for n, bf16, base, t in cases:
                ticket = self.peer.all_reduce(t, 1.0, bf16)
                t0 = time.monotonic()
                while not self.peer.query(ticket):
                    if time.monotonic() - t0 > _SELFTEST_TIMEOUT_S:
                        raise TimeoutError(f"peer all-reduce of {n} floats did not complete in {_SELFTEST_TIMEOUT_S:.0f} s")
                    time.sleep(0.0005)
                self.peer.wait(ticket)
                took = time.monotonic() - t0
                if self.peer.error_seq() != 0:
                    d = int(self.peer.error_detail())
                    raise RuntimeError(f"launch {self.peer.error_seq()}: barrier {'AB'[(d >> 16) & 1]} of block {(d >> 8) & 255} "
                                       f"never saw rank {d & 255} ({took:.1f} s; the ranks' kernels must run at the same time)")
                want = base * W + W * (W + 1) / 2.0
                bad = int((t != want).sum().item())
                if bad:
                    raise RuntimeError(f"peer all-reduce of {n} floats (bf16 wire {bf16}): {bad} wrong elements after {took:.2f} s")
            return True, ""
        except Exception as e:  # noqa: BLE001
            return False, f"{type(e).__name__}: {e}"

    def check_peer(self) -> None:
        """Raise if a bounded poll of the peer-mapped all-reduce ever ran out on this rank (a peer did not arrive at a
        barrier within DPA_PEER_TIMEOUT seconds: the kernel then finished with wrong sums instead of hanging the queue).
        One small device-to-host copy: for synchronisation points (end of an epoch, of a benchmark, before a checkpoint)."""
        if self.peer is None:
            return
        seq = int(self.peer.error_seq())
        if seq != 0:
            d = int(self.peer.error_detail())
            raise RuntimeError(f"rank {self.global_rank}: peer-mapped all-reduce launch {seq}: barrier {'AB'[(d >> 16) & 1]} of "
                               f"workgroup {(d >> 8) & 255} never saw rank {d & 255} within {_PEER_TIMEOUT_S:.0f} s; the sums "
                               f"since then are wrong")

    def _peer_all_reduce(self, tensor: torch.Tensor, op):
        if op != dist.ReduceOp.SUM or tensor.dtype != torch.float32 or not (tensor.is_cuda and tensor.is_contiguous()):
            return None
        if tensor.data_ptr() % 16 != 0:
            return None
        return _PeerWork(self.peer, self.peer.all_reduce(tensor, 1.0, self.grad_wire == "bf16"))

    def _on_native(self, *tensors: torch.Tensor) -> bool:
        if self.native is None:
            return False
        host_ok = getattr(self.native, "accepts_host", False)        # utils/fake_native.py (tests) takes CPU tensors too
        return all((t.is_cuda or host_ok) and t.is_contiguous() for t in tensors)

    def _native_all_reduce(self, tensor: torch.Tensor, op):
        code = {dist.ReduceOp.SUM: 0, dist.ReduceOp.MAX: 1, dist.ReduceOp.MIN: 2}.get(op)
        if code is None or not self._on_native(tensor):
            return None
        if self._direct and code == 0 and tensor.dtype == torch.float32 and self.world_size > 1:
            # one-shot reduce-scatter + all-gather over the direct links; scratch per message size, reused
            bf16 = self.grad_wire == "bf16"
            shard = (-(-tensor.numel() // self.world_size) + 7) // 8 * 8
            need = shard * ((2 * self.world_size - 1) if bf16 else (self.world_size - 1))
            key = (need, bf16)
            scratch = self._scratch.get(key)
            if scratch is None:
                scratch = self._scratch[key] = torch.empty(need, dtype=torch.bfloat16 if bf16 else torch.float32,
                                                           device=tensor.device)
            return _NativeWork(self.native, self.native.all_reduce_direct(tensor, scratch))
        return _NativeWork(self.native, self.native.all_reduce(tensor, code))

    # ---- queries (reference: dist.get_rank / get_world_size, DP.py:108,112) ----
    @property
    def rank(self) -> int:
        if self._static is not None:
            return self._static[0]
        return dist.get_rank(self.group) if self.initialized else 0

    @property
    def world_size(self) -> int:
        if self._static is not None:
            return self._static[1]
        return dist.get_world_size(self.group) if self.initialized else 1

    @property
    def global_rank(self) -> int:
        if self._static is not None:
            return self._ranks[self._static[0]] if self._ranks is not None else self._static[0]
        return dist.get_rank() if self.initialized else 0

    @property
    def backend(self) -> str:
        """The transport of DEVICE tensors: "nccl" (= RCCL: the native layer or c10d's ProcessGroupNCCL), "gloo", or
        "none" without a process group.  (The c10d group under a native communicator is a gloo control plane.)"""
        if self._static is not None or self.native is not None:
            return "nccl"
        return self._c10d_backend()

    def _c10d_backend(self) -> str:
        if self._static is not None or not self.initialized:
            return "none"
        return dist.get_backend(self.group)

    def _adopt_c10d_nccl(self) -> None:
        """Fallback of the world communicator when the native layer is unavailable: a c10d RCCL group over the same
        ranks becomes the carrier of device collectives (collective over the world: every rank calls it)."""
        try:
            self.group = dist.new_group(ranks=list(range(dist.get_world_size())), backend="nccl",
                                        timeout=datetime.timedelta(seconds=self._timeout_s))
        except TypeError:                      # older signature without timeout
            self.group = dist.new_group(ranks=list(range(dist.get_world_size())), backend="nccl")
        self._c10d_sub_backend = "nccl"

    def to_global(self, group_rank: int) -> int:
        if self._ranks is None:
            return group_rank
        return self._ranks[group_rank]

    # ---- collectives ----
    def _native_result(self, ticket: int, async_op: bool):
        work = _NativeWork(self.native, ticket)
        if _SYNC_DEBUG:
            self.native.synchronize()
        if async_op:
            return work
        work.wait()
        return None

    def _need_c10d(self, what: str) -> None:
        if self._static is not None:
            raise RuntimeError(f"{what}: this communicator was built without torch.distributed (init_native_group) "
                               f"and only moves contiguous device tensors")

    def _single(self) -> bool:
        """World size 1 with nothing forcing the real call path."""
        return self.world_size == 1 and not (_FORCE_COLLECTIVES and (self.native is not None or dist.is_initialized()))

    def all_reduce(self, tensor: torch.Tensor, op=dist.ReduceOp.SUM, async_op: bool = False):
        if self._single():
            return _DoneWork() if async_op else None
        if self.peer is not None:
            work = self._peer_all_reduce(tensor, op)
            if work is not None:
                if _SYNC_DEBUG:
                    self.peer.synchronize()
                if async_op:
                    return work
                work.wait()
                return None
        if self.native is not None:
            work = self._native_all_reduce(tensor, op)
            if work is not None:
                if _SYNC_DEBUG:
                    self.native.synchronize()
                if async_op:
                    return work
                work.wait()
                return None
        self._need_c10d("all_reduce")
        if self._staged(tensor) and not _GLOO_NATIVE_DEVICE:
            host = tensor.cpu()
            work = _StagedWork(dist.all_reduce(host, op=op, group=self.group, async_op=True), host, tensor)
            if async_op:
                return work
            work.wait()
            return None
        work = dist.all_reduce(tensor, op=op, group=self.group, async_op=True)
        if _SYNC_DEBUG:
            work.wait()
            if tensor.is_cuda:
                torch.cuda.synchronize(tensor.device)
        if async_op:
            return work
        work.wait()
        return None

    def reduce(self, tensor: torch.Tensor, dst: int, async_op: bool = False):
        """Sum onto group rank ``dst`` (in place there; other ranks' buffers are left in an unspecified state).
        The sharded optimizer's gradient path (parallel/dp.py, ``shard_optimizer``)."""
        if self.world_size == 1:
            return _DoneWork() if async_op else None
        if self._on_native(tensor):
            return self._native_result(self.native.reduce(tensor, dst), async_op)
        self._need_c10d("reduce")
        if self._staged(tensor) and not _GLOO_NATIVE_DEVICE:
            host = tensor.cpu()
            inner = dist.reduce(host, dst=self.to_global(dst), group=self.group, async_op=True)
            work = _StagedWork(inner, host, tensor if self.rank == dst else None)
        else:
            work = dist.reduce(tensor, dst=self.to_global(dst), group=self.group, async_op=True)
        if async_op and not _SYNC_DEBUG:
            return work
        work.wait()
        return _DoneWork() if async_op else None

    def reduce_scatter(self, output: torch.Tensor, input: torch.Tensor, async_op: bool = False):
        if self.world_size == 1:
            output.copy_(input)
            return _DoneWork() if async_op else None
        if self._on_native(output, input):
            return self._native_result(self.native.reduce_scatter(output, input), async_op)
        self._need_c10d("reduce_scatter")
        work = dist.reduce_scatter_tensor(output, input, group=self.group, async_op=True)
        if async_op and not _SYNC_DEBUG:
            return work
        work.wait()
        return _DoneWork() if async_op else None

    def all_gather(self, output: torch.Tensor, input: torch.Tensor, async_op: bool = False):
        if self.world_size == 1:
            output.copy_(input)
            return _DoneWork() if async_op else None
        if self._on_native(output, input):
            return self._native_result(self.native.all_gather(output, input), async_op)
        self._need_c10d("all_gather")
        work = dist.all_gather_into_tensor(output, input, group=self.group, async_op=True)
        if async_op and not _SYNC_DEBUG:
            return work
        work.wait()
        return _DoneWork() if async_op else None

    def broadcast(self, tensor: torch.Tensor, src: int = 0, async_op: bool = False):
        if self.world_size == 1:
            return _DoneWork() if async_op else None
        if self._on_native(tensor):
            return self._native_result(self.native.broadcast(tensor, src), async_op)
        self._need_c10d("broadcast")
        if self._staged(tensor) and not _GLOO_NATIVE_DEVICE:
            host = tensor.cpu()
            work = _StagedWork(dist.broadcast(host, src=self.to_global(src), group=self.group, async_op=True), host, tensor)
            if async_op:
                return work
            work.wait()
            return None
        work = dist.broadcast(tensor, src=self.to_global(src), group=self.group, async_op=True)
        if async_op and not _SYNC_DEBUG:
            return work
        work.wait()
        return _DoneWork() if async_op else None

    def barrier(self):
        if self.world_size <= 1:
            return
        if self._static is not None:         # no c10d: a one-element all-reduce and a host wait for it
            t = torch.zeros(1, dtype=torch.float32, device=self.device)
            self.native.wait(self.native.all_reduce(t, 0))
            torch.cuda.current_stream(self.device).synchronize()
            return
        if self._c10d_backend() == "nccl" and self.device is not None:
            dist.barrier(group=self.group, device_ids=[self.device.index])
        else:
            dist.barrier(group=self.group)        # host barrier (gloo control plane); callers synchronise their own streams

    # ---- point to point (reference: MP.py:140,144,153,167) ----
    def _staged(self, tensor: torch.Tensor) -> bool:
        """A device tensor under ``--backend gloo`` (the reference's default, and what its MP README prescribes,
        multinode-MP/README.md:11) goes through a host copy, as in MP.py:139-145: gloo's point-to-point moves host
        memory only, and its own device-tensor collectives stopped making progress on a 102 MB broadcast on ROCm
        (ResNet-50's flat parameter buffer, two ranks; small tensors worked).  ``DPA_GLOO_DEVICE=native`` hands
        device tensors to gloo's collectives anyway.  Under RCCL nothing ever leaves the device."""
        return tensor.is_cuda and self._c10d_backend() != "nccl"

    def _native_blocking_p2p(self, ops, what: str) -> None:
        """One grouped native transfer, waited for on the HOST with a watchdog.  ncclSend / ncclRecv are rendezvous
        operations on the one communication stream of each rank: a transfer whose peer never posts the matching call -
        or two ranks that each ``send`` first and ``recv`` second to one another as separate calls - would block that
        stream forever.  The blocking forms therefore fail loudly after DPA_P2P_TIMEOUT seconds; exchanges in BOTH
        directions between two ranks belong in ONE group: ``sendrecv`` / ``batch_p2p``."""
        import time
        ticket = self.native.batch_p2p(ops)
        t0 = time.monotonic()
        while not self.native.query(ticket):
            if time.monotonic() - t0 > _P2P_TIMEOUT_S:
                raise TimeoutError(
                    f"rank {self.global_rank}: {what} did not complete in {_P2P_TIMEOUT_S:.0f} s - unmatched, or both ranks "
                    f"sent first: pair transfers that cross between two ranks through Communicator.sendrecv / batch_p2p")
            time.sleep(0.0002)
        self.native.wait(ticket)
        if _SYNC_DEBUG:
            self.native.synchronize()

    def sendrecv(self, send_tensor: torch.Tensor, recv_tensor: torch.Tensor, peer: int, recv_peer: Optional[int] = None):
        """Send to ``peer`` and receive from ``recv_peer`` (default: the same rank) as ONE grouped call: the
        deadlock-free form of an exchange, whatever order the other side uses inside its own group."""
        src = peer if recv_peer is None else recv_peer
        for w in self.batch_p2p([("send", send_tensor, peer), ("recv", recv_tensor, src)]):
            w.wait()

    def send(self, tensor: torch.Tensor, dst: int):
        """Blocking send (reference: ``dist.send``, MP.py:140,167).  Between two ranks that exchange in both directions
        every rank must agree on the order (the reference: rank 0 sends then receives, rank 1 receives then sends);
        when both would send first, use ``sendrecv``."""
        if self._on_native(tensor):
            self._native_blocking_p2p([(True, tensor, dst)], f"send to rank {dst}")
            return
        self._need_c10d("send")
        if self._staged(tensor):
            dist.send(tensor.cpu(), dst=self.to_global(dst), group=self.group)
            return
        dist.send(tensor, dst=self.to_global(dst), group=self.group)
        if _SYNC_DEBUG and tensor.is_cuda:
            torch.cuda.synchronize(tensor.device)

    def recv(self, tensor: torch.Tensor, src: int):
        if self._on_native(tensor):
            self._native_blocking_p2p([(False, tensor, src)], f"recv from rank {src}")
            return
        self._need_c10d("recv")
        if self._staged(tensor):
            host = torch.empty(tensor.shape, dtype=tensor.dtype)
            dist.recv(host, src=self.to_global(src), group=self.group)
            tensor.copy_(host)
            return
        dist.recv(tensor, src=self.to_global(src), group=self.group)
        if _SYNC_DEBUG and tensor.is_cuda:
            torch.cuda.synchronize(tensor.device)

    def isend(self, tensor: torch.Tensor, dst: int):
        if self._on_native(tensor):
            return self._native_result(self.native.batch_p2p([(True, tensor, dst)]), True)
        self._need_c10d("isend")
        if self._staged(tensor):
            host = tensor.cpu()
            return _StagedWork(dist.isend(host, dst=self.to_global(dst), group=self.group), host, None)
        return dist.isend(tensor, dst=self.to_global(dst), group=self.group)

    def irecv(self, tensor: torch.Tensor, src: int):
        if self._on_native(tensor):
            return self._native_result(self.native.batch_p2p([(False, tensor, src)]), True)
        self._need_c10d("irecv")
        if self._staged(tensor):
            host = torch.empty(tensor.shape, dtype=tensor.dtype)
            return _StagedWork(dist.irecv(host, src=self.to_global(src), group=self.group), host, tensor)
        return dist.irecv(tensor, src=self.to_global(src), group=self.group)

    def batch_p2p(self, ops: List[tuple]):
        """``ops`` = [("send"|"recv", tensor, peer), ...] issued as one grouped RCCL
        call (ncclGroupStart/End) so a send and a recv to neighbours cannot deadlock.  Returns a list of handles
        (one per op; on the native layer they all stand for the one grouped call)."""
        if not ops:
            return []
        if self._on_native(*[t for _, t, _ in ops]):
            work = self._native_result(self.native.batch_p2p([(kind == "send", t, peer) for kind, t, peer in ops]), True)
            return [work] * len(ops)
        self._need_c10d("batch_p2p")
        if any(self._staged(t) for _, t, _ in ops):       # gloo: non-blocking host transfers, one handle per op
            return [self.isend(t, peer) if kind == "send" else self.irecv(t, peer) for kind, t, peer in ops]
        p2p = []
        for kind, tensor, peer in ops:
            fn = dist.isend if kind == "send" else dist.irecv
            p2p.append(dist.P2POp(fn, tensor, self.to_global(peer), self.group))
        works = dist.batch_isend_irecv(p2p)
        if _SYNC_DEBUG:
            for w in works:
                w.wait()
        return works

    # ---- object helpers (cold path: checkpoint merge, metrics) ----
    def _obj_key(self, what: str) -> str:
        self._obj_seq += 1                    # every rank of the communicator calls the object helpers in the same order
        name = "world" if self._ranks is None else "g" + "_".join(map(str, self._ranks))
        return f"obj/{name}/{what}/{self._obj_seq}"

    def gather_object(self, obj, dst: int = 0):
        if self.world_size == 1:
            return [obj]
        if self._static is not None:
            key = self._obj_key("gather")
            self.store.set_object(f"{key}/{self.rank}", obj, readers=1)          # read once, by dst: then gone
            if self.rank != dst:
                return None
            return [self.store.get_object(f"{key}/{r}") for r in range(self.world_size)]
        out = [None] * self.world_size if self.rank == dst else None
        dist.gather_object(obj, out, dst=self.to_global(dst), group=self.group)
        return out

    def broadcast_object(self, obj, src: int = 0):
        if self.world_size == 1:
            return obj
        if self._static is not None:
            key = self._obj_key("bcast")
            if self.rank == src:
                self.store.set_object(key, obj, readers=self.world_size - 1)
                return obj
            return self.store.get_object(key)
        box = [obj]
        dist.broadcast_object_list(box, src=self.to_global(src), group=self.group)
        return box[0]

    def all_reduce_scalar(self, value: float, op: str = "sum") -> float:
        if self.world_size == 1:
            return float(value)
        rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
        if self._static is not None:
            t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
            self.all_reduce(t, op=rop)
            return float(t.item())
        dev = self.device if (self.device is not None and self._c10d_backend() == "nccl") else torch.device("cpu")
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=rop, group=self.group)
        return float(t.item())

    # ---- sub-groups ----
    def split(self, ranks: Sequence[int]) -> Optional["Communicator"]:
        """Collective over THIS communicator (the world): every rank must call ``split`` with the same
        sequence of rank lists, in the same order.  Returns the communicator when
        this rank is a member, else None.  On the native layer the sub-communicator is an ``ncclCommSplit`` of this
        one (members pass one colour, the others NCCL_SPLIT_NOCOLOR)."""
        ranks = list(ranks)
        member = self.global_rank in ranks
        if self._static is not None:
            sub = Communicator(group=None, device=self.device, ranks=ranks,
                               static=(ranks.index(self.global_rank) if member else 0, len(ranks)), store=self.store)
            sub.attach_native(parent=self, member=member)
            if member and self.peer is not None and len(ranks) > 1:
                sub.attach_peer()
            return sub if member else None
        group = dist.new_group(ranks=ranks, backend=self._c10d_sub_backend)
        sub = Communicator(group=group, device=self.device, ranks=ranks) if member else None
        if sub is not None:
            sub._c10d_sub_backend, sub._timeout_s = self._c10d_sub_backend, self._timeout_s
        if self.native is not None:
            if member:
                sub.attach_native(parent=self, member=True)
            else:                                            # ncclCommSplit is collective over the parent
                Communicator(group=None, device=self.device, ranks=ranks, static=(0, 1)).attach_native(parent=self, member=False)
        if sub is not None and self.peer is not None and len(ranks) > 1:
            sub.attach_peer()                                # collective over the members only (their own control-plane group)
        return sub

    def close(self) -> None:
        """Release the native communicator and the rendezvous store of a c10d-free communicator."""
        if self.peer is not None:
            self.peer.synchronize()
            try:
                self.check_peer()
            finally:
                self.peer = None
        if self.native is not None:
            self.native.synchronize()
            self.native = None
        if self.store is not None and self._ranks is None:
            self.store.close()
            self.store = None


def build_grid(world: Communicator, dp: int, pp: int):
    """DP x PP process grid.  Rank layout: ``rank = replica * pp + stage`` so the
    stages of one pipeline replica are adjacent GPUs.  Returns
    ``(pp_comm, dp_comm, replica, stage)``; ``pp_comm`` spans the stages of this
    replica, ``dp_comm`` spans the replicas of this stage.  Ranks beyond ``dp * pp``
    (idle, like ranks >= 2 in the reference's MP script) get ``(None, None, -1, -1)``.
    Collective over the world: EVERY rank must call it (group creation is global).  When ranks idle, ``world.active``
    is set to a communicator of the working ranks (what end-of-run barriers must use: an idle rank has left by then)."""
    n = world.world_size
    if dp * pp > n:
        raise ValueError(f"dp({dp}) x pp({pp}) > world size {n}")
    me = world.global_rank
    if n == 1:
        return world, world, 0, 0
    active = me < dp * pp
    replica, stage = divmod(me, pp) if active else (-1, -1)
    pp_comm = dp_comm = None
    world.active = world.split(list(range(dp * pp))) if dp * pp < n else None
    for r in range(dp):
        c = world.split([r * pp + s for s in range(pp)])
        if r == replica:
            pp_comm = c
    for s in range(pp):
        c = world.split([r * pp + s for r in range(dp)])
        if s == stage:
            dp_comm = c
    return pp_comm, dp_comm, replica, stage
