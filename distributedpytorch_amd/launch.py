"""Per-GPU process launcher with the reference's node-topology rank math.

Reference behaviour being reproduced (multinode-DP/multinodeDP.py:164-183,
multinode-MP/multinodeMP.py:180-199): one launcher invocation per node,
``--GPU-list a,b,c`` gives the number of processes of every node,
``--node-index i`` selects this node, and

    local_size = gpu_list[i]; world_size = sum(gpu_list)
    offset = sum(gpu_list[:i]); global_rank = offset + local_rank

What is new: the children are started with the *spawn* start method (HIP cannot be
re-initialised after fork), the config travels as an argument instead of a module
global, and the parent monitors the children: when one dies, the rest are
terminated and the launcher raises instead of hanging in ``join()`` forever
(reference DP.py:182-183 has no failure handling).
"""
from __future__ import annotations

import os
import signal
import sys
import time
import traceback
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch.multiprocessing as mp


@dataclass(frozen=True)
class Topology:
    """Result of the rank math for one node."""
    gpu_list: tuple
    node_index: int

    @property
    def local_size(self) -> int:
        return self.gpu_list[self.node_index]

    @property
    def world_size(self) -> int:
        return sum(self.gpu_list)

    @property
    def offset(self) -> int:
        return sum(self.gpu_list[: self.node_index])

    def global_rank(self, local_rank: int) -> int:
        if not 0 <= local_rank < self.local_size:
            raise ValueError(f"local_rank {local_rank} outside node of size {self.local_size}")
        return self.offset + local_rank

    def local_ranks(self) -> List[int]:
        return list(range(self.local_size))

    def node_of(self, global_rank: int) -> int:
        acc = 0
        for node, n in enumerate(self.gpu_list):
            if global_rank < acc + n:
                return node
            acc += n
        raise ValueError(f"rank {global_rank} outside world of {self.world_size}")


def topology(gpu_list: Sequence[int], node_index: int) -> Topology:
    gpu_list = tuple(int(v) for v in gpu_list)
    if not 0 <= node_index < len(gpu_list):
        raise ValueError(f"node_index {node_index} outside GPU list {gpu_list}")
    return Topology(gpu_list, node_index)


class ChildFailed(RuntimeError):
    def __init__(self, local_rank: int, exitcode: int, tb: Optional[str] = None):
        self.local_rank, self.exitcode, self.tb = local_rank, exitcode, tb
        msg = f"process local_rank={local_rank} exited with code {exitcode}"
        if tb:
            msg += "\n" + tb
        super().__init__(msg)


def _child_entry(fn, local_rank, global_rank, world_size, args, err_queue):
    # Keep IPC in dmabuf mode for RCCL / tensor sharing on this driver stack.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:       # compute / side / RCCL streams on separate hardware queues
        os.environ["GPU_MAX_HW_QUEUES"] = "8"                   # (comm.rccl_env_defaults); before the child's first HIP call
    os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["RANK"] = str(global_rank)
    os.environ["WORLD_SIZE"] = str(world_size)
    try:
        fn(local_rank, global_rank, world_size, *args)
    except KeyboardInterrupt:
        sys.exit(130)
    except BaseException:  # noqa: BLE001 - report everything to the parent
        err_queue.put((local_rank, traceback.format_exc()))
        sys.exit(1)


def spawn(fn: Callable, gpu_list: Sequence[int] = (1,), node_index: int = 0, args: tuple = (),
          join: bool = True, poll_s: float = 0.05, start_method: str = "spawn"):
    """Start ``gpu_list[node_index]`` processes running
    ``fn(local_rank, global_rank, world_size, *args)``.

    Equivalent of the reference's ``Process(...).start()`` loop (DP.py:176-183) with
    ``mp.spawn``-style failure propagation.  Returns the context when ``join`` is
    False (call ``.join()`` on it).
    """
    topo = topology(gpu_list, node_index)
    if len(gpu_list) == 1 and "DPA_STORE_TOKEN" not in os.environ:
        # one node: the job's ranks are exactly this launcher's children, so a random per-job secret for the rendezvous
        # store (store.py: requests without it are dropped) can be drawn here and inherited.  Several nodes are started by
        # hand (multinode-DP/README.md:15-17): export the same DPA_STORE_TOKEN on each of them.
        import secrets
        os.environ["DPA_STORE_TOKEN"] = secrets.token_hex(16)
    ctx = mp.get_context(start_method)
    err_queue = ctx.SimpleQueue()
    procs = []
    for local_rank in topo.local_ranks():
        p = ctx.Process(target=_child_entry,
                        args=(fn, local_rank, topo.global_rank(local_rank), topo.world_size, args, err_queue),
                        daemon=False)
        p.start()
        procs.append(p)
    handle = SpawnContext(procs, err_queue, poll_s)
    if join:
        handle.join()
        return None
    return handle


class SpawnContext:
    def __init__(self, procs, err_queue, poll_s):
        self.procs, self.err_queue, self.poll_s = procs, err_queue, poll_s

    def pids(self):
        return [p.pid for p in self.procs]

    def _terminate_all(self):
        for p in self.procs:
            if p.is_alive():
                p.terminate()
        deadline = time.time() + 5.0
        for p in self.procs:
            p.join(max(0.0, deadline - time.time()))
        for p in self.procs:
            if p.is_alive():
                os.kill(p.pid, signal.SIGKILL)
                p.join()

    def join(self, timeout: Optional[float] = None):
        """Wait for all children; on the first failure kill the others and raise."""
        t0 = time.time()
        while True:
            alive = False
            for lr, p in enumerate(self.procs):
                if p.exitcode is None:
                    alive = True
                elif p.exitcode != 0:
                    tb = None
                    # a short grace period lets the traceback arrive
                    t_wait = time.time() + 0.5
                    while time.time() < t_wait and self.err_queue.empty():
                        time.sleep(0.01)
                    pending = []
                    while not self.err_queue.empty():
                        pending.append(self.err_queue.get())
                    for rank_, tb_ in pending:
                        if rank_ == lr:
                            tb = tb_
                    if tb is None and pending:
                        lr, tb = pending[0]
                    self._terminate_all()
                    raise ChildFailed(lr, p.exitcode, tb)
            if not alive:
                return True
            if timeout is not None and time.time() - t0 > timeout:
                self._terminate_all()
                raise TimeoutError(f"children still running after {timeout}s; terminated")
            time.sleep(self.poll_s)
