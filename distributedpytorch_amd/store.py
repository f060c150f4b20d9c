"""Rendezvous key-value store over TCP: the piece of c10d's ``TCPStore`` this framework needs, on plain sockets.

The reference rendezvouses through ``init_method="tcp://MASTER:PORT"`` (multinode-DP/multinodeDP.py:161,
multinode-MP/multinodeMP.py:177).  The native RCCL communicator (csrc/rccl_comm.cpp) only needs one thing from
a rendezvous: rank 0's 128-byte ``ncclUniqueId`` must reach every rank.  This store does that without c10d:
rank 0 runs a tiny server thread, every rank (rank 0 included) is a client.

Protocol: one request per line of JSON, values base64 - ``set`` / ``get`` (blocks until the key exists or the
timeout expires) / ``add`` (atomic counter, returns the new value) / ``wait`` (several keys).  Cold path only
(bootstrap, barriers of the bootstrap itself, small objects); tensors never go through it.
"""
from __future__ import annotations

import base64
import json
import pickle
import socket
import socketserver
import threading
import time
from typing import Any, Dict, List, Optional, Tuple


class _State:
    def __init__(self):
        self.data: Dict[str, bytes] = {}
        self.cond = threading.Condition()


class _Handler(socketserver.StreamRequestHandler):
    def handle(self):
        st: _State = self.server.state          # type: ignore[attr-defined]
        for line in self.rfile:
            try:
                req = json.loads(line.decode())
            except Exception:  # noqa: BLE001 - a broken client must not take the store down
                return
            op, key = req.get("op"), req.get("key", "")
            resp: Dict[str, Any] = {"ok": True}
            if op == "set":
                with st.cond:
                    st.data[key] = base64.b64decode(req["value"])
                    st.cond.notify_all()
            elif op == "get":
                deadline = time.monotonic() + float(req.get("timeout", 300.0))
                with st.cond:
                    while key not in st.data:
                        left = deadline - time.monotonic()
                        if left <= 0:
                            break
                        st.cond.wait(min(left, 1.0))
                    if key in st.data:
                        resp["value"] = base64.b64encode(st.data[key]).decode()
                    else:
                        resp = {"ok": False, "error": f"timeout waiting for key {key!r}"}
            elif op == "add":
                with st.cond:
                    cur = int(st.data.get(key, b"0").decode()) + int(req["amount"])
                    st.data[key] = str(cur).encode()
                    st.cond.notify_all()
                    resp["value"] = cur
            elif op == "delete":
                with st.cond:
                    st.data.pop(key, None)
            else:
                resp = {"ok": False, "error": f"unknown op {op!r}"}
            self.wfile.write((json.dumps(resp) + "\n").encode())
            self.wfile.flush()


class _Server(socketserver.ThreadingTCPServer):
    allow_reuse_address = True
    daemon_threads = True


class TCPKVStore:
    """``TCPKVStore(host, port, is_master)``.  ``is_master`` starts the server (a daemon thread) first."""

    def __init__(self, host: str, port: int, is_master: bool, timeout_s: float = 300.0):
        self.host, self.port, self.timeout_s = host, int(port), float(timeout_s)
        self._server: Optional[_Server] = None
        if is_master:
            self._server = _Server((host if host not in ("localhost",) else "127.0.0.1", self.port), _Handler)
            self._server.state = _State()        # type: ignore[attr-defined]
            self.port = self._server.server_address[1]           # port 0 = pick a free one
            threading.Thread(target=self._server.serve_forever, name="dpa-store", daemon=True).start()
        self._sock: Optional[socket.socket] = None
        self._file = None
        self._lock = threading.Lock()
        self._connect()

    def _connect(self) -> None:
        deadline = time.monotonic() + self.timeout_s
        last: Optional[Exception] = None
        while time.monotonic() < deadline:
            try:
                self._sock = socket.create_connection((self.host, self.port), timeout=self.timeout_s)
                self._sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._file = self._sock.makefile("rwb")
                return
            except OSError as e:         # the master may not be listening yet
                last = e
                time.sleep(0.05)
        raise TimeoutError(f"cannot reach the rendezvous store at {self.host}:{self.port}: {last}")

    def _call(self, **req) -> Dict[str, Any]:
        with self._lock:
            self._file.write((json.dumps(req) + "\n").encode())
            self._file.flush()
            line = self._file.readline()
        if not line:
            raise ConnectionError("rendezvous store closed the connection")
        resp = json.loads(line.decode())
        if not resp.get("ok"):
            raise TimeoutError(resp.get("error", "store error"))
        return resp

    # ---- raw bytes ----
    def set(self, key: str, value: bytes) -> None:
        self._call(op="set", key=key, value=base64.b64encode(bytes(value)).decode())

    def get(self, key: str, timeout_s: Optional[float] = None) -> bytes:
        t = self.timeout_s if timeout_s is None else timeout_s
        return base64.b64decode(self._call(op="get", key=key, timeout=t)["value"])

    def add(self, key: str, amount: int = 1) -> int:
        return int(self._call(op="add", key=key, amount=int(amount))["value"])

    def delete(self, key: str) -> None:
        self._call(op="delete", key=key)

    def wait(self, keys: List[str], timeout_s: Optional[float] = None) -> None:
        for k in keys:
            self.get(k, timeout_s)

    # ---- small Python objects (cold path: metrics, checkpoint merge) ----
    def set_object(self, key: str, obj: Any) -> None:
        self.set(key, pickle.dumps(obj))

    def get_object(self, key: str, timeout_s: Optional[float] = None) -> Any:
        return pickle.loads(self.get(key, timeout_s))      # only ever holds what this job's own ranks wrote

    def barrier(self, name: str, world_size: int, timeout_s: Optional[float] = None) -> None:
        """Every rank adds one to ``name``; the last arrival releases the others."""
        if self.add(f"{name}/count", 1) == world_size:
            self.set(f"{name}/done", b"1")
        self.get(f"{name}/done", timeout_s)

    def close(self) -> None:
        try:
            if self._file is not None:
                self._file.close()
            if self._sock is not None:
                self._sock.close()
        finally:
            self._file = self._sock = None
            if self._server is not None:
                self._server.shutdown()
                self._server.server_close()
                self._server = None


def parse_tcp_url(url: str) -> Tuple[str, int]:
    """``tcp://host:port`` (the reference's ``--dist-url``) -> (host, port)."""
    hostport = url.split("://", 1)[-1]
    host, port = hostport.rsplit(":", 1)
    return host, int(port)
