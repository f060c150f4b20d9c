"""Rendezvous key-value store over TCP: the piece of c10d's ``TCPStore`` this framework needs, on plain sockets.

The reference rendezvouses through ``init_method="tcp://MASTER:PORT"`` (multinode-DP/multinodeDP.py:161,
multinode-MP/multinodeMP.py:177).  The native RCCL communicator (csrc/rccl_comm.cpp) only needs one thing from
a rendezvous: rank 0's 128-byte ``ncclUniqueId`` must reach every rank.  This store does that without c10d:
rank 0 runs a tiny server thread, every rank (rank 0 included) is a client.

Protocol: one request per line of JSON, values base64 - ``set`` / ``get`` (blocks until the key exists or the
timeout expires) / ``add`` (atomic counter, returns the new value) / ``wait`` (several keys).  Cold path only
(bootstrap, barriers of the bootstrap itself, small objects); tensors never go through it.

Hygiene: every request carries the job's token (``DPA_STORE_TOKEN``; ``launch.spawn`` draws a random one
for its children) and the server drops connections that present another - ``get_object`` unpickles, so nobody outside
the job may plant a key; a value set with ``readers=n`` is deleted once it has been read n times (object gathers and
broadcasts leave nothing behind); barriers number themselves, so a name can be used again and the previous round's
keys are removed; the client socket's timeout is the request's own plus a margin, so a timed-out ``get`` surfaces as
the store's ``TimeoutError`` and not as ``socket.timeout``.
"""
from __future__ import annotations

import base64
import hmac
import json
import os
import pickle
import socket
import socketserver
import threading
import time
from typing import Any, Dict, List, Optional, Tuple


def job_token() -> str:
    """The shared secret of this job's ranks ("" = none configured: requests are not authenticated)."""
    return os.environ.get("DPA_STORE_TOKEN", "")


class _State:
    def __init__(self, token: str = ""):
        self.data: Dict[str, bytes] = {}
        self.readers: Dict[str, int] = {}       # key -> reads left before the server deletes it
        self.cond = threading.Condition()
        self.token = token


class _Handler(socketserver.StreamRequestHandler):
    def handle(self):
        st: _State = self.server.state          # type: ignore[attr-defined]
        for line in self.rfile:
            try:
                req = json.loads(line.decode())
            except Exception:  # noqa: BLE001 - a broken client must not take the store down
                return
            op, key = req.get("op"), req.get("key", "")
            if st.token and not hmac.compare_digest(str(req.get("token", "")), st.token):
                return                            # not one of this job's ranks: drop the connection, answer nothing
            resp: Dict[str, Any] = {"ok": True}
            if op == "set":
                with st.cond:
                    st.data[key] = base64.b64decode(req["value"])
                    n = int(req.get("readers", 0))
                    if n > 0:
                        st.readers[key] = n
                    else:
                        st.readers.pop(key, None)
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
                        if key in st.readers:     # the last expected reader takes the value with it
                            st.readers[key] -= 1
                            if st.readers[key] <= 0:
                                del st.readers[key]
                                del st.data[key]
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
                    st.readers.pop(key, None)
            elif op == "count":                   # tests / diagnostics: how many keys the store holds
                with st.cond:
                    resp["value"] = len(st.data)
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
        self._token = job_token()
        self._barrier_seq: Dict[str, int] = {}
        if is_master:
            self._server = _Server((host if host not in ("localhost",) else "127.0.0.1", self.port), _Handler)
            self._server.state = _State(self._token)        # type: ignore[attr-defined]
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
        if self._token:
            req["token"] = self._token
        with self._lock:
            # the server answers a blocking get within the request's own timeout: wait a little longer than that
            self._sock.settimeout(float(req.get("timeout", self.timeout_s)) + 15.0)
            self._file.write((json.dumps(req) + "\n").encode())
            self._file.flush()
            line = self._file.readline()
        if not line:
            raise ConnectionError("rendezvous store closed the connection (wrong DPA_STORE_TOKEN, or the master left)")
        resp = json.loads(line.decode())
        if not resp.get("ok"):
            raise TimeoutError(resp.get("error", "store error"))
        return resp

    # ---- raw bytes ----
    def set(self, key: str, value: bytes, readers: int = 0) -> None:
        """``readers`` > 0: the server deletes the key once it has been read that many times."""
        self._call(op="set", key=key, value=base64.b64encode(bytes(value)).decode(), readers=int(readers))

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
    def set_object(self, key: str, obj: Any, readers: int = 0) -> None:
        self.set(key, pickle.dumps(obj), readers)

    def num_keys(self) -> int:
        return int(self._call(op="count")["value"])

    def get_object(self, key: str, timeout_s: Optional[float] = None) -> Any:
        return pickle.loads(self.get(key, timeout_s))      # only ever holds what this job's own ranks wrote

    def barrier(self, name: str, world_size: int, timeout_s: Optional[float] = None) -> None:
        """Every rank adds one to ``name``; the last arrival releases the others.  A name may be used again: each use
        has its own round number (every rank calls the barriers of a name in the same order), and the last arrival of
        round k removes round k - 1's keys - everyone has left that round, or it could not have arrived here."""
        k = self._barrier_seq.get(name, 0)
        self._barrier_seq[name] = k + 1
        if self.add(f"{name}/{k}/count", 1) == world_size:
            if k > 0:
                self.delete(f"{name}/{k - 1}/count")
                self.delete(f"{name}/{k - 1}/done")
            self.set(f"{name}/{k}/done", b"1")
        self.get(f"{name}/{k}/done", timeout_s)

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
