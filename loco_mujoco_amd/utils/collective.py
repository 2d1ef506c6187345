"""
The one collective of the path: an all-reduce of a handful of float64 metrics at report time (environments
shard over the GPUs of a node with no data-path exchange). One process per GPU, launched by
``python -m torch.distributed.run`` or anything else that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.

* ``backend="rccl"``: ``ncclAllReduce`` on ``librccl.so`` through ctypes — RCCL over xGMI, one call site, no PyTorch. The
  128-byte ``ncclUniqueId`` travels from rank 0 to the others over a TCP socket on MASTER_ADDR:MASTER_PORT (the same
  rendezvous the reduction of the "tcp" backend uses). Device buffers come from ``hipMalloc`` of the HIP runtime that
  ``liblocohip.so`` links.
* ``backend="tcp"``: the same reduction over the rendezvous sockets on the host — for the multi-rank flow on a box with ONE
  GPU (RCCL refuses two ranks on one device) and for the CPU tests (world size 2, no GPU).

Reference counterpart: none — the reference runs one environment per process and has no collective.
"""

import ctypes as C
import os
import socket
import struct
import time

import numpy as np

SUM, MAX = "sum", "max"
_NCCL_OP = {SUM: 0, MAX: 2}          # ncclSum, ncclMax
_NCCL_FLOAT64 = 8


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return buf


class Collective:
    def __init__(self, backend="rccl", rank=None, world=None, device=None, timeout_s=120.0, require_rccl=False):
        """``require_rccl``: with ``backend="rccl"`` raise on every rank instead of falling back to the sockets."""
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else int(world)
        self.backend = backend if self.world > 1 else "none"
        self._peers, self._server, self._comm = [], None, None
        if self.world == 1:
            return
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(os.environ.get("MASTER_PORT", "29500"))
        # MASTER_PORT itself may be held by the launcher's own store (torch.distributed.run keeps a TCPStore there): the
        # rendezvous of THIS module listens on the first free port of a fixed list derived from it, and both sides check a
        # magic word so that nobody talks to a stranger
        ports = [1024 + (base + 1000 + 7 * k) % 60000 for k in range(8)]
        magic = struct.pack("<II", 0x4C4D4331, base)
        if self.rank == 0:
            srv = None
            for port in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, port))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("no free rendezvous port among %s" % ports)
            srv.listen(self.world)
            srv.settimeout(timeout_s)
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout_s)
                hello = _recv_exact(conn, 12)
                if hello[:8] != magic:
                    conn.close()
                    continue
                conn.sendall(magic)
                peers[struct.unpack("<i", hello[8:])[0]] = conn
            self._peers = [peers[r] for r in range(1, self.world)]
            self._server = srv
        else:
            deadline = time.time() + timeout_s
            s = None
            while s is None:
                for port in ports:
                    try:
                        c = socket.create_connection((addr, port), timeout=2.0)
                        c.settimeout(timeout_s)
                        c.sendall(magic + struct.pack("<i", self.rank))
                        if _recv_exact(c, 8) == magic:
                            s = c
                            break
                        c.close()
                    except (OSError, ConnectionError):
                        pass
                if s is None:
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous with rank 0 failed (%s ports %s)" % (addr, ports))
                    time.sleep(0.05)
            self._peers = [s]
        if self.backend == "rccl":
            # RCCL over xGMI. Should the communicator not come up on some rank (driver / IPC configuration of the node), every
            # rank falls back to the reduction over the rendezvous sockets that already exist — agreed on over those sockets,
            # so that no rank waits inside a collective the others never enter. The bench line reports which one was used.
            dev = int(os.environ.get("LOCAL_RANK", 0)) if device is None else int(device)
            self.backend = "tcp"
            err = None
            uid = (C.c_byte * 128)()
            try:                                         # phase A, local: libraries, device, (rank 0) the unique id
                self._load_libraries()
                self._check_hip(self._hip.hipSetDevice(dev), "hipSetDevice")
                if self.rank == 0:
                    self._check(self._nccl.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
            except Exception as e:                       # noqa: BLE001 - any failure of the optional fast path
                err = e
            if self.rank == 0:                           # the id always travels (zeros after a failure): nobody waits for it
                for p in self._peers:
                    p.sendall(bytes(uid))
            else:
                C.memmove(uid, _recv_exact(self._peers[0], 128), 128)
            agreed = self.all_reduce(np.array([0.0 if err is None else 1.0]), MAX)[0] == 0.0
            if agreed:
                try:                                     # phase B, collective: the communicator
                    self._init_rccl(uid)
                except Exception as e:                   # noqa: BLE001
                    err = e
                agreed = self.all_reduce(np.array([0.0 if err is None else 1.0]), MAX)[0] == 0.0
            if agreed:
                self.backend = "rccl"
            else:
                if self.rank == 0:
                    import sys
                    print("collective: RCCL communicator not available (%s): metric reduction over TCP sockets instead"
                          % (err or "failure on another rank"), file=sys.stderr)
                self._comm = None
                if require_rccl:
                    raise RuntimeError("RCCL was required for the metric reduction and is not available: %s" % (err or "failure on another rank"))

    # ------------------------------------------------------------------ RCCL (ctypes)
    def _load_libraries(self):
        try:                                             # the HIP runtime liblocohip.so has already mapped (RTLD_NOLOAD: no second copy) ...
            self._hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL | getattr(os, "RTLD_NOLOAD", 4))
        except OSError:                                  # ... or, before that library was loaded, the same soname
            self._hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
        self._nccl = C.CDLL("librccl.so")

    def _init_rccl(self, uid):
        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_byte * 128)]
        u = UniqueId()
        C.memmove(C.byref(u), uid, 128)
        comm = C.c_void_p()
        self._nccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        self._check(self._nccl.ncclCommInitRank(C.byref(comm), self.world, u, self.rank), "ncclCommInitRank")
        self._comm = comm
        self._dbuf = C.c_void_p()
        self._cap = 64
        self._check_hip(self._hip.hipMalloc(C.byref(self._dbuf), C.c_size_t(8 * self._cap)), "hipMalloc")
        self._nccl.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

    def _check(self, rc, what):
        if rc != 0:
            self._nccl.ncclGetErrorString.restype = C.c_char_p
            raise RuntimeError("%s failed: %s" % (what, self._nccl.ncclGetErrorString(rc).decode()))

    @staticmethod
    def _check_hip(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with HIP error %d" % (what, rc))

    def _reduce_rccl(self, v, op):
        assert v.size <= self._cap
        nbytes = C.c_size_t(8 * v.size)
        self._check_hip(self._hip.hipMemcpy(self._dbuf, v.ctypes.data_as(C.c_void_p), nbytes, 1), "hipMemcpy H2D")
        # THE call site: RCCL all-reduce over xGMI on the null stream
        self._check(self._nccl.ncclAllReduce(self._dbuf, self._dbuf, v.size, _NCCL_FLOAT64, _NCCL_OP[op], self._comm, None), "ncclAllReduce")
        self._check_hip(self._hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        out = np.empty_like(v)
        self._check_hip(self._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self._dbuf, nbytes, 2), "hipMemcpy D2H")
        return out

    # ------------------------------------------------------------------ the collective
    def all_reduce(self, values, op=SUM):
        """float64 vector, reduced over the ranks; every rank gets the result."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        if self.world == 1:
            return v.copy()
        if self.backend == "rccl":
            return self._reduce_rccl(v, op)
        # host reduction over the rendezvous sockets: gather on rank 0, reduce, send back
        if self.rank == 0:
            parts = [v] + [np.frombuffer(_recv_exact(p, 8 * v.size), dtype=np.float64) for p in self._peers]
            red = np.sum(parts, axis=0) if op == SUM else np.max(parts, axis=0)
            for p in self._peers:
                p.sendall(red.tobytes())
            return red
        self._peers[0].sendall(v.tobytes())
        return np.frombuffer(_recv_exact(self._peers[0], 8 * v.size), dtype=np.float64).copy()

    def barrier(self):
        self.all_reduce(np.zeros(1))

    def comm_count(self):
        """Ranks of the RCCL communicator as RCCL itself counts them (ncclCommCount); None when the reduction does not run on RCCL."""
        if self._comm is None:
            return None
        n = C.c_int(0)
        self._nccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._check(self._nccl.ncclCommCount(self._comm, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def close(self):
        if self._comm is not None:
            self._nccl.ncclCommDestroy.argtypes = [C.c_void_p]
            self._nccl.ncclCommDestroy(self._comm)
            self._hip.hipFree(self._dbuf)
            self._comm = None
        for p in self._peers:
            p.close()
        if self._server is not None:
            self._server.close()
        self._peers, self._server = [], None
