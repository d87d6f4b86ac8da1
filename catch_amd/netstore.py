"""A process group over plain TCP -- the plumbing of the multi-GPU paths without torch.

One process per GPU needs, besides the RCCL all-reduces of the data path (catchhip_shard_allreduce), a way to
rendezvous (the RCCL unique id of rank 0 on every rank), barriers around the timed region, and small host objects
passed around (plans, error agreement, the selections of the groups each rank solved).  Rounds 2-3 borrowed
torch.distributed's gloo backend for that; this module does it in ~150 lines of sockets and pickle so that the
product imports no torch (north_star: "no PyTorch").  The same object is also the FALLBACK transport of the solver
rounds' exchanges when RCCL cannot span the ranks (several ranks on one GPU): all-reduce of numpy arrays through
rank 0 -- slow, and only ever used where nothing faster exists.

Topology: a star.  Rank 0 listens on MASTER_ADDR : CATCHHIP_STORE_PORT (default MASTER_PORT + 1 ... + 16: under
torch.distributed.run the launcher's own store sits on MASTER_PORT), every other rank connects and identifies
itself; a collective = every rank sends its frame to rank 0, rank 0 combines and answers.  Frames are
length-prefixed pickles (protocol 5: numpy buffers travel without a copy on the sending side).
"""
import os
import pickle
import socket
import struct
import time

_HELLO = b"catchhip-store-1"


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=5)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("catch_amd.netstore: peer closed the connection")
        got += k
    return buf


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


class TcpGroup:
    """rank / size / barrier / allgather / broadcast / allreduce over TCP through rank 0."""

    def __init__(self, rank, size, addr="127.0.0.1", port=None, token="", timeout=300.0):
        self.rank, self.size = int(rank), int(size)
        self._peers = {}          # rank 0: rank -> socket
        self._root = None         # others: socket to rank 0
        self._listen = None
        if self.size <= 1:
            return
        ports = [int(port)] if port else []
        if not ports:
            base = int(os.environ.get("MASTER_PORT", "29500"))
            ports = [base + 1 + i for i in range(16)]
        token = (token or os.environ.get("TORCHELASTIC_RUN_ID", "") + ":" + os.environ.get("MASTER_PORT", "")).encode()
        deadline = time.time() + timeout
        if self.rank == 0:
            err = None
            for p in ports:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((addr if addr not in ("localhost",) else "127.0.0.1", p))
                except OSError as exc:
                    err = exc
                    s.close()
                    continue
                self._listen = s
                break
            if self._listen is None:
                raise OSError("catch_amd.netstore: no free port among %s (%s)" % (ports, err))
            self._listen.listen(self.size)
            self._listen.settimeout(1.0)
            while len(self._peers) < self.size - 1:
                if time.time() > deadline:
                    raise TimeoutError("catch_amd.netstore: %d of %d ranks connected" % (len(self._peers) + 1, self.size))
                try:
                    c, _ = self._listen.accept()
                except socket.timeout:
                    continue
                try:
                    c.settimeout(10.0)
                    hello = _recv(c)
                    if not (isinstance(hello, tuple) and hello[0] == _HELLO and hello[1] == token):
                        c.close()            # somebody else's client
                        continue
                    _send(c, (_HELLO, token))
                    c.settimeout(None)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._peers[int(hello[2])] = c
                except (OSError, pickle.UnpicklingError, EOFError, struct.error):
                    c.close()
        else:
            while self._root is None:
                if time.time() > deadline:
                    raise TimeoutError("catch_amd.netstore: rank %d could not reach rank 0 at %s:%s" % (self.rank, addr, ports))
                for p in ports:
                    try:
                        c = socket.create_connection((addr, p), timeout=2.0)
                    except OSError:
                        continue
                    try:
                        c.settimeout(10.0)
                        _send(c, (_HELLO, token, self.rank))
                        ack = _recv(c)
                        if isinstance(ack, tuple) and ack[0] == _HELLO and ack[1] == token:
                            c.settimeout(None)
                            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self._root = c
                            break
                    except (OSError, pickle.UnpicklingError, EOFError, struct.error, ConnectionError):
                        pass
                    c.close()
                if self._root is None:
                    time.sleep(0.05)

    # -- one collective: every rank's payload to rank 0, f(list of payloads) back to every rank -----------------
    def _collective(self, payload, combine):
        if self.size <= 1:
            return combine([payload])
        if self.rank == 0:
            parts = [payload] + [None] * (self.size - 1)
            for r in range(1, self.size):
                parts[r] = _recv(self._peers[r])
            out = combine(parts)
            for r in range(1, self.size):
                _send(self._peers[r], out)
            return out
        _send(self._root, payload)
        return _recv(self._root)

    def allgather(self, obj):
        """Every rank's object, in rank order."""
        return self._collective(obj, lambda parts: list(parts))

    def broadcast(self, obj, src=0):
        return self._collective(obj, lambda parts: parts[src])

    def barrier(self):
        self._collective(None, lambda parts: None)

    def allreduce(self, arr, op="sum"):
        """Element-wise sum / max of equally shaped numpy arrays over the ranks."""
        import numpy as np

        def combine(parts):
            acc = np.array(parts[0], copy=True)
            for b in parts[1:]:
                acc = acc + b if op == "sum" else np.maximum(acc, b)
            return acc
        return self._collective(arr, combine)

    def close(self):
        for c in list(self._peers.values()) + [self._root, self._listen]:
            if c is not None:
                try:
                    c.close()
                except OSError:
                    pass
        self._peers, self._root, self._listen = {}, None, None


class GlooGroup:
    """The same interface over torch.distributed (gloo): the second transport -- CATCHHIP_RENDEZVOUS=gloo, and
    what tests/test_multiproc_gloo.py drives the product's helpers with."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()

    def allgather(self, obj):
        out = [None] * self.size
        self.dist.all_gather_object(out, obj)
        return out

    def broadcast(self, obj, src=0):
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self):
        self.dist.barrier()

    def allreduce(self, arr, op="sum"):
        import numpy as np
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        return t.numpy()

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
