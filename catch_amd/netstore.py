"""A process group over plain TCP -- the plumbing of the multi-GPU paths without torch.

One process per GPU needs, besides the RCCL all-reduces of the data path (catchhip_shard_allreduce), a way to
rendezvous (the RCCL unique id of rank 0 on every rank), barriers around the timed region, and small host objects
passed around (plans, error agreement, the selections of the groups each rank solved).  Rounds 2-3 borrowed
torch.distributed's gloo backend for that; this module does it with sockets so that the product imports no torch
(north_star: "no PyTorch").  The same object is also the FALLBACK transport of the solver rounds' exchanges when
RCCL cannot span the ranks (several ranks on one GPU): all-reduce of numpy arrays through rank 0 -- slow, and only
ever used where nothing faster exists.

Topology: a star.  Rank 0 listens on MASTER_ADDR : CATCHHIP_STORE_PORT (default MASTER_PORT + 1 ... + 16: under
torch.distributed.run the launcher's own store sits on MASTER_PORT), every other rank connects and identifies
itself; a collective = every rank sends its frame to rank 0, rank 0 combines and answers.

Trust (round 5): nothing a stranger sends is ever unpickled, sized by or indexed with.
  * the handshake is FIXED-FORMAT bytes (no pickle, no length field): rank 0 sends magic + a 16-byte nonce; the
    client answers magic + rank (u32) + its own nonce + HMAC-SHA256(secret, "c" | both nonces | rank); rank 0
    checks it with hmac.compare_digest, checks 1 <= rank < size and that the rank is not connected yet, and
    answers HMAC(secret, "s" | ...) so the client knows it reached ITS rank 0;
  * the secret is CATCHHIP_STORE_SECRET when the launcher exports one -- REQUIRED whenever MASTER_ADDR is not a
    loopback address (round 6: the group refuses to start otherwise) --, else the launcher's run id + port, which only
    keeps apart jobs of one user on one host: without an explicit secret rank 0 binds to the loopback interface;
  * after the handshake every frame is  length (u64, capped) | HMAC(session key, direction | sequence number |
    payload) | payload;  the payload (a pickle, protocol 5) is only deserialised once the MAC holds, so only
    authenticated peers are ever unpickled, and a replayed or reordered frame fails;
  * a bad client costs rank 0 one closed socket (every per-client error is caught, the accept loop goes on);
  * collectives carry a generous socket timeout (CATCHHIP_STORE_TIMEOUT, default 1800 s): a live-but-stuck rank
    turns into a TimeoutError on every other rank instead of a hang.
"""
import hashlib
import hmac
import os
import pickle
import socket
import struct
import time

_MAGIC = b"catchhip-store-2"                 # 16 bytes
_NONCE = 16
_MAC = 32
_CLIENT_HELLO = len(_MAGIC) + 4 + _NONCE + _MAC
_MAX_FRAME = int(float(os.environ.get("CATCHHIP_STORE_MAX_FRAME_GB", "8")) * (1 << 30))


class StoreAuthError(ConnectionError):
    pass


def _mac(key, *parts):
    h = hmac.new(key, digestmod=hashlib.sha256)
    for p in parts:
        h.update(p)
    return h.digest()


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


class _Channel:
    """One authenticated connection: MAC-ed, sequence-numbered frames in both directions."""

    def __init__(self, sock, key, i_am_root):
        self.sock, self.key = sock, key
        self._tx_dir, self._rx_dir = (b"R", b"C") if i_am_root else (b"C", b"R")
        self._tx = self._rx = 0

    def send(self, obj):
        data = pickle.dumps(obj, protocol=5)
        tag = _mac(self.key, self._tx_dir, struct.pack("<Q", self._tx), data)
        self._tx += 1
        self.sock.sendall(struct.pack("<Q", len(data)) + tag)
        self.sock.sendall(data)

    def recv(self):
        try:
            head = _recv_exact(self.sock, 8 + _MAC)
            (n,) = struct.unpack_from("<Q", head)
            if n > _MAX_FRAME:
                raise StoreAuthError("catch_amd.netstore: frame of %d bytes exceeds the cap" % n)
            data = _recv_exact(self.sock, n)
        except socket.timeout as exc:
            raise TimeoutError("catch_amd.netstore: a collective timed out (a rank is stuck or gone)") from exc
        want = _mac(self.key, self._rx_dir, struct.pack("<Q", self._rx), data)
        if not hmac.compare_digest(want, bytes(head[8:])):
            raise StoreAuthError("catch_amd.netstore: frame authentication failed")
        self._rx += 1
        return pickle.loads(data)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass


def _is_closed(sock):
    """True when the peer has closed `sock` (a pending EOF or error), without consuming data."""
    import select
    try:
        readable, _, _ = select.select([sock], [], [], 0)
        if not readable:
            return False
        return sock.recv(1, socket.MSG_PEEK | socket.MSG_DONTWAIT) == b""
    except (BlockingIOError, InterruptedError):
        return False
    except OSError:
        return True


def _is_loopback(addr):
    try:
        return socket.gethostbyname(addr).startswith("127.")
    except OSError:
        return addr in ("localhost",)


class TcpGroup:
    """rank / size / barrier / allgather / broadcast / allreduce over TCP through rank 0."""

    def __init__(self, rank, size, addr="127.0.0.1", port=None, token="", timeout=300.0, secret=None):
        self.rank, self.size = int(rank), int(size)
        self._peers = {}          # rank 0: rank -> _Channel
        self._root = None         # others: _Channel to rank 0
        self._listen = None
        if self.size <= 1:
            return
        if not 0 <= self.rank < self.size:
            raise ValueError("catch_amd.netstore: rank %d outside [0, %d)" % (self.rank, self.size))
        ports = [int(port)] if port else []
        if not ports:
            base = int(os.environ.get("MASTER_PORT", "29500"))
            ports = [base + 1 + i for i in range(16)]
        explicit = secret if secret is not None else os.environ.get("CATCHHIP_STORE_SECRET")
        weak = token or (os.environ.get("TORCHELASTIC_RUN_ID", "") + ":" + os.environ.get("MASTER_PORT", ""))
        if not explicit and not _is_loopback(addr):
            #  the derived key is guessable -- torchrun's default run id is the literal "none" -- and a
            # peer that passes the handshake sends frames that are unpickled: off the loopback interface that is
            # remote code execution for anyone who can reach the port.  No secret, no multi-node rendezvous.
            raise RuntimeError("catch_amd.netstore: MASTER_ADDR %r is not a loopback address; set CATCHHIP_STORE_SECRET "
                               "(the same value on every rank) to run the rendezvous across hosts" % (addr,))
        key = hashlib.sha256(b"catchhip-store-key|" + (explicit if explicit else weak).encode()
                             + b"|" + struct.pack("<I", self.size)).digest()
        self._io_timeout = float(os.environ.get("CATCHHIP_STORE_TIMEOUT", "1800"))
        deadline = time.time() + timeout
        if self.rank == 0:
            self._serve(addr, ports, key, deadline, loopback_only=not explicit and _is_loopback(addr))
        else:
            self._connect(addr, ports, key, deadline)

    # -- rank 0 ---------------------------------------------------------------------------------------------
    def _serve(self, addr, ports, key, deadline, loopback_only):
        bind_addr = "127.0.0.1" if loopback_only or addr == "localhost" else addr
        err = None
        for p in ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((bind_addr, p))
            except OSError as exc:
                err = exc
                s.close()
                continue
            self._listen = s
            break
        if self._listen is None:
            raise OSError("catch_amd.netstore: no free port among %s (%s)" % (ports, err))
        self._listen.listen(max(self.size, 16))
        self._listen.settimeout(1.0)
        while len(self._peers) < self.size - 1:
            if time.time() > deadline:
                raise TimeoutError("catch_amd.netstore: %d of %d ranks connected" % (len(self._peers) + 1, self.size))
            try:
                c, _ = self._listen.accept()
            except socket.timeout:
                continue
            except OSError:
                continue
            try:
                chan = self._admit(c, key)
            except Exception:                     # whatever a stranger provokes costs one socket, not the rendezvous
                chan = None
            if chan is None:
                try:
                    c.close()
                except OSError:
                    pass

    def _admit(self, c, key):
        """Fixed-format challenge/response; returns the channel of a NEW valid rank or None."""
        c.settimeout(2.0)      # (a silent stranger stalls the serial accept loop this long, not 10 s)
        nonce_s = os.urandom(_NONCE)
        c.sendall(_MAGIC + nonce_s)
        hello = bytes(_recv_exact(c, _CLIENT_HELLO))
        if hello[:len(_MAGIC)] != _MAGIC:
            return None
        off = len(_MAGIC)
        rank_b = hello[off:off + 4]
        nonce_c = hello[off + 4:off + 4 + _NONCE]
        tag = hello[off + 4 + _NONCE:]
        if not hmac.compare_digest(_mac(key, b"c", nonce_s, nonce_c, rank_b), tag):
            return None
        (r,) = struct.unpack("<I", rank_b)
        if not 1 <= r < self.size:
            return None                           # rank 0, or a rank beyond the job
        if r in self._peers:
            # connected already -- unless that connection is dead (the client gave up on a slow handshake and came
            # back: its first socket then reads EOF): an authenticated retry replaces a dead channel
            if not _is_closed(self._peers[r].sock):
                return None
            self._peers.pop(r).close()
        c.sendall(_mac(key, b"s", nonce_c, nonce_s, rank_b))
        c.settimeout(self._io_timeout)
        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        chan = _Channel(c, _mac(key, b"k", nonce_s, nonce_c, rank_b), i_am_root=True)
        self._peers[r] = chan
        return chan

    # -- the other ranks ------------------------------------------------------------------------------------
    def _connect(self, addr, ports, key, deadline):
        rank_b = struct.pack("<I", self.rank)
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
                    first = bytes(_recv_exact(c, len(_MAGIC) + _NONCE))
                    if first[:len(_MAGIC)] == _MAGIC:
                        nonce_s = first[len(_MAGIC):]
                        nonce_c = os.urandom(_NONCE)
                        c.sendall(_MAGIC + rank_b + nonce_c + _mac(key, b"c", nonce_s, nonce_c, rank_b))
                        ack = bytes(_recv_exact(c, _MAC))
                        if hmac.compare_digest(_mac(key, b"s", nonce_c, nonce_s, rank_b), ack):
                            c.settimeout(self._io_timeout)
                            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self._root = _Channel(c, _mac(key, b"k", nonce_s, nonce_c, rank_b), i_am_root=False)
                            break
                except (OSError, ConnectionError):
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
                parts[r] = self._peers[r].recv()
            out = combine(parts)
            for r in range(1, self.size):
                self._peers[r].send(out)
            return out
        self._root.send(payload)
        return self._root.recv()

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
