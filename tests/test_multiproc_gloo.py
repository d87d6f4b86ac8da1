"""world_size-2 tests of the multi-rank host logic on CPU (gloo).

What runs here is the PRODUCT's multi-GPU host code (catch_amd/parallel.py):
`plan_with_sharding` / `split_universes` (who gets what), `sharded_solve` (the
round loop every rank drives) and `host_exchange` (the all-reduce of the
per-set gain and lost buffers, here over a real gloo process group).  Only the
device is missing, so the shard object behind the loop is a NumPy stand-in with
the interface of engine.Shard (count / claim_check / apply / picks /
buffer_to_host / buffer_from_host); its per-rank arithmetic restates what the
gs_* kernels do on their local rows (test infrastructure, like the oracle).
The result must equal the oracle's sequential greedy, on every rank.
"""
import os
import socket
import time
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ID_BITS, ID_MASK = 32, 0xFFFFFFFF


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class NumpyShard:
    """One rank's universes of an instance, on the CPU.  rows: (set, universe,
    start, end) in this rank's local universe numbering; ulen: their lengths.
    Ownership is per 64-base word, as on the device."""

    def __init__(self, rows, num_sets, ulen, ranks=None):
        self.num_sets = num_sets
        off = np.concatenate([[0], np.cumsum(ulen)]).astype(np.int64)
        self.rows = [(int(s), int(off[u] + a), int(off[u] + b), int(u)) for s, u, a, b in rows]
        self.cov = np.zeros(int(off[-1]) + 64, dtype=bool)
        for _, a, b, _ in self.rows:
            self.cov[a:b] = True
        self.usize = np.array([int(self.cov[off[u]:off[u + 1]].sum()) for u in range(len(ulen))],
                              dtype=np.int64)
        self.uoff = off
        self.rank = np.zeros(num_sets, dtype=np.int64) if ranks is None else np.asarray(ranks)
        self.rank_vals = sorted(set(int(x) for x in self.rank))
        self.cur = 0
        self.picked = np.zeros(num_sets, dtype=bool)
        self.gain = np.zeros(num_sets, dtype=np.int64)
        self.gainbuf = np.zeros(num_sets + 2, dtype=np.uint32)
        self.lostbuf = np.zeros(num_sets, dtype=np.uint8)
        self.accepted = []          # (set, key)
        self.done = 0

    # -- the engine.Shard interface ------------------------------------
    def count(self):
        self.gainbuf[:] = 0
        self.lostbuf[:] = 0
        if self.done:
            return
        for s, a, b, _ in self.rows:
            if not self.picked[s]:
                self.gainbuf[s] += int(self.cov[a:b].sum())
        self.gainbuf[self.num_sets] = int((self.usize > 0).sum())

    def claim_check(self):
        if self.done:
            return
        if self.gainbuf[self.num_sets] == 0:
            self.done = 1
            return
        self.gain = np.where(self.picked, 0, self.gainbuf[:self.num_sets].astype(np.int64))
        cur = self.rank_vals[self.cur]
        self.claimants = [s for s in range(self.num_sets)
                          if self.gain[s] > 0 and self.rank[s] == cur]
        key = {s: (int(self.gain[s]) << ID_BITS) | (ID_MASK - s) for s in self.claimants}
        owner = {}
        mine = set(self.claimants)
        for s, a, b, _ in self.rows:
            if s in mine:
                for w in range(a >> 6, ((b - 1) >> 6) + 1):
                    lo, hi = max(a, w << 6), min(b, (w + 1) << 6)
                    if self.cov[lo:hi].any():
                        owner[w] = max(owner.get(w, 0), key[s])
        for s, a, b, _ in self.rows:
            if s in mine:
                for w in range(a >> 6, ((b - 1) >> 6) + 1):
                    lo, hi = max(a, w << 6), min(b, (w + 1) << 6)
                    if self.cov[lo:hi].any() and owner[w] != key[s]:
                        self.lostbuf[s] = 1

    def apply(self):
        if self.done:
            return self.done
        if not self.claimants:
            self.cur += 1
            if self.cur >= len(self.rank_vals):
                self.done = -1
            return self.done
        for s in self.claimants:
            if self.lostbuf[s]:
                continue
            self.accepted.append((s, (int(self.gain[s]) << ID_BITS) | (ID_MASK - s)))
            self.picked[s] = True
            for ss, a, b, u in self.rows:
                if ss == s:
                    self.usize[u] -= int(self.cov[a:b].sum())
                    self.cov[a:b] = False
        return 0

    def picks(self):
        from catch_amd import parallel
        if self.done == -1:
            raise IndexError("ranks exhausted")
        dense = {v: i for i, v in enumerate(self.rank_vals)}
        return parallel.merge_picks([s for s, _ in self.accepted], [k for _, k in self.accepted],
                                    [dense[int(r)] for r in self.rank])

    def buffer_to_host(self, which):
        return (self.gainbuf if which == 0 else self.lostbuf).copy()

    def buffer_from_host(self, which, arr):
        if which == 0:
            self.gainbuf[:] = np.asarray(arr, dtype=np.uint32)
        else:
            self.lostbuf[:] = np.asarray(arr, dtype=np.uint8)


def _instance(seed, P, U, with_ranks):
    rng = np.random.Generator(np.random.PCG64(seed))
    glen = rng.integers(300, 1500, size=U)
    rows = []
    for s in range(P):
        for u in range(U):
            if rng.random() < 0.6:
                pos = int(rng.integers(0, glen[u] - 260))
                for _ in range(int(rng.integers(1, 3))):
                    ln = int(rng.integers(1, 200))
                    if pos + ln > glen[u]:
                        break
                    rows.append((s, u, pos, pos + ln))
                    pos += ln + int(rng.integers(1, 100))
    ranks = [int(x) for x in rng.integers(0, 3, size=P)] if with_ranks else None
    return sorted(rows), [int(x) for x in glen], ranks


def _worker(rank, world, port, q, transport="gloo"):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from catch_amd import netstore, parallel
    if transport == "gloo":
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from gloo_group import GlooGroup  # (torch stays on the test side: the product rendezvouses over netstore only)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        group = coll = GlooGroup(dist)
    else:
        # the product's own process group (catch_amd.netstore: plain TCP through rank 0, no torch)
        group = coll = netstore.TcpGroup(rank, world, "127.0.0.1", port)
        ok0 = coll.allgather(("r", rank)) == [("r", i) for i in range(world)]
        ok0 = ok0 and coll.broadcast({"x": rank}, src=world - 1) == {"x": world - 1}
        a = np.arange(5, dtype=np.int64) * (rank + 1)
        ok0 = ok0 and coll.allreduce(a, "sum").tolist() == (np.arange(5) * sum(range(1, world + 1))).tolist()
        ok0 = ok0 and coll.allreduce(a, "max").tolist() == (np.arange(5) * world).tolist()
        coll.barrier()
        if not ok0:
            q.put((rank, False))
            return
    from oracle import oracle as orc
    orc.build()
    ok = True
    # ---- level 2: one instance, universes sharded over the ranks ----------
    for seed, P, U, with_ranks in [(1, 40, 6, False), (2, 70, 9, True), (3, 25, 2, False),
                                   (4, 60, 5, True)]:
        rows, glen, ranks = _instance(seed, P, U, with_ranks)
        r = np.array(rows, dtype=np.int64)
        exp = orc.lazy_greedy(r[:, 0], r[:, 1], r[:, 2], r[:, 3], P, glen, None, ranks)
        b = parallel.split_universes(glen, world)
        g0, g1 = b[rank], b[rank + 1]
        local = [(s, u - g0, a, e) for s, u, a, e in rows if g0 <= u < g1]
        shard = NumpyShard(local, P, glen[g0:g1], ranks)
        got = parallel.sharded_solve(
            [shard], lambda which: parallel.host_exchange(group, [shard], which))
        ok = ok and got == exp and len(exp) > 3
    # ---- level 1 + 2 together: a plan over several groups ------------------
    costs = [900, 40, 35, 30, 20, 10]
    sharded, whole = parallel.plan_with_sharding(costs, world)
    ok = ok and sharded == [0] and sorted(i for w in whole for i in w) == [1, 2, 3, 4, 5]
    loads = [sum(costs[i] for i in w) for w in whole]
    ok = ok and max(loads) - min(loads) <= max(costs[1:])
    mine = {}
    for gi in whole[rank]:                           # my whole groups: solved alone
        rows, glen, ranks = _instance(100 + gi, 30, 3, False)
        r = np.array(rows, dtype=np.int64)
        mine[gi] = orc.lazy_greedy(r[:, 0], r[:, 1], r[:, 2], r[:, 3], 30, glen)
    gathered = coll.allgather(mine)
    merged = {}
    for part in gathered:
        merged.update(part)
    ok = ok and sorted(merged) == [1, 2, 3, 4, 5]
    for gi in merged:                                # every rank now holds every group's picks
        rows, glen, _ = _instance(100 + gi, 30, 3, False)
        r = np.array(rows, dtype=np.int64)
        ok = ok and merged[gi] == orc.lazy_greedy(r[:, 0], r[:, 1], r[:, 2], r[:, 3], 30, glen)
    q.put((rank, ok))
    coll.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_solve_and_plan_over_gloo(world):
    """world 2, and world 3 -- where the two-universe instance leaves one rank
    with an EMPTY shard that still has to take part in every exchange."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_solve_and_plan_over_the_tcp_group(world):
    """The same over catch_amd.netstore.TcpGroup -- the transport the product uses since round 4 (rendezvous,
    barriers, host objects, and the fallback exchange of the solver rounds): its collectives, then the sharded
    round loop and the two-level plan."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "tcp")) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def test_plan_helpers_single_process():
    sys.path.insert(0, REPO)
    from catch_amd import parallel
    assert parallel.lpt_assign([5, 9, 1, 7], 2) == [[1, 2], [3, 0]]
    assert parallel.shard_plan([3, 3, 3], 1) == [[0, 1, 2]]
    assert parallel.split_universes([10, 10, 10, 10], 2) == [0, 2, 4]
    assert parallel.split_universes([100], 4) in ([0, 0, 0, 1, 1], [0, 0, 1, 1, 1], [0, 1, 1, 1, 1])
    b = parallel.split_universes([5, 1, 1, 1, 20, 3, 3], 3)
    assert b[0] == 0 and b[-1] == 7 and all(x <= y for x, y in zip(b, b[1:]))
    sharded, whole = parallel.plan_with_sharding([10, 10, 10], 8)
    assert sharded == [0, 1, 2] and all(w == [] for w in whole)
    sharded, whole = parallel.plan_with_sharding([10, 10, 10], 1)
    assert sharded == [] and whole == [[0, 1, 2]]
    # the S4 shape (Mbases per group): two ranks balance without sharding, from three on the 265-Mbase group
    # is sharded and the busiest rank stays within 8 % of an even share
    s4 = [265.4, 37.1, 25.8, 13.4, 27.7, 2.0, 3.2, 43.0, 23.1, 27.4, 8.8, 10.7, 8.8, 10.4, 5.3, 26.5, 18.3, 14.1, 2.3, 18.9]
    for world in (2, 3, 4, 8):
        sharded, whole = parallel.plan_with_sharding(s4, world, min_cost=30)
        assert sharded == ([] if world == 2 else [0])
        assert sorted(i for w in whole for i in w) == [i for i in range(len(s4)) if i not in sharded]
        loads = [sum(s4[i] for i in w) + sum(s4[i] for i in sharded) / world for w in whole]
        assert max(loads) <= 1.08 * sum(s4) / world
    # groups that may not be sharded (fewer genomes than ranks) stay whole however large they are
    sharded, whole = parallel.plan_with_sharding([900, 40, 35], 4, eligible=[False, True, True])
    assert 0 not in sharded and any(0 in w for w in whole)
    # nothing large enough to shard: plain longest-first
    sharded, whole = parallel.plan_with_sharding([50, 40, 5], 2, min_cost=100)
    assert sharded == [] and whole == [[0], [1, 2]]
    assert parallel.merge_picks([7, 3, 9], [50, 90, 70]) == [3, 9, 7]
    assert parallel.merge_picks([7, 3, 9], [50, 90, 70], {7: 0, 3: 1, 9: 0}) == [9, 7, 3]


def test_tcp_group_rejects_strangers_and_bad_ranks(tmp_path):
    """nothing is unpickled before authentication, garbage and oversized length fields
    do not end the rendezvous, ranks outside [1, size) and duplicates are refused, a wrong secret is refused --
    and the real ranks still get through afterwards and their frames are MAC-checked."""
    import pickle
    import socket
    import struct
    import threading

    from catch_amd import netstore

    port = _free_port()
    out = {}

    def root():
        try:
            g = netstore.TcpGroup(0, 3, "127.0.0.1", port, secret="s3cret", timeout=60)
            out["gather"] = g.allgather("r0")
            out["sum"] = g.allreduce(np.arange(4, dtype=np.int64))
            g.close()
        except Exception as exc:                      # pragma: no cover
            out["root_error"] = repr(exc)

    t = threading.Thread(target=root)
    t.start()
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (open, (str(marker), "w"))

    def stranger(payload, read_first=True):
        for _ in range(100):
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                break
            except OSError:
                time.sleep(0.05)
        c.settimeout(5.0)
        try:
            if read_first:
                c.recv(64)
            c.sendall(payload)
            try:
                c.recv(64)
            except OSError:
                pass
        finally:
            c.close()

    evil = pickle.dumps(("catchhip-store-1", "", Evil()))
    stranger(b"GET / HTTP/1.1\r\nHost: x\r\n\r\n")
    stranger(struct.pack("<Q", len(evil)) + evil)                  # round 4's frame format: must not be unpickled
    stranger(struct.pack("<Q", 1 << 62) + b"x" * 64)               # a huge length field must not allocate
    stranger(netstore._MAGIC + struct.pack("<I", 1) + b"n" * 16 + b"m" * 32)    # right shape, wrong MAC
    # right secret, invalid ranks: rank 0 and a rank beyond the job
    for bad_rank in (0, 3, 7):
        with pytest.raises((TimeoutError, ConnectionError, OSError)):
            g = netstore.TcpGroup.__new__(netstore.TcpGroup)
            g.rank, g.size, g._peers, g._root, g._listen = bad_rank, 3, {}, None, None
            g._io_timeout = 5.0
            import hashlib
            key = hashlib.sha256(b"catchhip-store-key|" + b"s3cret" + b"|" + struct.pack("<I", 3)).digest()
            g._connect("127.0.0.1", [port], key, time.time() + 0.5)
    with pytest.raises(TimeoutError):                              # wrong secret
        netstore.TcpGroup(1, 3, "127.0.0.1", port, secret="other", timeout=0.5)
    assert not marker.exists()

    res = {}

    def member(r):
        g = netstore.TcpGroup(r, 3, "127.0.0.1", port, secret="s3cret", timeout=60)
        res[r] = (g.allgather("r%d" % r), g.allreduce(np.arange(4, dtype=np.int64) * (r + 1)))
        if r == 1:                                                  # a duplicate of a connected rank is refused
            with pytest.raises(TimeoutError):
                netstore.TcpGroup(1, 3, "127.0.0.1", port, secret="s3cret", timeout=0.5)
        g.close()

    # rank 1 first, alone, so that its duplicate attempt happens while rank 0 still accepts
    m1 = threading.Thread(target=member, args=(1,))
    m1.start()
    time.sleep(1.5)
    m2 = threading.Thread(target=member, args=(2,))
    m2.start()
    for th in (m1, m2, t):
        th.join(timeout=60)
    assert "root_error" not in out, out
    assert out["gather"] == ["r0", "r1", "r2"]
    assert out["sum"].tolist() == (np.arange(4) * 6).tolist()
    assert res[1][0] == res[2][0] == out["gather"] and res[1][1].tolist() == out["sum"].tolist()


def test_tcp_group_frames_are_authenticated():
    """A frame whose MAC does not hold (tampered, replayed or out of order) raises instead of being unpickled."""
    import socket

    from catch_amd import netstore

    a, b = socket.socketpair()
    key = b"k" * 32
    tx, rx = netstore._Channel(a, key, True), netstore._Channel(b, key, False)
    tx.send({"x": 1})
    assert rx.recv() == {"x": 1}
    tx.send("again")
    tx._tx -= 1                                      # replay the same sequence number
    assert rx.recv() == "again"
    tx.send("replayed")
    with pytest.raises(netstore.StoreAuthError):
        rx.recv()
    a.close(); b.close()


def test_tcp_group_needs_a_secret_off_the_loopback_interface(monkeypatch):
    """without CATCHHIP_STORE_SECRET the key is guessable, so a rendezvous on an address that is not a
    loopback address is refused outright (every rank, before any socket is opened); with a secret, or on 127.0.0.1,
    the constructor goes on as before."""
    from catch_amd import netstore

    monkeypatch.delenv("CATCHHIP_STORE_SECRET", raising=False)
    for rank in (0, 1):
        with pytest.raises(RuntimeError, match="CATCHHIP_STORE_SECRET"):
            netstore.TcpGroup(rank, 2, "192.0.2.7", _free_port(), timeout=0.2)
    # a secret: rank 1 now tries to connect (nobody listens on TEST-NET-1: it times out instead of being refused)
    with pytest.raises((TimeoutError, OSError)):
        netstore.TcpGroup(1, 2, "192.0.2.7", _free_port(), secret="s", timeout=0.3)
    assert netstore.TcpGroup(0, 1, "192.0.2.7").size == 1          # a single process opens nothing


def test_tcp_group_lets_a_rank_replace_its_dead_connection():
    """a rank whose first connection died after rank 0 registered it (it gave up on a slow handshake
    and came back) is admitted again -- rank 0 sees the first socket at EOF -- while a second LIVE connection of a
    rank stays refused (test_tcp_group_rejects_strangers_and_bad_ranks)."""
    import hashlib
    import socket
    import struct
    import threading

    from catch_amd import netstore

    # (with size 2 the accept loop ends at the first admission, so the replacement is exercised on _admit directly)
    ls = socket.socket(); ls.bind(("127.0.0.1", 0)); ls.listen(4)
    a1 = socket.create_connection(ls.getsockname()); s1, _ = ls.accept()
    g0 = netstore.TcpGroup.__new__(netstore.TcpGroup)
    g0.rank, g0.size, g0._peers, g0._root, g0._listen, g0._io_timeout = 0, 2, {}, None, None, 5.0
    g0._peers[1] = netstore._Channel(s1, b"k" * 32, True)
    assert not netstore._is_closed(s1)
    a1.close()                                       # the client gave up
    time.sleep(0.05)
    assert netstore._is_closed(s1)
    # a fresh, authenticated connection of rank 1 now replaces it
    res = {}

    def client():
        g = netstore.TcpGroup.__new__(netstore.TcpGroup)
        g.rank, g.size, g._peers, g._root, g._listen, g._io_timeout = 1, 2, {}, None, None, 5.0
        k = hashlib.sha256(b"catchhip-store-key|" + b"s3cret" + b"|" + struct.pack("<I", 2)).digest()
        g._connect("127.0.0.1", [ls.getsockname()[1]], k, time.time() + 5)
        res["ok"] = g._root is not None
        hold.wait(10)                                # (stays connected while a duplicate is tried)
        g.close()

    hold = threading.Event()
    th = threading.Thread(target=client)
    th.start()
    c2, _ = ls.accept()
    k = hashlib.sha256(b"catchhip-store-key|" + b"s3cret" + b"|" + struct.pack("<I", 2)).digest()
    chan = g0._admit(c2, k)
    time.sleep(0.2)
    assert chan is not None and g0._peers[1] is chan and res.get("ok")
    # ... and a third connection while that one is alive is refused
    th2 = threading.Thread(target=lambda: res.__setitem__("dup", _try_connect(netstore, ls.getsockname()[1])))
    th2.start()
    c3, _ = ls.accept()
    assert g0._admit(c3, k) is None
    c3.close()
    ls.close()                                       # (the refused client's retries now fail at once)
    hold.set()
    th.join(timeout=10)
    th2.join(timeout=15)
    assert res.get("dup") is not True
    g0.close()


def _try_connect(netstore, port):
    import hashlib
    import struct
    g = netstore.TcpGroup.__new__(netstore.TcpGroup)
    g.rank, g.size, g._peers, g._root, g._listen, g._io_timeout = 1, 2, {}, None, None, 5.0
    k = hashlib.sha256(b"catchhip-store-key|" + b"s3cret" + b"|" + struct.pack("<I", 2)).digest()
    try:
        g._connect("127.0.0.1", [port], k, time.time() + 0.5)
    except (TimeoutError, OSError):
        return False
    return True
