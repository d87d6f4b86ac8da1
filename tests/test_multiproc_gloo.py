"""world_size-2 test of the multi-rank host logic on CPU (gloo).

The data path of the N>1 bench shards whole groups across ranks with no
collective; what the ranks exchange is the timing/units reduction.  The probe-
sharded variant's arithmetic (packed (gain, ~id) keys, MAX all-reduce, sets
s % nranks == rank) is checked here with the oracle's gains standing in for
the gain kernel (tests may use the oracle; the product never does)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ID_BITS, ID_MASK = 24, 0xFFFFFF


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _greedy_sharded(rank, world, rows, P, U, glen):
    """Probe-sharded greedy with a MAX all-reduce of the packed key per pick."""
    import torch
    base = np.concatenate([[0], np.cumsum(glen)])
    cov = np.zeros(int(base[-1]), dtype=bool)
    for s, u, a, b in rows:
        cov[base[u] + a:base[u] + b] = True
    left = np.array([cov[base[u]:base[u + 1]].sum() for u in range(U)])
    picked, picks = set(), []
    while (left > 0).any():
        best = 0
        for s in range(rank, P, world):
            if s in picked:
                continue
            g = 0
            for u in range(U):
                c = sum(int(cov[base[u] + a:base[u] + b].sum())
                        for ss, uu, a, b in rows if ss == s and uu == u)
                g += min(int(left[u]), c)
            if g > 0:
                best = max(best, (g << ID_BITS) | (ID_MASK - s))
        t = torch.tensor([best], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        key = int(t[0])
        assert key >> ID_BITS > 0
        s = ID_MASK - (key & ID_MASK)
        picked.add(s)
        picks.append(s)
        for ss, u, a, b in rows:
            if ss == s:
                cov[base[u] + a:base[u] + b] = False
        left = np.array([cov[base[u]:base[u + 1]].sum() for u in range(U)])
    return picks


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    from oracle import oracle as orc
    rng = np.random.Generator(np.random.PCG64(11))
    P, U = 24, 3
    glen = [120, 90, 150]
    rows = []
    for s in range(P):
        for u in range(U):
            if rng.random() < 0.5:
                a = int(rng.integers(0, glen[u] - 30))
                rows.append((s, u, a, a + int(rng.integers(5, 30))))
    rows.sort()
    picks = _greedy_sharded(rank, world, rows, P, U, glen)
    r = np.array(rows)
    exp = orc.approx_multiuniverse(r[:, 0], r[:, 1], r[:, 2], r[:, 3], P, U)
    # weak-scaling bookkeeping of bench.py: max time over ranks, sum of units
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([10.0 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.barrier()
    q.put((rank, picks, exp, float(t[0]), float(u[0])))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, picks, exp, tmax, usum in res:
        assert picks == exp           # same picks, same order, on every rank
        assert tmax == 2.0 and usum == 30.0
