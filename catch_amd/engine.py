"""Thin Python objects over the C-ABI handles of libcatchhip.so.

Only marshals NumPy host buffers in and out (stdlib + NumPy, no torch); all
compute happens in the HIP kernels behind include/catchhip.h.
"""
import ctypes

import numpy as np

from catch_amd import _lib
from catch_amd._lib import (c_f32p, c_f64p, c_i32p, c_i64p, c_u16p, c_u32p,
                            c_u64p, c_u8p, check)

SCAN_AUTO, SCAN_GENERAL, SCAN_FAST, SCAN_SEED = 0, 1, 2, 3
PHASE_SCAN, PHASE_ROWS, PHASE_GREEDY, PHASE_NDF, PHASE_GREEDY_ROUNDS = 0, 1, 2, 3, 4
PHASE_VERIFY = 5
PHASE_CLAIM = 6
PHASE_VCOUNT = 7      # key-grouped join: the counting pass (PHASE_VERIFY: the writing pass)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _concat(strs):
    off = np.zeros(len(strs) + 1, dtype=np.int64)
    if strs:
        np.cumsum([len(s) for s in strs], out=off[1:])
    buf = np.frombuffer("".join(strs).encode("latin-1"), dtype=np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(buf), off


_as_utf8 = None


def _str_pointers(strs):
    """(void*[n], int64 lengths) of the strings' own byte storage, or None when
    some string is not plain ASCII (then the caller concatenates instead).
    CPython keeps an ASCII str as one byte per character and
    PyUnicode_AsUTF8AndSize hands out that buffer without a copy; it stays valid
    while the str is alive, i.e. for the duration of the call it is passed to."""
    global _as_utf8
    if isinstance(strs, FragmentTable):
        return strs.pointers()
    if _as_utf8 is None:
        _as_utf8 = ctypes.pythonapi.PyUnicode_AsUTF8AndSize
        _as_utf8.restype = ctypes.c_void_p
        _as_utf8.argtypes = [ctypes.py_object, ctypes.POINTER(ctypes.c_ssize_t)]
    n = len(strs)
    arr = (ctypes.c_void_p * n)()
    lens = np.zeros(n, dtype=np.int64)
    size = ctypes.c_ssize_t(0)
    ref = ctypes.byref(size)
    for i, s in enumerate(strs):
        if not isinstance(s, str) or not s.isascii():
            return None
        arr[i] = _as_utf8(s, ref)
        lens[i] = size.value
    return arr, lens


class FragmentTable:
    """Sequences as VIEWS of their parents' own byte storage: (address, length) per fragment, the parents kept alive
    beside them.  A clustered design cuts 3.5 Gbases of genomes into 224 k fragments (catch/filter/probe_designer.py:
    78-184 via Genome.break_into_fragments); slicing them out as str objects and wrapping each in a Genome cost 0.6 s of
    a 4.3-s step before a single base was looked at, although everything downstream -- the signatures, the targets of
    the clusters -- only ever hands (pointer, length) pairs to the library.  `parents`: plain-ASCII str objects
    (others: build() returns None and the caller slices as before); fragment i = parents[parent[i]][start[i] :
    start[i] + length[i]]."""

    def __init__(self, parents, parent, start, length, addr):
        self.parents, self.parent, self.start, self.length, self.addr = parents, parent, start, length, addr

    @staticmethod
    def build(parents, parent, start, length):
        global _as_utf8
        if _as_utf8 is None:
            _as_utf8 = ctypes.pythonapi.PyUnicode_AsUTF8AndSize
            _as_utf8.restype = ctypes.c_void_p
            _as_utf8.argtypes = [ctypes.py_object, ctypes.POINTER(ctypes.c_ssize_t)]
        n = len(parents)
        if n and (set(map(type, parents)) != {str} or not all(map(str.isascii, parents))):
            return None
        base = np.zeros(max(n, 1), dtype=np.uint64)
        size = ctypes.c_ssize_t(0)
        ref = ctypes.byref(size)
        # A plain-ASCII str keeps its characters right behind its header (CPython's "compact ASCII" layout), so the
        # address is id(s) + the header's size -- one vectorised add instead of 198 k calls through ctypes (0.2 s of a
        # 4.3-s step).  The header size is MEASURED on this interpreter, and the first and last parents are checked
        # against PyUnicode_AsUTF8AndSize; any surprise falls back to the call per string.
        fast = False
        if n:
            probe_s = "ACGT" * 3
            hdr = _as_utf8(probe_s, ref) - id(probe_s)
            if 0 < hdr <= 128:
                base[:n] = np.fromiter(map(id, parents), dtype=np.uint64, count=n) + np.uint64(hdr)
                fast = all(int(base[i]) == _as_utf8(parents[i], ref) for i in {0, n // 2, n - 1})
        if not fast:
            for i, s in enumerate(parents):
                base[i] = _as_utf8(s, ref)
        parent = np.ascontiguousarray(parent, dtype=np.int64)
        start = np.ascontiguousarray(start, dtype=np.int64)
        length = np.ascontiguousarray(length, dtype=np.int64)
        return FragmentTable(parents, parent, start, length, base[parent] + start.astype(np.uint64))

    def __len__(self):
        return int(self.length.size)

    def take(self, idx):
        """The fragments idx (an index array), in that order."""
        idx = np.asarray(idx, dtype=np.int64)
        return FragmentTable(self.parents, self.parent[idx], self.start[idx], self.length[idx], self.addr[idx])

    def string(self, i, lo=0, hi=None):
        """Fragment i (or its characters [lo, hi)) as a str."""
        a = int(self.start[i])
        b = a + int(self.length[i])
        return self.parents[int(self.parent[i])][a + lo:b if hi is None else a + hi]

    def pointers(self):
        """(void*[n], int64 lengths) as _str_pointers gives them."""
        arr = (ctypes.c_void_p * max(len(self), 1)).from_buffer_copy(
            np.ascontiguousarray(self.addr if len(self) else np.zeros(1, dtype=np.uint64)).tobytes())
        return arr, np.ascontiguousarray(self.length)


def pyset_order(hashes):
    """catchhip_pyset_order: indices in the order a CPython set iterates keys
    with these hashes after they were added in index order."""
    h = np.ascontiguousarray(hashes, dtype=np.int64)
    out = np.zeros(max(h.size, 1), dtype=np.int64)
    check(_lib.lib().catchhip_pyset_order(_ptr(h, c_i64p), int(h.size), _ptr(out, c_i64p)))
    return out[:h.size]


def pyset_order_strs(strs):
    """catchhip_pyset_order_strs: the order `list(set)` would give the distinct
    ASCII strings after they were added one by one (hash(str) of CPython <=
    3.10 under PYTHONHASHSEED=0)."""
    n = len(strs)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    buf, off = _concat(strs)
    out = np.zeros(n, dtype=np.int64)
    check(_lib.lib().catchhip_pyset_order_strs(_ptr(buf, c_u8p), _ptr(off, c_i64p), n, _ptr(out, c_i64p)))
    return out


def device_count():
    n = ctypes.c_int(0)
    rc = _lib.lib().catchhip_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def pool_trim():
    """catchhip_pool_trim: idle cached device blocks go back to the driver."""
    check(_lib.lib().catchhip_pool_trim())


def pool_stats():
    """catchhip_pool_stats -> dict (device-memory cache of the library)."""
    out = np.zeros(4, dtype=np.int64)
    check(_lib.lib().catchhip_pool_stats(_ptr(out, c_i64p)))
    return dict(zip(("hipmalloc_calls", "bytes_held", "bytes_cached_free",
                     "oom_retries"), (int(x) for x in out)))


class Context:
    """One HIP device + stream (catchhip_ctx)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        self._L = _lib.lib()
        check(self._L.catchhip_ctx_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h:
            self._L.catchhip_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._L.catchhip_ctx_sync(self._h))

    def kernel_ms(self, phase):
        ms = ctypes.c_double(0.0)
        n = ctypes.c_int64(0)
        check(self._L.catchhip_ctx_last_kernel_ms(self._h, phase,
                                                  ctypes.byref(ms),
                                                  ctypes.byref(n)))
        return ms.value, n.value

    def counters(self):
        """catchhip_ctx_last_counters -> dict of work counters."""
        out = np.zeros(8, dtype=np.int64)
        check(self._L.catchhip_ctx_last_counters(self._h, _ptr(out, c_i64p)))
        names = ["raw_hits", "seed_hits", "greedy_iters", "picks",
                 "winner_rows", "rows_recounted", "bitmap_words_read", "_"]
        d = dict(zip(names, (int(x) for x in out)))
        dropped = np.zeros(1, dtype=np.int64)
        check(self._L.catchhip_ctx_last_seeds_dropped(self._h, _ptr(dropped, c_i64p)))
        d["seeds_dropped"] = int(dropped[0])   # of seed_hits: left without a seed by the look-up's filter
        jc = np.zeros(4, dtype=np.int64)
        check(self._L.catchhip_ctx_last_join_counters(self._h, _ptr(jc, c_i64p)))
        d.update(zip(("join_hit_positions", "join_pairs", "join_lane_slots", "join_cut_tasks"), (int(x) for x in jc)))
        sc = np.zeros(4, dtype=np.int64)
        check(self._L.catchhip_ctx_last_solver_counters(self._h, _ptr(sc, c_i64p)))
        d.update(zip(("flat_rows_streamed", "flat_rows_recounted", "flat_bitmap_words",
                      "flat_owner_words"), (int(x) for x in sc)))
        return d

    def pyset_order(self, hashes):
        """catchhip_pyset_order_device: engine.pyset_order computed on the device."""
        h = np.ascontiguousarray(hashes, dtype=np.int64)
        out = np.zeros(max(h.size, 1), dtype=np.int64)
        check(self._L.catchhip_pyset_order_device(self._h, _ptr(h, c_i64p), int(h.size), _ptr(out, c_i64p)))
        return out[:h.size]

    def ndf_counters(self):
        """catchhip_ctx_last_ndf_counters -> dict (last Hamming filter)."""
        out = np.zeros(4, dtype=np.int64)
        check(self._L.catchhip_ctx_last_ndf_counters(self._h, _ptr(out, c_i64p)))
        return dict(zip(("probes", "tables", "pairs_compared", "edges"),
                        (int(x) for x in out)))

    # -- RCCL -----------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = np.zeros(128, dtype=np.uint8)
        check(_lib.lib().catchhip_comm_unique_id(_ptr(buf, c_u8p)))
        return buf.tobytes()

    @staticmethod
    def comm_info():
        """catchhip_comm_info: the RCCL copy (version, file) communicators use."""
        buf = ctypes.create_string_buffer(1024)
        check(_lib.lib().catchhip_comm_info(buf, 1024))
        return buf.value.decode("utf-8", "replace")

    def comm_init(self, unique_id, nranks, rank):
        import os
        import sys
        buf = np.frombuffer(unique_id, dtype=np.uint8).copy()
        # RCCL prints a version banner on stdout at first use; keep stdout
        # clean for callers that emit machine-readable output
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            rc = self._L.catchhip_comm_init(self._h, _ptr(buf, c_u8p), nranks,
                                            rank)
            ctypes.CDLL(None).fflush(None)   # the banner sits in C stdio's buffer
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        check(rc)

    def comm_destroy(self):
        check(self._L.catchhip_comm_destroy(self._h))

    def comm_selftest(self, nelem=1 << 20):
        """catchhip_comm_selftest: a checked SUM all-reduce on this context's communicator (collective)."""
        check(self._L.catchhip_comm_selftest(self._h, int(nelem)))

    # -- near-duplicate filter --------------------------------------------
    def ndf_hamming(self, probe_strs, L, positions, dist_thres):
        n = len(probe_strs)
        buf, _ = _concat(probe_strs)
        pos = np.ascontiguousarray(positions, dtype=np.int32)
        ntables, k = pos.shape
        keep = np.zeros(max(n, 1), dtype=np.uint8)
        check(self._L.catchhip_ndf_hamming(self._h, _ptr(buf, c_u8p), n, L,
                                           _ptr(pos, c_i32p), ntables, k,
                                           dist_thres, _ptr(keep, c_u8p)))
        return keep[:n].astype(bool)


    def ndf_minhash(self, probe_strs, kmer_size, params, dist_thres):
        """catchhip_ndf_minhash; params[table][fn] = (a, b)."""
        n = len(probe_strs)
        buf, off = _concat(probe_strs)
        ab = np.ascontiguousarray(params, dtype=np.int64)
        ntables, k = ab.shape[0], ab.shape[1]
        keep = np.zeros(max(n, 1), dtype=np.uint8)
        check(self._L.catchhip_ndf_minhash(
            self._h, _ptr(buf, c_u8p), _ptr(off, c_i64p), n, int(kmer_size),
            _ptr(ab, c_i64p), ntables, k, float(dist_thres), _ptr(keep, c_u8p)))
        return keep[:n].astype(bool)


    def ndf_minhash_many(self, groups, kmer_size, params, dist_thres):
        """catchhip_ndf_minhash_many; groups = lists of probe strings,
        params[group][table][fn] = (a, b).  Returns one bool array per group."""
        flat = [s for g in groups for s in g]
        n = len(flat)
        if n == 0:
            return [np.zeros(0, dtype=bool) for _ in groups]
        buf, off = _concat(flat)
        goff = np.zeros(len(groups) + 1, dtype=np.int64)
        np.cumsum([len(g) for g in groups], out=goff[1:])
        ab = np.ascontiguousarray(params, dtype=np.int64)
        assert ab.ndim == 4 and ab.shape[0] == len(groups)
        ntables, k = ab.shape[1], ab.shape[2]
        keep = np.zeros(n, dtype=np.uint8)
        check(self._L.catchhip_ndf_minhash_many(
            self._h, _ptr(buf, c_u8p), _ptr(off, c_i64p), n, _ptr(goff, c_i64p),
            len(groups), int(kmer_size), _ptr(ab, c_i64p), ntables, k,
            float(dist_thres), _ptr(keep, c_u8p)))
        keep = keep.astype(bool)
        return [keep[goff[g]:goff[g + 1]] for g in range(len(groups))]


class Targets:
    """Device-resident target sequences (catchhip_targets).
    genomes: list of genomes, each a list of sequence strings."""

    def __init__(self, ctx, genomes):
        self.ctx = ctx
        if isinstance(genomes, FragmentTable):       # every fragment a genome of one sequence
            seqs, sgn = genomes, np.arange(len(genomes), dtype=np.int32)
        else:
            seqs, sg = [], []
            for j, g in enumerate(genomes):
                for s in g:
                    seqs.append(s)
                    sg.append(j)
            sgn = np.asarray(sg, dtype=np.int32)
        if sgn.size == 0:
            sgn = np.zeros(1, dtype=np.int32)
        self.nseq = len(seqs)
        self.ngenomes = len(genomes)
        self._h = ctypes.c_void_p()
        ptrs = _str_pointers(seqs) if len(seqs) else None
        if ptrs is not None:
            # one pointer per sequence: the library gathers them into pinned
            # memory itself (no 600 MB join + encode on the Python side)
            parr, lens = ptrs
            off = np.zeros(len(seqs) + 1, dtype=np.int64)
            np.cumsum(lens, out=off[1:])
            check(ctx._L.catchhip_targets_create_ptrs(
                ctx._h, parr, _ptr(lens, c_i64p), _ptr(sgn, c_i32p), self.nseq,
                self.ngenomes, ctypes.byref(self._h)))
        else:
            buf, off = _concat(seqs)
            check(ctx._L.catchhip_targets_create(
                ctx._h, _ptr(buf, c_u8p), _ptr(off, c_i64p), _ptr(sgn, c_i32p),
                self.nseq, self.ngenomes, ctypes.byref(self._h)))
        self.total = int(off[-1])
        self.seq_off = off      # global start of every sequence (+ total)

    def rebind(self, ctx):
        """catchhip_targets_rebind: hand the object to another context of the
        same device (waits for the stream it was built on)."""
        check(ctx._L.catchhip_targets_rebind(self._h, ctx._h))
        self.ctx = ctx
        return self

    def set_groups(self, group_of_genome):
        """catchhip_targets_set_groups (one entry per genome)."""
        g = np.ascontiguousarray(group_of_genome, dtype=np.int32)
        assert g.size == self.ngenomes
        if g.size == 0:
            g = np.zeros(1, dtype=np.int32)
        check(self.ctx._L.catchhip_targets_set_groups(self.ctx._h, self._h,
                                                      _ptr(g, c_i32p)))

    def close(self):
        if self._h:
            self.ctx._L.catchhip_targets_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Probes:
    """Device-resident unique probes + anchor table (catchhip_probes)."""

    def __init__(self, ctx, uniq, owner, ent_probe, ent_pos, k):
        self.ctx = ctx
        self.n = len(uniq)
        buf, off = _concat(uniq)
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        ep = np.ascontiguousarray(ent_probe, dtype=np.int32)
        eo = np.ascontiguousarray(ent_pos, dtype=np.int32)
        if owner.size == 0:
            owner = np.zeros(1, dtype=np.int32)
        nent = int(ep.size) if self.n else 0
        if ep.size == 0:
            ep = np.zeros(1, dtype=np.int32)
            eo = np.zeros(1, dtype=np.int32)
        self._h = ctypes.c_void_p()
        check(ctx._L.catchhip_probes_create(
            ctx._h, _ptr(buf, c_u8p), _ptr(off, c_i64p), self.n,
            _ptr(owner, c_i32p), _ptr(ep, c_i32p), _ptr(eo, c_i32p), nent,
            int(k or 0), ctypes.byref(self._h)))

    def rebind(self, ctx):
        """catchhip_probes_rebind (see Targets.rebind)."""
        check(ctx._L.catchhip_probes_rebind(self._h, ctx._h))
        self.ctx = ctx
        return self

    def set_groups(self, group_of_probe):
        """catchhip_probes_set_groups (one entry per unique probe)."""
        g = np.ascontiguousarray(group_of_probe, dtype=np.int32)
        assert g.size == self.n
        if g.size == 0:
            g = np.zeros(1, dtype=np.int32)
        check(self.ctx._L.catchhip_probes_set_groups(self.ctx._h, self._h,
                                                     _ptr(g, c_i32p)))

    def close(self):
        if self._h:
            self.ctx._L.catchhip_probes_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Candidates:
    """Unique candidate probes of a Targets object, on the device
    (catchhip_candidates): sliding windows + exact de-duplication."""

    def __init__(self, ctx, targets, probe_length, probe_stride,
                 seq_length_to_skip=None):
        self.ctx = ctx
        self.targets = targets          # keeps the targets alive
        self.L = int(probe_length)
        self._h = ctypes.c_void_p()
        nc, nu = ctypes.c_int64(0), ctypes.c_int64(0)
        check(ctx._L.catchhip_candidates_create(
            ctx._h, targets._h, self.L, int(probe_stride),
            -1 if seq_length_to_skip is None else int(seq_length_to_skip),
            ctypes.byref(self._h), ctypes.byref(nc), ctypes.byref(nu)))
        self.ncandidates, self.n = nc.value, nu.value

    def rebind(self, ctx):
        """catchhip_candidates_rebind (see Targets.rebind)."""
        check(ctx._L.catchhip_candidates_rebind(self._h, ctx._h))
        self.ctx = ctx
        return self

    def positions(self, ids=None):
        """Global start (concatenated target coordinate) of unique candidates
        `ids` (default: all)."""
        if ids is None:
            n, idp = self.n, None
        else:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            n, idp = int(ids.size), _ptr(ids, c_i64p)
        out = np.zeros(max(n, 1), dtype=np.int64)
        check(self.ctx._L.catchhip_candidates_fetch(
            self.ctx._h, self._h, idp, n, _ptr(out, c_i64p)))
        return out[:n]

    def ndf_hamming(self, positions, dist_thres):
        """catchhip_candidates_ndf_hamming: the list becomes the candidates the
        Hamming near-duplicate filter keeps, in its priority order."""
        pos = np.ascontiguousarray(positions, dtype=np.int32)
        nk = ctypes.c_int64(0)
        check(self.ctx._L.catchhip_candidates_ndf_hamming(
            self.ctx._h, self._h, _ptr(pos, c_i32p), pos.shape[0], pos.shape[1],
            int(dist_thres), ctypes.byref(nk)))
        self.n = nk.value

    def ndf_minhash(self, kmer_size, params, dist_thres):
        """catchhip_candidates_ndf_minhash; params[table][fn] = (a, b)."""
        ab = np.ascontiguousarray(params, dtype=np.int64)
        nk = ctypes.c_int64(0)
        check(self.ctx._L.catchhip_candidates_ndf_minhash(
            self.ctx._h, self._h, int(kmer_size), _ptr(ab, c_i64p), ab.shape[0],
            ab.shape[1], float(dist_thres), ctypes.byref(nk)))
        self.n = nk.value

    def ndf_hamming_many(self, positions, dist_thres):
        """catchhip_candidates_ndf_hamming_many (grouped targets);
        positions[group][table][j]."""
        pos = np.ascontiguousarray(positions, dtype=np.int32)
        assert pos.ndim == 3
        nk = ctypes.c_int64(0)
        check(self.ctx._L.catchhip_candidates_ndf_hamming_many(
            self.ctx._h, self._h, _ptr(pos, c_i32p), pos.shape[0], pos.shape[1],
            pos.shape[2], int(dist_thres), ctypes.byref(nk)))
        self.n = nk.value

    def ndf_minhash_many(self, kmer_size, params, dist_thres):
        """catchhip_candidates_ndf_minhash_many (grouped targets);
        params[group][table][fn] = (a, b)."""
        ab = np.ascontiguousarray(params, dtype=np.int64)
        assert ab.ndim == 4
        nk = ctypes.c_int64(0)
        check(self.ctx._L.catchhip_candidates_ndf_minhash_many(
            self.ctx._h, self._h, int(kmer_size), _ptr(ab, c_i64p), ab.shape[0],
            ab.shape[1], ab.shape[2], float(dist_thres), ctypes.byref(nk)))
        self.n = nk.value

    def groups(self):
        """Group of every unique candidate (zeros without grouped targets)."""
        out = np.zeros(max(self.n, 1), dtype=np.int32)
        check(self.ctx._L.catchhip_candidates_groups(self.ctx._h, self._h,
                                                     _ptr(out, c_i32p)))
        return out[:self.n]

    def probes(self, k, ent_probe=None, ent_pos=None):
        """Probes object of the unique candidates; anchors given (sorted by
        (probe, position), unique) or the pigeonhole table when omitted."""
        p = Probes.__new__(Probes)
        p.ctx, p.n = self.ctx, self.n
        p._h = ctypes.c_void_p()
        if ent_probe is None:
            check(self.ctx._L.catchhip_probes_from_candidates(
                self.ctx._h, self._h, None, None, 0, int(k), ctypes.byref(p._h)))
        else:
            ep = np.ascontiguousarray(ent_probe, dtype=np.int32)
            eo = np.ascontiguousarray(ent_pos, dtype=np.int32)
            nent = int(ep.size)
            if nent == 0:
                ep = np.zeros(1, np.int32)
                eo = np.zeros(1, np.int32)
            check(self.ctx._L.catchhip_probes_from_candidates(
                self.ctx._h, self._h, _ptr(ep, c_i32p), _ptr(eo, c_i32p), nent,
                int(k), ctypes.byref(p._h)))
        return p

    def probes_from_draws(self, k, draws):
        """Probes object of the unique candidates with random anchors given as
        the positions np.random drew (uint8 [n][draws per probe]): the sorted
        distinct positions of every probe, built on the device."""
        p = Probes.__new__(Probes)
        p.ctx, p.n = self.ctx, self.n
        p._h = ctypes.c_void_p()
        d = np.ascontiguousarray(draws, dtype=np.uint8)
        if self.n == 0:
            d = np.zeros((1, 1), np.uint8)
        assert d.ndim == 2 and (self.n == 0 or d.shape[0] == self.n)
        check(self.ctx._L.catchhip_probes_from_candidates_draws(
            self.ctx._h, self._h, _ptr(d, c_u8p), int(d.shape[1]), int(k), ctypes.byref(p._h)))
        return p

    def close(self):
        if self._h:
            self.ctx._L.catchhip_candidates_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Rows:
    """Device-resident cover rows (catchhip_rows)."""

    def __init__(self, ctx, handle, n):
        self.ctx = ctx
        self._h = handle
        self.n = int(n)

    @staticmethod
    def scan(ctx, probes, targets, mismatches, lcf_thres, island=0,
             cover_extension=0, mode=SCAN_AUTO, merge=True):
        """catchhip_cover_scan, or with merge=False catchhip_cover_ranges
        (every distinct cover range of every probe, nothing merged)."""
        h = ctypes.c_void_p()
        n = ctypes.c_int64(0)
        fn = ctx._L.catchhip_cover_scan if merge else ctx._L.catchhip_cover_ranges
        check(fn(
            ctx._h, probes._h, targets._h, int(mismatches), int(lcf_thres),
            int(island), int(cover_extension), int(mode), ctypes.byref(h),
            ctypes.byref(n)))
        return Rows(ctx, h, n.value)

    @staticmethod
    def scan_first_seen(ctx, probes, targets, mismatches, lcf_thres, island=0,
                        cover_extension=0, mode=SCAN_AUTO, anchor_order=None):
        """catchhip_cover_scan_first_seen: merged rows that also carry, per
        (set, universe) group, the key of the first accepted seed
        (fetch_first_seen)."""
        h = ctypes.c_void_p()
        n = ctypes.c_int64(0)
        order = (None if anchor_order is None
                 else np.ascontiguousarray(anchor_order, dtype=np.uint32))
        check(ctx._L.catchhip_cover_scan_first_seen(
            ctx._h, probes._h, targets._h, int(mismatches), int(lcf_thres),
            int(island), int(cover_extension), int(mode),
            None if order is None else _ptr(order, c_u32p),
            ctypes.byref(h), ctypes.byref(n)))
        return Rows(ctx, h, n.value)

    def adapter_votes(self, multiplicity):
        """catchhip_adapter_votes -> (A votes, B votes) per set id."""
        mult = np.ascontiguousarray(multiplicity, dtype=np.int64)
        n = int(mult.size)
        a = np.zeros(max(n, 1), dtype=np.int64)
        b = np.zeros(max(n, 1), dtype=np.int64)
        check(self.ctx._L.catchhip_adapter_votes(
            self.ctx._h, self._h, n, _ptr(mult, c_i64p), _ptr(a, c_i64p),
            _ptr(b, c_i64p)))
        return a[:n], b[:n]

    def stats(self, ngenomes, num_sets=0):
        """catchhip_rows_stats -> (total_len[ngenomes], union_len[ngenomes],
        universes_per_set[num_sets])."""
        tl = np.zeros(max(ngenomes, 1), dtype=np.int64)
        ul = np.zeros(max(ngenomes, 1), dtype=np.int64)
        ps = np.zeros(max(num_sets, 1), dtype=np.int64)
        check(self.ctx._L.catchhip_rows_stats(
            self.ctx._h, self._h, _ptr(tl, c_i64p), _ptr(ul, c_i64p),
            int(num_sets), _ptr(ps, c_i64p)))
        return tl[:ngenomes], ul[:ngenomes], ps[:num_sets]

    def cover_check(self, num_sets, picks, universe_p=None):
        """catchhip_rows_cover_check: replays `picks` (set ids in pick order) over these rows with kernels that
        share nothing with the solvers -> dict(picks_without_gain, universes_short, bad_pick_ids, universe_bases,
        covered_bases); a correct solution has the first three at zero."""
        pk = np.ascontiguousarray(picks, dtype=np.int64)
        up = None if universe_p is None else np.ascontiguousarray(universe_p, dtype=np.float64)
        out = np.zeros(5, dtype=np.int64)
        check(self.ctx._L.catchhip_rows_cover_check(
            self.ctx._h, self._h, int(num_sets), _ptr(pk, c_i64p) if pk.size else None, int(pk.size),
            None if up is None else _ptr(up, c_f64p), _ptr(out, c_i64p)))
        return dict(picks_without_gain=int(out[0]), universes_short=int(out[1]), bad_pick_ids=int(out[2]),
                    universe_bases=int(out[3]), covered_bases=int(out[4]))

    def fetch_first_seen(self):
        """uint64[n]: (k-mer position in the universe << 32) | anchor order."""
        out = np.zeros(max(self.n, 1), dtype=np.uint64)
        check(self.ctx._L.catchhip_rows_fetch_first_seen(
            self.ctx._h, self._h, _ptr(out, c_u64p)))
        return out[:self.n]

    @staticmethod
    def from_host(ctx, set_id, universe, start, end, genome_len):
        si = np.ascontiguousarray(set_id, dtype=np.int32)
        un = np.ascontiguousarray(universe, dtype=np.int32)
        st = np.ascontiguousarray(start, dtype=np.int64)
        en = np.ascontiguousarray(end, dtype=np.int64)
        gl = np.ascontiguousarray(genome_len, dtype=np.int64)
        n = int(si.size)
        ng = int(gl.size)
        if n == 0:
            si = np.zeros(1, np.int32); un = np.zeros(1, np.int32)
            st = np.zeros(1, np.int64); en = np.zeros(1, np.int64)
        if ng == 0:
            gl = np.zeros(1, np.int64)
        h = ctypes.c_void_p()
        check(ctx._L.catchhip_rows_from_host(
            ctx._h, _ptr(si, c_i32p), _ptr(un, c_i32p), _ptr(st, c_i64p),
            _ptr(en, c_i64p), n, _ptr(gl, c_i64p), ng, ctypes.byref(h)))
        return Rows(ctx, h, n)

    def fetch(self):
        n = max(self.n, 1)
        si = np.zeros(n, np.int32); un = np.zeros(n, np.int32)
        st = np.zeros(n, np.int64); en = np.zeros(n, np.int64)
        check(self.ctx._L.catchhip_rows_fetch(
            self.ctx._h, self._h, _ptr(si, c_i32p), _ptr(un, c_i32p),
            _ptr(st, c_i64p), _ptr(en, c_i64p)))
        return si[:self.n], un[:self.n], st[:self.n], en[:self.n]

    def greedy(self, num_sets, ranks=None, universe_p=None):
        """catchhip_setcover_greedy -> picked set ids in pick order."""
        num_sets = int(num_sets)
        out = np.zeros(max(num_sets, 1), dtype=np.int64)
        n_out = ctypes.c_int64(0)
        rk = None if ranks is None else np.ascontiguousarray(ranks, np.int64)
        up = (None if universe_p is None
              else np.ascontiguousarray(universe_p, np.float64))
        check(self.ctx._L.catchhip_setcover_greedy(
            self.ctx._h, self._h, num_sets,
            None if rk is None else _ptr(rk, c_i64p),
            None if up is None else _ptr(up, c_f64p),
            _ptr(out, c_i64p), ctypes.byref(n_out)))
        return out[:n_out.value].tolist()

    def close(self):
        if self._h:
            self.ctx._L.catchhip_rows_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Shard:
    """One rank's part of a universe-sharded set cover instance
    (catchhip_shard): the frontier solver's state over this rank's cover
    rows.  Driven round by round by catch_amd.parallel.sharded_solve."""

    def __init__(self, rows, num_sets, ranks=None, universe_p=None, instance_partial=None):
        """universe_p: the fraction to cover of every universe of THIS shard
        (its own genomes, in order), or None = all of each.  With some below 1
        the rounds have a third step (verdict + a second exchange of the lost
        marks; catchhip_shard_create_pi).  instance_partial: whether any
        universe of the WHOLE instance, on whatever rank, is partial -- the
        same value on every rank, so that all of them build the same kind of
        shard (None: decided from this shard's universe_p alone)."""
        self.ctx = rows.ctx
        self.rows = rows                     # keeps the rows alive
        self.num_sets = int(num_sets)
        rk = None if ranks is None else np.ascontiguousarray(ranks, np.int64)
        up = None
        if universe_p is not None:
            # (one entry per universe of the shard's rows: the library reads exactly that many)
            up = np.ascontiguousarray(universe_p, dtype=np.float64)
            if up.size == 0:
                up = None
        self.partial = bool(up is not None and (up < 1.0).any())
        if instance_partial is not None:
            self.partial = bool(instance_partial)
        # whether ANY shard of the instance is partial decides the shape of a round (parallel.sharded_solve) and the
        # kernels of every shard
        self.partial_instance = self.partial
        self._h = ctypes.c_void_p()
        check(self.ctx._L.catchhip_shard_create_pi(
            self.ctx._h, rows._h, self.num_sets,
            None if rk is None else _ptr(rk, c_i64p),
            None if up is None else up.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            -1 if instance_partial is None else int(bool(instance_partial)), ctypes.byref(self._h)))

    def count(self):
        check(self.ctx._L.catchhip_shard_count(self._h))

    def claim_check(self):
        check(self.ctx._L.catchhip_shard_claim_check(self._h))

    def verdict(self):
        """Partial coverage: the local universe tests of this round's candidates
        (failures join the lost marks, which are exchanged once more)."""
        check(self.ctx._L.catchhip_shard_verdict(self._h))

    def apply(self):
        """-> 1 finished, -1 ranks exhausted, 0 another round."""
        done = ctypes.c_int32(0)
        check(self.ctx._L.catchhip_shard_apply(self._h, ctypes.byref(done)))
        return done.value

    def allreduce(self, which):
        """RCCL exchange on the context's communicator (0 gain, 1 lost)."""
        check(self.ctx._L.catchhip_shard_allreduce(self._h, int(which)))

    def _exchange_shape(self):
        """(gain elements, lost elements, lost dtype) of the NEXT exchange: shards
        that run the flat kernels pack their buffers (only the sets still alive
        travel), so the sizes change from round to round."""
        info = np.zeros(4, dtype=np.int64)
        check(self.ctx._L.catchhip_shard_info(self._h, _ptr(info, c_i64p)))
        return int(info[0]), int(info[1]), (np.uint32 if info[2] == 4 else np.uint8)

    def buffer_to_host(self, which):
        """The gain (uint32) or lost (uint8) exchange buffer."""
        ng, nl, ldt = self._exchange_shape()
        out = np.zeros(ng, dtype=np.uint32) if which == 0 else np.zeros(nl, dtype=ldt)
        if out.size:
            check(self.ctx._L.catchhip_shard_buffer_copy(
                self._h, int(which), out.ctypes.data_as(ctypes.c_void_p), 1))
        return out

    def buffer_from_host(self, which, arr):
        ng, nl, ldt = self._exchange_shape()
        a = np.ascontiguousarray(arr, dtype=np.uint32 if which == 0 else ldt)
        assert a.size == (ng if which == 0 else nl)
        if a.size:
            check(self.ctx._L.catchhip_shard_buffer_copy(
                self._h, int(which), a.ctypes.data_as(ctypes.c_void_p), 0))

    def picks(self):
        out = np.zeros(max(self.num_sets, 1), dtype=np.int64)
        n = ctypes.c_int64(0)
        check(self.ctx._L.catchhip_shard_picks(self._h, _ptr(out, c_i64p),
                                               ctypes.byref(n)))
        return out[:n.value].tolist()

    def close(self):
        if self._h:
            self.ctx._L.catchhip_shard_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shards_solve(shards, transport, rounds_per_sync=4):
    """catchhip_shard_solve: the whole round loop of the instance under the C ABI, `rounds_per_sync` rounds per host
    read-back.  transport "rccl": one shard, exchanges over its context's communicator; "local": the shards of this
    process (one context) exchange among themselves.  Returns the picks in the sequential pick order."""
    arr = (ctypes.c_void_p * len(shards))(*[s._h for s in shards])
    done = ctypes.c_int32(0)
    check(shards[0].ctx._L.catchhip_shard_solve(len(shards), arr, 0 if transport == "rccl" else 1, int(rounds_per_sync),
                                                 ctypes.byref(done)))
    out = [sh.picks() for sh in shards]
    if any(o != out[0] for o in out):
        raise RuntimeError("sharded solve: shards returned different picks")
    return out[0]


def shards_allreduce_local(shards, which):
    """catchhip_shard_allreduce_local: the exchange between shards that live
    in this process on one device."""
    arr = (ctypes.c_void_p * len(shards))(*[s._h for s in shards])
    check(shards[0].ctx._L.catchhip_shard_allreduce_local(len(shards), arr,
                                                         int(which)))


class Signatures:
    """Device-resident MinHash signatures of sequences (catchhip_sigs): the N
    smallest values of (a * md5(kmer) + b) mod (2^31 - 1) per sequence."""

    def __init__(self, ctx, seqs, kmer_size, N, a, b):
        self.ctx = ctx
        self.n = len(seqs)
        self.N = int(N)
        self._h = ctypes.c_void_p()
        ptrs = _str_pointers(seqs) if self.n else None
        if ptrs is not None:
            # one pointer per sequence: gathered by the library's host threads (no join + encode of gigabytes here)
            arr, lens = ptrs
            check(ctx._L.catchhip_sigs_create_ptrs(
                ctx._h, ctypes.cast(arr, ctypes.c_void_p), _ptr(lens, c_i64p), self.n, int(kmer_size), self.N,
                int(a), int(b), ctypes.byref(self._h)))
            return
        try:
            raw = "".join(seqs).encode("ascii")
        except UnicodeEncodeError:
            raise ValueError("sequences must be ASCII")
        buf = np.frombuffer(raw, dtype=np.uint8)
        if buf.size == 0:
            buf = np.zeros(1, dtype=np.uint8)
        off = np.zeros(self.n + 1, dtype=np.uint64)
        if seqs:
            np.cumsum([len(s) for s in seqs], out=off[1:])
        self._h = ctypes.c_void_p()
        check(ctx._L.catchhip_sigs_create(
            ctx._h, _ptr(np.ascontiguousarray(buf), c_u8p), _ptr(off, c_u64p),
            self.n, int(kmer_size), self.N, int(a), int(b),
            ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self.ctx._L.catchhip_sigs_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fetch(self):
        """(n, N) uint32, each row ascending."""
        out = np.zeros((max(self.n, 1), self.N), dtype=np.uint32)
        check(self.ctx._L.catchhip_sigs_fetch(self.ctx._h, self._h,
                                              _ptr(out, c_u32p)))
        return out[:self.n]

    def common_row(self, j):
        """values signature j shares with every signature under the N-step
        merge walk of estimate_jaccard_dist (uint16[n])."""
        out = np.zeros(max(self.n, 1), dtype=np.uint16)
        check(self.ctx._L.catchhip_sigs_common_row(self.ctx._h, self._h,
                                                   int(j), _ptr(out, c_u16p)))
        return out[:self.n]

    def neighbors(self, j, min_common):
        """catchhip_sigs_neighbors -> (indices ascending, their common counts):
        the sequences whose merge walk against signature j finds at least
        min_common shared values."""
        if not hasattr(self, "_nb_buf"):
            self._nb_buf = np.zeros(max(self.n, 1), dtype=np.uint64)
        cnt = ctypes.c_int64(0)
        check(self.ctx._L.catchhip_sigs_neighbors(
            self.ctx._h, self._h, int(j), int(min_common), _ptr(self._nb_buf, c_u64p),
            int(self._nb_buf.size), ctypes.byref(cnt)))
        got = np.sort(self._nb_buf[:cnt.value])
        return (got >> np.uint64(16)).astype(np.int64), (got & np.uint64(0xffff)).astype(np.int64)

    NEIGHBORS_MANY_MAX = 32        # queries per call (NEIGH_MAXQ), signatures of at most 112 values

    def neighbors_many(self, js, min_common):
        """catchhip_sigs_neighbors_many -> list of (indices ascending, their
        common counts), one per vertex of js (at most NEIGHBORS_MANY_MAX)."""
        js = np.ascontiguousarray(js, dtype=np.uint32)
        if not hasattr(self, "_nbm_buf"):
            self._nbm_buf = np.zeros(max(4 * self.n, 1 << 16), dtype=np.uint64)
        cnt = ctypes.c_int64(0)
        while True:
            rc = self.ctx._L.catchhip_sigs_neighbors_many(
                self.ctx._h, self._h, _ptr(js, c_u32p), int(js.size), int(min_common),
                _ptr(self._nbm_buf, c_u64p), int(self._nbm_buf.size), ctypes.byref(cnt))
            if rc != 0 and cnt.value > self._nbm_buf.size:      # more neighbours than room: grow and ask again
                self._nbm_buf = np.zeros(int(cnt.value) + (1 << 16), dtype=np.uint64)
                continue
            check(rc)
            break
        got = np.sort(self._nbm_buf[:cnt.value])                # by query, then by index
        q = (got >> np.uint64(48)).astype(np.int64)
        bounds = np.searchsorted(q, np.arange(js.size + 1))
        idx = ((got >> np.uint64(16)) & np.uint64(0xffffffff)).astype(np.int64)
        com = (got & np.uint64(0xffff)).astype(np.int64)
        return [(idx[bounds[i]:bounds[i + 1]], com[bounds[i]:bounds[i + 1]]) for i in range(js.size)]

    GRAPH_MAX_EDGES = 1 << 29      # ordered pairs kept as a graph (6 GB on the device while sorting, 4 GB on the host)

    def graph(self, min_common):
        """catchhip_sigs_graph + fetch -> (ptr int64[n + 1], idx uint32[E] ascending inside a row, common
        uint32[E]): the neighbour lists of every vertex in one device pass, or None when there are more than
        GRAPH_MAX_EDGES ordered pairs (the caller then asks list by list).  Signatures of at most 112 values."""
        cnt = ctypes.c_int64(0)
        check(self.ctx._L.catchhip_sigs_graph(self.ctx._h, self._h, int(min_common), int(self.GRAPH_MAX_EDGES),
                                              ctypes.byref(cnt)))
        e = int(cnt.value)
        if e > self.GRAPH_MAX_EDGES:
            return None
        ptr = np.zeros(self.n + 1, dtype=np.int64)
        idx = np.empty(max(e, 1), dtype=np.uint32)
        com = np.empty(max(e, 1), dtype=np.uint32)
        check(self.ctx._L.catchhip_sigs_graph_fetch(self.ctx._h, self._h, _ptr(ptr, c_i64p), _ptr(idx, c_u32p),
                                                    _ptr(com, c_u32p)))
        return ptr, idx[:e], com[:e]

    def condensed(self, lut):
        """float32[n(n-1)/2] in SciPy's condensed order; entry = lut[common]."""
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        assert lut.size == self.N + 1
        npairs = self.n * (self.n - 1) // 2
        out = np.zeros(max(npairs, 1), dtype=np.float32)
        check(self.ctx._L.catchhip_sigs_condensed(
            self.ctx._h, self._h, _ptr(lut, c_f32p), _ptr(out, c_f32p)))
        return out[:npairs]


def tolerant_bp(ctx, probes, targets, mismatches, lcf_thres, island, out):
    """catchhip_tolerant_bp: out (int64, one per unique probe) += bp."""
    check(ctx._L.catchhip_tolerant_bp(ctx._h, probes._h, targets._h,
                                      int(mismatches), int(lcf_thres),
                                      int(island), _ptr(out, c_i64p)))


def setcover_filter(ctx, probes, targets, mismatches, lcf_thres, island,
                    cover_extension, num_sets, ranks=None, universe_p=None,
                    mode=SCAN_AUTO, as_array=False):
    """catchhip_setcover_filter: scan + greedy in one call.
    Returns (picked set ids in pick order, number of cover rows); as_array: the
    ids as an int64 array instead of a list (a union instance makes 10^5 picks:
    list conversions there and back were 8 ms of a 100-ms step)."""
    return setcover_filter_many([(ctx, probes, targets, num_sets, ranks,
                                  universe_p)], mismatches, lcf_thres, island,
                                cover_extension, mode, as_array)[0]


def setcover_filter_many(groups, mismatches, lcf_thres, island,
                         cover_extension, mode=SCAN_AUTO, as_array=False):
    """catchhip_setcover_filter_many over independent groups, each a tuple
    (ctx, probes, targets, num_sets, ranks or None, universe_p or None) with
    its own Context.  Returns [(ids, nrows)] per group."""
    n = len(groups)
    if n == 0:
        return []
    L = groups[0][0]._L
    VP = ctypes.c_void_p
    ctxs = (VP * n)(*[g[0]._h for g in groups])
    prs = (VP * n)(*[g[1]._h for g in groups])
    tgs = (VP * n)(*[g[2]._h for g in groups])
    nsets = np.array([int(g[3]) for g in groups], dtype=np.int64)
    outs = [np.zeros(max(int(g[3]), 1), dtype=np.int64) for g in groups]
    rks = [None if g[4] is None else np.ascontiguousarray(g[4], np.int64)
           for g in groups]
    ups = [None if g[5] is None else np.ascontiguousarray(g[5], np.float64)
           for g in groups]
    out_p = (c_i64p * n)(*[_ptr(o, c_i64p) for o in outs])
    rk_p = (c_i64p * n)(*[None if r is None else _ptr(r, c_i64p) for r in rks])
    up_p = (c_f64p * n)(*[None if u is None else _ptr(u, c_f64p) for u in ups])
    n_out = np.zeros(n, dtype=np.int64)
    nrows = np.zeros(n, dtype=np.int64)
    if n == 1:
        check(L.catchhip_setcover_filter(
            ctxs[0], prs[0], tgs[0], int(mismatches), int(lcf_thres),
            int(island), int(cover_extension), int(mode), int(nsets[0]),
            rk_p[0], up_p[0], out_p[0], _ptr(n_out, c_i64p),
            _ptr(nrows, c_i64p)))
    else:
        check(L.catchhip_setcover_filter_many(
            n, ctxs, prs, tgs, int(mismatches), int(lcf_thres), int(island),
            int(cover_extension), int(mode), _ptr(nsets, c_i64p), rk_p, up_p,
            out_p, _ptr(n_out, c_i64p), _ptr(nrows, c_i64p)))
    res = [((outs[g][:n_out[g]].copy() if as_array else outs[g][:n_out[g]].tolist()), int(nrows[g])) for g in range(n)]
    if _solution_checks is not None:
        # bench / tests: every instance's picks replayed by the independent check kernels (one more scan, untimed)
        for g, (ids, _) in zip(groups, res):
            rows = Rows.scan(g[0], g[1], g[2], mismatches, lcf_thres, island, cover_extension, mode)
            try:
                r = rows.cover_check(g[3], ids, g[5])
                r.update(picks=len(ids), rows=rows.n, universes=int(g[2].ngenomes))
                _solution_checks.append(r)
            finally:
                rows.close()
    return res


_solution_checks = None


def collect_solution_checks(sink):
    """sink: a list -- from now on every fused scan + solve (setcover_filter / _many) is followed by
    catchhip_rows_cover_check on a second scan of the same instance and its verdict is appended; None: off."""
    global _solution_checks
    _solution_checks = sink


class Prefetch:
    """Runs build(item) for the items in order on a helper thread, at most
    `depth` finished results ahead of the consumer: the packing and upload of
    group i + 1 (host gather into pinned memory, H2D, the front-end kernels on
    the upload context's stream) overlap the scan and solve of group i on the
    compute context.  Iterating yields (item, result) in order; an exception in
    build() is re-raised at the consumer.  close() stops the helper and hands
    every result that was built but not consumed to `discard`."""

    def __init__(self, items, build, depth=2, discard=None):
        import queue
        import threading
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._discard = discard
        self._items = list(items)

        def run():
            for it in self._items:
                if self._stop.is_set():
                    break
                try:
                    res, err = build(it), None
                except BaseException as e:      # handed to the consumer
                    res, err = None, e
                taken = False
                while not self._stop.is_set():
                    try:
                        self._q.put((it, res, err), timeout=0.05)
                        taken = True
                        break
                    except queue.Full:
                        continue
                if not taken and res is not None and self._discard:
                    self._discard(res)          # stopped before it was handed over
                if err is not None:
                    break

        self._thread = threading.Thread(target=run, name="catchhip-prefetch",
                                        daemon=True)
        self._thread.start()

    def __iter__(self):
        for _ in range(len(self._items)):
            it, res, err = self._q.get()
            if err is not None:
                raise err
            yield it, res

    def close(self):
        import queue
        self._stop.set()
        self._thread.join()
        while True:
            try:
                _it, res, _err = self._q.get_nowait()
            except queue.Empty:
                break
            if res is not None and self._discard:
                self._discard(res)


class PrefetchPool:
    """Prefetch with several helper threads: build(item, worker) runs on `workers`
    threads, items are taken in order and results are handed to the consumer in
    order; at most workers + depth items are built or waiting at any time.  For
    stages whose calls leave the device idle much of the time (the MinHash
    filter's ~45 dependent rounds): two of them on two streams overlap."""

    def __init__(self, items, build, workers=2, depth=1, discard=None):
        import threading
        self._items = list(items)
        n = len(self._items)
        self._res = [None] * n
        self._ready = [threading.Event() for _ in range(n)]
        self._stop = threading.Event()
        self._discard = discard
        self._lock = threading.Lock()
        self._next = 0
        self._taken = 0                       # results the consumer has taken
        self._room = threading.Semaphore(max(1, int(workers)) + max(0, int(depth)))

        def run(w):
            while not self._stop.is_set():
                if not self._room.acquire(timeout=0.05):
                    continue
                with self._lock:
                    i = self._next
                    self._next += 1
                if i >= n or self._stop.is_set():
                    self._room.release()
                    return
                try:
                    res, err = build(self._items[i], w), None
                except BaseException as e:      # handed to the consumer
                    res, err = None, e
                self._res[i] = (res, err)
                self._ready[i].set()
                if err is not None:
                    self._stop.set()
                    return

        self._threads = [threading.Thread(target=run, args=(w,), name="catchhip-prefetch-%d" % w, daemon=True)
                         for w in range(max(1, int(workers)))]
        for t in self._threads:
            t.start()

    def __iter__(self):
        for i in range(len(self._items)):
            while not self._ready[i].wait(0.05):
                if self._stop.is_set() and not self._ready[i].is_set():
                    # a later item failed before this one was built: report that failure
                    for r in self._res:
                        if r is not None and r[1] is not None:
                            raise r[1]
            res, err = self._res[i]
            self._res[i] = None
            self._taken = i + 1
            self._room.release()
            if err is not None:
                raise err
            yield self._items[i], res

    def close(self):
        self._stop.set()
        for t in self._threads:
            t.join()
        for i in range(self._taken, len(self._items)):
            r, self._res[i] = self._res[i], None
            if r is not None and r[0] is not None and self._discard:
                self._discard(r[0])


_upload_ctxs = {}


def upload_context(device=None, index=0):
    """The context whose stream packs and uploads the NEXT group's inputs while
    the compute context works (made on first use; index: further ones for
    stages that run several builders side by side)."""
    if device is None:
        device = default_context().device
    if (device, index) not in _upload_ctxs:
        _upload_ctxs[(device, index)] = Context(device)
    return _upload_ctxs[(device, index)]


_default_ctx = None


def default_context():
    """Process-wide context on the device selected by CATCHHIP_DEVICE /
    LOCAL_RANK (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        import os
        dev = int(os.environ.get("CATCHHIP_DEVICE",
                                 os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
    return _default_ctx
