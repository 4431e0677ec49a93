"""Input pipeline on the engine's host parser: text files -> device batches.

Mirrors the reference's RecDataset readers (same constructor shape: file_list, config) for
  models/rank/deepfm/criteo_reader.py:21-103 / dcn_v2/reader.py  (slot text)   -> SlotTextReader
  models/rank/dnn/benchmark_reader.py:35-54                       (raw Criteo)  -> CriteoTsvReader
but parses whole files in C++ (librecengine.so, multi-threaded) into pinned host buffers and ships each
batch with one async copy, instead of a Python loop per line + a 28-array collate per sample.
Batches have the layout the kernels take: label [B,1] i64, ids [B,26] i64 (= concat(sparse_inputs,1)),
dense [B,13] f32.  drop_last=True as tools/utils/utils_single.py:104-110 builds the DataLoader.
"""
import ctypes as C
import mmap
import os
import queue
import threading
import time

import numpy as np
import torch

from ._lib import check, lib

CONT_MIN = [0, -3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]                      # benchmark_reader.py:23
CONT_DIFF = [20, 603, 100, 50, 64000, 500, 100, 50, 500, 10, 10, 10, 50]  # benchmark_reader.py:25
HASH_DIM = 1000001                                                        # benchmark_reader.py:26


def _alloc(n, S, Dn, pinned):
    pin = pinned and torch.cuda.is_available()
    label = torch.empty(n, dtype=torch.int64, pin_memory=pin)
    ids = torch.empty(n, S, dtype=torch.int64, pin_memory=pin)
    dense = torch.empty(n, max(Dn, 1), dtype=torch.float32, pin_memory=pin)
    return label, ids, dense


def _close_mmap(mm):
    """Closes a text mapping.  If the parser raised, the traceback's frames may still hold the NumPy view _buf()
    exported from it and mmap.close() raises BufferError('cannot close exported pointers exist') — which would mask
    the parse error that matters.  Then the mapping is left to the garbage collector."""
    try:
        mm.close()
    except BufferError:
        pass


def _buf(data):
    """(char pointer, length, keep-alive) of the text: a bytes object, or any buffer — the readers hand in an mmap of
    the file, so the parser threads read the page cache directly and no copy of the text is made."""
    if isinstance(data, bytes):
        return data, len(data), data
    arr = np.frombuffer(data, np.uint8)
    return C.c_char_p(arr.ctypes.data), arr.size, arr


def _count_lines(data, threads):
    n = C.c_int64(0)
    ptr, ln, _keep = _buf(data)
    check(lib().rec_count_lines(ptr, ln, threads, C.byref(n)), "rec_count_lines")
    return max(int(n.value), 1)


def blank_lines(data, threads=0):
    """Indices of the whitespace-only lines of the text (sorted numpy int64 array; usually empty)."""
    ptr, ln, _keep = _buf(data)
    n = C.c_int64(0)
    check(lib().rec_blank_lines(ptr, ln, threads, 0, None, C.byref(n)), "rec_blank_lines")
    out = np.empty(max(int(n.value), 1), np.int64)
    if n.value:
        check(lib().rec_blank_lines(ptr, ln, threads, int(n.value), out.ctypes.data_as(C.c_void_p), C.byref(n)),
              "rec_blank_lines")
    return out[: int(n.value)]


def csr_cut(pieces, num_slots, threads=0):
    """One batch from consecutive line ranges of parse_feasign_slots results: pieces = [(values, lod, base, l0, l1)].
    -> (values, lod [num_slots, lines + 1], base [num_slots + 1]) host tensors (rec_csr_cut)."""
    P = len(pieces)
    nl = sum(l1 - l0 for _, _, _, l0, l1 in pieces)
    total = sum(int((lod[:, l1] - lod[:, l0]).sum()) for _, lod, _, l0, l1 in pieces)
    out_v = torch.empty(max(total, 1), dtype=torch.int64)
    out_l = torch.empty(num_slots, nl + 1, dtype=torch.int64)
    out_b = torch.empty(num_slots + 1, dtype=torch.int64)
    arr = lambda xs: (C.c_void_p * P)(*xs)
    i64 = lambda xs: (C.c_int64 * P)(*xs)
    check(lib().rec_csr_cut(num_slots, P, arr([v.data_ptr() for v, _, _, _, _ in pieces]),
                            arr([l.data_ptr() for _, l, _, _, _ in pieces]), i64([l.stride(0) for _, l, _, _, _ in pieces]),
                            arr([b.data_ptr() for _, _, b, _, _ in pieces]), i64([p[3] for p in pieces]),
                            i64([p[4] for p in pieces]), threads, C.c_void_p(out_v.data_ptr()), out_v.numel(),
                            C.c_void_p(out_l.data_ptr()), C.c_void_p(out_b.data_ptr())), "rec_csr_cut")
    return out_v[:total], out_l, out_b


def parse_slot_text(data: bytes, n_sparse=26, n_dense=13, log1p_dense=False, threads=0, pinned=False):
    """-> (label [n] i64, ids [n,S] i64, dense [n,Dn] f32) host tensors."""
    cap = _count_lines(data, threads)
    label, ids, dense = _alloc(cap, n_sparse, n_dense, pinned)
    n = C.c_int64(0)
    ptr, ln, _keep = _buf(data)
    check(lib().rec_parse_slot_text(ptr, ln, n_sparse, n_dense, int(log1p_dense), cap, threads,
                                    C.c_void_p(label.data_ptr()), C.c_void_p(ids.data_ptr()),
                                    C.c_void_p(dense.data_ptr()), C.byref(n)), "rec_parse_slot_text")
    return label[: n.value], ids[: n.value], dense[: n.value, :n_dense]


def parse_criteo_tsv(data: bytes, n_dense=13, n_sparse=26, hash_dim=HASH_DIM, threads=0, pinned=False):
    cap = _count_lines(data, threads)
    label, ids, dense = _alloc(cap, n_sparse, n_dense, pinned)
    cmin = np.asarray(CONT_MIN[:n_dense], np.float32)
    cdiff = np.asarray(CONT_DIFF[:n_dense], np.float32)
    n = C.c_int64(0)
    ptr, ln, _keep = _buf(data)
    check(lib().rec_parse_criteo_tsv(ptr, ln, n_dense, n_sparse, cmin.ctypes.data_as(C.c_void_p),
                                     cdiff.ctypes.data_as(C.c_void_p), hash_dim, cap, threads,
                                     C.c_void_p(label.data_ptr()), C.c_void_p(ids.data_ptr()),
                                     C.c_void_p(dense.data_ptr()), C.byref(n)), "rec_parse_criteo_tsv")
    return label[: n.value], ids[: n.value], dense[: n.value, :n_dense]


def parse_feasign_slots(data: bytes, first_slot=1, num_slots=301, hash_rows=0, threads=0):
    """Multi-value slot lines (slot_dnn/queuedataset_reader.py:56-82) -> slot-major CSR, host tensors:
    (values [total] i64, lod [num_slots, n+1] i64, slot_base [num_slots+1] i64, n lines) — slot s of line b is
    values[slot_base[s] + lod[s,b] : slot_base[s] + lod[s,b+1]]; with hash_rows > 0 the values are table rows."""
    cap = _count_lines(data, threads)
    ptr, ln, _keep = _buf(data)
    nc = C.c_int64(0)
    check(lib().rec_count_byte(ptr, ln, ord(":"), threads, C.byref(nc)), "rec_count_byte")
    bound = int(nc.value) + cap * num_slots             # every token + one padding id per (line, slot)
    values = torch.empty(max(bound, 1), dtype=torch.int64)
    lod = torch.empty(num_slots, cap + 1, dtype=torch.int64)       # the parser writes every offset of the n lines
    base = torch.zeros(num_slots + 1, dtype=torch.int64)
    n, nv = C.c_int64(0), C.c_int64(0)
    check(lib().rec_parse_feasign_slots(ptr, ln, int(first_slot), int(num_slots), int(hash_rows), cap,
                                        values.numel(), threads, C.c_void_p(values.data_ptr()),
                                        C.c_void_p(lod.data_ptr()), C.c_void_p(base.data_ptr()), C.byref(n),
                                        C.byref(nv)), "rec_parse_feasign_slots")
    return values[: nv.value], lod[:, : n.value + 1], base, int(n.value)


class FeasignSlotReader:
    """slot_dnn/queuedataset_reader.py Reader over a file list: yields, per batch of `batch_size` lines (drop_last),
    (values, lod [num_slots, B+1], slot_base [num_slots+1]) on `device` — slot s of the batch feeds one
    rec_emb_gather_sumpool launch: ids = values[slot_base[s]:slot_base[s+1]], lod = lod[s]."""

    def __init__(self, file_list, batch_size, device="cuda", first_slot=1, num_slots=301, hash_rows=0, threads=0):
        self.file_list, self.batch_size, self.device = list(file_list), batch_size, device
        self.kw = dict(first_slot=first_slot, num_slots=num_slots, hash_rows=hash_rows, threads=threads)

    def __iter__(self):
        for values, lod, base in feasign_batches(self.file_list, self.batch_size, **self.kw):
            yield tuple(t.to(self.device, non_blocking=True) for t in (values, lod, base))


def feasign_batches(file_list, batch_size, first_slot=1, num_slots=301, hash_rows=0, threads=0):
    """Host batches (values, lod [num_slots, B+1], slot_base) of `batch_size` lines over a file list in the multi-value
    `feasign:slot` format: every file is parsed ONCE, whole (mmap'ed text, all threads), and the batches are cut from
    the parse by rec_csr_cut — lines are never split or re-joined in python.  Batching as the reference's datasets do
    it: lines accumulate across files, whitespace-only lines are skipped, the last partial batch is dropped."""
    B = batch_size
    pieces, have = [], 0              # (values, lod, base, first line, end line) ranges not yet batched
    for path in file_list:
        if os.path.getsize(path) == 0:
            continue
        with open(path, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                values, lod, base, n = parse_feasign_slots(mm, first_slot, num_slots, hash_rows, threads)
                blank = blank_lines(mm, threads).tolist()
            finally:
                _close_mmap(mm)
        edges = [-1] + [b for b in blank if b < n] + [n]       # runs of real lines between the (rare) blank ones
        for a, z in zip(edges[:-1], edges[1:]):
            lo = a + 1
            while have + (z - lo) >= B:
                take = B - have
                pieces.append((values, lod, base, lo, lo + take))
                yield csr_cut(pieces, num_slots, threads)
                pieces, have, lo = [], 0, lo + take
            if lo < z:
                pieces.append((values, lod, base, lo, z))
                have += z - lo


class _FileBatches:
    last_trace = None

    def __init__(self, file_list, batch_size, device, parse, shard=None):
        self.file_list = list(file_list)
        if shard is not None:                      # criteo_reader.py:30-43: files split by worker
            rank, world = shard
            if len(self.file_list) < world:
                raise ValueError("The number of data files is less than the number of workers")
            blk = len(self.file_list) // world
            mine = self.file_list[rank * blk:(rank + 1) * blk]
            if rank < len(self.file_list) - blk * world:
                mine.append(self.file_list[-(rank + 1)])
            self.file_list = mine
        self.batch_size, self.device, self.parse = batch_size, device, parse

    def _load(self, path):
        """One file -> host tensors.  The text is mmap'ed, not read(): the parser's threads walk the page cache
        directly (a read() of the file alone was 51 ms per 277 MB, four times the 12 ms the parse takes)."""
        with open(path, "rb") as f:
            size = os.fstat(f.fileno()).st_size
            if size == 0:
                return self.parse(b"")
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                return self.parse(mm)
            finally:
                _close_mmap(mm)

    def _files(self):
        """Parsed files in order; the NEXT files are read and parsed by background threads (the parser is a C call that
        releases the GIL) while the caller trains on the current one — started NOW, not at the first next(): iter(reader)
        begins the pass.  REC_READER_PRODUCERS (default 1) producers take the files round-robin (measured: the parser
        already uses every core, two producers only slow each other — profiles/r04_trainer_level.txt) and the consumer
        takes their results in file order."""
        nprod = max(1, min(int(os.environ.get("REC_READER_PRODUCERS", "1")), len(self.file_list) or 1))
        queues = [queue.Queue(maxsize=1) for _ in range(nprod)]
        stop = threading.Event()
        # REC_READER_TRACE=1: (file, load seconds, seconds blocked on the queue) of the newest pass over the files
        trace = _FileBatches.last_trace = [] if os.environ.get("REC_READER_TRACE") else None

        def put(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce(k):
            q = queues[k]
            try:
                for path in self.file_list[k::nprod]:
                    t0 = time.perf_counter()
                    item = self._load(path)
                    t1 = time.perf_counter()
                    if not put(q, item):
                        return
                    if trace is not None:
                        trace.append([os.path.basename(path), t1 - t0, time.perf_counter() - t1])
                item = None
            except BaseException as e:           # surfaces in the consumer
                item = e
            put(q, item)

        for k in range(nprod):
            threading.Thread(target=produce, args=(k,), daemon=True).start()
        return _StoppingIter(self._consume(queues, nprod, stop), stop)

    def _consume(self, queues, nprod, stop):
        try:
            for i in range(len(self.file_list)):
                item = queues[i % nprod].get()
                if isinstance(item, BaseException):
                    raise item
                if item is None:                 # a producer ran out early: cannot happen with a fixed file list
                    return
                yield item
            for q in queues:                     # every producer ends with None or an exception
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
        finally:
            stop.set()

    def __iter__(self):
        """iter(reader) starts reading and parsing the first files right away; the batches come from the returned
        generator.  (The trainer creates the NEXT epoch's iterator before it drains the device and writes the checkpoint
        of this one, so an epoch does not begin with an idle device waiting for its first file.)"""
        return self._batches(self._files())

    def _batches(self, files):
        B = self.batch_size
        carry = None
        for label, ids, dense in files:
            if carry is not None:
                label, ids, dense = (torch.cat([c, x]) for c, x in zip(carry, (label, ids, dense)))
            n = label.shape[0]
            for lo in range(0, n - B + 1, B):
                yield tuple(t[lo:lo + B].to(self.device, non_blocking=True) for t in
                            (label.view(-1, 1), ids, dense))
            rem = n % B
            carry = (label[n - rem:], ids[n - rem:], dense[n - rem:]) if rem else None


class _StoppingIter:
    """The file iterator of _FileBatches: its producers are started eagerly, so the stop flag must follow the lifetime of
    THIS object — an iterator that is dropped before (or between) next() calls stops them, where a bare generator's
    `finally` only runs once the generator has been started."""

    def __init__(self, gen, stop):
        self._gen, self._stop = gen, stop

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._gen)

    def close(self):
        self._stop.set()
        self._gen.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlotTextReader(_FileBatches):
    """criteo_reader.py RecDataset: yields (label [B,1], ids [B,26], dense [B,13]) on `device`."""

    def __init__(self, file_list, batch_size, device="cuda", log1p_dense=False, shard=None, threads=0):
        super().__init__(file_list, batch_size, device,
                         lambda data: parse_slot_text(data, 26, 13, log1p_dense, threads, pinned=True), shard)


class CriteoTsvReader(_FileBatches):
    """benchmark_reader.py Reader: raw Criteo TSV, min-max dense scaling, xxh32 feature hashing."""

    def __init__(self, file_list, batch_size, device="cuda", hash_dim=HASH_DIM, shard=None, threads=0):
        super().__init__(file_list, batch_size, device,
                         lambda data: parse_criteo_tsv(data, 13, 26, hash_dim, threads, pinned=True), shard)


class DinReader:
    """models/rank/din/dinReader.py RecDataset: lines "hist items;hist cats;target item;target cat;label".
    Groups of 20*batch_size samples are sorted by history length (stable) and cut into batches padded to the
    batch's longest history; yields (hist_item [B,T], hist_cat [B,T], target_item [B], target_cat [B],
    label [B,1] f32, mask [B,T,1] i64 (0 / -1e9), target_item_seq [B,T], target_cat_seq [B,T]) on `device` —
    the eight feeds of din/dygraph_model.py:47-56.  (The reference also scans the files for the longest
    history and writes it to ./tmp.txt, dinReader.py:29-43 — an unused side effect that is not reproduced.)"""

    def __init__(self, file_list, batch_size, device="cuda"):
        self.file_list, self.batch_size, self.device = list(file_list), int(batch_size), device

    def _emit(self, group, upto):
        B = self.batch_size
        lens = np.fromiter((len(g[0]) for g in group), dtype=np.int64, count=len(group))
        order = np.argsort(lens, kind="stable")
        for i in range(0, upto, B):
            idx = order[i:i + B]
            T = int(lens[idx].max())
            item = np.zeros((len(idx), T), np.int64)
            cat = np.zeros((len(idx), T), np.int64)
            for r, k in enumerate(idx):
                item[r, :lens[k]] = group[k][0]
                cat[r, :lens[k]] = group[k][1]
            ti = np.array([group[k][2] for k in idx], np.int64)
            tc = np.array([group[k][3] for k in idx], np.int64)
            label = np.array([group[k][4] for k in idx], np.float32).reshape(-1, 1)
            mask = np.where(np.arange(T)[None, :] < lens[idx][:, None], 0, -1000000000).astype(np.int64)
            arrs = (item, cat, ti, tc, label, mask.reshape(-1, T, 1), np.repeat(ti[:, None], T, 1),
                    np.repeat(tc[:, None], T, 1))
            yield tuple(torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=True) for a in arrs)

    def __iter__(self):
        group, gsz = [], self.batch_size * 20
        for path in self.file_list:
            with open(path, "r") as f:
                for line in f:
                    parts = line.strip().split(";")
                    if len(parts) < 5:
                        continue
                    group.append((np.array(parts[0].split(), np.int64), np.array(parts[1].split(), np.int64),
                                  int(parts[2]), int(parts[3]), float(parts[4])))
                    if len(group) == gsz:
                        yield from self._emit(group, gsz)
                        group = []
        if group:
            yield from self._emit(group, len(group) - len(group) % self.batch_size)
