"""Row-sharded DeepFM: one process per GPU, table rows split row-wise, exchanges by all-to-all.

Reference counterpart: the gpubox path — `core.PSGPU` pull/push inside `exe.train_from_dataset`
(/root/reference/tools/static_gpubox_trainer.py:152-160,244,256; models/rank/dnn/net.py:67-79), where
one trainer process drives 8 GPUs and the HeterPS hash table shards keys across them [EXT].  Here
(SURVEY.md §8(e)): owner(r) = r mod G, local row = r div G; every rank keeps its own batch of B samples
(data parallel) and ceil(N/G) rows of both tables plus their Adam state (model parallel).

One training step = three exchange rounds over RCCL (xGMI is fully connected, so all-to-all drives all
7 links of a GPU at once):
    ids  : all-to-all(v) of local row ids, grouped by owner            (int64)
    rows : owners gather W / W1 rows, all-to-all(v) back in send order  (f32 x D, f32 x 1)
    grads: row-grads / dy1 gathered into send order, all-to-all(v) to owners, merged + lazy Adam there
plus ONE all-reduce of the flat dense-gradient bucket (MLP + FM dense weights + the loss scalar).
The loss is the mean over the GLOBAL batch (G*B), so a sharded step is arithmetically one step of the
unsharded model on the concatenated batch (tests/test_sharded*.py check exactly that).
"""
import os

import torch

from . import ops
from .deepfm import NUM_THRESHOLDS, DeepFMLayer


class Comm:
    """The collectives a step needs, over a torch.distributed process group.
    backend nccl (= RCCL on ROCm): device buffers go straight to the collective.
    backend gloo: device buffers are staged through host memory (used by the CPU / single-GPU tests;
    a transport detail — compute never leaves the HIP kernels)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group if group is not None else dist.group.WORLD
        self.rank = dist.get_rank(self.group)
        self.world = dist.get_world_size(self.group)
        self.staged = dist.get_backend(self.group) != "nccl"
        self.trace = None       # tests set this to a list: the sequence of collectives this rank issued
        # bench.py switches this on for a few bracketed steps: {tag: [(bytes sent, bytes sent to OTHER ranks, start event,
        # end event)]} of every data-path collective — per-step bytes and microseconds of each exchange, per-GPU egress
        self.stats = None
        # backend nccl: the data-path collectives go through the engine's own C-ABI exchange layer
        # (rec_alltoall_exchange / rec_allreduce_sum_f32 over an RCCL communicator created here) — what a Paddle-side
        # binder gets; torch.distributed only carries the 128-byte communicator id and the small int64 metric sums.
        # REC_EMULATE_LINKS=<G> at world 1: every data-path collective is followed by a stand-in link launch
        # (rec_link_emulate) that holds REC_EMULATE_LINK_BLOCKS workgroups (default 8: RCCL's channels) for the time the
        # remote bytes of a G-GPU run would need on xGMI — does the exchange fit beside the dW GEMMs? (one-GPU evidence)
        self.emulate = int(os.environ.get("REC_EMULATE_LINKS", "0")) if dist.get_world_size(self.group) == 1 else 0
        self._emu_ring = None
        self.native = None
        self.native_init_timed_out = False
        self.native_ranks = 0   # ncclCommCount of the C-ABI communicator (0: exchange carried by torch.distributed)
        if not self.staged and os.environ.get("REC_NATIVE_EXCHANGE", "1") != "0":
            self._init_native()
        # ONE communicator: RCCL collectives of a communicator must execute in the same order on every rank, and
        # torch issues a blocking collective on the caller's current stream — so every collective of a training
        # step is issued on ONE stream (the side stream) or behind a stream wait on it (see train_step).

    def _init_native(self):
        """Creates the RCCL communicator of the C-ABI exchange.  Every step that can fail on ONE rank alone (RCCL not
        resolvable by dlsym, ncclGetUniqueId) happens BEFORE an agreed all-reduce(MIN) over torch.distributed, so that
        no rank enters ncclCommInitRank (which blocks for its peers) unless all of them will; a failure after that
        point (communicator init, self-test) is agreed on the same way."""
        import ctypes as C
        import sys
        from ._lib import check, lib
        dev = torch.device("cuda", torch.cuda.current_device())

        def agree(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
            return int(t.item()) == 1

        def warn(what, e):
            print("[recengine] rank %d: native RCCL exchange %s FAILED (%s) - using torch.distributed's RCCL "
                  "collectives for the exchange" % (self.rank, what, e), file=sys.stderr, flush=True)

        idbuf = torch.zeros(128, dtype=torch.uint8, device=dev)
        ok = True
        try:
            if not lib().rec_comm_available():
                raise RuntimeError("RCCL entry points do not resolve in this process")
            if self.rank == 0:
                raw = (C.c_char * 128)()
                check(lib().rec_comm_unique_id(C.cast(raw, C.c_void_p)), "rec_comm_unique_id")
                idbuf.copy_(torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8))
        except Exception as e:
            warn("set-up", e)
            ok = False
        if not agree(ok):
            return
        self.dist.broadcast(idbuf, src=self.dist.get_global_rank(self.group, 0), group=self.group)
        raw = bytes(idbuf.cpu().numpy().tobytes())
        h = C.c_void_p()
        try:
            # ncclCommInitRank blocks until every peer has arrived: run it in a worker thread with a time limit
            # (REC_NATIVE_INIT_TIMEOUT seconds), so that a rank that never comes back from it does not take the job
            # down — the ranks then agree to carry the exchange over torch.distributed's communicator instead
            import threading
            res = {}

            def work():
                try:
                    torch.cuda.set_device(dev)
                    check(lib().rec_comm_init(raw, self.world, self.rank, C.byref(h)), "rec_comm_init")
                    res["ok"] = True
                except Exception as e:      # noqa: BLE001
                    res["err"] = e
            th = threading.Thread(target=work, daemon=True)
            th.start()
            th.join(float(os.environ.get("REC_NATIVE_INIT_TIMEOUT", "120")))
            if th.is_alive():
                self.native_init_timed_out = True
                raise RuntimeError("rec_comm_init did not return within REC_NATIVE_INIT_TIMEOUT")
            if "err" in res:
                raise res["err"]
            self.native = h
            n = C.c_int32(0)
            check(lib().rec_comm_size(h, C.byref(n)), "rec_comm_size")
            self.native_ranks = int(n.value)
            if self.native_ranks != self.world:
                raise RuntimeError("communicator has %d ranks, group has %d" % (self.native_ranks, self.world))
        except Exception as e:
            warn("communicator init", e)
            ok = False
        def drop():
            """Fallback: release a communicator this rank did create (ranks where the init succeeded while a peer's
            failed).  Best effort — while the init thread is still blocked inside ncclCommInitRank there is nothing to
            destroy yet and the handle it may write later is abandoned (the thread is a daemon; documented limitation of
            the timeout path)."""
            if h.value and not th.is_alive():
                try:
                    lib().rec_comm_destroy(h)
                except Exception:      # noqa: BLE001
                    pass
            self.native = None

        if not agree(ok):       # a rank whose ncclCommInitRank failed makes every rank fall back together
            drop()
            return
        try:
            self._selftest_native(dev)
        except Exception as e:   # a broken exchange must not take the step down with it: RCCL through torch instead
            warn("self-test", e)
            ok = False
        if not agree(ok):
            drop()

    def _selftest_native(self, dev):
        """One all-to-all (one int64 per peer, value = 1000*src + dst) and one all-reduce over the new communicator,
        checked on the host, before any training data depends on it."""
        G, r = self.world, self.rank
        send = torch.tensor([1000 * r + d for d in range(G)], dtype=torch.int64, device=dev)
        recv = torch.full((G,), -1, dtype=torch.int64, device=dev)
        self._native_a2a(recv, send, [1] * G, [1] * G)
        one = torch.ones(3, dtype=torch.float32, device=dev)
        import ctypes as C
        from ._lib import check, lib
        check(lib().rec_allreduce_sum_f32(self.native, C.c_void_p(one.data_ptr()), one.numel(),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rec_allreduce_sum_f32")
        torch.cuda.synchronize()
        want = [1000 * s_ + r for s_ in range(G)]
        if recv.tolist() != want:
            raise RuntimeError("all-to-all returned %s, expected %s" % (recv.tolist(), want))
        if one.tolist() != [float(G)] * 3:
            raise RuntimeError("all-reduce returned %s, expected %d" % (one.tolist(), G))

    def _native_a2a(self, out, inp, out_splits, in_splits):
        import ctypes as C
        from ._lib import check, lib
        G = self.world
        row_bytes = inp.element_size() * (inp.shape[1] if inp.dim() > 1 else 1)
        sc = (C.c_int64 * G)(*[int(x) for x in in_splits])
        rc = (C.c_int64 * G)(*[int(x) for x in out_splits])
        check(lib().rec_alltoall_exchange(self.native, C.c_void_p(inp.data_ptr() if inp.numel() else 0), sc,
                                          C.c_void_p(out.data_ptr() if out.numel() else 0), rc, int(row_bytes),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "rec_alltoall_exchange")
        return out

    def exchange_counts(self, send_counts):
        """send_counts[d] = entries this rank sends to d  ->  recv_counts[s] = entries s sends here."""
        self._log("all_to_all:counts")
        dev = "cpu" if self.staged else torch.device("cuda", torch.cuda.current_device())
        t_in = torch.tensor(list(send_counts), dtype=torch.int64, device=dev)
        t_out = torch.empty_like(t_in)
        self.dist.all_to_all_single(t_out, t_in, group=self.group)
        return [int(x) for x in t_out.tolist()]

    def exchange_counts_device(self, send_counts_dev):
        """Device-side variant for backend nccl: no host sync here; returns the receive counts (device)."""
        self._log("all_to_all:counts")
        out = torch.empty_like(send_counts_dev)
        if self.native is not None:
            return self._native_a2a(out, send_counts_dev.contiguous(), [1] * self.world, [1] * self.world)
        self.dist.all_to_all_single(out, send_counts_dev.contiguous(), group=self.group)
        return out

    def _stat_begin(self, tag, sent, remote, on_gpu):
        if self.stats is None:
            return None
        ev = None
        if on_gpu:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.stats.setdefault(tag, []).append([int(sent), int(remote), ev])
        return ev

    def all_to_all(self, out, inp, out_splits, in_splits, tag=None):
        """Rows (dim 0) of `inp` split by in_splits go to the ranks; `out` receives out_splits rows.
        tag: name of the exchange in Comm.stats (ids / rows / grads ...)."""
        what = "all_to_all:%s x%d" % (str(inp.dtype).replace("torch.", ""), inp.shape[1] if inp.dim() > 1 else 1)
        self._log(what)
        ev = None
        if self.stats is not None:
            rb = inp.element_size() * (inp.shape[1] if inp.dim() > 1 else 1)
            ev = self._stat_begin(tag or what, sum(in_splits) * rb, (sum(in_splits) - in_splits[self.rank]) * rb, out.is_cuda)
        try:
            r = self._all_to_all(out, inp, out_splits, in_splits)
            if self.emulate > 1 and out.is_cuda:
                rb = inp.element_size() * (inp.shape[1] if inp.dim() > 1 else 1)
                G = self.emulate
                self._emulate_link(sum(in_splits) * rb * (G - 1) // G, min(G - 1, 7) * 153.0)
            return r
        finally:
            if ev is not None:
                ev[1].record()

    def _all_to_all(self, out, inp, out_splits, in_splits):
        if self.native is not None and out.is_cuda:
            if not (inp.is_contiguous() and out.is_contiguous()):
                raise ops.RecError("exchange buffers must be contiguous")
            return self._native_a2a(out, inp, out_splits, in_splits)
        if self.staged and out.is_cuda:
            h_in = inp.cpu()
            h_out = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_to_all_single(h_out, h_in, list(out_splits), list(in_splits), group=self.group)
            out.copy_(h_out)
        else:
            self.dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=self.group)
        return out

    def _log(self, what):
        if self.trace is not None:
            self.trace.append(what)

    def broadcast(self, t, src=0):
        self._log("broadcast")
        g = self.group
        if self.staged and t.is_cuda:
            h = t.cpu()
            self.dist.broadcast(h, src=self.dist.get_global_rank(g, src), group=g)
            t.copy_(h)
        else:
            self.dist.broadcast(t, src=self.dist.get_global_rank(g, src), group=g)
        return t

    def all_reduce_sum(self, t, tag="all_reduce"):
        self._log("all_reduce")
        ev = None
        if self.stats is not None:     # ring all-reduce: every rank sends 2 (G-1)/G of the buffer
            nb = t.numel() * t.element_size()
            ev = self._stat_begin(tag, nb, int(2 * nb * (self.world - 1) / max(self.world, 1)), t.is_cuda)
        try:
            r = self._all_reduce_sum(t)
            if self.emulate > 1 and t.is_cuda:         # ring all-reduce: 2 (G-1)/G of the buffer over ONE link
                G = self.emulate
                self._emulate_link(int(2 * t.numel() * t.element_size() * (G - 1) / G), 153.0)
            return r
        finally:
            if ev is not None:
                ev[1].record()

    def _emulate_link(self, remote_bytes, gbs):
        import ctypes as C
        from ._lib import check, lib
        if remote_bytes <= 0:
            return
        if self._emu_ring is None:
            self._emu_ring = (torch.empty(32 << 20, dtype=torch.uint8, device="cuda"),
                              torch.empty(32 << 20, dtype=torch.uint8, device="cuda"))
        a, b = self._emu_ring
        check(lib().rec_link_emulate(int(remote_bytes), float(gbs), 8.0, int(os.environ.get("REC_EMULATE_LINK_BLOCKS", "8")),
                                     C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), a.numel(),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rec_link_emulate")

    def stats_summary(self, steps):
        """{tag: {calls_per_step, bytes_per_step, remote_bytes_per_step, us_per_step, egress_GBs}} of the collectives
        recorded since Comm.stats = {} (call after a device synchronize)."""
        out = {}
        for tag, recs in (self.stats or {}).items():
            us = sum(e[0].elapsed_time(e[1]) * 1e3 for _, _, e in recs if e is not None)
            sent, remote = sum(r[0] for r in recs), sum(r[1] for r in recs)
            out[tag] = {"calls_per_step": len(recs) / steps, "bytes_per_step": sent / steps,
                        "remote_bytes_per_step": remote / steps, "us_per_step": us / steps if us else None,
                        "egress_GBs": (remote / 1e9) / (us * 1e-6) if us else None}
        return out

    def _all_reduce_sum(self, t):
        g = self.group
        if self.native is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
            import ctypes as C
            from ._lib import check, lib
            check(lib().rec_allreduce_sum_f32(self.native, C.c_void_p(t.data_ptr()), t.numel(),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "rec_allreduce_sum_f32")
            return t
        if self.staged and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, group=g)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=g)
        return t


class _Lookup:
    """Everything one routed lookup leaves behind for the backward exchange."""
    __slots__ = ("route", "plan", "slot_of_pos", "send_splits", "recv_splits", "n_send", "n_recv", "recv_rows",
                 "reply", "reply1")


class ShardedDeepFMLayer(DeepFMLayer):
    """DeepFMLayer (deepfm/net.py:21-49) with both embedding tables row-sharded over `group`.
    sparse_feature_number is the GLOBAL row count (after slot offsets)."""

    supports_padded_feat = False     # the routed FM path below writes a dense feat

    def __init__(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim,
                 sparse_num_field, layer_sizes, device="cuda", slot_offset=None, group=None,
                 comm=None, kernels=None, table="adam", accessor=None, hash_keys=False, scale_sparse_grad=True,
                 dedup=None, dedup_cap=None):
        """table 'adam': the record layout + lazy Adam of the unsharded layer.  table 'ps': the gpubox feature value
        (tools/static_gpubox_trainer.py:152-160; accessor = slot_dnn/config_online.yaml:57-89) — ONE 128-B record
        line per row [W(16) | W1 | show | click | g2sum_w | g2sum_x | state | delta_score | unseen_days], no second
        optimizer-state line, rows born lazily from zeroed memory: 10^10 rows are 160 GB per GPU on 8 GPUs.
        accessor: kwargs of ops.PsTable.  scale_sparse_grad: the push carries the gradient of the SUMMED loss
        (mean-loss gradient x GLOBAL batch), as Paddle's PS trainers do (scale_sparse_gradient_with_batch_size [EXT]).
        hash_keys: the ids are uint64 feasigns, hashed to rows [1, N) on the device (row = 1 + mix64(f) % (N-1))."""
        self.comm = comm if comm is not None else Comm(group)
        G = self.comm.world
        self.global_rows = int(sparse_feature_number)
        self.local_rows = (self.global_rows + G - 1) // G
        self.hash_keys = bool(hash_keys)
        self.scale_sparse_grad = bool(scale_sparse_grad)
        self.ps = None
        kk = kernels if kernels is not None else ops
        if table == "ps":
            self.ps = kk.PsTable(self.local_rows, sparse_feature_dim, torch.device(device), kind="deepfm",
                                 row_mul=G, row_add=self.comm.rank, **(accessor or {}))
        elif table != "adam":
            raise ValueError("table must be 'adam' or 'ps'")
        # Every rank draws ITS shard from its own generator state (seed + a function of the rank): with the launcher
        # seeding all ranks alike, the shards would otherwise be bit-identical copies — global rows i*G .. i*G+G-1
        # all starting from the same values instead of an i.i.d. table.  (The dense parameters drawn here too are
        # replaced by rank 0's below.)
        dev = torch.device(device)
        base_seed = torch.initial_seed()
        with torch.random.fork_rng(devices=[dev] if dev.type == "cuda" else []):
            torch.manual_seed((base_seed + 1000003 * (self.comm.rank + 1)) % (2 ** 63 - 1))
            # global padding row 0 lives on rank 0 as local row 0 (only meaningful without slot offsets)
            super().__init__(sparse_feature_number, sparse_feature_dim, dense_feature_dim,
                             sparse_num_field, layer_sizes, device=device, slot_offset=slot_offset,
                             table_rows=self.local_rows, zero_padding_row=(self.comm.rank == 0),
                             kernels=kernels, extra_dense=(("__loss__", (1,)),),
                             table_rec=self.ps.rec if self.ps is not None else None)
        # data-parallel replicas of the dense parameters (MLP, FM dense weights) must start identical: every rank
        # drew its own random initialisation, rank 0's wins (one flat buffer, one broadcast)
        if G > 1:
            self.comm.broadcast(self.dense.data, src=0)
        self.ws_route = self.k.Workspace(self.device)
        self._routes, self._route_flip, self._pending = [], 0, None
        self._groups = None
        self._replies, self._reply_flip = [None, None], 0
        # how the tail of a step (exchange chain vs dW GEMMs) shares the chip: "overlap" two streams on all CUs,
        # "serial" chain then GEMMs, "partition" disjoint CU ranges (side_cus for the chain)
        self.tail_mode = os.environ.get("REC_SHARD_TAIL", "overlap")
        self.side_cus = int(os.environ.get("REC_SHARD_SIDE_CUS", "64"))
        self._gemm_stream, self._gemm_cus = None, 0
        self._xbuf, self._pinned = {}, [None, None]
        self._next_lookup = None
        # Deduplicated exchange (REC_SHARD_DEDUP=1 / dedup=True): a rank asks every owner for its DISTINCT rows only,
        # expands the reply locally and merges its row gradients locally before they travel — what HeterPS does per pass
        # (tools/static_gpubox_trainer.py:237-259: load_into_memory -> begin_pass builds the pass's key set).  The
        # exchange sizes are a fixed capacity per owner (dedup_cap x the mean, REC_SHARD_DEDUP_CAP, default 1.25; capped
        # by the lookups of a batch and the rows of a shard), so nothing on the step path reads the device back.
        self.dedup = (os.environ.get("REC_SHARD_DEDUP", "0") == "1") if dedup is None else bool(dedup)
        self.dedup_cap = float(os.environ.get("REC_SHARD_DEDUP_CAP", "1.25")) if dedup_cap is None else float(dedup_cap)
        self._plans_d, self._plan_flip = [None, None], 0

    # -- parameters: global <-> shard --------------------------------------------------------------
    def set_dict(self, sd):
        """Accepts GLOBAL tables ([N,D] / [N,1]); keeps rows r with r % G == rank."""
        sd = dict(sd)
        self._next_lookup = None          # a prefetched lookup holds rows of the old tables
        G, r = self.comm.world, self.comm.rank
        for key, dst in (("fm.embedding.weight", self.fm.embedding),
                         ("fm.embedding_one.weight", self.fm.embedding_one)):
            if key in sd:
                src = torch.as_tensor(sd.pop(key)).reshape(self.global_rows, -1)[r::G]
                dst[: src.shape[0]].copy_(src.to(dst.device))
                if self.ps is not None:     # explicitly set rows EXIST (with their embedx part): otherwise the first
                    self.ps.rec[: src.shape[0], self.ps.state_col] = 2.0   # push would create them over the loaded values
        super().set_dict(sd)

    def gather_global_tables(self):
        """(W [N,D], W1 [N,1]) assembled on every rank — checkpoint export / tests."""
        G = self.comm.world
        out = []
        for t in (self.fm.embedding, self.fm.embedding_one):
            full = torch.zeros(G * self.local_rows, t.shape[1], dtype=t.dtype, device=t.device)
            parts = full.view(self.local_rows, G, t.shape[1])     # row r = local*G + owner
            mine = torch.zeros_like(parts)
            mine[:, self.comm.rank] = t
            self.comm.all_reduce_sum(mine)
            out.append(mine.reshape(G * self.local_rows, -1)[: self.global_rows])
        return out

    # -- routed lookup (ids exchange + rows exchange) ----------------------------------------------
    def _route_async(self, ids):
        """Partition `ids` by owner and start moving the per-owner counts to the host WITHOUT waiting:
        the routing of the next batch depends on its ids only, so it can be issued a step ahead
        (train_step(next_sparse_inputs=...)) and the lookup that consumes it never stalls on a host sync."""
        B, S = ids.shape
        n, G = B * S, self.comm.world
        k = self.k
        ids_key = ids                                # identity of the batch (the prefetch is matched by object)
        if not self._routes or self._routes[0].n != n:
            self._routes = [k.ShardRoute(n, G, self.device), k.ShardRoute(n, G, self.device)]
        self._route_flip ^= 1
        route = self._routes[self._route_flip]       # double-buffered: the previous one is still in use
        if self.hash_keys:                           # uint64 feasigns -> rows of the hashed global table (device)
            ids = k.feasign_rows(ids, self.global_rows, out=self._fit("hashed_ids%d" % self._route_flip, n, 0,
                                                                      torch.int64).view(B, S))
        k.shard_route(ids, self.global_rows, self.fm.padding_idx, G, self.ws_route, self.fm.slot_offset,
                      self.status, route)
        pend = dict(ids=ids_key, route=route, host=None, ev=None)
        if self.device.type == "cuda" and not self.comm.staged:
            recv_dev = self.comm.exchange_counts_device(route.send_counts[:G])
            host = self._pinned[self._route_flip]           # persistent pinned staging, one per route buffer
            if host is None or host.shape[1] != G:
                host = self._pinned[self._route_flip] = torch.empty(2, G, dtype=torch.int64, pin_memory=True)
            host[0].copy_(route.send_counts[:G], non_blocking=True)
            host[1].copy_(recv_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pend.update(host=host, ev=ev, keep=recv_dev)
        return pend

    def dedup_capacity(self, n):
        """Send slots per owner of a deduplicated lookup of n positions."""
        G = self.comm.world
        cap = min(n, self.local_rows)
        if self.dedup_cap > 0:
            cap = min(cap, int(self.dedup_cap * n / G) + 64)
        return max(cap, 1)

    def _lookup_dedup(self, ids):
        """The deduplicated lookup: plan (distinct rows per owner, fixed-capacity slots) -> all-to-all of the local
        rows -> the owners gather -> all-to-all back; the forward reads the reply through slot_of_pos.  No host read."""
        B, S = ids.shape
        n, G, D = B * S, self.comm.world, self.sparse_feature_dim
        k = self.k
        cap = self.dedup_capacity(n)
        if self.hash_keys:
            ids = k.feasign_rows(ids, self.global_rows, out=self._fit("hashed_ids_d", n, 0, torch.int64).view(B, S))
        self._plan_flip ^= 1                         # double-buffered like the routes: the next batch's lookup may be
        plan, _ = k.dedup_plan(ids, self.global_rows, self.fm.padding_idx, G, self.local_rows, cap, self.ws_route,
                               self.fm.slot_offset, self.status, self._plans_d[self._plan_flip])     # issued during this step
        self._plans_d[self._plan_flip] = plan
        L = _Lookup()
        L.route, L.plan = None, plan
        L.send_splits = L.recv_splits = [cap] * G
        L.n_send = L.n_recv = G * cap
        L.recv_rows = self._fit("recv_rows_d%d" % self._plan_flip, G * cap, 0, torch.int64)
        self.comm.all_to_all(L.recv_rows, plan.send_rows, L.recv_splits, L.send_splits, tag="a2a_ids")
        g_rows = self._fit("g_rows", G * cap, D)
        g_w1 = self._fit("g_w1", G * cap, 1)
        # empty slots carry the sentinel local_rows: they gather row 0, which nobody reads (no position maps to them)
        rows_g = torch.where(L.recv_rows == plan.sentinel, torch.zeros_like(L.recv_rows), L.recv_rows)
        k.record_gather(rows_g, self.fm.rec, D, g_rows, g_w1, self.status, table=self.ps)
        self._reply_flip ^= 1
        rep = self._replies[self._reply_flip]
        if rep is None or rep[0].shape[0] != G * cap + 1:
            f32 = dict(dtype=torch.float32, device=self.device)
            rep = self._replies[self._reply_flip] = (torch.zeros(G * cap + 1, D, **f32), torch.zeros(G * cap + 1, 1, **f32))
        L.reply, L.reply1 = rep
        self.comm.all_to_all(L.reply[1:], g_rows, L.send_splits, L.recv_splits, tag="a2a_rows")
        self.comm.all_to_all(L.reply1[1:], g_w1, L.send_splits, L.recv_splits, tag="a2a_rows")
        L.slot_of_pos = plan.slot_of_pos[:n]
        return L

    def _lookup(self, ids):
        if self.dedup:
            return self._lookup_dedup(ids)
        B, S = ids.shape
        n, G, D = B * S, self.comm.world, self.sparse_feature_dim
        k = self.k
        pend = self._pending
        self._pending = None
        if pend is None or pend["ids"] is not ids:
            pend = self._route_async(ids)
        route = pend["route"]
        L = _Lookup()
        L.route, L.plan = route, None
        L.slot_of_pos = route.slot_of_pos
        if pend["ev"] is not None:
            pend["ev"].synchronize()                  # normally long since complete (issued a step ahead)
            L.send_splits = [int(x) for x in pend["host"][0].tolist()]
            L.recv_splits = [int(x) for x in pend["host"][1].tolist()]
        else:
            L.send_splits = [int(x) for x in route.send_counts[:G].tolist()]       # host sync (G ints)
            L.recv_splits = self.comm.exchange_counts(L.send_splits)
        L.n_send, L.n_recv = sum(L.send_splits), sum(L.recv_splits)
        f32 = dict(dtype=torch.float32, device=self.device)
        # exchange buffers are persistent (grown with slack, never allocated per step): no allocator traffic and
        # no cross-stream block recycling on the exchange path
        L.recv_rows = self._fit("recv_rows%d" % (self._reply_flip ^ 1), L.n_recv, 0, torch.int64)
        self.comm.all_to_all(L.recv_rows, route.send_local_row[: L.n_send], L.recv_splits, L.send_splits, tag="a2a_ids")
        g_rows = self._fit("g_rows", L.n_recv, D)
        g_w1 = self._fit("g_w1", L.n_recv, 1)
        if L.n_recv:       # both embeddings of a row from its one record line (unborn PS rows: creation values)
            k.record_gather(L.recv_rows, self.fm.rec, D, g_rows, g_w1, self.status, table=self.ps)
        self._reply_flip ^= 1            # double-buffered: the next batch's lookup may be issued during this step
        rep = self._replies[self._reply_flip]
        if rep is None or rep[0].shape[0] != n + 1:
            rep = self._replies[self._reply_flip] = (torch.zeros(n + 1, D, **f32),
                                                     torch.zeros(n + 1, 1, **f32))  # row 0 stays 0
        L.reply, L.reply1 = rep
        self.comm.all_to_all(L.reply[1:1 + L.n_send], g_rows, L.send_splits, L.recv_splits, tag="a2a_rows")
        self.comm.all_to_all(L.reply1[1:1 + L.n_send], g_w1, L.send_splits, L.recv_splits, tag="a2a_rows")
        return L

    def _fm_fwd_routed(self, L, B, S, dense_inputs):
        # the reply buffer is a (n+1)-row table read through slot_of_pos; 0 = padding -> zero row
        return self.k.deepfm_fm_fwd(L.slot_of_pos.view(B, S), dense_inputs, L.reply, L.reply1,
                                    self.dense.p["fm.dense_w"], self.dense.p["fm.dense_w_one"], 0, None,
                                    self.status, compact=self.compact)

    def forward(self, sparse_inputs, dense_inputs):
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        L = self._lookup(ids)
        y1, y2, feat, _, _ = self._fm_fwd_routed(L, B, S, dense_inputs)
        y_dnn, _ = self.k.mlp_forward(feat.view(B, -1), self._mlp_weights()[0], self.mlp_b, self.ws_mlp)
        return torch.sigmoid(y1 + y2 + y_dnn)

    __call__ = forward

    # -- one full training step --------------------------------------------------------------------
    def train_step(self, sparse_inputs, dense_inputs, label, lr=1e-3, auc_stats=None,
                   next_sparse_inputs=None):
        """Returns (loss [1] = mean over the GLOBAL batch, pred [B,1] of the local samples).
        next_sparse_inputs: ids of the NEXT batch (the same tensor object must be passed to the next call):
        its routing is issued now, under this step's GEMMs."""
        k = self.k
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        G, D = self.comm.world, self.sparse_feature_dim
        if self.ps is None:
            self._ensure_sparse_state()
        self.step_count += 1
        t = self.step_count
        on_gpu = self.device.type == "cuda"
        cur = torch.cuda.current_stream() if on_gpu else None
        mode = self.tail_mode if on_gpu else "overlap"
        if on_gpu and self._side is None:
            if mode == "partition":
                # the chip is split for the tail of the step: HBM/xGMI-bound chain on the first side_cus compute
                # units, dW GEMMs on the others (two kernels sharing all CUs do not co-schedule)
                ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
                self._side = k.cu_range_stream(self.device, 0, self.side_cus)
                self._gemm_stream, self._gemm_cus = k.cu_range_stream(self.device, self.side_cus, ncu), ncu - self.side_cus
            else:
                self._side = k.concurrent_stream(self.device)   # verified to overlap with the main stream

        class _Side:          # "with side:" = run on the side stream after everything issued so far (GPU only)
            def __enter__(s_):
                if on_gpu:
                    self._side.wait_stream(cur)
                    s_.ctx = torch.cuda.stream(self._side)
                    s_.ctx.__enter__()

            def __exit__(s_, *a):
                if on_gpu:
                    s_.ctx.__exit__(*a)

        with self._timed("lookup_exchange"):
            nl, self._next_lookup = self._next_lookup, None
            # prefetched behind the previous step's sparse Adam (see the end of this function)
            L = nl[1] if nl is not None and nl[0] is ids else self._lookup(ids)
        groups = None
        if L.n_recv:
            # merge keys of the rows this rank owns: sorted on the side stream under the forward GEMMs
            if self._groups is None or self._groups.n < L.n_recv:
                self._groups = k.IdGroups(L.n_recv + L.n_recv // 8, self.device)
            with _Side():
                if L.plan is not None:        # empty slots hold the sentinel local_rows: dropped as the padding key
                    groups, _ = k.ids_group(L.recv_rows, self.local_rows + 1, self.local_rows, self.ws_group, None,
                                            self.status, self._groups)
                else:
                    groups, _ = k.ids_group(L.recv_rows, self.local_rows, None, self.ws_group, None,
                                            self.status, self._groups)
        if next_sparse_inputs is not None and not self.dedup:
            with _Side():
                self._pending = self._route_async(self._concat_ids(next_sparse_inputs))
        with self._timed("fm_fwd"):
            y1, y2, feat, sum_emb, _ = self._fm_fwd_routed(L, B, S, dense_inputs)
        mlp_w, mlp_dw = self._mlp_weights()
        loss_slot = self.dense.g["__loss__"]
        # the tower's tail and its backward in one pass (rec_ctr_head_fwd_bwd), as the unsharded step does
        fused_head = self.n_linear > 1 and hasattr(k, "ctr_head") and k.ctr_head_ok(mlp_w[-1], mlp_dw[-1])
        with self._timed("mlp_fwd"):
            if fused_head:
                h, acts = k.mlp_forward(feat.view(B, -1), mlp_w[:-1], self.mlp_b[:-1], self.ws_mlp, relu_last=True)
                pred, dz, _, g_head = k.ctr_head(h, mlp_w[-1], self.mlp_b[-1], y1, y2, label, self.ws, mlp_dw[-1],
                                                 self.mlp_db[-1], mean_over=G * B,
                                                 out=(self._buf("pred", (B, 1)), self._buf("dz", (B, 1)), loss_slot,
                                                      self._buf("g_head", (B, mlp_w[-1].shape[0]))))
            else:
                y_dnn, acts = self.k.mlp_forward(feat.view(B, -1), mlp_w, self.mlp_b, self.ws_mlp)
        if not fused_head:
            pred, dz, _ = k.sigmoid_logloss(y1, y2, y_dnn, label, self.ws, mean_over=G * B,
                                            out=(self._buf("pred", (B, 1)), self._buf("dz", (B, 1)),
                                                 loss_slot))
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        with self._timed("mlp_bwd"):
            # dX chain only; every dW / db GEMM is deferred to the tail so that the exchange-bound work
            # (gradient all-to-all, sparse Adam, next batch's lookup) has ~1.2 ms of MFMA work to hide under
            if fused_head:
                d_flat, finish_dw0 = self.k.mlp_backward(g_head, acts, mlp_w[:-1], mlp_dw[:-1], self.mlp_db[:-1],
                                                         self.ws_mlp, defer_all=True)
            else:
                d_flat, finish_dw0 = self.k.mlp_backward(dz, acts, mlp_w, mlp_dw, self.mlp_db,
                                                         self.ws_mlp, defer_all=True)
        with self._timed("fm_bwd"):
            row_grad, _, _ = k.deepfm_fm_bwd(
                dense_inputs, feat, sum_emb, d_flat.view(B, self.fp, -1), dz, dz, S, self.ws,
                out=(self._row_grad_buf(B * S),
                     self.dense.g["fm.dense_w"].view(self.dense_feature_dim, -1),
                     self.dense.g["fm.dense_w_one"]),
                dense_w=self.dense.p["fm.dense_w"], compact=self.compact)
        # row-gradient exchange + lazy sparse optimizer + next lookup (xGMI / HBM bound) on the side stream,
        # underneath the MFMA-bound dW GEMMs and the dense all-reduce on the main stream
        with _Side():
            with self._timed("grad_exchange"):
                f32 = dict(dtype=torch.float32, device=self.device)
                C1 = 2 if self.ps is not None else 1     # PS: the label rides with dz (click counter of the accessor)
                if L.plan is not None:
                    # merged per distinct row on THIS rank (ascending position order), laid into the plan's send slots;
                    # PS: the occurrence and click counts of a distinct row travel as exact small floats beside dz
                    send_g = k.dedup_merge(L.plan, row_grad, D, out=self._xbuf.get("send_g_d"))
                    self._xbuf["send_g_d"] = send_g
                    send_g = send_g[: L.n_send]
                    if C1 == 1:
                        send_g1 = k.dedup_merge(L.plan, dz, 1, grad_div=S, out=self._xbuf.get("send_g1_d"))
                        self._xbuf["send_g1_d"] = send_g1
                        send_g1 = send_g1[: L.n_send]
                    else:
                        trip = torch.cat([dz.reshape(-1, 1), torch.ones_like(dz).reshape(-1, 1),
                                          label.to(torch.float32).reshape(-1, 1)], dim=1).contiguous()
                        send_g1 = k.dedup_merge(L.plan, trip, 3, grad_div=S, out=self._xbuf.get("send_g3_d"))
                        self._xbuf["send_g3_d"] = send_g1
                        send_g1 = send_g1[: L.n_send]
                        C1 = 3                       # dz | occurrences | clicks of the distinct row
                else:
                    send_g = self._fit("send_g", L.n_send, D)
                    send_g1 = self._fit("send_g1", L.n_send, C1)
                    if L.n_send:
                        k.emb_gather(L.route.send_pos[: L.n_send], row_grad, None, self.status, out=send_g)
                        k.emb_gather(L.route.send_sample[: L.n_send], dz, None, self.status, out=send_g1,
                                     out_group=1, out_group_stride=C1)
                        if C1 == 2:
                            k.emb_gather(L.route.send_sample[: L.n_send], label.to(torch.float32), None, self.status,
                                         out=send_g1[:, 1:], out_group=1, out_group_stride=C1)
                recv_g = self._fit("recv_g", max(L.n_recv, 1), D)
                recv_g1 = self._fit("recv_g1", max(L.n_recv, 1), C1)
                self.comm.all_to_all(recv_g[: L.n_recv], send_g, L.recv_splits, L.send_splits, tag="a2a_grads")
                self.comm.all_to_all(recv_g1[: L.n_recv], send_g1, L.recv_splits, L.send_splits, tag="a2a_grads")
            with self._timed("sparse_adam"):
                if L.n_recv and self.ps is not None:
                    # the accessor's push: counters, AdaGrad rule per part, lazy birth / embedx creation
                    if self.scale_sparse_grad:      # the loss is the mean over the GLOBAL batch
                        self.ps.accessor.grad_scale = float(label.shape[0] * self.comm.world)
                    if C1 == 3:                     # deduplicated: a received row carries its rank's merged counters
                        show = recv_g1[: L.n_recv, 1].round().contiguous().to(torch.int64)
                        click = recv_g1[: L.n_recv, 2].round().contiguous().to(torch.int64)
                        k.ps_push_rows(self.ps, groups, recv_g, 1, grad1=recv_g1, grad1_pitch=3, show=show, click=click)
                    else:
                        click = recv_g1[: L.n_recv, 1].contiguous().to(torch.int64)
                        k.ps_push_rows(self.ps, groups, recv_g, 1, grad1=recv_g1, grad1_pitch=2, click=click)
                elif L.n_recv:
                    st = self.sparse_state
                    pp = self._pp = k.segment_partials(groups, recv_g, D, out=getattr(self, "_pp", None))
                    pp1 = self._pp1 = k.segment_partials(groups, recv_g1, 1, out=getattr(self, "_pp1", None))
                    k.sparse_adam_record(groups, recv_g, recv_g1, 1, self.fm.rec, st["mv"], D, t, lr,
                                         v_offset=(D + 3) // 4 * 4, partials=pp, partials1=pp1)
            # The next batch's lookup needs nothing of this step but the table rows the sparse optimizer just
            # wrote: it goes on the same side stream right behind it — ids / rows exchange over xGMI while the
            # main stream runs the dW GEMMs, the dense all-reduce and the dense Adam.
            if self._pending is not None and next_sparse_inputs is not None:
                with self._timed("next_lookup"):
                    nids = self._pending["ids"]
                    self._next_lookup = (nids, self._lookup(nids))
            elif self.dedup and next_sparse_inputs is not None:      # no routing a step ahead: the plan needs no host sizes
                with self._timed("next_lookup"):
                    nids = self._concat_ids(next_sparse_inputs)
                    self._next_lookup = (nids, self._lookup(nids))
        if mode == "serial":
            cur.wait_stream(self._side)                          # exchange chain first, GEMMs after it
        if mode == "partition":
            self._gemm_stream.wait_stream(cur)
            with torch.cuda.stream(self._gemm_stream):
                with self._timed("mlp_bwd_dw0"):
                    finish_dw0(num_cus=self._gemm_cus)
                    self._fold_backward()
            cur.wait_stream(self._gemm_stream)
        else:
            with self._timed("mlp_bwd_dw0"):
                finish_dw0()
                self._fold_backward()
        # one bucket: dense grads + loss.  Issued on the side stream, behind the exchange all-to-alls already queued
        # there and behind the dW GEMMs of the main stream: all collectives of the step stay totally ordered.
        if on_gpu:
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self.comm.all_reduce_sum(self.dense.grad, tag="allreduce_dense")
            cur.wait_stream(self._side)
        else:
            self.comm.all_reduce_sum(self.dense.grad, tag="allreduce_dense")
        loss = loss_slot.clone()
        loss_slot.zero_()                                         # not a parameter: keep Adam off it
        k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr)
        self._w_key = None          # the folded layer-0 weight belongs to the old parameters
        if on_gpu:
            cur.wait_stream(self._side)
        return loss, pred

    def _fit(self, name, rows, cols, dtype=torch.float32):
        """Persistent exchange buffer `name`, first `rows` rows (cols = 0: 1-D)."""
        b = self._xbuf.get(name)
        if b is None or b.shape[0] < rows:
            cap = rows + rows // 8 + 1
            b = self._xbuf[name] = torch.empty((cap, cols) if cols else (cap,), dtype=dtype, device=self.device)
        return b[:rows]

    def _buf(self, name, shape):
        b = getattr(self, "_b_" + name, None)
        if b is None or tuple(b.shape) != tuple(shape):
            b = torch.empty(*shape, dtype=torch.float32, device=self.device)
            setattr(self, "_b_" + name, b)
        return b
