"""BenchmarkDNNLayer (slot_dnn / the gpubox model) on the ROW-SHARDED PS accessor table: one process per GPU.

Reference counterpart: the 8-GPU gpubox run (/root/reference/tools/run_gpubox.sh:21-24,
tools/static_gpubox_trainer.py:152-160,237-259): ONE trainer process drives 8 GPUs, `core.PSGPU` shards the feature
keys across them and pulls / pushes through the inter-GPU links [EXT HeterPS]; the model is
models/rank/slot_dnn/net.py:55-85 (dnn/net.py:67-82 for the 26-slot config).  Here: every rank keeps
its own batch (data parallel) and the rows r with r % G == rank of the hashed table (model parallel); per step

    pull : values -> rows (device hash) -> partition by owner -> all-to-all(v) of local rows -> owners gather the
           record's W -> all-to-all(v) back in send order -> the REQUESTER pools them per (sample, slot)
           (rec_multislot_sumpool_fwd over the reply buffer: the sums never cross a link, only D floats per value do)
    push : d pooled [B, S*D] -> one gradient row per value (the value's (sample, slot) row), its sample's show / click
           -> all-to-all(v) to the owners -> SelectedRows merge -> the accessor's push (rec_ps_push_rows)
    dense: ONE all-reduce of the flat dense-gradient buffer (+ the loss scalar)

The loss is the mean over the GLOBAL batch (G x B) and the pushed gradient is that of the SUMMED loss
(grad_scale = G x B), so a G-rank step is arithmetically one unsharded step on the concatenated batch — which is what
tests/test_sharded_slot_dnn.py checks against one oracle run.  Kept simple on purpose: the exchanges are issued in
program order on the current stream (no look-ahead routing as in ShardedDeepFMLayer); this is the multi-slot host
logic over the same kernels and the same Comm."""
import torch

from . import ops
from .deepfm import NUM_THRESHOLDS
from .sharded import Comm
from .slot_dnn import CLIP, BenchmarkDNNLayer


class ShardedBenchmarkDNNLayer(BenchmarkDNNLayer):
    def __init__(self, dict_dim, emb_dim, slot_num, layer_sizes, device="cuda", kernels=None, key_mode=1,
                 accessor=None, comm=None, group=None, scale_sparse_grad=True):
        self.comm = comm if comm is not None else Comm(group)
        G = self.comm.world
        self.global_rows = int(dict_dim)
        self.local_rows = (self.global_rows + G - 1) // G
        acc = dict(accessor or {})
        acc.update(row_mul=G, row_add=self.comm.rank)       # a key is created with the same values on any sharding
        super().__init__(self.local_rows, emb_dim, slot_num, layer_sizes, device=device, kernels=kernels,
                         sparse_optimizer="ps", key_mode=key_mode, accessor=acc, scale_sparse_grad=scale_sparse_grad)
        self.dict_dim = self.global_rows                     # the hash modulus is the GLOBAL row count
        if G > 1:                                            # data-parallel replicas of the MLP start identical
            self.comm.broadcast(self.dense.data, src=0)
        self.ws_route = self.k.Workspace(self.device)
        self._bufs, self._route = {}, None

    def _fit(self, name, n, d, dtype=torch.float32):
        """Persistent exchange buffer [>= n(, d)], grown with slack."""
        b = self._bufs.get(name)
        shape = (max(n, 1), d) if d else (max(n, 1),)
        if b is None or b.shape[0] < max(n, 1) or b.dtype != dtype:
            cap = int(max(n, 1) * 1.25) + 16
            b = self._bufs[name] = torch.zeros((cap, d) if d else (cap,), dtype=dtype, device=self.device)
        return b[: max(n, 1)] if n else b[:0]

    # -- pull -----------------------------------------------------------------------------------------
    def _pull(self, mb):
        """-> (reply [nnz+1, D] with row 0 = zeros, the requester-side batch over it, route, splits)"""
        k, G, D = self.k, self.comm.world, self.emb_dim
        nnz = mb.nnz
        rows_g = k.feasign_rows(mb.values[:nnz], self.global_rows) if self.key_mode == 1 else mb.values[:nnz]
        route = self._route
        if route is None or route.n != nnz:
            route = self._route = k.ShardRoute(nnz, G, self.device)
        k.shard_route(rows_g.reshape(nnz, 1).contiguous(), self.global_rows, 0, G, self.ws_route, None, self.status,
                      route)
        send_splits = [int(x) for x in route.send_counts[:G].tolist()]          # host sync (G ints)
        recv_splits = self.comm.exchange_counts(send_splits)
        n_send, n_recv = sum(send_splits), sum(recv_splits)
        recv_rows = self._fit("recv_rows", n_recv, 0, torch.int64)
        self.comm.all_to_all(recv_rows, route.send_local_row[:n_send].contiguous(), recv_splits, send_splits)
        g_rows = self._fit("g_rows", n_recv, D)
        if n_recv:     # the owners' lookup: W of the record (a key that does not exist yet reads as zeros)
            k.emb_gather(recv_rows, self.table.W, None, self.status, out=g_rows)
        reply = self._fit("reply", nnz + 1, D)
        reply[0].zero_()
        self.comm.all_to_all(reply[1:1 + n_send], g_rows, send_splits, recv_splits)
        # requester-side batch: value i -> row slot_of_pos[i] of the reply buffer (0 = padding -> the zero row)
        local = k.MultislotBatch(route.slot_of_pos[:nnz].contiguous(), mb.lod, mb.slot_base)
        return reply, local, route, (send_splits, recv_splits, n_send, n_recv, recv_rows)

    def forward(self, mb):
        reply, local, _, _ = self._pull(mb)
        x, _, _, _, _ = self.k.multislot_sumpool(local, reply[: mb.nnz + 1], mb.nnz + 1, 0, 0, self.status,
                                                 want_backward=False)
        y, _ = self.k.mlp_forward(x, self.mlp_w, self.mlp_b, self.ws_mlp)
        return torch.sigmoid(torch.clamp(y, CLIP[0], CLIP[1]))

    __call__ = forward

    # -- one full training step ----------------------------------------------------------------------
    def train_step(self, mb, label, lr=1e-3, auc_stats=None, show=None):
        """-> (loss [1] = mean over the GLOBAL batch, pred [B,1] of the local samples)."""
        k, D, S, G = self.k, self.emb_dim, self.slot_num, self.comm.world
        if mb.num_slots != S:
            raise ops.RecError("batch has %d slots, the net %d" % (mb.num_slots, S))
        B, nnz = label.shape[0], mb.nnz
        self.step_count += 1
        t = self.step_count
        reply, local, route, (send_splits, recv_splits, n_send, n_recv, recv_rows) = self._pull(mb)
        x, counts, seg, _, _ = k.multislot_sumpool(local, reply[: nnz + 1], nnz + 1, 0, 0, self.status,
                                                   want_backward=True)
        self.last_counts = counts
        y, acts = k.mlp_forward(x, self.mlp_w, self.mlp_b, self.ws_mlp)
        pred, dz, loss = k.sigmoid_logloss(y, None, None, label, self.ws, clip=CLIP, mean_over=G * B)
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        dx, finish_dw0 = k.mlp_backward(dz, acts, self.mlp_w, self.mlp_dw, self.mlp_db, self.ws_mlp, defer_first=True)
        finish_dw0()
        # ---- push: one gradient row per SENT value (its (sample, slot) row of dx), show / click of its sample
        send_seg = seg[:nnz].to(torch.int64)[route.send_pos[:n_send]]            # b * S + s of every sent value
        send_g = self._fit("send_g", n_send, D)
        if n_send:
            k.emb_gather(send_seg, dx.view(B * S, D), None, self.status, out=send_g)
        smp = torch.div(send_seg, S, rounding_mode="floor")
        lab = label.reshape(-1)
        send_sc = self._fit("send_sc", n_send, 2, torch.int64)
        if n_send:
            send_sc[:, 0] = show.reshape(-1)[smp] if show is not None else 1
            send_sc[:, 1] = lab[smp]
        recv_g = self._fit("recv_g", n_recv, D)
        recv_sc = self._fit("recv_sc", n_recv, 2, torch.int64)
        self.comm.all_to_all(recv_g, send_g, recv_splits, send_splits)
        self.comm.all_to_all(recv_sc, send_sc, recv_splits, send_splits)
        if n_recv:
            if self._groups is None or self._groups.n < n_recv:
                self._groups = k.IdGroups(int(n_recv * 1.25) + 1, self.device)
            groups = self._groups
            k.ids_group(recv_rows[:n_recv], self.local_rows, None, self.ws_group, None, self.status, groups)
            if self.scale_sparse_grad:          # the loss is the mean over the GLOBAL batch
                self.table.accessor.grad_scale = float(G * B)
            k.ps_push_rows(self.table, groups, recv_g, 1, show=recv_sc[:, 0].contiguous(),
                           click=recv_sc[:, 1].contiguous())
        # ---- dense: one all-reduce of the flat gradient buffer, the loss with it
        if G > 1:
            self.comm.all_reduce_sum(self.dense.grad)
            self.comm.all_reduce_sum(loss)
        k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr)
        return loss, pred
