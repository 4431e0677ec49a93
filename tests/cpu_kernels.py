"""Oracle-backed stand-in for paddlerec_amd.ops on CPU tensors — TEST INFRASTRUCTURE ONLY.

Lets the host orchestration of paddlerec_amd.sharded (routing, split bookkeeping, exchange order,
unscrambling) run under world_size-2 gloo without a GPU.  The product never imports this module:
paddlerec_amd.ops is the only operator backend it ships, and that one refuses CPU tensors.
"""
import numpy as np
import torch

from oracle import deepfm_ref as R
from oracle import shard_ref


def _n(t):
    return None if t is None else t.detach().numpy()


class Workspace:
    def __init__(self, device):
        self.device = device


def new_status(device):
    return torch.zeros(1, dtype=torch.int32)


class ShardRoute:
    def __init__(self, n, num_shards, device):
        self.n, self.num_shards = n, num_shards
        self.send_local_row = torch.zeros(max(n, 1), dtype=torch.int64)
        self.send_pos = torch.zeros(max(n, 1), dtype=torch.int64)
        self.send_sample = torch.zeros(max(n, 1), dtype=torch.int64)
        self.slot_of_pos = torch.zeros(max(n, 1), dtype=torch.int64)
        self.send_counts = torch.zeros(num_shards + 1, dtype=torch.int64)


def shard_route(ids, num_rows, padding_idx, num_shards, ws, slot_offset=None, status=None, route=None):
    r = shard_ref.shard_route(_n(ids), num_shards, padding_idx, _n(slot_offset))
    k = len(r["send_pos"])
    route.send_local_row[:k] = torch.from_numpy(r["send_local_row"])
    route.send_pos[:k] = torch.from_numpy(r["send_pos"])
    route.send_sample[:k] = torch.from_numpy(r["send_sample"])
    route.slot_of_pos[: ids.numel()] = torch.from_numpy(r["slot_of_pos"])
    route.send_counts[:] = torch.from_numpy(r["send_counts"])
    return route, status


class DedupPlan:
    def __init__(self, n, num_shards, cap, device):
        self.n, self.num_shards, self.cap = n, num_shards, cap
        self.groups = IdGroups(n, device)
        self.send_rows = torch.empty(num_shards * cap, dtype=torch.int64)
        self.slot_of_pos = torch.zeros(max(n, 1), dtype=torch.int64)
        self.slot_of_uniq = torch.zeros(max(n, 1), dtype=torch.int64)
        self.counts = torch.zeros(num_shards, dtype=torch.int64)
        self.sentinel = None


def dedup_plan(ids, num_rows, padding_idx, num_shards, local_rows, cap, ws, slot_offset=None, status=None, plan=None):
    """Oracle-side statement of ops.dedup_plan (the distinct (owner, local row) pairs of a batch, owner-major ascending,
    in fixed-capacity slots)."""
    idn = _n(ids).astype(np.int64)
    n, G = idn.size, int(num_shards)
    S = idn.shape[-1]
    if plan is None or plan.n != n or plan.cap != cap:
        plan = DedupPlan(n, G, cap, "cpu")
    rows = idn if slot_offset is None else idn + _n(slot_offset).reshape(1, -1)
    pad = (idn == padding_idx) if padding_idx is not None else np.zeros_like(idn, bool)
    oob = ((rows < 0) | (rows >= num_rows)) & ~pad
    if oob.any() and status is not None:
        status |= 1
    valid = ~(pad | oob).reshape(-1)
    key = ((rows % G) * local_rows + rows // G).reshape(-1)
    g = plan.groups
    g.spos, g.uniq, g.offs = R.group_ids(np.where(valid, key, 0), valid)
    none = G * cap
    plan.sentinel = int(local_rows)
    send = np.full(none, local_rows, np.int64)
    slot_u = np.full(max(n, 1), none, np.int64)
    slot_p = np.zeros(max(n, 1), np.int64)
    fill = np.zeros(G, np.int64)
    for u, k in enumerate(g.uniq):
        o = int(k // local_rows)
        j = fill[o]
        fill[o] += 1
        if j < cap:
            slot_u[u] = o * cap + j
            send[o * cap + j] = k - o * local_rows
            for kk in range(g.offs[u], g.offs[u + 1]):
                slot_p[int(g.spos[kk])] = 1 + o * cap + j
        elif status is not None:
            status |= 2
    plan.counts = torch.from_numpy(fill.copy())
    plan.send_rows, plan.slot_of_uniq, plan.slot_of_pos = torch.from_numpy(send), torch.from_numpy(slot_u), torch.from_numpy(slot_p)
    return plan, status


def dedup_merge(plan, grad, emb_dim, grad_div=1, out=None):
    none = plan.num_shards * plan.cap
    merged = _merged_rows(plan.groups, grad, int(emb_dim), grad_div)
    o = np.zeros((none + 1, int(emb_dim)), np.float32)
    su = _n(plan.slot_of_uniq)
    for u in range(len(plan.groups.uniq)):
        o[su[u]] = merged[u]
    return torch.from_numpy(o)


def emb_gather(ids, W, padding_idx=None, status=None, out=None, out_group=0, out_group_stride=0):
    res = torch.from_numpy(R.embedding_lookup(_n(W), _n(ids).reshape(-1, 1), padding_idx)[:, 0, :])
    if out is None:
        return res, status
    if out_group > 0:          # lookup i -> out + (i // group) * stride + (i % group) * D  (out may be a strided view)
        n, D = res.shape
        torch.as_strided(out, (n // out_group, out_group * D), (out_group_stride, 1), out.storage_offset()) \
            .copy_(res.reshape(n // out_group, out_group * D))
        return out, status
    out.view(res.shape).copy_(res)
    return out, status


def gemm(A, B, ws, trans_a=False, trans_b=False, epilogue="none", bias=None, aux0=None, aux1=None, out=None,
         split_k=0, b_colsum=None, row_scale=None, out2=None, num_cus=0):
    """NumPy statement of rec_gemm_f32 and its epilogues (include/recengine.h: rec_epilogue)."""
    a = _n(A).T if trans_a else _n(A)
    b = _n(B).T if trans_b else _n(B)
    acc = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
    bv = _n(bias).reshape(1, -1) if bias is not None else np.float32(0)
    a0 = _n(aux0) if aux0 is not None else None
    a1 = _n(aux1) if aux1 is not None else None
    if epilogue == "none":
        c = acc
    elif epilogue == "bias":
        c = acc + bv
    elif epilogue == "bias_relu":
        c = np.maximum(acc + bv, 0)
    elif epilogue == "relu_mask":
        c = np.where(a0 > 0, acc, 0)
    elif epilogue == "cross":
        u = acc + bv
        c = a1 + a0 * u
        if out2 is not None:
            out2.copy_(torch.from_numpy(np.ascontiguousarray(u, dtype=np.float32)))
    elif epilogue == "bias_sigmoid":
        c = R.sigmoid(acc + bv)
    elif epilogue == "bias_tanh":
        c = np.tanh(acc + bv)
    elif epilogue == "add":
        c = acc + a1 + bv + (a0 if a0 is not None else 0)
    elif epilogue == "dtanh":
        c = acc * (1 - a0 * a0)
    elif epilogue == "dsigmoid":
        c = acc * a0 * (1 - a0)
    elif epilogue == "moe":
        c = a1 + a0 * _n(row_scale).reshape(-1, 1) * (acc + bv)
    else:
        raise NotImplementedError(epilogue)
    if b_colsum is not None:
        b_colsum.copy_(torch.from_numpy(b.sum(axis=0, dtype=np.float32)).reshape(b_colsum.shape))
    r = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))
    return r if out is None else out.copy_(r)


def cross_bwd_prep(dX, X0, U, dU, dX0_acc, accumulate):
    dx, x0, u = _n(dX), _n(X0), _n(U)
    dU.copy_(torch.from_numpy(dx * x0))
    acc = _n(dX0_acc).copy() if accumulate else 0
    dX0_acc.copy_(torch.from_numpy((acc + dx * u).astype(np.float32)))


def moe_bwd_prep(dX, X0, U, prob_e, dU, dX0_acc, accumulate, dp_e):
    dx, x0, u, pe = _n(dX), _n(X0), _n(U), _n(prob_e).reshape(-1, 1)
    dU.copy_(torch.from_numpy(dx * x0 * pe))
    acc = _n(dX0_acc).copy() if accumulate else 0
    dX0_acc.copy_(torch.from_numpy((acc + dx * pe * u).astype(np.float32)))
    dp_e.copy_(torch.from_numpy((dx * x0 * u).sum(axis=1, dtype=np.float32)))


def softmax_rows(x, out=None):
    z = _n(x)
    e = np.exp(z - z.max(axis=1, keepdims=True))
    r = torch.from_numpy((e / e.sum(axis=1, keepdims=True)).astype(np.float32))
    return r if out is None else out.copy_(r)


def softmax_rows_bwd(p, dp, out=None):
    pn, dpn = _n(p), _n(dp)
    r = torch.from_numpy((pn * (dpn - (pn * dpn).sum(axis=1, keepdims=True))).astype(np.float32))
    return r if out is None else out.copy_(r)


def sumsq(x, out, ws, accumulate=False):
    v = np.float32((_n(x).astype(np.float64) ** 2).sum())
    out.copy_(torch.tensor([v + (float(out[0]) if accumulate else 0.0)], dtype=torch.float32))
    return out


def _merged_rows(groups, grad, D, grad_div=1, grad_group=0, grad_group_stride=0, grad_index=None):
    """Sum of the duplicate gradient rows of every unique id, rows addressed through rec_grad_layout: position pos ->
    q = pos // div; offset = group > 0 ? (q // group) * stride + (q % group) * D : q * D (from the view's first element)."""
    if grad_group > 0:
        base = grad.storage_offset()
        flat = torch.as_strided(grad, (grad.untyped_storage().nbytes() // 4 - base,), (1,), base).numpy()
        row = lambda q: flat[(q // grad_group) * grad_group_stride + (q % grad_group) * D:][:D]
    else:
        g = grad.numpy().reshape(-1, D)
        row = lambda q: g[q]
    merged = np.zeros((len(groups.uniq), D), np.float32)
    for u in range(len(groups.uniq)):
        acc = np.zeros(D, np.float32)
        for kk in range(groups.offs[u], groups.offs[u + 1]):
            pos = int(groups.spos[kk])
            if grad_index is not None:
                pos = int(grad_index[pos])
            acc = acc + row(pos // grad_div)
        merged[u] = acc
    return merged


def sparse_sgd_rows(groups, grad, P, lr, grad_div=1, grad_group=0, grad_group_stride=0, partials=None):
    merged = _merged_rows(groups, grad, P.shape[1], grad_div, grad_group, grad_group_stride)
    Pn = P.numpy()
    Pn[groups.uniq] = Pn[groups.uniq] - np.float32(lr) * merged


SMALL_MERGE_MAX = 15360


def sparse_sgd_small(ids, grad, P, lr, padding_idx=None, status=None, grad_div=1, grad_group=0, grad_group_stride=0):
    g = IdGroups(ids.numel(), "cpu")
    ids_group(ids.reshape(-1), P.shape[0], padding_idx, None, None, status, g)
    sparse_sgd_rows(g, grad, P, lr, grad_div, grad_group, grad_group_stride)
    return status


def sgd_dense(p, g, lr):
    p.sub_(g.reshape(p.shape) * float(lr))


def bce_with_logits(logit, label, ws):
    """binary_cross_entropy_with_logits(reduction='mean') (din/dygraph_model.py:58-61) -> (pred, dz, loss)."""
    from oracle import din_ref
    z, t = _n(logit).astype(np.float32), _n(label).astype(np.float32)
    pred = din_ref.sigmoid(z).astype(np.float32)
    dz = ((pred - t) / np.float32(z.shape[0])).astype(np.float32)
    loss = np.float32(din_ref.bce_with_logits_mean(z, t))
    f = torch.from_numpy
    return f(pred), f(dz), torch.tensor([loss], dtype=torch.float32)


def _din_hq(hist_item, hist_cat, tis, tcs, w_hi, w_hc, w_ti, w_tc):
    h = np.concatenate([_n(w_hi)[_n(hist_item)], _n(w_hc)[_n(hist_cat)]], axis=2)
    q = np.concatenate([_n(w_ti)[_n(tis)], _n(w_tc)[_n(tcs)]], axis=2)
    return h, q


def din_attention_pool(hist_item, hist_cat, tgt_item_seq, tgt_cat_seq, mask, w_hist_item, w_hist_cat,
                       w_tgt_item_seq, w_tgt_cat_seq, att_w, att_b, status=None, want_weights=True, saved=None):
    from oracle import din_ref
    h, q = _din_hq(hist_item, hist_cat, tgt_item_seq, tgt_cat_seq, w_hist_item, w_hist_cat, w_tgt_item_seq,
                   w_tgt_cat_seq)
    out, w = din_ref.attention_pool(h, q, _n(mask).astype(np.float32), [_n(x) for x in att_w], [_n(x) for x in att_b],
                                    return_weights=True)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return f(out), (f(w) if want_weights else None), status


def din_attention_pool_bwd(hist_item, hist_cat, tgt_item_seq, tgt_cat_seq, w_hist_item, w_hist_cat, w_tgt_item_seq,
                           w_tgt_cat_seq, att_w, att_b, att_weight, d_out, saved=None):
    """d h, d q of the attention-pool given the forward's softmax weights (oracle/din_ref.py:66-90 with p = att_weight;
    the attention MLP's own gradients are not produced, App. B-9)."""
    from oracle import din_ref
    h, q = _din_hq(hist_item, hist_cat, tgt_item_seq, tgt_cat_seq, w_hist_item, w_hist_cat, w_tgt_item_seq,
                   w_tgt_cat_seq)
    aw, ab = [_n(x) for x in att_w], [_n(x) for x in att_b]
    E = h.shape[2]
    x = np.concatenate([h, q, h - q, h * q], axis=2)
    a1 = din_ref.sigmoid(x @ aw[0] + ab[0])
    a2 = din_ref.sigmoid(a1 @ aw[1] + ab[1])
    pw, dout = _n(att_weight), _n(d_out)
    dp = (h * dout[:, None, :]).sum(axis=2)
    dh = pw[..., None] * dout[:, None, :]
    ds = pw * (dp - (pw * dp).sum(axis=1, keepdims=True))
    dl = (ds * np.float32(E ** -0.5))[..., None]
    dz2 = (dl @ aw[2].T) * a2 * (1 - a2)
    dz1 = (dz2 @ aw[1].T) * a1 * (1 - a1)
    dx = dz1 @ aw[0].T
    dh = dh + dx[..., :E] + dx[..., 2 * E:3 * E] + dx[..., 3 * E:] * q
    dq = dx[..., E:2 * E] - dx[..., 2 * E:3 * E] + dx[..., 3 * E:] * h
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return f(dh), f(dq)


def sparse_rows_sumsq(groups, grad, D, out, ws, accumulate=False, grad_div=1, grad_group=0, grad_group_stride=0,
                      partials=None):
    m = _merged_rows(groups, grad, D, grad_div, grad_group, grad_group_stride)
    v = np.float32((m.astype(np.float64) ** 2).sum())
    out.copy_(torch.tensor([v + (float(out[0]) if accumulate else 0.0)], dtype=torch.float32))
    return out


def dropout(x, p, seed, stream_a, stream_b=None, out=None, step_stride=0):
    from oracle import dcn_v2_ref as X
    xn = _n(x)
    keep = X.dropout_keep(xn.shape, p, seed, stream_a)
    s = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    if stream_b is not None:
        keep = keep & X.dropout_keep(xn.shape, p, seed, stream_b)
        s = s * s
    y = torch.from_numpy(np.where(keep, xn * s, np.float32(0)).astype(np.float32))
    (x if out is None else out).copy_(y)
    return x if out is None else out


def l2_decay_grad(grad, w, coeff, grad_scale=None):
    c = np.float32(coeff) / (np.float32(float(grad_scale[0])) if grad_scale is not None else np.float32(1))
    grad.add_(w, alpha=float(c))
    return grad


def clip_scale(sumsq_t, clip_norm, out):
    out.copy_(torch.tensor([clip_norm / max(float(np.sqrt(float(sumsq_t[0]))), clip_norm)], dtype=torch.float32))
    return out


def deepfm_fm_fwd(ids, dense, W, W1, dense_w, dense_w_one, padding_idx=0, slot_offset=None,
                  status=None, out=None, compact=False):
    y1, y2, feat = R.fm_forward(_n(ids), _n(dense), _n(W1).reshape(-1, 1), _n(W), _n(dense_w_one),
                                _n(dense_w).reshape(1, dense.shape[1], -1), padding_idx, _n(slot_offset))
    sum_emb = feat.sum(axis=1, dtype=np.float32)
    if compact:
        B, S = ids.shape
        D, Dn = feat.shape[2], dense.shape[1]
        packed = np.zeros((B, 1, D), np.float32)
        packed[:, 0, :Dn] = _n(dense)
        feat = np.concatenate([feat[:, :S], packed], axis=1)
    f = torch.from_numpy
    return f(y1), f(y2), f(np.ascontiguousarray(feat)), f(sum_emb), status


def deepfm_fm_bwd(dense, feat, sum_emb, d_feat_dnn, dy1, dy2, S, ws, out=None, dense_w=None, compact=False):
    B, F, D = feat.shape
    ids = np.ones((B, S), np.int64)     # padding handling happens at the merge, not here
    featn, dfn = _n(feat), _n(d_feat_dnn).reshape(B, F, D)
    if compact:      # rebuild the full tensors: dense embeddings recomputed, their d_dnn part is zero here
        Dn = dense.shape[1]
        de = _n(dense)[:, :, None] * _n(dense_w).reshape(1, Dn, D)
        featn = np.concatenate([featn[:, :S], de.astype(np.float32)], axis=1)
        dfn = np.concatenate([dfn[:, :S], np.zeros((B, Dn, D), np.float32)], axis=1)
    g = R.fm_backward(ids, _n(dense), featn, dfn, _n(dy1).reshape(B, 1), _n(dy2).reshape(B, 1), None)
    row_grad, ddw, ddw1 = out
    row_grad.copy_(torch.from_numpy(g["row_grad"]))
    ddw.copy_(torch.from_numpy(g["d_dense_w"][0]))
    ddw1.copy_(torch.from_numpy(g["d_dense_w_one"]))
    return row_grad, ddw, ddw1


def sigmoid_logloss(y1, y2, y_dnn, label, ws, eps=1e-4, want_dz=True, out=None, mean_over=0, clip=None):
    z = sum(_n(t) for t in (y1, y2, y_dnn) if t is not None)         # NULL addends are allowed by the entry point
    open_ = np.float32(1)
    if clip is not None:                                             # paddle.clip in front of the sigmoid
        open_ = ((z > clip[0]) & (z < clip[1])).astype(np.float32)
        z = np.clip(z, np.float32(clip[0]), np.float32(clip[1]))
    p = R.sigmoid(z).astype(np.float32)
    t = _n(label).astype(np.float32)
    B = p.shape[0]
    den = np.float32(mean_over if mean_over else B)
    cost = -t * np.log(p + np.float32(eps)) - (1 - t) * np.log(1 - p + np.float32(eps))
    dz = ((-t / (p + np.float32(eps)) + (1 - t) / (1 - p + np.float32(eps))) / den) * (p * (1 - p)) * open_
    pred, dzo, loss = out if out is not None else (torch.empty(B, 1), torch.empty(B, 1), torch.empty(1))
    pred.copy_(torch.from_numpy(p))
    dzo.copy_(torch.from_numpy(dz.astype(np.float32)))
    loss.copy_(torch.tensor([cost.sum(dtype=np.float32) / den]))
    return pred, dzo, loss


def colsum(G, ws, out=None):
    r = torch.from_numpy(_n(G).sum(axis=0, dtype=np.float32))
    return r if out is None else out.copy_(r.reshape(out.shape))


def adam_dense(p, m, v, g, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=None):
    gn = g.numpy() if grad_scale is None else g.numpy() * np.float32(float(grad_scale[0]))
    R.adam_update(p.numpy(), m.numpy(), v.numpy(), gn, step, lr, beta1, beta2, eps)


class IdGroups:
    def __init__(self, n, device):
        self.n = n


def ids_group(ids, num_rows, padding_idx, ws, slot_offset=None, status=None, groups=None, payload=None):
    rows, valid = R.effective_rows(_n(ids).reshape(-1, 1), padding_idx, None)
    groups.spos, groups.uniq, groups.offs = R.group_ids(rows.reshape(-1), valid.reshape(-1))
    if payload is not None:             # rec_ids_group_payload: the payload travels instead of the position
        groups.spos = _n(payload).astype(np.int64)[groups.spos]
    return groups, status


def segment_partials(groups, grad, D, grad_div=1, grad_group=0, grad_group_stride=0, out=None, grad_index=None):
    return None     # a pure speed-up of the device kernels (hot rows); the merge below is position by position


def sparse_adam_rows(groups, grad, grad_div, P, M, V, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                     partials=None, grad_group=0, grad_group_stride=0, grad_scale=None, grad_index=None):
    merged = _merged_rows(groups, grad, P.shape[1], grad_div, grad_group, grad_group_stride, grad_index)
    if grad_scale is not None:
        merged = merged * np.float32(float(grad_scale[0]))
    R.adam_update_rows(P.numpy(), M.numpy(), V.numpy(), groups.uniq, merged, step, lr=lr, beta1=beta1,
                       beta2=beta2, eps=eps)


def auc_histogram(pred, label, stat_pos, stat_neg, num_thresholds=4095):
    pos, neg = R.auc_histogram(pred.numpy(), label.numpy(), num_thresholds)
    stat_pos += torch.from_numpy(pos)
    stat_neg += torch.from_numpy(neg)


def mlp_forward(x, weights, biases, ws, relu_last=False, out_last=None):
    acts = []
    n = len(weights)
    for i in range(n):
        acts.append(x)
        x = torch.addmm(biases[i], x, weights[i])
        if i < n - 1 or relu_last:
            x = torch.relu_(x)
        if i == n - 1 and out_last is not None:
            x = out_last.copy_(x)
    return x, acts + [x]


def copy_f32(dst, src):
    return dst.copy_(src)


def mlp_backward(dy, acts, weights, dws, dbs, ws, defer_first=False, defer_all=False, dw_stream=None, dw_ws=None,
                 defer_split=0, defer_cus=0):
    if defer_first or defer_all:
        return mlp_backward(dy, acts, weights, dws, dbs, ws), (lambda: None)
    n = len(weights)
    g = dy
    for i in reversed(range(n)):
        if i < n - 1:
            g = g * (acts[i + 1] > 0).to(g.dtype)
        torch.mm(acts[i].t(), g, out=dws[i])
        torch.sum(g, dim=0, out=dbs[i])
        g = torch.mm(g, weights[i].t())
    return g


def dense_fold_fwd(S, dense_w, W0, M):
    dw = dense_w.numpy().reshape(-1, dense_w.shape[-1])
    Dn, D = dw.shape
    w = W0.numpy()[S * D:(S + Dn) * D].reshape(Dn, D, -1)
    M.copy_(torch.from_numpy(np.einsum("jd,jdn->jn", dw, w).astype(np.float32)))
    return M


def dense_fold_bwd(S, dense_w, W0, dM, dW0, d_dense_w, accumulate=True):
    dw = dense_w.numpy().reshape(-1, dense_w.shape[-1])
    Dn, D = dw.shape
    w = W0.numpy()[S * D:(S + Dn) * D].reshape(Dn, D, -1)
    dm = dM.numpy()
    dW0.numpy()[S * D:(S + Dn) * D] = (dw[:, :, None] * dm[:, None, :]).reshape(Dn * D, -1)
    add = np.einsum("jn,jdn->jd", dm, w).astype(np.float32)
    tgt = d_dense_w.numpy().reshape(Dn, D)
    tgt[...] = (tgt + add) if accumulate else add


def sparse_adam_record(groups, grad, grad1, grad1_div, rec, mv, D, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                       v_offset=None, grad_scale=None, partials=None, partials1=None):
    """rec_sparse_adam_record: both embeddings of a DeepFM row (record layout W(D) | W1 | m1 | v1, state m | v)."""
    if v_offset is None:
        v_offset = (D + 3) // 4 * 4
    kw = dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps, grad_scale=grad_scale)
    sparse_adam_rows(groups, grad, 1, rec[:, :D], mv[:, :D], mv[:, v_offset:v_offset + D], step, **kw)
    sparse_adam_rows(groups, grad1, grad1_div, rec[:, D:D + 1], rec[:, D + 1:D + 2], rec[:, D + 2:D + 3], step, **kw)


def sparse_adam_record_small(ids, slot_offset, padding_idx, grad, grad1, grad1_div, rec, mv, D, step, lr=1e-3,
                             beta1=0.9, beta2=0.999, eps=1e-8, v_offset=None, grad_scale=None, status=None):
    g = IdGroups(ids.numel(), "cpu")
    ids_group(ids, rec.shape[0], padding_idx, None, slot_offset, status, g)
    sparse_adam_record(g, grad, grad1, grad1_div, rec, mv, D, step, lr, beta1, beta2, eps, v_offset, grad_scale)
    return status


class MultislotBatch:
    def __init__(self, values, lod, slot_base):
        self.values, self.lod, self.slot_base = values, lod, slot_base
        self.num_slots, self.batch = lod.shape[0], lod.shape[1] - 1
        self.nnz = values.numel()


def multislot_sumpool(mb, W, num_rows=None, padding_idx=0, key_mode=0, status=None, out=None, want_counts=True,
                      want_backward=True, lazy_init=None):
    from oracle import slot_dnn_ref as M
    Wn = _n(W).copy()
    if lazy_init is not None and lazy_init[2] > 0:       # unborn PS rows read as their creation values
        from oracle import ps_ref
        so, dims, rng_range, seed = lazy_init
        base = torch.as_strided(W, (W.shape[0], so + 1), (W.stride(0), 1), W.storage_offset()).numpy()
        for row in np.nonzero(base[:, so] == 0)[0]:
            Wn[row] = [ps_ref.init_value(seed, row, d, rng_range) if d < dims else 0 for d in range(W.shape[1])]
    o, cnt, seg, rows = M.multislot_sumpool(_n(mb.values), _n(mb.lod), _n(mb.slot_base), Wn, padding_idx, key_mode,
                                            num_rows)
    f = torch.from_numpy
    return f(o), f(cnt), f(seg), f(rows), status


from paddlerec_amd.ops import PsTable  # noqa: E402,F401  (ctypes structs + a torch buffer: no kernel involved)


def _acc_dict(table):
    A = table.accessor
    return dict(lr=A.lr, initial_g2sum=A.initial_g2sum, bounds=(A.min_bound, A.max_bound),
                initial_range=A.initial_range, embedx_lr=A.x_lr, embedx_initial_g2sum=A.x_initial_g2sum,
                embedx_bounds=(A.x_min_bound, A.x_max_bound), embedx_initial_range=A.x_initial_range,
                embedx_threshold=A.embedx_threshold, nonclk_coeff=A.nonclk_coeff,
                click_coeff=A.click_coeff, grad_scale=A.grad_scale, show_scale=bool(A.show_scale),
                embed_zero_init=bool(A.embed_zero_init), seed=A.seed, row_mul=A.row_mul, row_add=A.row_add)


def _lay_dict(table):
    L = table.layout
    return dict(embed_off=L.embed_off, embedx_off=L.embedx_off, embedx_dim=L.embedx_dim, stat_off=L.stat_off)


def record_gather(rows, rec, D, out_w, out_w1, status, table=None):
    from oracle import ps_ref
    r = _n(rows)
    if table is not None:
        W, W1 = ps_ref.pull_deepfm(rec.numpy(), _lay_dict(table), r, _acc_dict(table))
    else:
        W, W1 = rec.numpy()[r, :D], rec.numpy()[r, D]
    out_w.copy_(torch.from_numpy(np.ascontiguousarray(W)).reshape(out_w.shape))
    out_w1.copy_(torch.from_numpy(np.ascontiguousarray(W1)).reshape(out_w1.shape))
    return out_w, out_w1


def feasign_rows(keys, num_rows, out=None):
    from oracle import slot_dnn_ref
    r = torch.from_numpy(slot_dnn_ref.feasign_rows(_n(keys).astype(np.uint64), num_rows))
    return r if out is None else out.copy_(r.reshape(out.shape))


def ps_push_rows(table, groups, grad, num_slots, grad_pitch=None, grad_index=None, grad1=None, grad1_div=1,
                 show=None, click=None, grad1_pitch=1):
    from oracle import ps_ref
    D, L, A = table.emb_dim, table.layout, table.accessor
    gi = _n(grad_index) if grad_index is not None else None
    full = _merged_rows(groups, grad, int(grad_pitch or D), grad_index=gi)
    if table.kind == "slot":
        g_w, g_x = full[:, 0], full[:, 1:D]
    else:
        g_x = full[:, :D]
        g1 = grad1.reshape(-1, int(grad1_pitch))[:, :1].contiguous()
        g_w = _merged_rows(groups, g1, 1, grad_div=grad1_div)[:, 0]
    U = len(groups.uniq)
    dshow, dclick = np.zeros(U), np.zeros(U)
    for u in range(U):
        for kk in range(groups.offs[u], groups.offs[u + 1]):
            pos = int(groups.spos[kk])
            smp = (int(gi[pos]) if gi is not None else pos) // num_slots
            dshow[u] += float(show[smp]) if show is not None else 1.0
            dclick[u] += float(click[smp]) if click is not None else 0.0
    lay, acc = _lay_dict(table), _acc_dict(table)
    ps_ref.push_rows(table.rec.numpy(), lay, groups.uniq, g_w, g_x, dshow, dclick, acc)


def adam_rows_all(groups, grad, grad_div, P, M, V, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                  grad_group=0, grad_group_stride=0, grad_scale=None, partials=None):
    """rec_adam_rows_all: lazy_mode=False Adam — every row moves, rows absent from the merged gradient with g = 0."""
    merged = _merged_rows(groups, grad, P.shape[1], grad_div, grad_group, grad_group_stride)
    if grad_scale is not None:
        merged = merged * np.float32(float(grad_scale[0]))
    Pn, Mn, Vn = P.numpy().copy(), M.numpy().copy(), V.numpy().copy()      # strided views: work on copies
    R.adam_update_dense_equivalent(Pn, Mn, Vn, groups.uniq, merged, step, lr=lr, beta1=beta1, beta2=beta2, eps=eps)
    P.copy_(torch.from_numpy(Pn)); M.copy_(torch.from_numpy(Mn)); V.copy_(torch.from_numpy(Vn))


# ------------------------------------------------------------------ xDeepFM CIN (include/recengine.h: rec_cin_*)
def cin_view(t, kind):
    return ("bfd", None) if kind == "bfd" else ("xt", t[1])


def _cin_bsd(t, view, B, D):
    """any feature tensor as a numpy [B, J, D] VIEW (writes go through)."""
    a = _n(t)
    if view[0] == "bfd":
        return a
    return a.reshape(B, D, a.shape[1]).transpose(0, 2, 1)


def cin_outer_fwd(B, D, F, S, X0, v0, Xk, vk, Z):
    x0, xk = _cin_bsd(X0, v0, B, D), _cin_bsd(Xk, vk, B, D)
    z = np.einsum("bfd,bsd->bdfs", x0, xk).reshape(B * D, F * S)
    Z.copy_(torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32)))
    return Z


def cin_outer_bwd(B, D, F, S, dZ, X0, v0, Xk, vk, dX0, dv0, acc0, dXk, dvk, acck, dpool=None):
    x0, xk = _cin_bsd(X0, v0, B, D), _cin_bsd(Xk, vk, B, D)
    dz = _n(dZ).reshape(B, D, F, S)
    a = np.einsum("bdfs,bsd->bfd", dz, xk).astype(np.float32)
    bk = np.einsum("bdfs,bfd->bsd", dz, x0).astype(np.float32)
    if dpool is not None:
        bk = bk + _n(dpool)[:, :, None]
    o0, ok = _cin_bsd(dX0, dv0, B, D), _cin_bsd(dXk, dvk, B, D)
    o0[...] = (o0 if acc0 else 0) + a
    ok[...] = (ok if acck else 0) + bk


def cin_contract_fwd(B, D, F, Y, X0, v0, XT):
    x0 = _cin_bsd(X0, v0, B, D)                                    # [B,F,D]
    y = _n(Y).reshape(B, D, -1, F)                                 # [B,D,C,F]
    XT.copy_(torch.from_numpy(np.einsum("bfd,bdcf->bdc", x0, y).reshape(B * D, -1).astype(np.float32)))
    return XT


def cin_contract_bwd(B, D, F, Y, dXT, X0, v0, dY, dX0, dv0, acc0):
    x0 = _cin_bsd(X0, v0, B, D)
    y = _n(Y).reshape(B, D, -1, F)
    g = _n(dXT).reshape(B, D, -1)
    dY.copy_(torch.from_numpy(np.einsum("bdc,bfd->bdcf", g, x0).reshape(B * D, -1).astype(np.float32)))
    a = np.einsum("bdc,bdcf->bfd", g, y).astype(np.float32)
    o0 = _cin_bsd(dX0, dv0, B, D)
    o0[...] = (o0 if acc0 else 0) + a


def cin_sumpool(B, D, XT, out):
    x = _n(XT)
    out.copy_(torch.from_numpy(x.reshape(B, D, -1).sum(axis=1, dtype=np.float32)))
    return out


def cin_sumpool_bwd(B, D, dpool, dXT):
    dXT.copy_(torch.from_numpy(np.repeat(_n(dpool)[:, None, :], D, axis=1).reshape(B * D, -1).copy()))
    return dXT


def ps_shrink_rows(table, decay=0.98, delete_threshold=0.8, delete_after_unseen_days=float("inf")):
    from oracle import ps_ref
    return ps_ref.shrink_rows(table.rec.numpy(), _lay_dict(table), _acc_dict(table), decay, delete_threshold,
                              delete_after_unseen_days)


def ps_save_select(table, param, base_threshold=1.5, delta_threshold=0.25, delta_keep_days=16.0):
    from oracle import ps_ref
    return torch.from_numpy(ps_ref.save_select(table.rec.numpy(), _lay_dict(table), _acc_dict(table), param,
                                               base_threshold, delta_threshold, delta_keep_days))


# ------------------------------------------------------------------ DLRM (include/recengine.h: rec_batchnorm_*, rec_dot_interact_*)
def batchnorm_fwd(X, gamma, beta, running_mean, running_var, ws, training=True, momentum=0.9, eps=1e-5, out=None):
    from oracle import dlrm_ref
    y, mean, invstd = dlrm_ref.batchnorm_forward(_n(X), _n(gamma), _n(beta), running_mean.numpy(), running_var.numpy(),
                                                 training, momentum, eps)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    y = f(y)
    return (y if out is None else out.copy_(y)), f(mean), f(invstd)


def batchnorm_bwd(X, dY, gamma, save_mean, save_invstd, ws, relu_mask=False, dgamma=None, dbeta=None, out=None):
    from oracle import dlrm_ref
    dx, dg, db = dlrm_ref.batchnorm_backward(_n(X), _n(dY), _n(gamma), _n(save_mean), _n(save_invstd))
    if relu_mask:
        dx = dx * (_n(X) > 0)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    dx, dg, db = f(dx), f(dg), f(db)
    return (dx if out is None else out.copy_(dx)), (dg if dgamma is None else dgamma.copy_(dg)), \
        (db if dbeta is None else dbeta.copy_(db))


def dot_interact_fwd(T, out=None):
    from oracle import dlrm_ref
    r = torch.from_numpy(dlrm_ref.dot_interact(_n(T)))
    return r if out is None else out.copy_(r)


def dot_interact_bwd(T, dR, out=None):
    from oracle import dlrm_ref
    r = torch.from_numpy(dlrm_ref.dot_interact_backward(_n(T), _n(dR)))
    return r if out is None else out.copy_(r)


def accuracy_count(pred, label, counts):
    ok = ((_n(pred).reshape(-1) > 0.5) == (_n(label).reshape(-1) != 0)).sum()
    counts[0] += int(ok)
    counts[1] += int(pred.numel())
    return counts


# ------------------------------------------------------------------ CrossNet layers (rec_crossnet_*_layer_*)
def crossnet_v2_layer_fwd(x0, xl, W, bias, ws, out=None, u=None):
    return gemm(xl, W, ws, epilogue="cross", bias=bias, aux0=x0, aux1=xl, out2=u, out=out)


def crossnet_v2_layer_bwd(x0, xl, W, u, dxnext, dx0_acc, accumulate_dx0, fold_dx0, dW, db, ws, out=None):
    du = torch.empty(xl.shape, dtype=torch.float32)
    cross_bwd_prep(dxnext, x0, u, du, dx0_acc, accumulate=accumulate_dx0)
    gemm(xl, du, ws, trans_a=True, out=dW, b_colsum=db)
    return gemm(du, W, ws, trans_b=True, epilogue="add", aux1=dxnext, aux0=dx0_acc if fold_dx0 else None, out=out)


def crossnet_mix_layer_fwd(x0, xl, U, V, Cm, bias, gate_w, gate_b, ws, out=None):
    B, d = xl.shape
    E, _, r = U.shape
    gate = gemm(xl, gate_w, ws, epilogue="bias", bias=gate_b)
    prob = softmax_rows(gate)
    t1 = torch.empty(B, E * r, dtype=torch.float32)
    t2 = torch.empty(B, E * r, dtype=torch.float32)
    x_next = out if out is not None else torch.empty(B, d, dtype=torch.float32)
    for e in range(E):
        gemm(xl, V[e], ws, epilogue="bias_tanh", out=t1[:, e * r:(e + 1) * r])
        gemm(t1[:, e * r:(e + 1) * r], Cm[e], ws, trans_b=True, epilogue="bias_tanh", out=t2[:, e * r:(e + 1) * r])
    for e in range(E):
        gemm(t2[:, e * r:(e + 1) * r], U[e], ws, trans_b=True, epilogue="moe", bias=bias, aux0=x0,
             aux1=xl if e == 0 else x_next, row_scale=prob[:, e], out=x_next)
    return x_next, t1, t2, prob


def crossnet_mix_layer_bwd(x0, xl, U, V, Cm, bias, gate_w, t1, t2, prob, dxnext, dx0_acc, accumulate_dx0, fold_dx0,
                           gU, gV, gC, gbias, g_gate_w, g_gate_b, accumulate_gate, ws, out=None):
    B, d = xl.shape
    E, _, r = U.shape
    u = torch.empty(B, d, dtype=torch.float32)
    du = torch.empty(B, d, dtype=torch.float32)
    dp = torch.empty(B, E, dtype=torch.float32)
    dc = torch.empty(B, r, dtype=torch.float32)
    da = torch.empty(B, r, dtype=torch.float32)
    dxl = out if out is not None else torch.empty(B, d, dtype=torch.float32)
    acc = accumulate_dx0
    for e in range(E):
        t1e, t2e = t1[:, e * r:(e + 1) * r], t2[:, e * r:(e + 1) * r]
        gemm(t2e, U[e], ws, trans_b=True, epilogue="bias", bias=bias, out=u)
        moe_bwd_prep(dxnext, x0, u, prob[:, e], du, dx0_acc, acc, dp[:, e])
        acc = True
        dbias_e = colsum(du, ws)
        if e == 0:
            gbias.copy_(dbias_e)
        else:
            gbias.add_(dbias_e)
        gemm(du, t2e, ws, trans_a=True, out=gU[e])
        gemm(du, U[e], ws, epilogue="dtanh", aux0=t2e, out=dc)
        gemm(dc, t1e, ws, trans_a=True, out=gC[e])
        gemm(dc, Cm[e], ws, epilogue="dtanh", aux0=t1e, out=da)
        gemm(xl, da, ws, trans_a=True, out=gV[e])
        gemm(da, V[e], ws, trans_b=True, epilogue="add", aux1=dxnext if e == 0 else dxl, out=dxl)
    dgate = softmax_rows_bwd(prob, dp)
    gw = gemm(xl, dgate, ws, trans_a=True)
    gb = colsum(dgate, ws)
    if accumulate_gate:
        g_gate_w.add_(gw)
        g_gate_b.add_(gb)
    else:
        g_gate_w.copy_(gw)
        g_gate_b.copy_(gb)
    gemm(dgate, gate_w, ws, trans_b=True, epilogue="add", aux1=dxl, out=dxl, aux0=dx0_acc if fold_dx0 else None)
    return dxl


# ------------------------------------------------------------------ custom operators (compat paddle.utils.cpp_extension)
# CUSTOM_OPS[name] = (forward, gradient) with the calling convention of the shim's kernel functions
# (paddlerec_amd/paddle_ops/rec_paddle_ops.cc): tensors in the op's declared input order + the attribute dict -> the
# op's declared outputs.  GPU-less runs of the reference's trainer on a net.py patched by integration/*.patch use these.
def _co_deepfm_fwd(x, attrs):
    ids, dense, W, W1, dense_w, dense_w_one = x
    y1, y2, feat, sum_emb, _ = deepfm_fm_fwd(ids, dense, W, W1, dense_w, dense_w_one, attrs["padding_idx"])
    return [y1.reshape(-1, 1), y2.reshape(-1, 1), feat, sum_emb, torch.zeros(1, dtype=torch.int32)]


def _co_deepfm_bwd(x, attrs):
    ids, dense, feat, sum_emb, dense_w, d_feat, dy1, dy2 = x
    B, S = ids.shape
    D, Dn = feat.shape[2], dense.shape[1]
    out = (torch.empty(B * S, D), torch.empty(Dn, D), torch.empty(Dn))
    deepfm_fm_bwd(dense, feat, sum_emb, d_feat, dy1, dy2, S, None, out=out)
    return [out[0], dy1.reshape(B, 1).clone(), out[1].reshape(1, Dn, D), out[2]]


def _co_cross_v2_fwd(x, attrs):
    x0, xl, W, b = x
    u = torch.empty_like(xl)
    out = crossnet_v2_layer_fwd(x0, xl, W, b, Workspace("cpu"), u=u)
    return [out, u]


def _co_cross_v2_bwd(x, attrs):
    x0, xl, W, u, d_out = x
    dx0, dW, db = torch.empty_like(xl), torch.empty_like(W), torch.empty(W.shape[1])
    dxl = crossnet_v2_layer_bwd(x0, xl, W, u, d_out, dx0, False, False, dW, db, Workspace("cpu"))
    return [dx0, dxl, dW, db]


def _co_cross_mix_fwd(x, attrs):
    x0, xl, U, V, Cm, bias, gate_w, gate_b = x
    out, t1, t2, prob = crossnet_mix_layer_fwd(x0, xl, U, V, Cm, bias.reshape(-1), gate_w, gate_b, Workspace("cpu"))
    return [out, t1, t2, prob]


def _co_cross_mix_bwd(x, attrs):
    x0, xl, U, V, Cm, bias, gate_w, t1, t2, prob, d_out = x
    dx0 = torch.empty_like(xl)
    gU, gV, gC = torch.empty_like(U), torch.empty_like(V), torch.empty_like(Cm)
    gbias, ggw, ggb = torch.empty(bias.numel()), torch.empty_like(gate_w), torch.empty(gate_w.shape[1])
    dxl = crossnet_mix_layer_bwd(x0, xl, U, V, Cm, bias.reshape(-1), gate_w, t1, t2, prob, d_out, dx0, False, False, gU, gV,
                                 gC, gbias, ggw, ggb, False, Workspace("cpu"))
    return [dx0, dxl, gU, gV, gC, gbias.reshape(bias.shape), ggw, ggb]


def _co_din_fwd(x, attrs):
    hi, hc, ti, tc, mask, w_hi, w_hc, w_ti, w_tc, w1, b1, w2, b2, w3, b3 = x
    out, att, _ = din_attention_pool(hi, hc, ti, tc, mask, w_hi, w_hc, w_ti, w_tc, [w1, w2, w3], [b1, b2, b3])
    import ctypes as C
    from paddlerec_amd import _lib
    d = _lib.DinDesc(hi.shape[0], hi.shape[1], w_hi.shape[1], w_hc.shape[1], w1.shape[1], w2.shape[1], w_hi.shape[0],
                     w_hc.shape[0], w_hi.shape[1], w_hc.shape[1])
    saves = _lib.lib().rec_din_saves_act1(C.byref(d)) == 1                   # host query: the shape the shim allocates
    act1 = torch.zeros(hi.shape[0], hi.shape[1], w1.shape[1]) if saves else torch.zeros(1)
    return [out, att, act1, torch.zeros(1, dtype=torch.int32)]


def _co_din_bwd(x, attrs):
    hi, hc, ti, tc, w_hi, w_hc, w_ti, w_tc, w1, b1, w2, b2, w3, out, att, act1, d_out = x
    dh, dq = din_attention_pool_bwd(hi, hc, ti, tc, w_hi, w_hc, w_ti, w_tc, [w1, w2, w3], [b1, b2, None], att, d_out)
    Ei, n = w_hi.shape[1], hi.numel()
    dh, dq = dh.reshape(n, -1), dq.reshape(n, -1)
    return [dh[:, :Ei].contiguous(), dh[:, Ei:].contiguous(), dq[:, :Ei].contiguous(), dq[:, Ei:].contiguous()]


def _co_multislot_fwd(x, attrs):
    values, offsets, W = x
    S, D = offsets.shape[0], int(attrs["emb_dim"])
    mb = MultislotBatch(values.reshape(-1), offsets, torch.zeros(S + 1, dtype=torch.int64))      # absolute offsets
    out, cnt, seg, rows, _ = multislot_sumpool(mb, W[:, :D], W.shape[0], attrs["padding_idx"], attrs["key_mode"])
    n = values.numel()
    return [out, cnt, seg[:n].to(torch.int32), rows[:n].to(torch.int64), torch.zeros(1, dtype=torch.int32)]


def _co_multislot_bwd(x, attrs):
    seg, d_out = x
    D = int(attrs["emb_dim"])
    return [d_out.reshape(-1, D)[seg.to(torch.int64)].contiguous()]


def _co_ps_table(rec, attrs):
    a = [float(v) for v in attrs["accessor"]]
    t = PsTable(1, int(attrs["emb_dim"]), "cpu", kind="slot", row_stride=rec.shape[1], lr=a[0], initial_g2sum=a[1],
                bounds=(a[2], a[3]), initial_range=a[4], embedx_lr=a[5], embedx_initial_g2sum=a[6],
                embedx_bounds=(a[7], a[8]), embedx_initial_range=a[9], embedx_threshold=a[10], nonclk_coeff=a[11],
                click_coeff=a[12], seed=int(a[13]))
    t.rec, t.num_rows = rec, rec.shape[0]
    return t


def _co_ps_pull_fwd(x, attrs):
    keys, rec, show_click, anchor = x
    B, S = keys.shape
    D = int(attrs["emb_dim"])
    rows = feasign_rows(keys.reshape(-1).contiguous(), rec.shape[0])
    return [rec[rows, :D].reshape(B, S, D).contiguous(), rows, torch.zeros(1, dtype=torch.int32)]


def _co_ps_pull_bwd(x, attrs):
    rows, rec, show_click, d_out = x
    B, S, D = d_out.shape
    table = _co_ps_table(rec, attrs)
    table.accessor.grad_scale = float(B)
    groups = IdGroups(rows.numel(), "cpu")
    ids_group(rows, rec.shape[0], 0, None, None, None, groups)
    click = show_click[:, 1].round().to(torch.int64)
    ps_push_rows(table, groups, d_out.reshape(B * S, D).contiguous(), S, click=click)      # in place on rec
    return [torch.zeros(1)]


CUSTOM_OPS = {
    "rec_multislot_sumpool": (_co_multislot_fwd, _co_multislot_bwd),
    "rec_ps_pull": (_co_ps_pull_fwd, _co_ps_pull_bwd),
    "rec_deepfm_fm": (_co_deepfm_fwd, _co_deepfm_bwd),
    "rec_crossnet_v2_layer": (_co_cross_v2_fwd, _co_cross_v2_bwd),
    "rec_crossnet_mix_layer": (_co_cross_mix_fwd, _co_cross_mix_bwd),
    "rec_din_attention_pool": (_co_din_fwd, _co_din_bwd),
}
