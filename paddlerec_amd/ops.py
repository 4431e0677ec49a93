"""Operator layer: torch tensors in, C-ABI calls out.

torch is plumbing here (device memory, streams); every op below is a hand-written HIP kernel in
librecengine.so.  Tensors must live on a ROCm device — there is deliberately no CPU path.
Reference call sites are cited in include/recengine.h next to each entry point.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (EPI, AdagradHyper, AdamHyper, CinView, DeepFMDesc, DinDesc, GemmBImage, GemmDesc, GemmEpilogueArgs, GradLayout,
                   GradSrc, LazyInit, MultislotDesc, PsAccessor, PsLayout, RecError, check)

_recorder = None        # paddlerec_amd.plan.CallPlan while a step is being recorded


def lib():
    """The loaded library — or, while a step is recorded (plan.py), a proxy that also lists every call."""
    h = _lib.lib()
    return h if _recorder is None else _recorder.proxy(h)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() builds a Stream object through four python
    layers (7.7 of the 24 profiled microseconds of one ops.gemm call); the raw accessors are one C call each."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    p = C.c_void_p(0 if t is None else t.data_ptr())
    if _recorder is not None and t is not None:
        _recorder.note_pointer(p, t)
    return p


_T_CACHE = {}


def _transposed(w):
    """w.t().contiguous(), kept while THIS tensor object is unchanged (the attention MLP is frozen in DIN's dygraph
    mode: one transpose per set_attention instead of one torch kernel per step)."""
    import weakref
    hit = _T_CACHE.get(id(w))
    if hit is None or hit[0]() is not w or hit[1] != w._version:
        if len(_T_CACHE) > 64:
            _T_CACHE.clear()
        hit = _T_CACHE[id(w)] = (weakref.ref(w), w._version, w.t().contiguous())
    return hit[2]


def copy_f32(dst, src):
    """dst[...] = src[...] (contiguous f32 device tensors of one size) as a C-ABI call (rec_copy_async)."""
    _chk(dst, torch.float32, "dst")
    _chk(src, torch.float32, "src")
    if dst.numel() != src.numel():
        raise RecError("copy_f32: %d != %d elements" % (dst.numel(), src.numel()))
    check(lib().rec_copy_async(_p(dst), _p(src), C.c_size_t(dst.numel() * 4), _stream()), "rec_copy_async")
    return dst


def _chk(t, dtype, name, shape=None):
    if t is None:
        return
    if not t.is_cuda:
        raise RecError("%s must be a device tensor (no CPU fallback)" % name)
    if t.dtype != dtype:
        raise RecError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RecError("%s must be contiguous" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RecError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))


def _chk_table(t, name):
    """A table is f32 [N,D] (or [N]) on the device with unit column stride; rows may be strided
    (a column slice of a wider record buffer).  Returns (D, row_stride)."""
    if not t.is_cuda:
        raise RecError("%s must be a device tensor (no CPU fallback)" % name)
    if t.dtype != torch.float32:
        raise RecError("%s must be float32, got %s" % (name, t.dtype))
    if t.dim() == 1:
        return 1, (t.stride(0) if t.numel() > 1 else 1)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RecError("%s must be [N,D] with unit column stride" % name)
    D, rs = t.shape[1], (t.stride(0) if t.shape[0] > 1 else t.shape[1])
    if rs < D:
        raise RecError("%s: row stride %d < D %d" % (name, rs, D))
    if D % 4 == 0 and rs % 4 == 0 and t.data_ptr() % 16 != 0:
        raise RecError("%s: rows must be 16-byte aligned" % name)
    return D, rs


def new_status(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def raise_on_status(status, what="lookup"):
    """Host sync: raises if a kernel flagged an out-of-range id (Paddle raises on OOB [EXT])."""
    s = int(status.item())
    if s & _lib.REC_FLAG_INDEX_OOB:
        raise RecError("%s: index out of range [0, num_rows)" % what)
    if s & _lib.REC_FLAG_EXCHANGE_OVERFLOW:
        raise RecError("%s: a rank needed more distinct rows of one owner than the deduplicated exchange's capacity "
                       "(REC_SHARD_DEDUP_CAP)" % what)


class Workspace:
    """Grow-only device scratch owned by the caller of the C-ABI (the engine never allocates)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
        return self.buf


def deepfm_train_step(net, ids, dense, label, step, ws, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, auc_stats=None,
                      num_thresholds=4095, status=None, out=None, side_stream=None):
    """The whole DeepFM train step through ONE C-ABI call (rec_deepfm_train_step).  net: a filled _lib.DeepFMNet (the
    caller keeps the tensors it points into alive).  side_stream (torch stream or None): the large-batch schedule of the
    mirror (grouping and sparse update beside the main stream's GEMMs).  -> (loss [1], pred [B,1])."""
    _chk(ids, torch.int64, "ids")
    _chk(dense, torch.float32, "dense")
    _chk(label, torch.int64, "label")
    B = ids.shape[0]
    if ids.shape[1] != net.num_slots or dense.shape[0] != B or label.numel() != B:
        raise RecError("ids / dense / label shapes do not fit the net")
    dev = ids.device
    if out is None:
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, 1, dtype=torch.float32, device=dev)
    else:
        loss, pred = out
    nbytes = C.c_size_t(0)
    check(lib().rec_deepfm_train_step_workspace_bytes(C.byref(net), B, C.byref(nbytes)),
          "rec_deepfm_train_step_workspace_bytes")
    w = ws.get(nbytes.value)
    h = _hyper(lr, beta1, beta2, eps, step)
    pos, neg = (auc_stats[0], auc_stats[1]) if auc_stats is not None else (None, None)
    check(lib().rec_deepfm_train_step(C.byref(net), B, _p(ids), _p(dense), _p(label), C.byref(h), _p(pos), _p(neg),
                                      int(num_thresholds), _p(loss), _p(pred), _p(status), _p(w),
                                      C.c_size_t(w.numel()), _stream(),
                                      C.c_void_p(side_stream.cuda_stream) if side_stream is not None else None),
          "rec_deepfm_train_step")
    return loss, pred


def din_train_step(net, hist_item, hist_cat, target_item, target_cat, label, mask, target_item_seq, target_cat_seq, lr, ws,
                   status=None, out=None, side_stream=None):
    """The whole DIN train step through ONE C-ABI call (rec_din_train_step).  net: a filled _lib.DinNet (the caller keeps
    the tensors it points into alive).  ids / mask [B,T] i64, targets [B] i64, label [B] f32.  -> (loss [1], pred [B,1])."""
    B, T = hist_item.shape
    for t, n in ((hist_item, "hist_item"), (hist_cat, "hist_cat"), (target_item_seq, "target_item_seq"),
                 (target_cat_seq, "target_cat_seq"), (mask, "mask")):
        _chk(t, torch.int64, n, (B, T))
    for t, n in ((target_item, "target_item"), (target_cat, "target_cat")):
        _chk(t, torch.int64, n)
        if t.numel() != B:
            raise RecError("%s must have one id per sample" % n)
    _chk(label, torch.float32, "label")
    if label.numel() != B:
        raise RecError("label must have one value per sample")
    dev = hist_item.device
    if out is None:
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, 1, dtype=torch.float32, device=dev)
    else:
        loss, pred = out
    if status is None:
        status = new_status(dev)
    nbytes = C.c_size_t(0)
    check(lib().rec_din_train_step_workspace_bytes(C.byref(net), B, T, C.byref(nbytes)), "rec_din_train_step_workspace_bytes")
    w = ws.get(nbytes.value)
    check(lib().rec_din_train_step(C.byref(net), B, T, _p(hist_item), _p(hist_cat), _p(target_item), _p(target_cat),
                                   _p(label), _p(mask), _p(target_item_seq), _p(target_cat_seq), float(lr), _p(loss),
                                   _p(pred), _p(status), _p(w), C.c_size_t(w.numel()), _stream(),
                                   None if side_stream is None else C.c_void_p(side_stream.cuda_stream)), "rec_din_train_step")
    return loss, pred


def dcn_v2_train_step(net, ids, dense, label, step, lr, ws, auc_stats=None, num_thresholds=4095, status=None, out=None,
                      beta1=0.9, beta2=0.999, eps=1e-8):
    """The whole DCN-v2 train step through ONE C-ABI call (rec_dcn_v2_train_step).  net: a filled _lib.DcnV2Net (the
    caller keeps the tensors it points into alive).  ids [B,S] i64, dense [B,Dn] f32, label [B(,1)] i64; step = Adam step
    count (1-based).  -> (loss [1], pred [B,1])."""
    _chk(ids, torch.int64, "ids")
    if ids.dim() != 2 or ids.shape[1] != net.num_slots:
        raise RecError("ids must be [batch, num_slots]")
    B = ids.shape[0]
    _chk(dense, torch.float32, "dense", (B, net.dense_dim))
    _chk(label, torch.int64, "label")
    if label.numel() != B:
        raise RecError("label must have one value per sample")
    dev = ids.device
    if out is None:
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, 1, dtype=torch.float32, device=dev)
    else:
        loss, pred = out
    if status is None:
        status = new_status(dev)
    pos = neg = None
    if auc_stats is not None:
        pos, neg = auc_stats
        _chk(pos, torch.int64, "stat_pos", (num_thresholds + 1,))
        _chk(neg, torch.int64, "stat_neg", (num_thresholds + 1,))
    h = _hyper(lr, beta1, beta2, eps, step)
    nbytes = C.c_size_t(0)
    check(lib().rec_dcn_v2_train_step_workspace_bytes(C.byref(net), B, C.byref(nbytes)),
          "rec_dcn_v2_train_step_workspace_bytes")
    w = ws.get(nbytes.value)
    check(lib().rec_dcn_v2_train_step(C.byref(net), B, _p(ids), _p(dense), _p(label), C.byref(h), _p(pos), _p(neg),
                                      int(num_thresholds), _p(loss), _p(pred), _p(status), _p(w), C.c_size_t(w.numel()),
                                      _stream()), "rec_dcn_v2_train_step")
    return loss, pred


SUPPORTS_FEAT_LD = True       # deepfm_fm_fwd / _bwd take feat_ld (a padded sample stride of feat)


def make_desc(B, S, Dn, D, num_rows, padding_idx, row_stride=None, w1_stride=1, compact=False, feat_stride=0):
    return DeepFMDesc(int(B), int(S), int(Dn), int(D), int(row_stride or D), int(num_rows),
                      -1 if padding_idx is None else int(padding_idx), int(w1_stride), int(bool(compact)),
                      int(feat_stride))


# ------------------------------------------------------------------ DeepFM FM block
def deepfm_fm_fwd(ids, dense, W, W1, dense_w, dense_w_one, padding_idx=0, slot_offset=None,
                  status=None, out=None, compact=False, feat_ld=0):
    """ids [B,S] i64, dense [B,Dn] f32, W [N,D], W1 [N,1]|[N], dense_w [Dn,D]|[1,Dn,D], dense_w_one [Dn]
    -> y1 [B,1], y2 [B,1], feat [B,S+Dn,D], sum_emb [B,D], status
    compact: feat is [B,S+1,D] — S embedding rows + one row of the raw dense values (rec_deepfm_desc.compact_dense)."""
    B, S = ids.shape
    Dn = dense.shape[1]
    N, D = W.shape
    dev = ids.device
    _chk(ids, torch.int64, "ids")
    _chk(dense, torch.float32, "dense", (B, Dn))
    _, w_stride = _chk_table(W, "W")
    _, w1_stride = _chk_table(W1, "W1")
    _chk(dense_w, torch.float32, "dense_w")
    _chk(dense_w_one, torch.float32, "dense_w_one", (Dn,))
    _chk(slot_offset, torch.int64, "slot_offset", (S,))
    if W1.numel() != N or dense_w.numel() != Dn * D:
        raise RecError("W1 / dense_w shape mismatch")
    if out is None:
        y1 = torch.empty(B, 1, dtype=torch.float32, device=dev)
        y2 = torch.empty(B, 1, dtype=torch.float32, device=dev)
        feat = torch.empty(B, S + 1 if compact else S + Dn, D, dtype=torch.float32, device=dev)
        sum_emb = torch.empty(B, D, dtype=torch.float32, device=dev)
    else:
        y1, y2, feat, sum_emb = out
    if status is None:
        status = new_status(dev)
    if feat_ld:          # feat_ld: feat is a caller-kept zero-initialised [B, feat_ld] buffer (rec_deepfm_desc.feat_stride)
        if out is None or feat.dim() != 2 or feat.shape[1] != feat_ld or not feat.is_contiguous():
            raise RecError("feat_ld needs out=(y1, y2, feat [B, feat_ld], sum_emb)")
    desc = make_desc(B, S, Dn, D, N, padding_idx, w_stride, w1_stride, compact, feat_ld)
    check(lib().rec_deepfm_fm_fwd(C.byref(desc), _p(ids), _p(dense), _p(W), _p(W1), _p(dense_w),
                                  _p(dense_w_one), _p(slot_offset), _p(y1), _p(y2), _p(feat),
                                  _p(sum_emb), _p(status), _stream()), "rec_deepfm_fm_fwd")
    return y1, y2, feat, sum_emb, status


def deepfm_fm_bwd(dense, feat, sum_emb, d_feat_dnn, dy1, dy2, S, ws, out=None, dense_w=None, compact=False,
                  row_rank=None, feat_ld=0):
    """-> row_grad [B*S,D], d_dense_w [Dn,D], d_dense_w_one [Dn].
    dense_w ([Dn,D] / [1,Dn,D], optional): recompute the dense part of feat instead of re-reading it.
    compact: feat / d_feat_dnn are [B,S+1,D] and d_dense_w is the FM part only (see deepfm_fm_fwd).
    row_rank [B*S] i32 (IdGroups.rank): row_grad is written in SORTED order (rec_deepfm_fm_bwd_sorted)."""
    if feat_ld:          # feat / d_feat_dnn are [B, feat_ld] (padded sample stride)
        B, D = sum_emb.shape
        Dn = dense.shape[1]
        F = S + 1 if compact else S + Dn
        if feat.shape != (B, feat_ld) or d_feat_dnn.shape != (B, feat_ld) or feat_ld < F * D:
            raise RecError("feat / d_feat_dnn must be [B, feat_ld] with feat_ld >= fields x emb_dim")
    else:
        B, F, D = feat.shape
        Dn = dense.shape[1] if compact else F - S
    dev = feat.device
    for t, n in ((dense, "dense"), (feat, "feat"), (sum_emb, "sum_emb"), (d_feat_dnn, "d_feat_dnn"),
                 (dy1, "dy1"), (dy2, "dy2")):
        _chk(t, torch.float32, n)
    if d_feat_dnn.numel() != feat.numel() or dy1.numel() != B or dy2.numel() != B:
        raise RecError("gradient shape mismatch")
    if not (feat.is_contiguous() and d_feat_dnn.is_contiguous()):
        raise RecError("feat / d_feat_dnn must be contiguous")
    if dense_w is not None:
        _chk(dense_w, torch.float32, "dense_w")
        if dense_w.numel() != Dn * D:
            raise RecError("dense_w shape mismatch")
    if out is None:
        row_grad = torch.empty(B * S, D, dtype=torch.float32, device=dev)
        d_dense_w = torch.empty(Dn, D, dtype=torch.float32, device=dev)
        d_dense_w_one = torch.empty(Dn, dtype=torch.float32, device=dev)
    else:
        row_grad, d_dense_w, d_dense_w_one = out
    desc = make_desc(B, S, Dn, D, 1, None, compact=compact, feat_stride=feat_ld)
    nbytes = C.c_size_t(0)
    check(lib().rec_deepfm_fm_bwd_workspace_bytes(C.byref(desc), C.byref(nbytes)))
    w = ws.get(nbytes.value)
    if row_rank is not None:
        _chk(row_rank, torch.int32, "row_rank")
        if row_rank.numel() < B * S:
            raise RecError("row_rank shorter than B*S")
        check(lib().rec_deepfm_fm_bwd_sorted(C.byref(desc), _p(dense), _p(feat), _p(sum_emb), _p(d_feat_dnn),
                                             _p(dy1), _p(dy2), _p(dense_w), _p(row_rank), _p(row_grad),
                                             _p(d_dense_w), _p(d_dense_w_one), _p(w), C.c_size_t(w.numel()),
                                             _stream()), "rec_deepfm_fm_bwd_sorted")
        return row_grad, d_dense_w, d_dense_w_one
    check(lib().rec_deepfm_fm_bwd(C.byref(desc), _p(dense), _p(feat), _p(sum_emb), _p(d_feat_dnn),
                                  _p(dy1), _p(dy2), _p(dense_w), _p(row_grad), _p(d_dense_w),
                                  _p(d_dense_w_one),
                                  _p(w), C.c_size_t(w.numel()), _stream()), "rec_deepfm_fm_bwd")
    return row_grad, d_dense_w, d_dense_w_one


def dense_fold_fwd(S, dense_w, W0, M):
    """M[j,:] = dense_w[j,:] @ W0[(S+j)*D:(S+j+1)*D, :]   (dense embeddings folded into MLP layer 0)."""
    Dn, D = dense_w.shape[-2], dense_w.shape[-1]
    for t, n in ((dense_w, "dense_w"), (W0, "W0")):
        _chk(t, torch.float32, n)
    if not M.is_cuda or M.dtype != torch.float32 or not M.is_contiguous() or tuple(M.shape) != (Dn, W0.shape[1]):
        raise RecError("M must be a contiguous float32 device tensor [Dn, n_out]")
    check(lib().rec_dense_fold_fwd(int(S), Dn, D, W0.shape[1], _p(dense_w), _p(W0), _p(M), _stream()),
          "rec_dense_fold_fwd")
    return M


def dense_fold_bwd(S, dense_w, W0, dM, dW0, d_dense_w, accumulate=True):
    """dW0 dense rows = dense_w (x) dM;  d_dense_w (+)= dM @ W0_dense^T."""
    Dn, D = dense_w.shape[-2], dense_w.shape[-1]
    for t, n in ((dense_w, "dense_w"), (W0, "W0"), (dM, "dM"), (dW0, "dW0"), (d_dense_w, "d_dense_w")):
        _chk(t, torch.float32, n)
    check(lib().rec_dense_fold_bwd(int(S), Dn, D, W0.shape[1], _p(dense_w), _p(W0), _p(dM), _p(dW0),
                                   _p(d_dense_w), int(accumulate), _stream()), "rec_dense_fold_bwd")


def dense_fold_fwd_full(S, dense_w, W0, W0_folded):
    """W0_folded [(S+1)*D, n_out]: sparse rows of W0 + the folded dense rows M, one launch."""
    Dn, D = dense_w.shape[-2], dense_w.shape[-1]
    for t, n in ((dense_w, "dense_w"), (W0, "W0"), (W0_folded, "W0_folded")):
        _chk(t, torch.float32, n)
    if tuple(W0_folded.shape) != ((S + 1) * D, W0.shape[1]) or W0.shape[0] != (S + Dn) * D or Dn > D:
        raise RecError("W0 must be [(S+Dn)*D, n_out] and W0_folded [(S+1)*D, n_out] with Dn <= D")
    check(lib().rec_dense_fold_fwd_full(int(S), Dn, D, W0.shape[1], _p(dense_w), _p(W0), _p(W0_folded), _stream()),
          "rec_dense_fold_fwd_full")
    return W0_folded


def dense_fold_bwd_full(S, dense_w, W0, dW0_folded, dW0, d_dense_w, accumulate=True):
    """dW0_folded = feat'^T dZ0 [(S+1)*D, n_out] (scratch) -> dW0 (sparse rows copied, dense rows = dense_w (x) dM) and
    d_dense_w (+)= dM @ W0_dense^T, one launch."""
    Dn, D = dense_w.shape[-2], dense_w.shape[-1]
    for t, n in ((dense_w, "dense_w"), (W0, "W0"), (dW0_folded, "dW0_folded"), (dW0, "dW0"), (d_dense_w, "d_dense_w")):
        _chk(t, torch.float32, n)
    if (tuple(dW0_folded.shape) != ((S + 1) * D, W0.shape[1]) or tuple(dW0.shape) != tuple(W0.shape)
            or W0.shape[0] != (S + Dn) * D or Dn > D):
        raise RecError("dW0 must be [(S+Dn)*D, n_out] and dW0_folded [(S+1)*D, n_out] with Dn <= D")
    check(lib().rec_dense_fold_bwd_full(int(S), Dn, D, W0.shape[1], _p(dense_w), _p(W0), _p(dW0_folded), _p(dW0),
                                        _p(d_dense_w), int(accumulate), _stream()), "rec_dense_fold_bwd_full")


# ------------------------------------------------------------------ lookups
def emb_gather(ids, W, padding_idx=None, status=None, out=None, out_group=0, out_group_stride=0):
    """out[i,:] = W[ids[i],:] (zero row where ids[i]==padding_idx).  W [N,D] f32.
    out_group/out_group_stride: write lookup i at out + (i//group)*stride + (i%group)*D (floats)."""
    _chk(ids, torch.int64, "ids")
    _, w_stride = _chk_table(W, "W")
    N, D = W.shape
    if out is None:
        out = torch.empty(*ids.shape, D, dtype=torch.float32, device=ids.device)
    else:
        if out_group <= 0:
            _chk(out, torch.float32, "out")
            if out.numel() != ids.numel() * D:
                raise RecError("out has %d elements, expected %d" % (out.numel(), ids.numel() * D))
        elif not out.is_cuda or out.dtype != torch.float32:
            raise RecError("out must be a float32 device tensor")
    if status is None:
        status = new_status(ids.device)
    check(lib().rec_emb_gather(ids.numel(), D, w_stride, N,
                               -1 if padding_idx is None else padding_idx, _p(ids), _p(W), _p(out),
                               int(out_group), int(out_group_stride), _p(status), _stream()),
          "rec_emb_gather")
    return out, status


def emb_gather_sumpool(ids, lod, W, padding_idx=0, status=None):
    """ids [nnz] i64, lod [B+1] i64 -> (bow [B,D], counts [B] i32, status)"""
    _chk(ids, torch.int64, "ids")
    _chk(lod, torch.int64, "lod")
    _chk(W, torch.float32, "W")
    N, D = W.shape
    B = lod.numel() - 1
    out = torch.empty(B, D, dtype=torch.float32, device=W.device)
    counts = torch.empty(B, dtype=torch.int32, device=W.device)
    if status is None:
        status = new_status(W.device)
    check(lib().rec_emb_gather_sumpool(B, D, W.stride(0), N,
                                       -1 if padding_idx is None else padding_idx, _p(ids), _p(lod),
                                       _p(W), _p(out), _p(counts), _p(status), _stream()),
          "rec_emb_gather_sumpool")
    return out, counts, status


def emb_sumpool_bwd(lod, d_out, nnz):
    _chk(lod, torch.int64, "lod")
    _chk(d_out, torch.float32, "d_out")
    B, D = d_out.shape
    row_grad = torch.empty(nnz, D, dtype=torch.float32, device=d_out.device)
    check(lib().rec_emb_sumpool_bwd(B, D, _p(lod), _p(d_out), _p(row_grad), _stream()),
          "rec_emb_sumpool_bwd")
    return row_grad


class MultislotBatch:
    """Device-resident slot-major CSR of one batch (the layout rec_parse_feasign_slots writes):
    values [nnz] i64, lod [S, B+1] i64, slot_base [S+1] i64."""

    def __init__(self, values, lod, slot_base):
        _chk(values, torch.int64, "values")
        _chk(lod, torch.int64, "lod")
        _chk(slot_base, torch.int64, "slot_base")
        if lod.dim() != 2 or slot_base.numel() != lod.shape[0] + 1:
            raise RecError("lod must be [S, B+1] and slot_base [S+1]")
        self.values, self.lod, self.slot_base = values, lod, slot_base
        self.num_slots, self.batch = lod.shape[0], lod.shape[1] - 1
        self.nnz = values.numel()


def multislot_sumpool(mb, W, num_rows=None, padding_idx=0, key_mode=0, status=None, out=None, want_counts=True,
                      want_backward=True, lazy_init=None):
    """All slots of a batch in one launch (slot_dnn/net.py:63-77): -> (out [B, S*D], counts [B,S] i32 | None,
    seg_of_value [nnz] i32 | None, rows [nnz] i64 | None, status).  W [N,D] table view (row stride allowed);
    key_mode 1: values are uint64 feasigns hashed to rows on the device (feasign_rows).
    lazy_init = (state_offset, init_dims, init_range, seed): PS rows are born at their first pull (PsTable.lazy_init)."""
    D, stride = _chk_table(W, "W")
    N = int(num_rows if num_rows is not None else W.shape[0])
    B, S = mb.batch, mb.num_slots
    dev = W.device
    if B * S >= 2 ** 31:
        raise RecError("batch x slots must stay below 2^31")
    if out is None:
        out = torch.empty(B, S * D, dtype=torch.float32, device=dev)
    else:       # [B, ld] with ld >= S * D: the kernel writes the first S * D columns of a row (rec_multislot_desc.out_stride)
        _chk(out, torch.float32, "out")
        if out.dim() != 2 or out.shape[0] != B or out.shape[1] < S * D:
            raise RecError("out must be [batch, >= slots * emb_dim]")
    counts = torch.empty(B, S, dtype=torch.int32, device=dev) if want_counts else None
    seg = torch.empty(max(mb.nnz, 1), dtype=torch.int32, device=dev) if want_backward else None
    rows = torch.empty(max(mb.nnz, 1), dtype=torch.int64, device=dev) if want_backward else None
    if status is None:
        status = new_status(dev)
    d = MultislotDesc(B, S, D, stride, int(key_mode), N, -1 if padding_idx is None else int(padding_idx),
                      mb.lod.stride(0), out.shape[1], *(lazy_init if lazy_init is not None else (0, 0, 0.0, 0)))
    check(lib().rec_multislot_sumpool_fwd(C.byref(d), _p(mb.values), _p(mb.lod), _p(mb.slot_base), _p(W), _p(out),
                                          _p(counts), _p(seg), _p(rows), _p(status), _stream()),
          "rec_multislot_sumpool_fwd")
    return out, counts, seg, rows, status


def multislot_sumpool_bwd(seg_of_value, d_out, num_slots, emb_dim, nnz=None):
    """Rows form of the pool's gradient: row_grad [nnz, D], row k = d_out[b, s*D:(s+1)*D] of value k's segment b*S+s
    (what a binder that needs the SelectedRows value as a dense tensor hands to the optimizer; rows = the forward's)."""
    _chk(seg_of_value, torch.int32, "seg_of_value")
    _chk(d_out, torch.float32, "d_out")
    S, D = int(num_slots), int(emb_dim)
    if d_out.dim() != 2 or d_out.shape[1] < S * D or d_out.stride(1) != 1:
        raise RecError("d_out must be [batch, >= slots * emb_dim]")
    n = int(seg_of_value.numel() if nnz is None else nnz)
    row_grad = torch.empty(n, D, dtype=torch.float32, device=d_out.device)
    d = MultislotDesc(d_out.shape[0], S, D, D, 0, 1, -1, 0, d_out.stride(0), 0, 0, 0.0, 0)
    check(lib().rec_multislot_sumpool_bwd(C.byref(d), n, _p(seg_of_value), _p(d_out), _p(row_grad), _stream()),
          "rec_multislot_sumpool_bwd")
    return row_grad


def feasign_rows(keys, num_rows, out=None):
    """uint64 feasign bit patterns (int64 tensor) -> rows of a hashed table: 0 -> 0, f -> 1 + mix64(f) % (N-1)."""
    _chk(keys, torch.int64, "keys")
    if out is None:
        out = torch.empty_like(keys)
    check(lib().rec_feasign_rows(keys.numel(), int(num_rows), _p(keys), _p(out), _stream()), "rec_feasign_rows")
    return out


def feasign_rows_host(keys, num_rows):
    """Host variant over a numpy uint64 array (bit-identical to the device kernel)."""
    import numpy as np
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    out = np.empty(k.shape, np.int64)
    check(lib().rec_feasign_rows_host(k.size, int(num_rows), k.ctypes.data_as(C.c_void_p),
                                      out.ctypes.data_as(C.c_void_p)), "rec_feasign_rows_host")
    return out


# ------------------------------------------------------------------ SelectedRows merge + optimizers
class IdGroups:
    """Result buffers of rec_ids_group (device)."""

    def __init__(self, n, device):
        self.n = n
        self.sorted_pos = torch.empty(max(n, 1), dtype=torch.int32, device=device)
        self.uniq_rows = torch.empty(max(n, 1), dtype=torch.int64, device=device)
        self.seg_offset = torch.empty(n + 1, dtype=torch.int32, device=device)
        self.n_uniq = torch.zeros(4, dtype=torch.int32, device=device)
        self.rank = None            # [n] i32, filled by ids_group_slots / ids_rank when asked for

    def want_rank(self):
        if self.rank is None:
            self.rank = torch.empty(max(self.n, 1), dtype=torch.int32, device=self.sorted_pos.device)
        return self.rank

    def host(self):
        """(sorted_pos[:n_valid], uniq_rows[:U], seg_offset[:U+1]) on the CPU — host sync."""
        U, nv = (int(x) for x in self.n_uniq.tolist()[:2])
        return (self.sorted_pos[:nv].cpu().numpy(), self.uniq_rows[:U].cpu().numpy(),
                self.seg_offset[:U + 1].cpu().numpy())


def ids_group(ids, num_rows, padding_idx, ws, slot_offset=None, status=None, groups=None, payload=None):
    """payload [n] int32 (or None): what groups.sorted_pos holds for every lookup instead of its position
    (rec_ids_group_payload) — the multi-slot path passes the (sample, slot) segment of every value."""
    _chk(ids, torch.int64, "ids")
    if payload is not None:
        _chk(payload, torch.int32, "payload")
        if payload.numel() < ids.numel():
            raise RecError("payload shorter than ids")
    n = ids.numel()
    S = ids.shape[-1] if ids.dim() > 1 else 1
    dev = ids.device
    if groups is None:
        groups = IdGroups(n, dev)
    if status is None:
        status = new_status(dev)
    nbytes = C.c_size_t(0)
    check(lib().rec_ids_group_workspace_bytes(n, num_rows, C.byref(nbytes)))
    w = ws.get(nbytes.value)
    check(lib().rec_ids_group_payload(n, S, num_rows, -1 if padding_idx is None else padding_idx, _p(ids),
                                      _p(slot_offset), _p(payload), _p(groups.sorted_pos), _p(groups.uniq_rows),
                                      _p(groups.seg_offset), _p(groups.n_uniq), _p(status), _p(w),
                                      C.c_size_t(w.numel()), _stream()), "rec_ids_group_payload")
    return groups, status


def ids_group_slots(ids, slot_rows, padding_idx, ws, status=None, groups=None, want_rank=False):
    """ids [B,S] whose slot s owns rows [s*slot_rows, (s+1)*slot_rows) (rec_ids_group_slots): the grouping of
    ids_group(ids, S*slot_rows, ..., slot_offset = arange(S)*slot_rows), slot-local sort; want_rank: groups.rank[pos] =
    sorted index of lookup pos (-1 = dropped)."""
    _chk(ids, torch.int64, "ids")
    if ids.dim() != 2:
        raise RecError("ids must be [B, S]")
    B, S = ids.shape
    dev = ids.device
    if groups is None:
        groups = IdGroups(B * S, dev)
    if status is None:
        status = new_status(dev)
    rank = groups.want_rank() if want_rank else None
    nbytes = C.c_size_t(0)
    check(lib().rec_ids_group_slots_workspace_bytes(B, S, int(slot_rows), C.byref(nbytes)))
    w = ws.get(nbytes.value)
    check(lib().rec_ids_group_slots(B, S, int(slot_rows), -1 if padding_idx is None else padding_idx, _p(ids),
                                    _p(groups.sorted_pos), _p(groups.uniq_rows), _p(groups.seg_offset),
                                    _p(groups.n_uniq), _p(rank), _p(status), _p(w), C.c_size_t(w.numel()), _stream()),
          "rec_ids_group_slots")
    return groups, status


def group_slots_eligible(B, S, slot_rows):
    """True when rec_ids_group_slots sorts slot by slot for this shape (csrc/ids_group_slots.hip sg::eligible)."""
    return (B >= 8192 and 1 <= S <= 60 and 2 <= slot_rows <= (1 << 20) and B * S < 2 ** 31 - 1 and B <= 32 * 8192
            and S * slot_rows < 2 ** 32 - 1 and os.environ.get("REC_GROUP_SLOTS", "1") != "0")


def ids_rank(groups):
    """groups.rank[pos] = k with sorted_pos[k] = pos (-1 for dropped lookups) for a grouping made without payload."""
    rank = groups.want_rank()
    check(lib().rec_ids_rank(groups.n, _p(groups.n_uniq), _p(groups.sorted_pos), _p(rank), _stream()), "rec_ids_rank")
    return rank


def _gl(div, group, group_stride, partials=None, index=None, sorted_=False):
    if index is not None:
        _chk(index, torch.int32, "grad index")
    return GradLayout(int(div), int(group), int(group_stride), partials.data_ptr() if partials is not None else None,
                      index.data_ptr() if index is not None else None, int(bool(sorted_)))


def segment_partials(groups, grad, D, grad_div=1, grad_group=0, grad_group_stride=0, out=None, grad_index=None,
                     grad_sorted=False):
    """Tile partial sums of the long segments (hot rows) of `grad` under `groups` -> tensor to pass as
    `partials=` to the row-update ops together with the SAME grad and layout."""
    nbytes = C.c_size_t(0)
    check(lib().rec_segment_partials_bytes(groups.n, int(D), C.byref(nbytes)))
    need = max(nbytes.value // 4, 1)
    if out is None or out.numel() < need:
        out = torch.empty(need, dtype=torch.float32, device=grad.device)
    check(lib().rec_segment_partials(groups.n, int(D), _p(groups.n_uniq), _p(groups.seg_offset),
                                     _p(groups.sorted_pos), _p(grad),
                                     C.byref(_gl(grad_div, grad_group, grad_group_stride, None, grad_index, grad_sorted)),
                                     _p(out), _stream()),
          "rec_segment_partials")
    return out


def _hyper(lr, beta1, beta2, eps, step):
    return AdamHyper(float(lr), float(beta1), float(beta2), float(eps), int(step))


def sparse_adam_rows(groups, grad, grad_div, P, M, V, step, lr=1e-3, beta1=0.9, beta2=0.999,
                     eps=1e-8, grad_group=0, grad_group_stride=0, grad_scale=None, partials=None, grad_index=None):
    """grad_div / grad_group / grad_group_stride: rec_grad_layout (where position pos's row lives in
    grad); grad_scale: device float[1] clipping coefficient or None."""
    if grad_group <= 0:
        _chk(grad, torch.float32, "grad")
    elif not grad.is_cuda or grad.dtype != torch.float32:
        raise RecError("grad must be a float32 device tensor")
    D, stride = _chk_table(P, "P")
    sstride = _chk_table(M, "M")[1]
    for t, n in ((M, "M"), (V, "V")):
        if _chk_table(t, n) != (D, sstride) or t.shape[0] != P.shape[0]:
            raise RecError("%s must have the shape of P (M and V share one row stride)" % n)
    h = _hyper(lr, beta1, beta2, eps, step)
    check(lib().rec_sparse_adam_rows(groups.n, D, stride, sstride, _p(groups.n_uniq), _p(groups.uniq_rows),
                                     _p(groups.seg_offset), _p(groups.sorted_pos), _p(grad),
                                     C.byref(_gl(grad_div, grad_group, grad_group_stride, partials, grad_index)),
                                     _p(grad_scale), _p(P), _p(M), _p(V), C.byref(h), _stream()),
          "rec_sparse_adam_rows")


def sparse_adam_record(groups, grad, grad1, grad1_div, rec, mv, D, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                       v_offset=None, grad_scale=None, partials=None, partials1=None, grad_sorted=False):
    """Lazy Adam on BOTH embeddings of a DeepFM row in one pass: rec [N, stride] = W(D) | W1 | m1 | v1 | pad,
    mv [N, sstride] = m(D) | v(D) at v_offset.  grad [n,D] row gradients, grad1 = dz with layout {grad1_div,0,0}."""
    _chk(grad, torch.float32, "grad")
    _chk(grad1, torch.float32, "grad1")
    for t, n in ((rec, "rec"), (mv, "mv")):
        if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
            raise RecError("%s must be a 2-D float32 device tensor with unit column stride" % n)
    if mv.shape[0] != rec.shape[0]:
        raise RecError("rec and mv must have the same number of rows")
    if v_offset is None:
        v_offset = (D + 3) // 4 * 4
    h = _hyper(lr, beta1, beta2, eps, step)
    check(lib().rec_sparse_adam_record(groups.n, int(D), rec.stride(0), mv.stride(0), int(v_offset),
                                       _p(groups.n_uniq), _p(groups.uniq_rows), _p(groups.seg_offset),
                                       _p(groups.sorted_pos), _p(grad),
                                       C.byref(_gl(1, 0, 0, partials, None, grad_sorted)), _p(grad1),
                                       C.byref(_gl(grad1_div, 0, 0, partials1)), _p(grad_scale), _p(rec), _p(mv),
                                       C.byref(h), _stream()), "rec_sparse_adam_record")


def adam_record_all(groups, grad, grad1, grad1_div, rec, mv, D, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                    v_offset=None, grad_scale=None, partials=None, partials1=None):
    """lazy_mode=False Adam (dygraph default) on BOTH embeddings of every row of the record layout of sparse_adam_record in
    one pass (rec_adam_record_all)."""
    _chk(grad, torch.float32, "grad")
    _chk(grad1, torch.float32, "grad1")
    for t, n in ((rec, "rec"), (mv, "mv")):
        if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
            raise RecError("%s must be a 2-D float32 device tensor with unit column stride" % n)
    if mv.shape[0] != rec.shape[0]:
        raise RecError("rec and mv must have the same number of rows")
    if v_offset is None:
        v_offset = (D + 3) // 4 * 4
    h = _hyper(lr, beta1, beta2, eps, step)
    check(lib().rec_adam_record_all(rec.shape[0], int(D), rec.stride(0), mv.stride(0), int(v_offset),
                                    _p(groups.n_uniq), _p(groups.uniq_rows), _p(groups.seg_offset),
                                    _p(groups.sorted_pos), _p(grad), C.byref(_gl(1, 0, 0, partials)), _p(grad1),
                                    C.byref(_gl(grad1_div, 0, 0, partials1)), _p(grad_scale), _p(rec), _p(mv),
                                    C.byref(h), _stream()), "rec_adam_record_all")


def adam_rows_all(groups, grad, grad_div, P, M, V, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                  grad_group=0, grad_group_stride=0, grad_scale=None, partials=None):
    """lazy_mode=False Adam (dygraph default): every row of P/M/V moves, absent rows with g = 0."""
    D, stride = _chk_table(P, "P")
    sstride = _chk_table(M, "M")[1]
    for t, n in ((M, "M"), (V, "V")):
        if _chk_table(t, n) != (D, sstride) or t.shape[0] != P.shape[0]:
            raise RecError("%s must have the shape of P (M and V share one row stride)" % n)
    h = _hyper(lr, beta1, beta2, eps, step)
    check(lib().rec_adam_rows_all(P.shape[0], D, stride, sstride, _p(groups.n_uniq), _p(groups.uniq_rows),
                                  _p(groups.seg_offset), _p(groups.sorted_pos), _p(grad),
                                  C.byref(_gl(grad_div, grad_group, grad_group_stride, partials)),
                                  _p(grad_scale), _p(P), _p(M), _p(V), C.byref(h), _stream()),
          "rec_adam_rows_all")


def sparse_adagrad_rows(groups, grad, rec, emb_dim, num_slots, label=None, lr=0.05, initial_g2sum=3.0,
                        bounds=(-10.0, 10.0), grad_div=1, grad_group=0, grad_group_stride=0, partials=None,
                        grad_index=None):
    """PS accessor rule (SparseAdaGradSGDRule + show/click) on the touched rows of a record table
    rec [N, stride] = [show | click | g2sum_w | g2sum_x | W(D) | pad]."""
    if rec.dim() != 2 or rec.dtype != torch.float32 or not rec.is_cuda or rec.stride(1) != 1:
        raise RecError("rec must be a 2-D float32 device tensor with unit column stride")
    if label is not None:
        _chk(label, torch.int64, "label")
    h = AdagradHyper(float(lr), float(initial_g2sum), float(bounds[0]), float(bounds[1]))
    check(lib().rec_sparse_adagrad_rows(groups.n, int(emb_dim), rec.stride(0), int(num_slots), _p(groups.n_uniq),
                                        _p(groups.uniq_rows), _p(groups.seg_offset), _p(groups.sorted_pos),
                                        _p(grad),
                                        C.byref(_gl(grad_div, grad_group, grad_group_stride, partials, grad_index)),
                                        _p(label), _p(rec), C.byref(h), _stream()), "rec_sparse_adagrad_rows")


class PsTable:
    """A PS / gpubox sparse table on the device: record rows + the accessor parameters
    (slot_dnn/config_online.yaml:57-89; arithmetic = Paddle's published CtrCommonAccessor / SparseAdaGradSGDRule,
    see include/recengine.h and oracle/ps_ref.py).  kind "slot": W = [embed_w, embedx(D-1)] (slot_dnn / dnn: the
    looked-up vector is the whole W); kind "deepfm": embedx = the D-dim embedding at 0, embed_w = the first-order
    weight behind it.  Rows are zero memory until their first push (state float); a key that does not exist reads as
    zeros, as PullSparse returns it."""

    NUM_STATS = 7   # show, click, g2sum_w, g2sum_x, state, delta_score, unseen_days

    def __init__(self, num_rows, emb_dim, device, kind="slot", lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0),
                 initial_range=1e-4, embedx_threshold=10.0, nonclk_coeff=0.1, click_coeff=1.0, seed=2025,
                 row_stride=None, row_mul=1, row_add=0, embedx_lr=None, embedx_initial_g2sum=None,
                 embedx_bounds=None, embedx_initial_range=None, grad_scale=1.0, show_scale=True,
                 embed_zero_init=True):
        D = int(emb_dim)
        if kind == "slot":      # [W(D) | 7 statistics]: D = 9 -> 16 floats = one 64-B half line
            need = D + self.NUM_STATS
            stride = int(row_stride or (16 if need <= 16 else (need + 31) // 32 * 32))
            self.layout = PsLayout(stride, 0, 1, D - 1, D)
            self.w_cols = slice(0, D)
            stat0 = D
        elif kind == "deepfm":  # [W(D) | W1 | 7 statistics | pad]
            need = D + 1 + self.NUM_STATS
            stride = int(row_stride or (need + 31) // 32 * 32)
            self.layout = PsLayout(stride, D, 0, D, D + 1)
            self.w_cols = slice(0, D)
            stat0 = D + 1
        else:
            raise RecError("kind must be 'slot' or 'deepfm'")
        if stride < need:
            raise RecError("row_stride %d < %d" % (stride, need))
        self.stat = slice(stat0, stat0 + 4)                  # show, click, g2sum_w, g2sum_x
        self.stat_all = slice(stat0, stat0 + self.NUM_STATS)
        self.state_col = stat0 + 4
        self.delta_col, self.unseen_col = stat0 + 5, stat0 + 6
        self.kind, self.emb_dim, self.num_rows = kind, D, int(num_rows)
        self.rec = torch.zeros(int(num_rows), stride, dtype=torch.float32, device=device)
        self.W = self.rec[:, self.w_cols]
        xb = embedx_bounds if embedx_bounds is not None else bounds
        self.accessor = PsAccessor(
            float(lr), float(initial_g2sum), float(bounds[0]), float(bounds[1]), float(initial_range),
            float(lr if embedx_lr is None else embedx_lr),
            float(initial_g2sum if embedx_initial_g2sum is None else embedx_initial_g2sum), float(xb[0]), float(xb[1]),
            float(initial_range if embedx_initial_range is None else embedx_initial_range),
            float(embedx_threshold), float(nonclk_coeff), float(click_coeff), float(grad_scale),
            1 if show_scale else 0, 1 if embed_zero_init else 0, int(seed), int(row_mul), int(row_add))

    @property
    def lazy_init(self):
        """(state_offset, init_dims, init_range, seed) for the lookups: what a key that does not exist reads as.
        Paddle's default (embed_zero_init): zeros = the memory itself, lazy init off (range 0).  Otherwise embed_w
        (element 0 of a 'slot' vector) reads as its creation value, which the first push stores."""
        a = self.accessor
        rng = 0.0 if a.embed_zero_init else a.initial_range
        return (self.state_col - self.w_cols.start, 1, rng, a.seed)


def record_gather(rows, rec, D, out_w, out_w1, status, table=None):
    """Owner-side lookup: both embeddings of every row from its one record line (rec [N, stride] = W(D) | W1 | ...).
    table (PsTable, kind 'deepfm'): unborn rows read as their creation values."""
    _chk(rows, torch.int64, "rows")
    if rec.dim() != 2 or rec.dtype != torch.float32 or not rec.is_cuda or rec.stride(1) != 1:
        raise RecError("rec must be a 2-D float32 device tensor with unit column stride")
    n = rows.numel()
    lz = None
    if table is not None and not table.accessor.embed_zero_init and table.accessor.initial_range > 0:
        a = table.accessor      # embed_w (W1) of a key that does not exist yet reads as its creation value
        lz = LazyInit(table.state_col, 1, a.initial_range, a.seed, a.row_mul, a.row_add)
    check(lib().rec_record_gather(n, int(D), rec.stride(0), rec.shape[0], _p(rows), _p(rec), _p(out_w), _p(out_w1),
                                  C.byref(lz) if lz is not None else None, _p(status), _stream()),
          "rec_record_gather")
    return out_w, out_w1


def ps_push_rows(table, groups, grad, num_slots, grad_pitch=None, grad_index=None, grad1=None, grad1_div=1,
                 show=None, click=None, grad1_pitch=1, grad_group=0, grad_group_stride=0):
    """CtrCommonAccessor::Update on the touched rows of `table` (PsTable).  kind 'slot': grad rows [*, D] hold
    [g_embed_w, g_embedx...]; kind 'deepfm': grad = the D-dim row gradients, grad1 = dz [B] (layout {grad1_div}).
    grad_group / grad_group_stride: rec_grad_layout (rows of `grad_group` segments at a padded stride)."""
    D = table.emb_dim
    pitch = int(grad_pitch or D)
    if table.kind == "slot":
        gx = GradSrc(grad.data_ptr(), _gl(1, grad_group, grad_group_stride, None, grad_index), pitch, 1)
        gw = GradSrc(grad.data_ptr(), _gl(1, grad_group, grad_group_stride, None, grad_index), pitch, 0)
    else:
        if grad1 is None:
            raise RecError("kind 'deepfm' needs grad1 (the first-order gradient)")
        gx = GradSrc(grad.data_ptr(), _gl(1, 0, 0, None, grad_index), pitch, 0)
        gw = GradSrc(grad1.data_ptr(), _gl(grad1_div, 0, 0, None, None), int(grad1_pitch), 0)
    for t, n in ((show, "show"), (click, "click")):
        if t is not None:
            _chk(t, torch.int64, n)
    check(lib().rec_ps_push_rows(groups.n, int(num_slots), C.byref(table.layout), _p(groups.n_uniq),
                                 _p(groups.uniq_rows), _p(groups.seg_offset), _p(groups.sorted_pos), C.byref(gx),
                                 C.byref(gw), _p(show), _p(click), _p(table.rec), C.byref(table.accessor), _stream()),
          "rec_ps_push_rows")


def ps_shrink_rows(table, decay=0.98, delete_threshold=0.8, delete_after_unseen_days=float("inf")):
    """CtrCommonAccessor::Shrink: counters decay; values below delete_threshold or unseen for more than
    delete_after_unseen_days are deleted.  -> number deleted (host sync)."""
    n = torch.zeros(1, dtype=torch.int64, device=table.rec.device)
    check(lib().rec_ps_shrink_rows(table.num_rows, C.byref(table.layout), _p(table.rec), float(decay),
                                   float(delete_threshold), float(min(delete_after_unseen_days, 3.0e38)),
                                   C.byref(table.accessor), _p(n), _stream()),
          "rec_ps_shrink_rows")
    return int(n.item())


def ps_save_select(table, param, base_threshold=1.5, delta_threshold=0.25, delta_keep_days=16.0):
    """CtrCommonAccessor::Save(param) + UpdateStatAfterSave(param) over the table -> bool [N] device mask of the rows a
    save of this kind writes (0 all, 1 delta, 2 base, 3 all + unseen_days += 1)."""
    sel = torch.zeros(table.num_rows, dtype=torch.uint8, device=table.rec.device)
    check(lib().rec_ps_save_select(table.num_rows, C.byref(table.layout), _p(table.rec), int(param),
                                   float(base_threshold), float(delta_threshold), float(delta_keep_days),
                                   C.byref(table.accessor), _p(sel), None, _stream()), "rec_ps_save_select")
    return sel.bool()


def adam_dense(p, m, v, g, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=None):
    for t, n in ((p, "p"), (m, "m"), (v, "v"), (g, "g")):
        _chk(t, torch.float32, n)
    h = _hyper(lr, beta1, beta2, eps, step)
    check(lib().rec_adam_dense(p.numel(), _p(p), _p(m), _p(v), _p(g), _p(grad_scale), C.byref(h),
                               _stream()), "rec_adam_dense")


def _sumsq_ws(ws):
    nbytes = C.c_size_t(0)
    check(lib().rec_sumsq_workspace_bytes(C.byref(nbytes)))
    return ws.get(nbytes.value)


def sumsq(x, out, ws, accumulate=False):
    """out[0] (+)= sum(x^2), fixed reduction order."""
    _chk(x, torch.float32, "x")
    _chk(out, torch.float32, "out")
    w = _sumsq_ws(ws)
    check(lib().rec_sumsq(x.numel(), _p(x), _p(out), int(accumulate), _p(w), C.c_size_t(w.numel()),
                          _stream()), "rec_sumsq")
    return out


def sparse_rows_sumsq(groups, grad, D, out, ws, accumulate=False, grad_div=1, grad_group=0,
                      grad_group_stride=0, partials=None):
    """out[0] (+)= sum over merged rows of |sum of the row's duplicate gradients|^2."""
    _chk(out, torch.float32, "out")
    w = _sumsq_ws(ws)
    check(lib().rec_sparse_rows_sumsq(groups.n, int(D), _p(groups.n_uniq), _p(groups.seg_offset),
                                      _p(groups.sorted_pos), _p(grad),
                                      C.byref(_gl(grad_div, grad_group, grad_group_stride, partials)),
                                      _p(out), int(accumulate), _p(w), C.c_size_t(w.numel()), _stream()),
          "rec_sparse_rows_sumsq")
    return out


def dropout(x, p, seed, stream_a, stream_b=None, out=None, step_stride=0):
    """Train-mode Dropout (upscale_in_train) on a 2-D float32 matrix (row stride allowed), in place by default.
    stream_b: a second mask stream applied in the same pass (keep = keepA & keepB, scale 1/(1-p)^2).
    step_stride: by how much the caller's mask streams advance per training step (a recorded step re-derives them)."""
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.stride(1) != 1:
        raise RecError("x must be a 2-D float32 device tensor with unit column stride")
    if out is None:
        out = x
    sa, sb = C.c_uint64(int(stream_a)), C.c_uint64(int(stream_b or 0))
    if _recorder is not None:
        if not step_stride:
            raise RecError("dropout inside a recorded step needs step_stride")
        _recorder.note_step_arg(sa, step_stride)
        if stream_b is not None:
            _recorder.note_step_arg(sb, step_stride)
    check(lib().rec_dropout(x.shape[0], x.shape[1], x.stride(0), out.stride(0), _p(x), _p(out), float(p), int(seed),
                            sa, sb, 2 if stream_b is not None else 1, _stream()),
          "rec_dropout")
    return out


def l2_decay_grad(grad, w, coeff, grad_scale=None):
    """grad += (coeff / grad_scale) * w  (L2Decay appended after clipping; grad / w contiguous float32 views)."""
    _chk(grad, torch.float32, "grad")
    _chk(w, torch.float32, "w")
    check(lib().rec_l2_decay_grad(grad.numel(), _p(grad), _p(w), float(coeff), _p(grad_scale), _stream()),
          "rec_l2_decay_grad")
    return grad


def clip_scale(sumsq_t, clip_norm, out):
    """out[0] = clip_norm / max(sqrt(sumsq), clip_norm)  (ClipGradByGlobalNorm coefficient)."""
    check(lib().rec_clip_scale(_p(sumsq_t), float(clip_norm), _p(out), _stream()), "rec_clip_scale")
    return out


def din_attention_pool(hist_item, hist_cat, tgt_item_seq, tgt_cat_seq, mask, w_hist_item, w_hist_cat,
                       w_tgt_item_seq, w_tgt_cat_seq, att_w, att_b, status=None, want_weights=True, saved=None, ws=None):
    """Fused DIN attention-pool forward (din/net.py:141-173).  ids/mask [B,T] i64; att_w/att_b: the three
    attention Linear layers ([4E,H1],[H1,H2],[H2,1] / biases).  -> (out [B,E], att_weight [B,T] | None, status)
    saved: a dict the training caller hands in; it receives what the backward can reuse ("out", and "act1"
    [B,T,H1] when the engine saves layer-1 activations for this shape) — pass it on to din_attention_pool_bwd.
    ws (ops.Workspace, optional): lets a batch of few samples run one block per history tile
    (rec_din_attention_pool_fwd_ws)."""
    B, T = hist_item.shape
    for t, n in ((hist_item, "hist_item"), (hist_cat, "hist_cat"), (tgt_item_seq, "tgt_item_seq"),
                 (tgt_cat_seq, "tgt_cat_seq"), (mask, "mask")):
        _chk(t, torch.int64, n, (B, T))
    Ei, ldi = _chk_table(w_hist_item, "w_hist_item")
    Ec, ldc = _chk_table(w_hist_cat, "w_hist_cat")
    if _chk_table(w_tgt_item_seq, "w_tgt_item_seq") != (Ei, ldi) or _chk_table(w_tgt_cat_seq, "w_tgt_cat_seq") != (Ec, ldc):
        raise RecError("target tables must have the shape/stride of the history tables")
    E, H1, H2 = Ei + Ec, att_w[0].shape[1], att_w[1].shape[1]
    _chk(att_w[0], torch.float32, "att_w1", (4 * E, H1))
    _chk(att_w[1], torch.float32, "att_w2", (H1, H2))
    for t, n in ((att_w[2], "att_w3"), (att_b[0], "att_b1"), (att_b[1], "att_b2"), (att_b[2], "att_b3")):
        _chk(t, torch.float32, n)
    dev = hist_item.device
    out = torch.empty(B, E, dtype=torch.float32, device=dev)
    attw = torch.empty(B, T, dtype=torch.float32, device=dev) if want_weights else None
    if status is None:
        status = new_status(dev)
    d = DinDesc(B, T, Ei, Ec, H1, H2, w_hist_item.shape[0], w_hist_cat.shape[0], ldi, ldc)
    act1 = None
    if saved is not None and lib().rec_din_saves_act1(C.byref(d)):
        act1 = saved.get("act1")
        if act1 is None or act1.shape != (B, T, H1) or act1.device != dev:
            act1 = torch.empty(B, T, H1, dtype=torch.float32, device=dev)
    wk, nb = None, 0
    if ws is not None:
        nbytes = C.c_size_t(0)
        check(lib().rec_din_attention_pool_fwd_workspace_bytes(C.byref(d), C.byref(nbytes)))
        if nbytes.value:
            wk = ws.get(nbytes.value)
            nb = wk.numel()
    check(lib().rec_din_attention_pool_fwd_ws(
        C.byref(d), _p(hist_item), _p(hist_cat), _p(tgt_item_seq), _p(tgt_cat_seq), _p(mask),
        _p(w_hist_item), _p(w_hist_cat), _p(w_tgt_item_seq), _p(w_tgt_cat_seq), _p(att_w[0]), _p(att_b[0]),
        _p(att_w[1]), _p(att_b[1]), _p(att_w[2]), _p(att_b[2]), _p(out), _p(attw), _p(act1), _p(status), _p(wk),
        C.c_size_t(nb), _stream()), "rec_din_attention_pool_fwd_ws")
    if saved is not None:
        saved["out"], saved["act1"] = out, act1
    return out, attw, status


def din_attention_pool_bwd(hist_item, hist_cat, tgt_item_seq, tgt_cat_seq, w_hist_item, w_hist_cat,
                           w_tgt_item_seq, w_tgt_cat_seq, att_w, att_b, att_weight, d_out, saved=None, ws=None):
    """-> (d_hist [B,T,E], d_tgt_seq [B,T,E]): per-position gradients of the gathered rows.
    saved: the dict the forward filled (optional; without it the hidden activations are recomputed).
    ws (ops.Workspace, optional): large batches draw their samples from a ticket counter in it
    (rec_din_attention_pool_bwd_ws)."""
    B, T = hist_item.shape
    Ei, ldi = _chk_table(w_hist_item, "w_hist_item")
    Ec, ldc = _chk_table(w_hist_cat, "w_hist_cat")
    E, H1, H2 = Ei + Ec, att_w[0].shape[1], att_w[1].shape[1]
    _chk(att_weight, torch.float32, "att_weight", (B, T))
    _chk(d_out, torch.float32, "d_out", (B, E))
    w1t = _transposed(att_w[0])
    dev = hist_item.device
    dh = torch.empty(B, T, E, dtype=torch.float32, device=dev)
    dq = torch.empty(B, T, E, dtype=torch.float32, device=dev)
    d = DinDesc(B, T, Ei, Ec, H1, H2, w_hist_item.shape[0], w_hist_cat.shape[0], ldi, ldc)
    out_s = act1_s = None
    if saved is not None and saved.get("act1") is not None and saved.get("out") is not None:
        out_s, act1_s = saved["out"], saved["act1"]
        _chk(out_s, torch.float32, "saved out", (B, E))
        _chk(act1_s, torch.float32, "saved act1", (B, T, H1))
    wk, nb = None, 0
    if ws is not None:
        nbytes = C.c_size_t(0)
        check(lib().rec_din_attention_pool_bwd_workspace_bytes(C.byref(d), C.byref(nbytes)))
        if nbytes.value:
            wk = ws.get(nbytes.value)
            nb = wk.numel()
    check(lib().rec_din_attention_pool_bwd_ws(
        C.byref(d), _p(hist_item), _p(hist_cat), _p(tgt_item_seq), _p(tgt_cat_seq), _p(w_hist_item),
        _p(w_hist_cat), _p(w_tgt_item_seq), _p(w_tgt_cat_seq), _p(att_w[0]), _p(w1t), _p(att_b[0]),
        _p(att_w[1]), _p(att_b[1]), _p(att_w[2]), _p(att_weight), _p(out_s), _p(act1_s), _p(d_out), _p(dh), _p(dq),
        _p(wk), C.c_size_t(nb), _stream()), "rec_din_attention_pool_bwd_ws")
    return dh, dq


# ------------------------------------------------------------------ xDeepFM CIN (the passes around the GEMM)
def _chk_f32(t, name):
    if not t.is_cuda:
        raise RecError("%s must be a device tensor (no CPU fallback)" % name)
    if t.dtype != torch.float32:
        raise RecError("%s must be float32, got %s" % (name, t.dtype))


def cin_view(t, kind):
    """rec_cin_view of a feature tensor: kind "bfd" = [B,F,D] (feat_embeddings, any strides), "xt" = (tensor [B*D, C],
    D): a CIN layer's d-major GEMM output (row stride allowed, unit column stride)."""
    if kind == "bfd":
        _chk_f32(t, "feature tensor")
        return CinView(t.stride(0), t.stride(1), t.stride(2))
    x, D = t
    return CinView(D * _chk_mat(x, "XT"), 1, _chk_mat(x, "XT"))


def cin_outer_fwd(B, D, F, S, X0, v0, Xk, vk, Z):
    """Z[(b,d), f*S+s] = X0[b,f,d] * Xk[b,s,d]   (xdeepfm/net.py:163-175).  Z [B*D, F*S] f32."""
    _chk_f32(X0, "X0")
    _chk_f32(Xk, "Xk")
    ldz = _chk_mat(Z, "Z")
    if tuple(Z.shape) != (B * D, F * S):
        raise RecError("Z must be [B*D, F*S]")
    check(lib().rec_cin_outer_fwd(B, D, F, S, _p(X0), C.byref(v0), _p(Xk), C.byref(vk), _p(Z), ldz, _stream()),
          "rec_cin_outer_fwd")
    return Z


def cin_outer_bwd(B, D, F, S, dZ, X0, v0, Xk, vk, dX0, dv0, acc0, dXk, dvk, acck, dpool=None):
    """dX0 (+)= dZ . Xk over s;  dXk (+)= dZ . X0 over f (+ dpool[b,s] on every d row)."""
    for t, n in ((X0, "X0"), (Xk, "Xk"), (dX0, "dX0"), (dXk, "dXk")):
        _chk_f32(t, n)
    ldz = _chk_mat(dZ, "dZ")
    if tuple(dZ.shape) != (B * D, F * S):
        raise RecError("dZ must be [B*D, F*S]")
    ldp = 0
    if dpool is not None:
        ldp = _chk_mat(dpool, "dpool")
        if tuple(dpool.shape) != (B, S):
            raise RecError("dpool must be [B, S]")
    check(lib().rec_cin_outer_bwd(B, D, F, S, _p(dZ), ldz, _p(X0), C.byref(v0), _p(Xk), C.byref(vk), _p(dX0),
                                  C.byref(dv0), int(acc0), _p(dXk), C.byref(dvk), int(acck), _p(dpool), ldp, _stream()),
          "rec_cin_outer_bwd")


def cin_contract_fwd(B, D, F, Y, X0, v0, XT):
    """XT[(b,d), c] = sum_f X0[b,f,d] * Y[(b,d), c*F+f]   (the C < S association of a CIN layer)."""
    _chk_f32(X0, "X0")
    ldy, ldx = _chk_mat(Y, "Y"), _chk_mat(XT, "XT")
    Cc = XT.shape[1]
    if tuple(Y.shape) != (B * D, Cc * F) or XT.shape[0] != B * D:
        raise RecError("Y must be [B*D, C*F] and XT [B*D, C]")
    check(lib().rec_cin_contract_fwd(B, D, F, Cc, _p(Y), ldy, _p(X0), C.byref(v0), _p(XT), ldx, _stream()),
          "rec_cin_contract_fwd")
    return XT


def cin_contract_bwd(B, D, F, Y, dXT, X0, v0, dY, dX0, dv0, acc0):
    """dY = dXT (x) X0;  dX0 (+)= sum_c dXT[.,c] * Y[., c*F+f]."""
    for t, n in ((X0, "X0"), (dX0, "dX0")):
        _chk_f32(t, n)
    ldy, ldx, lddy = _chk_mat(Y, "Y"), _chk_mat(dXT, "dXT"), _chk_mat(dY, "dY")
    Cc = dXT.shape[1]
    if tuple(Y.shape) != (B * D, Cc * F) or tuple(dY.shape) != (B * D, Cc * F) or dXT.shape[0] != B * D:
        raise RecError("Y / dY must be [B*D, C*F] and dXT [B*D, C]")
    check(lib().rec_cin_contract_bwd(B, D, F, Cc, _p(Y), ldy, _p(dXT), ldx, _p(X0), C.byref(v0), _p(dY), lddy,
                                     _p(dX0), C.byref(dv0), int(acc0), _stream()), "rec_cin_contract_bwd")


def cin_sumpool(B, D, XT, out):
    """out[b,c] = sum_d XT[(b,d), c]   (net.py:195-198); out may be a column window of the pooled-feature row."""
    ldx, ldo = _chk_mat(XT, "XT"), _chk_mat(out, "out")
    Cc = XT.shape[1]
    if XT.shape[0] != B * D or tuple(out.shape) != (B, Cc):
        raise RecError("XT must be [B*D, C] and out [B, C]")
    check(lib().rec_cin_sumpool(B, D, Cc, _p(XT), ldx, _p(out), ldo, _stream()), "rec_cin_sumpool")
    return out


def cin_sumpool_bwd(B, D, dpool, dXT):
    """dXT[(b,d), c] = dpool[b,c]."""
    ldp, ldx = _chk_mat(dpool, "dpool"), _chk_mat(dXT, "dXT")
    Cc = dpool.shape[1]
    if tuple(dXT.shape) != (B * D, Cc) or dpool.shape[0] != B:
        raise RecError("dXT must be [B*D, C] and dpool [B, C]")
    check(lib().rec_cin_sumpool_bwd(B, D, Cc, _p(dpool), ldp, _p(dXT), ldx, _stream()), "rec_cin_sumpool_bwd")
    return dXT


# ------------------------------------------------------------------ DLRM: BatchNorm1D, dot interaction, accuracy
def batchnorm_fwd(X, gamma, beta, running_mean, running_var, ws, training=True, momentum=0.9, eps=1e-5, out=None):
    """nn.BatchNorm1D on X [m, n] (dlrm/net.py:151-153).  -> (Y, save_mean [n], save_invstd [n]); the running
    statistics are updated in place when training."""
    ldx = _chk_mat(X, "X")
    m, n = X.shape
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, torch.float32, nm, (n,))
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=X.device)
    ldy = _chk_mat(out, "out")
    sm = torch.empty(n, dtype=torch.float32, device=X.device)
    si = torch.empty(n, dtype=torch.float32, device=X.device)
    nbytes = C.c_size_t(0)
    check(lib().rec_batchnorm_workspace_bytes(m, n, C.byref(nbytes)))
    w = ws.get(nbytes.value)
    check(lib().rec_batchnorm_fwd(m, n, _p(X), ldx, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                  float(momentum), float(eps), int(bool(training)), _p(out), ldy, _p(sm), _p(si), _p(w),
                                  C.c_size_t(w.numel()), _stream()), "rec_batchnorm_fwd")
    return out, sm, si


def batchnorm_bwd(X, dY, gamma, save_mean, save_invstd, ws, relu_mask=False, dgamma=None, dbeta=None, out=None):
    """-> (dX, dgamma, dbeta).  relu_mask: X was a ReLU output — dX is zeroed where X <= 0."""
    ldx, lddy = _chk_mat(X, "X"), _chk_mat(dY, "dY")
    m, n = X.shape
    dev = X.device
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=dev)
    if dgamma is None:
        dgamma = torch.empty(n, dtype=torch.float32, device=dev)
    if dbeta is None:
        dbeta = torch.empty(n, dtype=torch.float32, device=dev)
    for t, nm in ((gamma, "gamma"), (save_mean, "save_mean"), (save_invstd, "save_invstd"), (dgamma, "dgamma"),
                  (dbeta, "dbeta")):
        _chk(t, torch.float32, nm, (n,))
    nbytes = C.c_size_t(0)
    check(lib().rec_batchnorm_workspace_bytes(m, n, C.byref(nbytes)))
    w = ws.get(nbytes.value)
    check(lib().rec_batchnorm_bwd(m, n, _p(X), ldx, _p(dY), lddy, _p(gamma), _p(save_mean), _p(save_invstd),
                                  int(bool(relu_mask)), _p(out), _chk_mat(out, "dX"), _p(dgamma), _p(dbeta), _p(w),
                                  C.c_size_t(w.numel()), _stream()), "rec_batchnorm_bwd")
    return out, dgamma, dbeta


def dot_interact_fwd(T, out=None):
    """T [B,F,D] = [emb_1 .. emb_{F-1}, x]  ->  R [B, D + F(F-1)/2] = [x | strictly-upper-triangle dots, row-major]
    (dlrm/net.py:96-123)."""
    _chk(T, torch.float32, "T")
    B, F, D = T.shape
    P = F * (F - 1) // 2
    if out is None:
        out = torch.empty(B, D + P, dtype=torch.float32, device=T.device)
    check(lib().rec_dot_interact_fwd(B, F, D, _p(T), F * D, _p(out), _chk_mat(out, "R"), _stream()),
          "rec_dot_interact_fwd")
    return out


def dot_interact_bwd(T, dR, out=None):
    _chk(T, torch.float32, "T")
    B, F, D = T.shape
    ldr = _chk_mat(dR, "dR")
    if out is None:
        out = torch.empty(B, F, D, dtype=torch.float32, device=T.device)
    _chk(out, torch.float32, "dT", (B, F, D))
    check(lib().rec_dot_interact_bwd(B, F, D, _p(T), F * D, _p(dR), ldr, _p(out), F * D, _stream()),
          "rec_dot_interact_bwd")
    return out


def accuracy_count(pred, label, counts):
    """counts [2] int64: += (#correct top-1 of two classes, n)   (paddle.metric.Accuracy)."""
    _chk(pred, torch.float32, "pred")
    _chk(label, torch.int64, "label")
    _chk(counts, torch.int64, "counts", (2,))
    check(lib().rec_accuracy_count(pred.numel(), _p(pred), _p(label), _p(counts), _stream()), "rec_accuracy_count")
    return counts


def sparse_sgd_rows(groups, grad, P, lr, grad_div=1, grad_group=0, grad_group_stride=0, partials=None):
    D, stride = _chk_table(P, "P")
    check(lib().rec_sparse_sgd_rows(groups.n, D, stride, _p(groups.n_uniq), _p(groups.uniq_rows),
                                    _p(groups.seg_offset), _p(groups.sorted_pos), _p(grad),
                                    C.byref(_gl(grad_div, grad_group, grad_group_stride, partials)),
                                    _p(P), float(lr), _stream()), "rec_sparse_sgd_rows")


SMALL_MERGE_MAX = 15360     # rec_sparse_sgd_small: lookups per call


def sparse_sgd_small(ids, grad, P, lr, padding_idx=None, status=None, grad_div=1, grad_group=0, grad_group_stride=0):
    """Merge + SGD row update of a small SelectedRows gradient in ONE launch (ids.numel() <= SMALL_MERGE_MAX,
    emb_dim <= 256): P[row] -= lr * (sum of the row's gradient rows, ascending position)."""
    _chk(ids, torch.int64, "ids")
    D, stride = _chk_table(P, "P")
    if status is None:
        status = new_status(ids.device)
    check(lib().rec_sparse_sgd_small(ids.numel(), D, stride, P.shape[0], -1 if padding_idx is None else padding_idx,
                                     _p(ids), _p(grad), C.byref(_gl(grad_div, grad_group, grad_group_stride)), _p(P),
                                     float(lr), _p(status), _stream()), "rec_sparse_sgd_small")
    return status


def sparse_sgd_small_multi(jobs, lr, status):
    """Up to 8 sparse_sgd_small jobs in ONE launch: jobs = [(ids, grad, P, grad_group, grad_group_stride), ...] (tables
    independent of each other: rec_sparse_sgd_small_multi)."""
    arr = (_lib.SmallSgdJob * len(jobs))()
    for a, (ids, grad, P, group, gstride) in zip(arr, jobs):
        _chk(ids, torch.int64, "ids")
        D, stride = _chk_table(P, "P")
        a.n, a.emb_dim, a.row_stride, a.num_rows, a.padding_idx = ids.numel(), D, stride, P.shape[0], -1
        a.ids, a.grad, a.P = ids.data_ptr(), grad.data_ptr(), P.data_ptr()
        a.grad_layout = _gl(1, group, gstride)
        if _recorder is not None:
            _recorder.note_field(a, "ids", ids)
            _recorder.keep.extend((ids, grad, P))
    check(lib().rec_sparse_sgd_small_multi(len(jobs), arr, float(lr), _p(status), _stream()),
          "rec_sparse_sgd_small_multi")
    return status


_SMALL_SCRATCH = {}


def sparse_adam_record_small(ids, slot_offset, padding_idx, grad, grad1, grad1_div, rec, mv, D, step, lr=1e-3,
                             beta1=0.9, beta2=0.999, eps=1e-8, v_offset=None, grad_scale=None, status=None):
    """sparse_adam_record with the SelectedRows merge done inside the launch (ids.numel() <= SMALL_MERGE_MAX): ids
    [B,S] (or flat [B*S]), row = id + slot_offset[s]; grad [B*S, D] row gradients; grad1 = dz with layout {grad1_div}."""
    _chk(ids, torch.int64, "ids")
    _chk(grad, torch.float32, "grad")
    _chk(grad1, torch.float32, "grad1")
    S = ids.shape[-1] if ids.dim() > 1 else 1
    if v_offset is None:
        v_offset = (D + 3) // 4 * 4
    if status is None:
        status = new_status(ids.device)
    h = _hyper(lr, beta1, beta2, eps, step)
    # one int per (device, stream): the launch's "every id stays in its slot's span" decision — written by one kernel
    # of the call and read by the next on the SAME stream, so calls on different streams must not share it
    key = (ids.device, _stream().value if ids.device.type == "cuda" else None)
    scratch = _SMALL_SCRATCH.get(key)
    if scratch is None:
        if len(_SMALL_SCRATCH) > 256:
            _SMALL_SCRATCH.clear()
        scratch = _SMALL_SCRATCH[key] = torch.zeros(1, dtype=torch.int32, device=ids.device)
    check(lib().rec_sparse_adam_record_small(ids.numel(), S, int(D), rec.stride(0), mv.stride(0), int(v_offset),
                                             rec.shape[0], -1 if padding_idx is None else int(padding_idx), _p(ids),
                                             _p(slot_offset), _p(grad), C.byref(_gl(1, 0, 0)), _p(grad1),
                                             C.byref(_gl(grad1_div, 0, 0)), _p(grad_scale), _p(rec), _p(mv),
                                             C.byref(h), _p(status), _p(scratch), _stream()),
          "rec_sparse_adam_record_small")
    return status


def sgd_dense(p, g, lr):
    _chk(p, torch.float32, "p")
    _chk(g, torch.float32, "g")
    check(lib().rec_sgd_dense(p.numel(), _p(p), _p(g), float(lr), _stream()), "rec_sgd_dense")


def bce_with_logits(logit, label, ws, mean_over=0):
    """-> pred [B,1], dz [B,1], loss [1]   (label float32 [B,1])"""
    B = logit.numel()
    _chk(logit, torch.float32, "logit")
    _chk(label, torch.float32, "label")
    dev = logit.device
    pred = torch.empty(B, 1, dtype=torch.float32, device=dev)
    dz = torch.empty(B, 1, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    nbytes = C.c_size_t(0)
    check(lib().rec_logloss_workspace_bytes(B, C.byref(nbytes)))
    w = ws.get(nbytes.value)
    check(lib().rec_bce_with_logits(B, int(mean_over), _p(logit), _p(label), _p(pred), _p(dz), _p(loss), _p(w),
                                    C.c_size_t(w.numel()), _stream()), "rec_bce_with_logits")
    return pred, dz, loss


def softmax_rows(x, out=None):
    ldx = _chk_mat(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().rec_softmax_rows(x.shape[0], x.shape[1], _p(x), ldx, _p(out), _chk_mat(out, "out"),
                                 _stream()), "rec_softmax_rows")
    return out


def moe_bwd_prep(dX, X0, U, prob_e, dU, dX0_acc, accumulate, dp_e):
    """CrossNetMix backward glue for one expert: dU = dX*X0*p_e; dX0_acc (+)= dX*p_e*U; dp_e = rowsum(dX*X0*U)."""
    M, N = dX.shape
    lds = [_chk_mat(t, n) for t, n in ((dX, "dX"), (X0, "X0"), (U, "U"), (dU, "dU"), (dX0_acc, "dX0_acc"))]
    check(lib().rec_moe_bwd_prep(M, N, _p(dX), lds[0], _p(X0), lds[1], _p(U), lds[2], _p(prob_e),
                                 prob_e.stride(0) if M > 1 else 1, _p(dU), lds[3], _p(dX0_acc), lds[4],
                                 int(accumulate), _p(dp_e), dp_e.stride(0) if M > 1 else 1, _stream()),
          "rec_moe_bwd_prep")


def softmax_rows_bwd(p, dp, out=None):
    if out is None:
        out = torch.empty(p.shape, dtype=torch.float32, device=p.device)
    check(lib().rec_softmax_rows_bwd(p.shape[0], p.shape[1], _p(p), _chk_mat(p, "p"), _p(dp), _chk_mat(dp, "dp"),
                                     _p(out), _chk_mat(out, "out"), _stream()), "rec_softmax_rows_bwd")
    return out


def cross_bwd_prep(dX, X0, U, dU, dX0_acc, accumulate):
    """dU = dX*X0; dX0_acc (+)= dX*U  (CrossNetV2 backward glue)."""
    M, N = dX.shape
    lds = [_chk_mat(t, n) for t, n in ((dX, "dX"), (X0, "X0"), (U, "U"), (dU, "dU"), (dX0_acc, "dX0_acc"))]
    check(lib().rec_cross_bwd_prep(M, N, _p(dX), lds[0], _p(X0), lds[1], _p(U), lds[2], _p(dU), lds[3],
                                   _p(dX0_acc), lds[4], int(accumulate), _stream()), "rec_cross_bwd_prep")


# ------------------------------------------------------------------ CrossNet layers (one C-ABI call per layer)
def _ld(t, name):
    return _chk_mat(t, name)


def crossnet_v2_layer_fwd(x0, xl, W, bias, ws, out=None, u=None):
    """x_{l+1} = x_l + x_0 * (x_l W + b) (dcn_v2/net.py:222-226) -> out; u (optional) receives x_l W + b."""
    B, d = xl.shape
    if out is None:
        out = torch.empty(B, d, dtype=torch.float32, device=xl.device)
    desc = _lib.CrossV2Desc(B, d, _ld(x0, "x0"), _ld(xl, "xl"), _ld(out, "out"), _ld(u, "u") if u is not None else d)
    nb = C.c_size_t(0)
    check(lib().rec_crossnet_v2_layer_workspace_bytes(C.byref(desc), C.byref(nb), None))
    w = ws.get(nb.value)
    check(lib().rec_crossnet_v2_layer_fwd(C.byref(desc), _p(x0), _p(xl), _p(W), _p(bias), _p(out), _p(u), _p(w),
                                          C.c_size_t(w.numel()), _stream()), "rec_crossnet_v2_layer_fwd")
    return out


def crossnet_v2_layer_bwd(x0, xl, W, u, dxnext, dx0_acc, accumulate_dx0, fold_dx0, dW, db, ws, out=None):
    """-> d x_l ([B,d]); dW, db written; dx0_acc (+)= dxnext * u."""
    B, d = xl.shape
    if out is None:
        out = torch.empty(B, d, dtype=torch.float32, device=xl.device)
    desc = _lib.CrossV2Desc(B, d, _ld(x0, "x0"), _ld(xl, "xl"), d, _ld(u, "u"))
    nb = C.c_size_t(0)
    check(lib().rec_crossnet_v2_layer_workspace_bytes(C.byref(desc), None, C.byref(nb)))
    w = ws.get(nb.value)
    check(lib().rec_crossnet_v2_layer_bwd(C.byref(desc), _p(x0), _p(xl), _p(W), _p(u), _p(dxnext), _ld(dxnext, "dxnext"),
                                          _p(dx0_acc), _ld(dx0_acc, "dx0_acc"), int(accumulate_dx0), int(fold_dx0),
                                          _p(out), _ld(out, "out"), _p(dW), _p(db), _p(w), C.c_size_t(w.numel()),
                                          _stream()), "rec_crossnet_v2_layer_bwd")
    return out


def crossnet_mix_layer_fwd(x0, xl, U, V, Cm, bias, gate_w, gate_b, ws, out=None):
    """One CrossNetMix layer (dcn_v2/net.py:278-320) -> (x_next, t1, t2, prob); t1 / t2 / prob are what the backward
    needs."""
    B, d = xl.shape
    E, _, r = U.shape
    dev = xl.device
    if out is None:
        out = torch.empty(B, d, dtype=torch.float32, device=dev)
    t1 = torch.empty(B, E * r, dtype=torch.float32, device=dev)
    t2 = torch.empty(B, E * r, dtype=torch.float32, device=dev)
    prob = torch.empty(B, E, dtype=torch.float32, device=dev)
    desc = _lib.CrossMixDesc(B, d, r, E, _ld(x0, "x0"), _ld(xl, "xl"), _ld(out, "out"))
    nb = C.c_size_t(0)
    check(lib().rec_crossnet_mix_layer_workspace_bytes(C.byref(desc), C.byref(nb), None))
    w = ws.get(nb.value)
    check(lib().rec_crossnet_mix_layer_fwd(C.byref(desc), _p(x0), _p(xl), _p(U), _p(V), _p(Cm), _p(bias), _p(gate_w),
                                           _p(gate_b), _p(out), _p(t1), _p(t2), _p(prob), _p(w),
                                           C.c_size_t(w.numel()), _stream()), "rec_crossnet_mix_layer_fwd")
    return out, t1, t2, prob


def crossnet_mix_layer_bwd(x0, xl, U, V, Cm, bias, gate_w, t1, t2, prob, dxnext, dx0_acc, accumulate_dx0, fold_dx0,
                           gU, gV, gC, gbias, g_gate_w, g_gate_b, accumulate_gate, ws, out=None):
    """-> d x_l; gU / gV / gC / gbias written, the shared gating gradients written or accumulated."""
    B, d = xl.shape
    E, _, r = U.shape
    if out is None:
        out = torch.empty(B, d, dtype=torch.float32, device=xl.device)
    desc = _lib.CrossMixDesc(B, d, r, E, _ld(x0, "x0"), _ld(xl, "xl"), d)
    nb = C.c_size_t(0)
    check(lib().rec_crossnet_mix_layer_workspace_bytes(C.byref(desc), None, C.byref(nb)))
    w = ws.get(nb.value)
    check(lib().rec_crossnet_mix_layer_bwd(C.byref(desc), _p(x0), _p(xl), _p(U), _p(V), _p(Cm), _p(bias), _p(gate_w),
                                           _p(t1), _p(t2), _p(prob), _p(dxnext), _ld(dxnext, "dxnext"), _p(dx0_acc),
                                           _ld(dx0_acc, "dx0_acc"), int(accumulate_dx0), int(fold_dx0), _p(out),
                                           _ld(out, "out"), _p(gU), _p(gV), _p(gC), _p(gbias), _p(g_gate_w),
                                           _p(g_gate_b), int(accumulate_gate), _p(w), C.c_size_t(w.numel()),
                                           _stream()), "rec_crossnet_mix_layer_bwd")
    return out


# ------------------------------------------------------------------ row-sharded tables
class ShardRoute:
    """Result buffers of rec_shard_route (device)."""

    def __init__(self, n, num_shards, device):
        self.n, self.num_shards = n, num_shards
        i64 = dict(dtype=torch.int64, device=device)
        self.send_local_row = torch.empty(max(n, 1), **i64)
        self.send_pos = torch.empty(max(n, 1), **i64)
        self.send_sample = torch.empty(max(n, 1), **i64)
        self.slot_of_pos = torch.empty(max(n, 1), **i64)
        self.send_counts = torch.zeros(num_shards + 1, **i64)


class DedupPlan:
    """One deduplicated lookup of a row-sharded table (paddlerec_amd/sharded.py, REC_SHARD_DEDUP): the DISTINCT rows this
    rank needs of every owner, in fixed-capacity send slots (owner o owns slots [o * cap, (o + 1) * cap)), and the maps
    between lookup positions, distinct rows and slots.  Everything lives on the device; nothing here reads it back."""

    def __init__(self, n, num_shards, cap, device):
        self.n, self.num_shards, self.cap = n, num_shards, cap
        self.groups = IdGroups(n, device)                  # grouping of the lookups by (owner, local row)
        i64 = dict(dtype=torch.int64, device=device)
        self.send_rows = torch.empty(num_shards * cap, **i64)       # local row per slot; empty slots hold `sentinel`
        self.slot_of_pos = torch.empty(max(n, 1), **i64)            # 1 + slot of the position's row; 0: padding / dropped
        self.slot_of_uniq = torch.empty(max(n, 1), **i64)           # slot of distinct row u; num_shards * cap: none
        self.counts = torch.zeros(num_shards, **i64)                # distinct rows per owner (before the capacity cut)
        self.sentinel = None


def dedup_plan(ids, num_rows, padding_idx, num_shards, local_rows, cap, ws, slot_offset=None, status=None, plan=None):
    """Distinct (owner, local row) pairs of a batch of lookups, owner-major and ascending — HeterPS dedups a pass's keys
    before it builds the per-GPU tables (tools/static_gpubox_trainer.py:237-246).  ids [B,S] int64; row = id +
    slot_offset[s]; owner = row % num_shards, local row = row // num_shards.  rec_dedup_plan: device only, no host read:
    sizes are the fixed capacity `cap` rows per owner; a rank that needs more of one owner sets REC_FLAG_EXCHANGE_OVERFLOW
    and loses the rows behind the capacity."""
    _chk(ids, torch.int64, "ids")
    n, G, dev = ids.numel(), int(num_shards), ids.device
    if plan is None or plan.n != n or plan.cap != cap or plan.num_shards != G:
        plan = DedupPlan(n, G, cap, dev)
    if status is None:
        status = new_status(dev)
    S = ids.shape[-1] if ids.dim() > 1 else 1
    nbytes = C.c_size_t(0)
    check(lib().rec_dedup_plan_workspace_bytes(n, G, int(local_rows), C.byref(nbytes)), "rec_dedup_plan_workspace_bytes")
    w = ws.get(nbytes.value)
    g = plan.groups
    plan.sentinel = int(local_rows)
    check(lib().rec_dedup_plan(n, S, int(num_rows), -1 if padding_idx is None else int(padding_idx), G, int(local_rows),
                               int(cap), _p(ids), _p(slot_offset), _p(g.sorted_pos), _p(g.uniq_rows), _p(g.seg_offset),
                               _p(g.n_uniq), _p(plan.send_rows), _p(plan.slot_of_pos), _p(plan.slot_of_uniq),
                               _p(plan.counts), _p(status), _p(w), C.c_size_t(w.numel()), _stream()), "rec_dedup_plan")
    return plan, status


def dedup_merge(plan, grad, emb_dim, grad_div=1, out=None):
    """The gradient rows of a deduplicated lookup merged per distinct row (duplicates summed in ascending position
    order, as every merge of the engine) and laid into the plan's send slots: out [num_shards * cap, emb_dim], zeros in
    the empty slots.  grad: [n / grad_div, emb_dim] (grad_div = S: one row per sample serving its S lookups)."""
    G, cap, D = plan.num_shards, plan.cap, int(emb_dim)
    none = G * cap
    if out is None or out.shape[0] != none + 1:
        out = torch.empty(none + 1, D, dtype=torch.float32, device=grad.device)      # + one row the unplaced rows land in
    out.zero_()

    class _G:                                   # the plan's grouping with the send slot in the place of the table row
        pass
    g, f = plan.groups, _G()
    f.n, f.n_uniq, f.seg_offset, f.sorted_pos, f.uniq_rows = g.n, g.n_uniq, g.seg_offset, g.sorted_pos, plan.slot_of_uniq
    sparse_sgd_rows(f, grad, out, -1.0, grad_div=grad_div)          # 0 - (-1) * sum = sum, exactly
    return out


def shard_route(ids, num_rows, padding_idx, num_shards, ws, slot_offset=None, status=None,
                route=None):
    """Partition the lookups by owner shard (row % num_shards).  -> (ShardRoute, status)"""
    _chk(ids, torch.int64, "ids")
    n = ids.numel()
    S = ids.shape[-1] if ids.dim() > 1 else 1
    dev = ids.device
    if route is None:
        route = ShardRoute(n, num_shards, dev)
    if status is None:
        status = new_status(dev)
    nbytes = C.c_size_t(0)
    check(lib().rec_shard_route_workspace_bytes(n, num_shards, C.byref(nbytes)))
    w = ws.get(nbytes.value)
    check(lib().rec_shard_route(n, S, num_rows, -1 if padding_idx is None else padding_idx,
                                num_shards, _p(ids), _p(slot_offset), _p(route.send_local_row),
                                _p(route.send_pos), _p(route.send_sample), _p(route.slot_of_pos),
                                _p(route.send_counts), _p(status), _p(w), C.c_size_t(w.numel()),
                                _stream()), "rec_shard_route")
    return route, status


# ------------------------------------------------------------------ f32 MFMA GEMM
def _chk_mat(t, name):
    if not t.is_cuda:
        raise RecError("%s must be a device tensor (no CPU fallback)" % name)
    if t.dtype != torch.float32 or t.dim() != 2:
        raise RecError("%s must be a 2-D float32 tensor" % name)
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise RecError("%s must have unit column stride" % name)
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


_gemm_ws_cache = {}


class GemmImages:
    """bf16 x 3 images of a set of GEMM weights, made in ONE launch (rec_gemm_b_images) ahead of the GEMMs that consume
    them: entries (B, trans_b) in the orientation the consuming call passes.  get(i) -> the image tensor (or None where
    the shape has no image form: the call then splits for itself, or runs the exact-f32 kernel)."""

    def __init__(self, specs, device):
        self.specs, self.images = list(specs), []
        for B, trans_b in self.specs:
            n, k = (B.shape if trans_b else B.shape[::-1])
            ok, nb = C.c_int32(0), C.c_size_t(0)
            check(lib().rec_gemm_b_image_bytes(int(k), int(n), C.byref(ok), C.byref(nb)), "rec_gemm_b_image_bytes")
            self.images.append(torch.empty(nb.value, dtype=torch.uint8, device=device) if ok.value else None)
        live = [(B, t, img) for (B, t), img in zip(self.specs, self.images) if img is not None]
        self._items = (GemmBImage * max(1, len(live)))()
        self._n = len(live)
        for it, (B, t, img) in zip(self._items, live):
            n, k = (B.shape if t else B.shape[::-1])
            it.B, it.ldb, it.k, it.n, it.trans_b, it.image = B.data_ptr(), _chk_mat(B, "B"), int(k), int(n), int(t), img.data_ptr()

    def refresh(self):
        """Re-split every weight from its current values on the current stream."""
        if self._n:
            check(lib().rec_gemm_b_images(self._n, self._items, _stream()), "rec_gemm_b_images")

    def get(self, i):
        return self.images[i]


def gemm(A, B, ws, trans_a=False, trans_b=False, epilogue="none", bias=None, aux0=None, aux1=None,
         out=None, split_k=0, b_colsum=None, row_scale=None, out2=None, num_cus=0, b_image=None, relu_bits=None):
    """out[M,N] = epi(op(A) @ op(B)); A/B/out row-major (row strides allowed).
    trans_a: A is given as [K,M]; trans_b: B is given as [N,K] (a torch Linear.weight, or W for dX).
    num_cus > 0: the current stream is confined to that many compute units (cu_range_stream) — size the
    split-K for them instead of the whole chip.
    b_image: op(B)'s bf16 x 3 image from GemmImages (current values of B): the call skips its own split launch.
    relu_bits: the ReLU mask as bits (rec_gemm_epilogue_args.relu_bits).  epilogue "bias_relu": a LIST — the call appends
    the mask tensor it wrote beside `out` (or None when this call has no bit form); "relu_mask": such a tensor (or None):
    read instead of aux0 when this call has the bit form."""
    d, x, out, need = _gemm_prepare(A, B, trans_a, trans_b, epilogue, bias, aux0, aux1, out, split_k, b_colsum, row_scale,
                                    out2, num_cus, b_image)
    if relu_bits is not None:
        nb = _relu_bits_bytes(d)
        if epilogue == "bias_relu":
            t = torch.empty(nb // 8, dtype=torch.int64, device=A.device) if nb else None
            relu_bits.append(t)
            relu_bits = t
        elif nb == 0 or relu_bits.numel() * 8 < nb:
            relu_bits = None
        if relu_bits is not None:
            x.relu_bits = relu_bits.data_ptr()
            if _recorder is not None:
                _recorder.keep.append(relu_bits)
    w = ws.get(need)
    check(lib().rec_gemm_f32(C.byref(d), _p(A), _p(B), _p(out), C.byref(x), _p(w),
                             C.c_size_t(w.numel()), _stream()), "rec_gemm_f32")
    return out


def linear_backward(X, G, W, ws, dW, db, relu_src=None, b_image=None, epilogue=None, aux0=None, relu_bits=None):
    """The backward of one Linear in ONE call (rec_gemm_f32_pair): dW = X^T G, db = colsum(G) and
    dX = G W^T (masked by relu_src > 0 — the layer's own input — when given; or any dX epilogue of gemm(): epilogue=
    "dsigmoid", aux0=the layer's input).  -> dX.  At launch-bound sizes the two GEMMs are one launch; otherwise exactly the
    two gemm() calls of mlp_backward, dW first."""
    if epilogue is None:
        epilogue, aux0 = ("relu_mask", relu_src) if relu_src is not None else ("none", None)
    d0, x0, _, need0 = _gemm_prepare(X, G, True, False, "none", None, None, None, dW, 0, db, None, None, 0, None)
    d1, x1, dX, need1 = _gemm_prepare(G, W, False, True, epilogue, None, aux0, None, None, 0, None, None, None, 0, b_image)
    if relu_bits is not None and epilogue == "relu_mask":      # relu_src's mask as bits (mlp_forward), where dX has the bit form
        nb = _relu_bits_bytes(d1)
        if nb and relu_bits.numel() * 8 >= nb:
            x1.relu_bits = relu_bits.data_ptr()
            if _recorder is not None:
                _recorder.keep.append(relu_bits)
    w = ws.get(max(need0, need1))
    check(lib().rec_gemm_f32_pair(C.byref(d0), _p(X), _p(G), _p(dW), C.byref(x0), C.byref(d1), _p(G), _p(W), _p(dX),
                                  C.byref(x1), _p(w), C.c_size_t(w.numel()), _stream()), "rec_gemm_f32_pair")
    return dX


_relu_bits_cache = {}


def _relu_bits_bytes(d):
    """Bytes of the ReLU bit mask of this call, 0 when it has no bit form (or REC_RELU_BITS=0)."""
    if os.environ.get("REC_RELU_BITS") == "0":
        return 0
    key = (d.m, d.n, d.k, d.lda, d.ldb, d.ldc, d.trans_a, d.trans_b, d.epilogue, d.split_k, os.environ.get("REC_GEMM_BF16X3"))
    nb = _relu_bits_cache.get(key)
    if nb is None:
        ok, nbytes = C.c_int32(0), C.c_size_t(0)
        check(lib().rec_gemm_relu_bits_bytes(C.byref(d), C.byref(ok), C.byref(nbytes)), "rec_gemm_relu_bits_bytes")
        nb = _relu_bits_cache[key] = nbytes.value if ok.value else 0
    return nb


def _gemm_prepare(A, B, trans_a, trans_b, epilogue, bias, aux0, aux1, out, split_k, b_colsum, row_scale, out2, num_cus,
                  b_image):
    """Checks of one rec_gemm_f32 call -> (descriptor, epilogue arguments, out, workspace bytes)."""
    lda, ldb = _chk_mat(A, "A"), _chk_mat(B, "B")
    K, M = (A.shape if trans_a else A.shape[::-1])
    N, K2 = (B.shape if trans_b else B.shape[::-1])
    if K != K2:
        raise RecError("inner dimensions differ: %d vs %d" % (K, K2))
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    ldc = _chk_mat(out, "out")
    if tuple(out.shape) != (M, N):
        raise RecError("out has shape %s, expected %s" % (tuple(out.shape), (M, N)))
    ld0 = ld1 = 0
    if aux0 is not None:
        ld0 = _chk_mat(aux0, "aux0")
    if aux1 is not None:
        ld1 = _chk_mat(aux1, "aux1")
    if bias is not None:
        _chk(bias, torch.float32, "bias")
        if bias.numel() != N:
            raise RecError("bias must have N elements")
    if b_colsum is not None:
        _chk(b_colsum, torch.float32, "b_colsum")
        if b_colsum.numel() != N:
            raise RecError("b_colsum must have N elements")
    rs_stride = 0
    if row_scale is not None:
        if not row_scale.is_cuda or row_scale.dtype != torch.float32 or row_scale.shape[0] != M:
            raise RecError("row_scale must be a float32 device tensor with M rows")
        rs_stride = row_scale.stride(0) if M > 1 else 1
    ld2 = 0
    if out2 is not None:
        ld2 = _chk_mat(out2, "out2")
        if tuple(out2.shape) != (M, N):
            raise RecError("out2 must have the shape of out")
    d = GemmDesc(M, N, K, lda, ldb, ldc, int(trans_a), int(trans_b), EPI[epilogue], int(split_k), int(max(num_cus, 0)))
    if num_cus > 0 and split_k <= 0:
        sp = C.c_int32(0)
        check(lib().rec_gemm_plan_splits(C.byref(d), int(num_cus), C.byref(sp)), "rec_gemm_plan_splits")
        d.split_k = sp.value
    pv = lambda t: None if t is None else t.data_ptr()
    x = GemmEpilogueArgs(pv(bias), pv(aux0), ld0, pv(aux1), ld1, pv(row_scale), rs_stride, pv(out2),
                         ld2, pv(b_colsum), pv(b_image))
    if _recorder is not None:      # a recorded step holds these addresses too
        _recorder.keep.extend(t for t in (bias, aux0, aux1, row_scale, out2, b_colsum, b_image) if t is not None)
    key = (M, N, K, lda, ldb, ldc, d.trans_a, d.trans_b, d.epilogue, d.split_k, d.num_cus,
           os.environ.get("REC_GEMM_BF16X3"), os.environ.get("REC_GEMM_BF16X3_DW"))
    need = _gemm_ws_cache.get(key)
    if need is None:       # a pure function of the descriptor: one C call per distinct GEMM, not per launch
        nbytes = C.c_size_t(0)
        check(lib().rec_gemm_f32_workspace_bytes(C.byref(d), C.byref(nbytes)))
        need = _gemm_ws_cache[key] = nbytes.value
    return d, x, out, need


def colsum(G, ws, out=None):
    ld = _chk_mat(G, "G")
    M, N = G.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=G.device)
    nbytes = C.c_size_t(0)
    check(lib().rec_colsum_workspace_bytes(M, N, C.byref(nbytes)))
    w = ws.get(nbytes.value)
    check(lib().rec_colsum(M, N, ld, _p(G), _p(out), _p(w), C.c_size_t(w.numel()), _stream()),
          "rec_colsum")
    return out


# ------------------------------------------------------------------ top MLP on the GEMM above
class _Acts(list):
    """mlp_forward's list of layer inputs; relu_bits[j] = the ReLU mask of acts[j] as bits, for mlp_backward's dX GEMMs."""
    relu_bits = None


def mlp_forward(x, weights, biases, ws, relu_last=False, out_last=None, images=None):
    """Linear(+bias)->ReLU ... ->Linear (deepfm/net.py:142-174) with Paddle-layout weights [in,out];
    bias and ReLU run in the GEMM epilogue.  relu_last: ReLU after the last layer too (the DNN tower of
    dcn_v2/net.py:161-184); out_last: buffer (view) for the last layer's output.
    images[i]: the bf16 x 3 image of weights[i] (GemmImages, current values) or None.
    Returns (y, acts): acts[i] = input of layer i."""
    acts = _Acts()
    bits = [None]                      # bits[j]: the ReLU mask of acts[j] as bits (gemm(relu_bits=)), or None
    n = len(weights)
    for i in range(n):
        acts.append(x)
        last = i == n - 1
        relu = not last or relu_last
        x = gemm(x, weights[i], ws, epilogue="bias_relu" if relu else "bias",
                 bias=biases[i], out=out_last if last else None, b_image=images[i] if images is not None else None,
                 relu_bits=bits if relu else None)
        if not relu:
            bits.append(None)
    acts.append(x)
    acts.relu_bits = bits
    return x, acts


def mlp_head_bwd(act, dz, w, ws, dw, db, relu=True, out=None):
    """Backward of a one-logit head Linear(n -> 1) behind a ReLU in ONE pass over act [B, n]: -> dx [B, n]; dw [n] (or
    [n,1]) and db [1] are written.  (rec_mlp_head_bwd)"""
    B, n = act.shape
    if out is None:
        out = torch.empty(B, n, dtype=torch.float32, device=act.device)
    nbytes = C.c_size_t(0)
    check(lib().rec_mlp_head_bwd_workspace_bytes(B, n, C.byref(nbytes)))
    wk = ws.get(nbytes.value)
    check(lib().rec_mlp_head_bwd(B, n, _p(act), act.stride(0), _p(dz), _p(w), 1 if relu else 0, _p(out), out.stride(0),
                                 _p(dw), _p(db), _p(wk), C.c_size_t(wk.numel()), _stream()), "rec_mlp_head_bwd")
    return out


def ctr_head_ok(w, dw):
    """The fused CTR head (rec_ctr_head_fwd_bwd) takes Linear(n -> 1) with n % 4 == 0, n <= 512 and an aligned contiguous
    weight (its input is the previous GEMM's own output buffer: contiguous, aligned)."""
    return (os.environ.get("REC_CTR_HEAD_FUSED", "1") != "0" and w.dim() == 2 and w.shape[1] == 1 and
            w.shape[0] % 4 == 0 and w.shape[0] <= 512 and w.is_contiguous() and w.data_ptr() % 16 == 0 and
            dw.is_contiguous())


def ctr_head(act, w, bias, y1, y2, label, ws, dw, db, eps=1e-4, clip=None, mean_over=0, relu=True, out=None):
    """Last Linear(n -> 1) + sigmoid + log_loss + mean AND their backward in one pass over act [B, n]
    (rec_ctr_head_fwd_bwd).  -> (pred [B,1], dz [B,1], loss [1], dx [B,n]); dw [n(,1)] and db [1] are written."""
    B, n = act.shape
    dev = act.device
    _chk(label, torch.int64, "label")
    for t, nm in ((y1, "y1"), (y2, "y2"), (bias, "bias")):
        _chk(t, torch.float32, nm)
    if label.numel() != B or (y1 is not None and y1.numel() != B) or (y2 is not None and y2.numel() != B):
        raise RecError("ctr_head: y1 / y2 / label must have one entry per row of act")
    if clip is not None and not clip[0] < clip[1]:
        raise RecError("clip must be (lo, hi) with lo < hi")
    if out is None:
        pred = torch.empty(B, 1, dtype=torch.float32, device=dev)
        dz = torch.empty(B, 1, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dx = torch.empty(B, n, dtype=torch.float32, device=dev)
    else:
        pred, dz, loss, dx = out
    nbytes = C.c_size_t(0)
    check(lib().rec_ctr_head_workspace_bytes(B, n, C.byref(nbytes)))
    wk = ws.get(nbytes.value)
    lo, hi = (float(clip[0]), float(clip[1])) if clip is not None else (0.0, 0.0)
    check(lib().rec_ctr_head_fwd_bwd(B, n, int(mean_over), _p(act), act.stride(0), _p(w), _p(bias), _p(y1), _p(y2),
                                     _p(label), float(eps), lo, hi, 1 if relu else 0, None, _p(pred), _p(dz), _p(loss),
                                     _p(dx), dx.stride(0), _p(dw), _p(db), _p(wk), C.c_size_t(wk.numel()), _stream()),
          "rec_ctr_head_fwd_bwd")
    return pred, dz, loss, dx


def _head_ok(act, w, dw, db, dy):
    """A one-logit head the fused backward takes: Linear(n -> 1), n % 4 == 0, n <= 512, aligned contiguous operands."""
    return (os.environ.get("REC_MLP_HEAD_FUSED", "1") != "0" and w.dim() == 2 and w.shape[1] == 1 and
            w.shape[0] % 4 == 0 and w.shape[0] <= 512 and act.dim() == 2 and act.stride(1) == 1 and
            act.stride(0) % 4 == 0 and act.data_ptr() % 16 == 0 and w.is_contiguous() and w.data_ptr() % 16 == 0 and
            dw.is_contiguous() and dy.is_contiguous() and dy.shape[-1] == 1 and act.shape[0] >= 64)


def mlp_backward(dy, acts, weights, dws, dbs, ws, defer_first=False, defer_all=False, dw_stream=None, dw_ws=None,
                 defer_split=0, images_t=None, defer_cus=0):
    """Backward of mlp_forward: dW_i -> dws[i], db_i -> dbs[i] (preallocated views); returns d(input).
    ReLU' is applied in the epilogue of the dX GEMM (mask = layer input > 0).
    defer_first: compute d(input) BEFORE dW_0 and return (d_input, finish) where finish() launches the
    dW_0 / db_0 GEMM — lets the caller start the HBM-bound consumers of d(input) on another stream
    underneath that MFMA-bound GEMM.
    defer_all: run the whole dX chain first and return (d_input, finish) with EVERY dW / db GEMM in finish()
    — the row-sharded step hides its gradient exchange, sparse optimizer and the next batch's lookup under them."""
    n = len(weights)
    g = dy
    imt = (lambda i: images_t[i]) if images_t is not None else (lambda i: None)   # images of weights[i]^T (the dX GEMMs)
    rbl = getattr(acts, "relu_bits", None)
    rb = (lambda i: rbl[i]) if rbl is not None else (lambda i: None)              # ReLU mask of acts[i] as bits
    head = n > 1 and _head_ok(acts[n - 1], weights[n - 1], dws[n - 1], dbs[n - 1], dy)
    if head:
        # the one-logit head: dX (with the ReLU mask of the layer in front), dW and db in ONE pass over its input
        g = mlp_head_bwd(acts[n - 1], dy, weights[n - 1], ws, dws[n - 1], dbs[n - 1], relu=True)
        n_run = n - 1
    else:
        n_run = n
    if defer_all:
        gs = [None] * n_run
        for i in reversed(range(n_run)):
            gs[i] = g
            g = gemm(g, weights[i], ws, trans_b=True, b_image=imt(i),
                     **(dict(epilogue="relu_mask", aux0=acts[i], relu_bits=rb(i)) if i > 0 else {}))

        def finish(num_cus=0):
            for i in reversed(range(n_run)):
                gemm(acts[i], gs[i], ws, trans_a=True, out=dws[i], b_colsum=dbs[i], num_cus=num_cus)
        return g, finish
    cur = torch.cuda.current_stream() if dw_stream is not None else None
    keep = []          # dw_stream: tensors the other stream still reads (the caching allocator must not recycle them)

    def join():
        if dw_stream is not None:
            cur.wait_stream(dw_stream)
            keep.clear()
    for i in reversed(range(n_run)):
        if i == 0 and defer_first:
            g0 = g
            d_in = gemm(g0, weights[0], ws, trans_b=True, b_image=imt(0))
            join()
            # defer_cus: the deferred dW_0's grid sized for that many CUs (the bf16 x 3 kernel: one block per CU)
            # defer_split: K split of the deferred dW_0 GEMM (0 = the planner's: one resident round of blocks) — the
            # caller that runs an HBM-bound kernel beside it asks for fewer, longer blocks, which leave it wave slots
            return d_in, (lambda: gemm(acts[0], g0, ws, trans_a=True, out=dws[0], b_colsum=dbs[0], split_k=defer_split,
                                    num_cus=defer_cus))
        if dw_stream is not None:
            # dW_i and dX_i both consume g_i and nothing of each other: two streams, so that the half-empty last
            # round of blocks of one GEMM is filled by the other
            dw_stream.wait_stream(cur)
            keep.append(g)
            with torch.cuda.stream(dw_stream):
                gemm(acts[i], g, dw_ws if dw_ws is not None else ws, trans_a=True, out=dws[i], b_colsum=dbs[i])
        else:
            # dW = X^T G, db = colsum(G) and dX = G W^T (+ ReLU') in one call: one launch at launch-bound sizes
            g = linear_backward(acts[i], g, weights[i], ws, dws[i], dbs[i], relu_src=acts[i] if i > 0 else None,
                                b_image=imt(i), relu_bits=rb(i) if i > 0 else None)
            continue
        if i > 0:
            g = gemm(g, weights[i], ws, trans_b=True, epilogue="relu_mask", aux0=acts[i], b_image=imt(i), relu_bits=rb(i))
        else:
            g = gemm(g, weights[i], ws, trans_b=True, b_image=imt(i))
    join()
    return g


# ------------------------------------------------------------------ loss head / metric
def sigmoid_logloss(y1, y2, y_dnn, label, ws, eps=1e-4, want_dz=True, out=None, mean_over=0, clip=None):
    """-> pred [B,1], dz [B,1] (or None), loss [1].  mean_over: denominator of the mean (0 -> B).
    clip = (lo, hi): paddle.clip on the logit in front of the sigmoid (slot_dnn/net.py:84)."""
    B = y1.numel()
    dev = y1.device
    _chk(y1, torch.float32, "y1")
    _chk(y2, torch.float32, "y2")
    _chk(y_dnn, torch.float32, "y_dnn")
    _chk(label, torch.int64, "label")
    if clip is not None and not clip[0] < clip[1]:
        raise RecError("clip must be (lo, hi) with lo < hi")
    if out is None:
        pred = torch.empty(B, 1, dtype=torch.float32, device=dev)
        dz = torch.empty(B, 1, dtype=torch.float32, device=dev) if want_dz else None
        loss = torch.empty(1, dtype=torch.float32, device=dev)
    else:
        pred, dz, loss = out
    nbytes = C.c_size_t(0)
    check(lib().rec_logloss_workspace_bytes(B, C.byref(nbytes)))
    w = ws.get(nbytes.value)
    lo, hi = (float(clip[0]), float(clip[1])) if clip is not None else (0.0, 0.0)
    check(lib().rec_sigmoid_logloss(B, int(mean_over), _p(y1), _p(y2), _p(y_dnn), _p(label), float(eps), lo, hi,
                                    _p(pred), _p(dz), _p(loss), _p(w), C.c_size_t(w.numel()), _stream()),
          "rec_sigmoid_logloss")
    return pred, dz, loss


def auc_histogram(pred, label, stat_pos, stat_neg, num_thresholds=4095):
    _chk(pred, torch.float32, "pred")
    _chk(label, torch.int64, "label")
    _chk(stat_pos, torch.int64, "stat_pos", (num_thresholds + 1,))
    _chk(stat_neg, torch.int64, "stat_neg", (num_thresholds + 1,))
    check(lib().rec_auc_histogram(pred.numel(), _p(pred), _p(label), num_thresholds, _p(stat_pos),
                                  _p(stat_neg), _stream()), "rec_auc_histogram")


def fill_uniform(buf, lo, hi, seed):
    _chk(buf, torch.float32, "buf")
    check(lib().rec_fill_uniform(buf.numel(), _p(buf), float(lo), float(hi), int(seed), _stream()),
          "rec_fill_uniform")
    return buf


_SIDE_STREAMS = {}


def concurrent_stream(device, tries=8, micros=200, priority=None, index=0):
    """A side stream whose kernels really run CONCURRENTLY with the current stream's.  HIP multiplexes streams
    onto a few hardware queues (4 by default, shared with the streams RCCL and rocPRIM create); two streams on
    one queue execute strictly in submission order, which silently turns 'sparse optimizer underneath the dW
    GEMM' into 'after it'.  Probe: spin `micros` on a candidate, then on the current stream, and accept the
    candidate if the second spin did not have to wait for the first (events).  Cached per (device, main stream).
    index > 0: a further stream that also overlaps with the side streams of the lower indices (its own hardware queue)."""
    device = torch.device(device)
    main = torch.cuda.current_stream(device)
    if priority is None:
        priority = int(os.environ.get("REC_SIDE_PRIORITY", "0"))      # -1 = high priority queue (measured: no effect)
    key = (device.index, main.cuda_stream, priority, int(index))
    if key in _SIDE_STREAMS:
        return _SIDE_STREAMS[key]
    others = [main] + [concurrent_stream(device, tries, micros, priority, j) for j in range(int(index))]

    def overlaps(s, o):
        for st in (s, o):                                         # first use of a stream binds its queue
            check(lib().rec_stream_spin(1, C.c_void_p(st.cuda_stream)), "rec_stream_spin")
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(o)
        check(lib().rec_stream_spin(micros, C.c_void_p(s.cuda_stream)), "rec_stream_spin")
        check(lib().rec_stream_spin(micros, C.c_void_p(o.cuda_stream)), "rec_stream_spin")
        e1.record(o)
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) * 1e3 < 1.5 * micros
    s, keep = None, []
    with torch.cuda.device(device):
        for _ in range(tries):
            s = torch.cuda.Stream(device=device, priority=priority)
            keep.append(s)          # rejected candidates stay alive until the search ends: a freed stream's queue slot
            if all(overlaps(s, o) for o in others):               # would be handed to the next candidate again
                break
    _SIDE_STREAMS[key] = s
    return s


_CU_STREAMS = {}


def cu_range_stream(device, cu_begin, cu_end):
    """torch stream (ExternalStream over rec_stream_create_cu_range) whose kernels only run on compute units
    [cu_begin, cu_end).  Cached for the life of the process."""
    device = torch.device(device)
    key = (device.index, int(cu_begin), int(cu_end))
    if key not in _CU_STREAMS:
        h = C.c_void_p()
        with torch.cuda.device(device):
            check(lib().rec_stream_create_cu_range(int(cu_begin), int(cu_end), C.byref(h)),
                  "rec_stream_create_cu_range")
        _CU_STREAMS[key] = torch.cuda.ExternalStream(h.value, device=device)
    return _CU_STREAMS[key]


def cu_stride_stream(device, first, stride, cu_total=256):
    """torch stream whose kernels only run on every `stride`-th compute unit (rec_stream_create_cu_stride)."""
    device = torch.device(device)
    key = (device.index, "stride", int(first), int(stride), int(cu_total))
    if key not in _CU_STREAMS:
        h = C.c_void_p()
        with torch.cuda.device(device):
            check(lib().rec_stream_create_cu_stride(int(first), int(stride), int(cu_total), C.byref(h)),
                  "rec_stream_create_cu_stride")
        _CU_STREAMS[key] = torch.cuda.ExternalStream(h.value, device=device)
    return _CU_STREAMS[key]


# ------------------------------------------------------------------ host: feature hash
def xxh32(data: bytes, seed: int = 0) -> int:
    return int(lib().rec_xxh32(data, len(data), seed))


def hash_features(field_idx, values, hash_dim=1000001):
    """xxh32(str(idx)+value) % hash_dim for each (idx, value) — dnn/benchmark_reader.py:52."""
    n = len(values)
    arr = (C.c_char_p * n)(*[v.encode("utf-8") for v in values])
    idx = (C.c_int32 * n)(*[int(i) for i in field_idx])
    out = (C.c_int64 * n)()
    check(lib().rec_xxh32_hash_mod(arr, idx, n, hash_dim, out), "rec_xxh32_hash_mod")
    return list(out)
