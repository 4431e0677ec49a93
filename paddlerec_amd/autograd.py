"""torch.autograd adapter over the C-ABI (SURVEY.md §8(b) "Python adapters (i)").

The host mirrors (deepfm.py, ...) run an explicit backward chain; a maintainer who keeps a torch (or, through the
same pattern, a Paddle `PyLayer`) autograd graph binds the fused block like this instead:

    y1, y2, feat = deepfm_fm(ids, dense, W, W1, dense_w, dense_w_one)          # deepfm/net.py:105-139
    pred = torch.sigmoid(y1 + y2 + mlp(feat.flatten(1)));  loss.backward()

forward  = rec_deepfm_fm_fwd, backward = rec_deepfm_fm_bwd.  The two tables get SPARSE gradients — the
SelectedRows of `nn.Embedding(sparse=True)` (deepfm/net.py:62-70,80): rows = the flattened ids (padding dropped),
values = the per-lookup gradient rows, unmerged; `.coalesce()` is Paddle's MergeAdd.  There is no CPU fallback: the
operator backend is paddlerec_amd.ops (tests inject an oracle-backed stand-in to check the autograd plumbing).
"""
import torch

from . import ops


class DeepFMFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, dense, W, W1, dense_w, dense_w_one, padding_idx, k):
        y1, y2, feat, sum_emb, status = k.deepfm_fm_fwd(ids, dense, W, W1, dense_w, dense_w_one, padding_idx, None,
                                                        None)
        ctx.save_for_backward(ids, dense, feat, sum_emb, dense_w)
        ctx.k, ctx.padding_idx, ctx.shapes = k, padding_idx, (tuple(W.shape), tuple(W1.shape), tuple(dense_w.shape))
        ctx.status = status
        return y1, y2, feat

    @staticmethod
    def backward(ctx, dy1, dy2, dfeat):
        ids, dense, feat, sum_emb, dense_w = ctx.saved_tensors
        k = ctx.k
        B, S = ids.shape
        D, Dn = feat.shape[2], dense.shape[1]
        dev = feat.device
        out = (torch.empty(B * S, D, dtype=torch.float32, device=dev),
               torch.empty(Dn, D, dtype=torch.float32, device=dev), torch.empty(Dn, dtype=torch.float32, device=dev))
        zeros = lambda ref, g: torch.zeros_like(ref) if g is None else g.contiguous()
        dy1 = zeros(feat.new_empty(B, 1), dy1)
        row_grad, d_dense_w, d_dense_w_one = k.deepfm_fm_bwd(dense, feat, sum_emb, zeros(feat, dfeat), dy1,
                                                             zeros(feat.new_empty(B, 1), dy2), S, k.Workspace(dev),
                                                             out=out)
        rows = ids.reshape(-1)
        keep = torch.ones_like(rows, dtype=torch.bool) if ctx.padding_idx is None else rows != ctx.padding_idx
        idx = rows[keep].unsqueeze(0)
        shp_w, shp_w1, shp_dw = ctx.shapes
        gW = torch.sparse_coo_tensor(idx, row_grad[keep], shp_w)
        gW1 = torch.sparse_coo_tensor(idx, dy1.reshape(B, 1).expand(B, S).reshape(-1, 1)[keep], shp_w1)
        return None, None, gW, gW1, d_dense_w.reshape(shp_dw), d_dense_w_one, None, None


def deepfm_fm(ids, dense, W, W1, dense_w, dense_w_one, padding_idx=0, kernels=None):
    """FM.forward of deepfm/net.py:105-139 as one differentiable op -> (y_first_order [B,1], y_second_order [B,1],
    feat_embeddings [B,S+Dn,D]).  ids [B,S] int64; W [N,D], W1 [N,1] leaf tables (sparse gradients)."""
    return DeepFMFunction.apply(ids, dense, W, W1, dense_w, dense_w_one, padding_idx, kernels if kernels is not None else ops)
