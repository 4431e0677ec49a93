"""Host-side mirror of the reference's DIN plugin on the recengine HIP kernels (forward / inference).

Mirrors /root/reference/models/rank/din/net.py:20-184 (DINLayer) with the reference's parameter names
(SURVEY.md App. C): three independent item tables, three cat tables, item_b, the attention MLP (kept in
`attention_layer`, NOT in state_dict — App. B-9), linearCon and the top MLP `linear_{0,1,2}`.
The attention-pool (net.py:141-173) is ONE fused kernel (rec_din_attention_pool_fwd); linearCon and the
top MLP run on rec_gemm_f32 with the sigmoid in the epilogue; `logit + item_b` is the last GEMM's
epilogue.  Training (backward of the attention-pool) is not built yet — see DESIGN.md.
"""
import math

import torch

from . import ops


def _xavier_uniform_(t, fan_in, fan_out):
    b = math.sqrt(6.0 / (fan_in + fan_out))
    return t.uniform_(-b, b)


class DINLayer:
    """din/net.py:20-184.  forward(...) -> logit [B,1]."""

    def __init__(self, item_emb_size, cat_emb_size, act, is_sparse, use_DataLoader, item_count, cat_count,
                 device="cuda"):
        self.device = torch.device(device)
        self.item_emb_size, self.cat_emb_size = item_emb_size, cat_emb_size
        self.item_count, self.cat_count = item_count, cat_count
        f32 = dict(dtype=torch.float32, device=self.device)
        E = item_emb_size + cat_emb_size
        self.firInDim = E
        self.params = {}
        for name, rows, dim in (("hist_item_emb_attr", item_count, item_emb_size),
                                ("hist_cat_emb_attr", cat_count, cat_emb_size),
                                ("target_item_emb_attr", item_count, item_emb_size),
                                ("target_cat_emb_attr", cat_count, cat_emb_size),
                                ("target_item_seq_emb_attr", item_count, item_emb_size),
                                ("target_cat_seq_emb_attr", cat_count, cat_emb_size)):
            self.params[name + ".weight"] = _xavier_uniform_(torch.empty(rows, dim, **f32), rows, dim)
        self.params["item_b_attr.weight"] = torch.zeros(item_count, 1, **f32)          # net.py:77-82
        # attention MLP (net.py:84-102): forward uses it, state_dict does not list it (App. B-9)
        sizes = [4 * E, 80, 40, 1]
        self.attention_w = [_xavier_uniform_(torch.empty(sizes[i], sizes[i + 1], **f32), sizes[i], sizes[i + 1])
                            for i in range(3)]
        self.attention_b = [torch.zeros(sizes[i + 1], **f32) for i in range(3)]
        self.params["linearCon.weight"] = _xavier_uniform_(torch.empty(E, E, **f32), E, E)  # net.py:109-117
        self.params["linearCon.bias"] = torch.zeros(E, **f32)
        con = [2 * E, 80, 40, 1]                                                         # net.py:119-137
        for i in range(3):
            self.params["linear_%d.weight" % i] = _xavier_uniform_(torch.empty(con[i], con[i + 1], **f32),
                                                                   con[i], con[i + 1])
            self.params["linear_%d.bias" % i] = torch.zeros(con[i + 1], **f32)
        self.ws = ops.Workspace(self.device)
        self.status = ops.new_status(self.device)

    def state_dict(self):
        return dict(self.params)

    def set_dict(self, sd):
        for k, v in sd.items():
            self.params[k].copy_(torch.as_tensor(v).to(self.device).reshape(self.params[k].shape))

    def set_attention(self, weights, biases):
        for dst, src in zip(self.attention_w + self.attention_b, list(weights) + list(biases)):
            dst.copy_(torch.as_tensor(src).to(self.device).reshape(dst.shape))

    def forward(self, hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq,
                target_cat_seq):
        p, E = self.params, self.firInDim
        B, T = hist_item_seq.shape
        mask2 = mask.reshape(B, T).contiguous()
        pooled, _, _ = ops.din_attention_pool(
            hist_item_seq, hist_cat_seq, target_item_seq, target_cat_seq, mask2,
            p["hist_item_emb_attr.weight"], p["hist_cat_emb_attr.weight"],
            p["target_item_seq_emb_attr.weight"], p["target_cat_seq_emb_attr.weight"],
            self.attention_w, self.attention_b, self.status, want_weights=False)       # net.py:141-173
        emb = torch.empty(B, 2 * E, dtype=torch.float32, device=self.device)           # net.py:178
        ops.gemm(pooled, p["linearCon.weight"], self.ws, epilogue="bias", bias=p["linearCon.bias"],
                 out=emb[:, :E])                                                         # net.py:175-176
        ti, tc = target_item.reshape(-1).contiguous(), target_cat.reshape(-1).contiguous()
        ops.emb_gather(ti, p["target_item_emb_attr.weight"], None, self.status, out=emb[:, E:],
                       out_group=1, out_group_stride=2 * E)                              # net.py:143,152
        ops.emb_gather(tc, p["target_cat_emb_attr.weight"], None, self.status,
                       out=emb[:, E + self.item_emb_size:], out_group=1, out_group_stride=2 * E)
        item_b, _ = ops.emb_gather(ti, p["item_b_attr.weight"], None, self.status)      # net.py:147
        x = ops.gemm(emb, p["linear_0.weight"], self.ws, epilogue="bias_sigmoid", bias=p["linear_0.bias"])
        x = ops.gemm(x, p["linear_1.weight"], self.ws, epilogue="bias_sigmoid", bias=p["linear_1.bias"])
        return ops.gemm(x, p["linear_2.weight"], self.ws, epilogue="add", bias=p["linear_2.bias"],
                        aux1=item_b)                                                     # net.py:180-183

    __call__ = forward
