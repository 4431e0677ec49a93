"""NumPy restatement of the PS / gpubox sparse path (oracle — test infrastructure only).

Reference call sites: models/rank/dnn/net.py:67-82 (sparse_embedding(size=[N, D+2]) +
continuous_value_model(emb, show_click, use_cvm=False)), dnn/static_model.py:86-94 (show = ones, click =
label), slot_dnn/config_online.yaml:57-79 (SparseAdaGradSGDRule lr 0.05, initial_g2sum 3.0, bounds +-10).
The accessor arithmetic itself is in the un-vendored PaddlePaddle PS code — **parity unpinned**; the
formula is the one fixed by SURVEY.md App. B-13.
"""
import numpy as np


def cvm_lookup(rec, ids, D, padding_idx=None):
    """sparse_embedding returns [show, click, embed...]; CVM(use_cvm=False) drops the two CVM columns (App. B-8)."""
    out = rec[ids][..., 4:4 + D]      # engine record: [show | click | g2sum_w | g2sum_x | W(D) | pad]
    if padding_idx is not None:
        out = out * (ids != padding_idx)[..., None]
    return out


def adagrad_rows(rec, D, uniq, merged, shows, clicks, lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0)):
    """In place on rec [N, stride] rows `uniq` with merged gradients [U,D], show/click increments [U]."""
    f = np.float32
    for u, row in enumerate(uniq):
        r = rec[row]
        r[0] += f(shows[u])
        r[1] += f(clicks[u])
        g2w, g2x = r[2], r[3]
        sw = np.sqrt(f(initial_g2sum) / (f(initial_g2sum) + g2w))
        sx = np.sqrt(f(initial_g2sum) / (f(initial_g2sum) + g2x))
        g = merged[u].astype(f)
        scale = np.full(D, sx, f)
        scale[0] = sw
        r[4:4 + D] = np.clip(r[4:4 + D] - f(lr) * g * scale, f(bounds[0]), f(bounds[1]))
        r[2] = g2w + g[0] * g[0]
        if D > 1:
            acc = f(0)
            for d in range(1, D):
                acc = acc + g[d] * g[d]
            r[3] = g2x + acc / f(D - 1)
