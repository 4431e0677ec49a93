"""Row P + the BenchmarkDNNLayer net (oracle — TEST INFRASTRUCTURE ONLY, never imported by paddlerec_amd/).

NumPy restatement of
    /root/reference/models/rank/slot_dnn/net.py:55-85      BenchmarkDNNLayer.forward
        for every slot: sparse_embedding(padding_idx=0, ONE shared table "embedding") -> sequence_pool('sum')
        concat(axis=1) -> Linear/ReLU stack -> sigmoid(clip(y, -15, 15))
    /root/reference/models/rank/slot_dnn/static_model.py:104-108   log_loss + mean
over the slot-major CSR batch layout of rec_parse_feasign_slots (values | lod [S, B+1] | slot_base).

Parity status: pinned to tests/golden/slot_dnn_D9.npz, which oracle/make_golden.py writes by executing the
reference's unmodified slot_dnn/net.py over oracle/paddle_shim (sparse_embedding / sequence_pool / clip follow
SURVEY App. B-7 [EXT]); the PS hash map of the reference (exact, keyed by the uint64 feasign) is replaced by the
engine's hashed table — `feasign_rows` below is the restatement of that mapping (murmur3 fmix64, public domain).
"""
import numpy as np

from . import deepfm_ref as R

M64 = (1 << 64) - 1


def mix64(k):
    """murmur3 fmix64 on Python ints (the published finaliser)."""
    k &= M64
    k ^= k >> 33
    k = (k * 0xff51afd7ed558ccd) & M64
    k ^= k >> 33
    k = (k * 0xc4ceb9fe1a85ec53) & M64
    k ^= k >> 33
    return k


def feasign_rows(keys, num_rows):
    """uint64 feasigns -> rows of a hashed table: 0 -> 0 (padding row), f -> 1 + mix64(f) % (num_rows - 1)."""
    k = np.asarray(keys).astype(np.uint64).reshape(-1)
    out = np.empty(k.shape, np.int64)
    for i, f in enumerate(k.tolist()):
        out[i] = 0 if f == 0 else 1 + mix64(f) % (num_rows - 1)
    return out.reshape(np.asarray(keys).shape)


def csr_from_samples(samples, num_slots):
    """samples[b][s] = list of ids of slot s in sample b -> (values, lod [S,B+1], slot_base [S+1]) slot-major."""
    B = len(samples)
    values, lod, base = [], np.zeros((num_slots, B + 1), np.int64), np.zeros(num_slots + 1, np.int64)
    for s in range(num_slots):
        base[s] = len(values)
        for b in range(B):
            values.extend(int(v) for v in samples[b][s])
            lod[s, b + 1] = len(values) - base[s]
    base[num_slots] = len(values)
    # uint64 feasigns travel as int64 bit patterns
    return np.array(values, dtype=np.uint64).astype(np.int64), lod, base


def multislot_sumpool(values, lod, slot_base, W, padding_idx=0, key_mode=0, num_rows=None):
    """slot_dnn/net.py:63-77: out[b, s*D:(s+1)*D] = sum of W[row(id)] over the ids of (sample b, slot s), padding ids
    skipped (ascending order).  Returns (out [B, S*D], counts [B,S] int32, seg_of_value [nnz] int32, rows [nnz])."""
    S, B = lod.shape[0], lod.shape[1] - 1
    D = W.shape[1]
    N = int(num_rows if num_rows is not None else W.shape[0])
    out = np.zeros((B, S * D), np.float32)
    counts = np.zeros((B, S), np.int32)
    seg = np.zeros(len(values), np.int32)
    rows = np.zeros(len(values), np.int64)
    for s in range(S):
        for b in range(B):
            acc = np.zeros(D, np.float32)
            for k in range(int(slot_base[s] + lod[s, b]), int(slot_base[s] + lod[s, b + 1])):
                v = int(values[k])
                seg[k] = b * S + s
                if padding_idx is not None and padding_idx >= 0 and v == padding_idx:
                    rows[k] = 0 if key_mode else padding_idx
                    continue
                r = (0 if v == 0 else 1 + mix64(v) % (N - 1)) if key_mode else v
                rows[k] = r
                acc = acc + W[r]
                counts[b, s] += 1
            out[b, s * D:(s + 1) * D] = acc
    return out, counts, seg, rows


def forward(values, lod, slot_base, W, mlp_w, mlp_b, padding_idx=0, key_mode=0, num_rows=None, clip=(-15.0, 15.0)):
    """-> (pred [B,1], cache).  net.py:77-84."""
    x, counts, seg, rows = multislot_sumpool(values, lod, slot_base, W, padding_idx, key_mode, num_rows)
    y, acts = R.dnn_forward(x, mlp_w, mlp_b, return_acts=True)
    yc = np.clip(y, np.float32(clip[0]), np.float32(clip[1]))
    pred = R.sigmoid(yc).astype(np.float32)
    return pred, dict(x=x, y=y, acts=acts, counts=counts, seg=seg, rows=rows)


def loss_and_grads(values, lod, slot_base, label, W, mlp_w, mlp_b, padding_idx=0, key_mode=0, num_rows=None,
                   clip=(-15.0, 15.0)):
    """static_model.py:104-108 loss + what backward produces: dense MLP gradients, d_pool [B, S*D] and the merged
    (per unique row, ascending) sparse gradient of the shared table."""
    pred, c = forward(values, lod, slot_base, W, mlp_w, mlp_b, padding_idx, key_mode, num_rows, clip)
    loss = R.log_loss_mean(pred, label)
    dz = R.log_loss_mean_grad_z(pred, label).astype(np.float32)
    dz = dz * ((c["y"] > clip[0]) & (c["y"] < clip[1]))        # paddle.clip gradient [EXT]: open interval
    d_pool, dws, dbs = R.dnn_backward(dz, c["acts"], mlp_w)
    D = W.shape[1]
    rows, seg = c["rows"], c["seg"]
    live = np.ones(len(values), bool)
    if padding_idx is not None and padding_idx >= 0:
        live = np.asarray(values) != padding_idx          # the padding id is compared with the VALUE (before hashing)
    uniq = np.unique(rows[live])
    merged = np.zeros((len(uniq), D), np.float32)
    pos = {int(r): i for i, r in enumerate(uniq)}
    dp = d_pool.reshape(-1, D)
    for k in np.nonzero(live)[0]:                      # ascending position order, as the engine's merge
        merged[pos[int(rows[k])]] += dp[seg[k]]
    return dict(loss=loss, pred=pred, dz=dz, d_pool=d_pool, dws=dws, dbs=dbs, uniq=uniq, merged=merged,
                counts=c["counts"], pool=c["x"], seg=seg, rows=rows)
