"""NumPy restatement of the row-sharded table routing (oracle — test infrastructure only).

The reference's multi-GPU sparse path is `core.PSGPU` (HeterPS) driven from
tools/static_gpubox_trainer.py:152-160,256 and models/rank/dnn/net.py:71-79; its key->GPU sharding
lives in the un-vendored PaddlePaddle build [EXT].  The engine's contract is
owner(r) = r mod G, local row = r div G; this file states it in the plainest possible form so the
HIP partition kernel (rec_shard_route) has a bit-exact target.
"""
import numpy as np

from .deepfm_ref import PADDING_IDX, effective_rows


def shard_route(ids, num_shards, padding_idx=PADDING_IDX, slot_offsets=None):
    """ids [B,S] int64 -> dict(send_local_row, send_pos, send_sample [n_send] int64 grouped by owner
    with ascending position inside a group; slot_of_pos [B*S] int64 (1 + send index, 0 = padding);
    send_counts [G+1] int64 ([G] = dropped positions))."""
    ids = np.asarray(ids, dtype=np.int64)
    B, S = ids.shape
    rows, valid = effective_rows(ids, padding_idx, slot_offsets)
    rows, valid = rows.reshape(-1), valid.reshape(-1)
    G = int(num_shards)
    pos = np.nonzero(valid)[0]
    owner = rows[pos] % G
    order = np.argsort(owner, kind="stable")
    spos = pos[order]
    slot_of_pos = np.zeros(B * S, dtype=np.int64)
    slot_of_pos[spos] = np.arange(1, len(spos) + 1)
    counts = np.zeros(G + 1, dtype=np.int64)
    counts[:G] = np.bincount(owner, minlength=G)
    counts[G] = B * S - len(pos)
    return dict(send_local_row=rows[spos] // G, send_pos=spos.astype(np.int64),
                send_sample=(spos // S).astype(np.int64), slot_of_pos=slot_of_pos,
                send_counts=counts)


def shard_of_table(W, rank, num_shards):
    """Rows of a global table owned by `rank`: r % G == rank, in local-row order r // G."""
    return W[rank::num_shards]
