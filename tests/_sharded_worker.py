"""One rank of a sharded DeepFM run (spawned by tests/test_sharded*.py).

    python tests/_sharded_worker.py <rank> <world> <port> <cpu|gpu> <outdir> [tables]

cpu : gloo, CPU tensors, operator backend = tests/cpu_kernels.py (oracle) — exercises the host
      orchestration only.
gpu : gloo with host staging, all ranks on cuda:0, operator backend = the HIP kernels.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from helpers import deepfm_state_dict, make_deepfm_problem  # noqa: E402

CFG = dict(B=48, N=257, D=16, fc=(32, 16), seed=77, pad_frac=0.08, steps=2, lr=1e-2)
ZIPF = os.environ.get("REC_TEST_ZIPF", "0") == "1"       # hot rows: one row owning a large share of a batch's lookups


def later_ids(rng, c, world, zipf=ZIPF):
    """ids of the steps after the first (the same global batch on every rank and in the oracle run)."""
    if zipf:
        return np.minimum(rng.zipf(1.3, size=(c["B"] * world, 26)), c["N"] - 1).astype(np.int64)
    return rng.integers(0, c["N"], (c["B"] * world, 26), dtype=np.int64)


def main():
    rank, world, port, mode, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    tables = len(sys.argv) > 6 and sys.argv[6] == "tables"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from paddlerec_amd.sharded import Comm, ShardedDeepFMLayer
    if mode == "cpu":
        import cpu_kernels
        dev, kernels = "cpu", cpu_kernels
    else:
        dev, kernels = "cuda:0", None
    c = CFG
    pr = make_deepfm_problem(B=c["B"] * world, N=c["N"], D=c["D"], fc=c["fc"], seed=c["seed"],
                             pad_frac=c["pad_frac"], tables=tables, zipf=ZIPF)
    so = pr["slot_offsets"]
    comm = Comm()
    comm.trace = []
    torch.manual_seed(1000 + rank)          # every rank draws a DIFFERENT random init ...
    m = ShardedDeepFMLayer(pr["N"], c["D"], 13, 26, list(c["fc"]), device=dev,
                           slot_offset=None if so is None else torch.as_tensor(so),
                           comm=comm, kernels=kernels)
    # ... and the constructor must have replaced the dense replica by rank 0's (data-parallel invariant)
    mine = m.dense.data.detach().cpu().clone()
    ref0 = mine.clone()
    dist.broadcast(ref0, src=0)
    assert torch.equal(mine, ref0), "dense parameters differ across ranks after construction"
    m.set_dict(deepfm_state_dict(pr["params"], len(c["fc"]) + 1))
    rng = np.random.default_rng(c["seed"] + 1)
    out = {}
    lo, hi = rank * c["B"], (rank + 1) * c["B"]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)
    batches = []
    for step in range(c["steps"]):
        if step == 0:
            ids, dense, label = pr["ids"], pr["dense"], pr["label"]
        else:      # later steps: fresh global batch, same on every rank
            ids = later_ids(rng, c, world)
            dense = rng.random((c["B"] * world, 13), dtype=np.float32)
            label = (rng.random((c["B"] * world, 1)) < 0.3).astype(np.int64)
        batches.append((t(ids[lo:hi]), t(dense[lo:hi]), t(label[lo:hi])))
    no_reads = os.environ.get("REC_TEST_NO_HOST_READS", "0") == "1"
    for step, (ids_t, dense_t, label_t) in enumerate(batches):
        nxt = batches[step + 1][0] if step + 1 < len(batches) else None     # routed a step ahead
        if no_reads:       # the deduplicated exchange: nothing on the step path reads a device value back to size anything
            def boom(*a, **k):
                raise AssertionError("host read (.item() / .tolist()) on the step path")
            saved = torch.Tensor.item, torch.Tensor.tolist
            torch.Tensor.item = torch.Tensor.tolist = boom
            try:
                loss, pred = m.train_step(ids_t, dense_t, label_t, lr=c["lr"], next_sparse_inputs=nxt)
            finally:
                torch.Tensor.item, torch.Tensor.tolist = saved
        else:
            loss, pred = m.train_step(ids_t, dense_t, label_t, lr=c["lr"], next_sparse_inputs=nxt)
        out["loss%d" % step] = loss.cpu().numpy().copy()
        out["pred%d" % step] = pred.cpu().numpy().copy()   # pred is a reused buffer
    pred_eval = m.forward(t(pr["ids"][lo:hi]), t(pr["dense"][lo:hi]))
    out["pred_eval"] = pred_eval.cpu().numpy()
    W, W1 = m.gather_global_tables()
    out["W"], out["W1"] = W.cpu().numpy(), W1.cpu().numpy()
    out["mlp_w0"] = m.dense.p["dnn.linear_0.weight"].cpu().numpy()
    from helpers import layer_moments
    mom = layer_moments(m)
    out["m_mlp_w0"], out["v_mlp_w0"] = mom["dnn.linear_0.weight"]
    out["m_dense_w"], out["v_dense_w"] = mom["fm.dense_w"]
    # this rank's rows (r, r+G, ...) of the table moments
    out["mW_local"] = m.sparse_state["m"].cpu().numpy()
    out["vW_local"] = m.sparse_state["v"].cpu().numpy()
    out["dense_w"] = m.dense.p["fm.dense_w"].cpu().numpy()
    out["status"] = m.status.cpu().numpy()
    out["trace"] = np.asarray(m.comm.trace)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
