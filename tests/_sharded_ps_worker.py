"""One rank of a row-sharded DeepFM run on the PS / gpubox table (spawned by tests/test_sharded.py).

    python tests/_sharded_ps_worker.py <rank> <world> <port> <cpu|gpu> <outdir>

uint64 feasign ids hashed on the "device", the accessor table (AdaGrad rule, show/click, lazy birth, embedx
threshold) sharded row-wise; cpu: gloo + the oracle-backed operator stand-in (host orchestration only)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from helpers import deepfm_state_dict, make_deepfm_problem  # noqa: E402

CFG = dict(B=40, N=199, D=16, fc=(32, 16), seed=91, steps=3, lr=1e-2,
           accessor=dict(lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0), initial_range=1e-2, embedx_threshold=4.0,
                         nonclk_coeff=0.1, click_coeff=1.0, seed=4242))


def make_batches(world):
    """Global batches of uint64 feasigns (a small pool, so that rows repeat within and across steps; 5 % padding)."""
    c = CFG
    rng = np.random.default_rng(c["seed"])
    pool = rng.integers(1, 2 ** 64, size=600, dtype=np.uint64)
    out = []
    for _ in range(c["steps"]):
        keys = pool[rng.integers(0, len(pool), (c["B"] * world, 26))]
        keys[rng.random(keys.shape) < 0.05] = 0
        dense = rng.random((c["B"] * world, 13), dtype=np.float32)
        label = (rng.random((c["B"] * world, 1)) < 0.3).astype(np.int64)
        out.append((keys.astype(np.int64), dense, label))
    return out


def main():
    rank, world, port, mode, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from paddlerec_amd.sharded import Comm, ShardedDeepFMLayer
    if mode == "cpu":
        import cpu_kernels
        dev, kernels = "cpu", cpu_kernels
    else:
        dev, kernels = "cuda:0", None
    c = CFG
    pr = make_deepfm_problem(B=4, N=c["N"], D=c["D"], fc=c["fc"], seed=c["seed"])
    comm = Comm()
    comm.trace = []
    torch.manual_seed(5 + rank)
    m = ShardedDeepFMLayer(c["N"], c["D"], 13, 26, list(c["fc"]), device=dev, comm=comm, kernels=kernels,
                           table="ps", accessor=c["accessor"], hash_keys=True)
    sd = deepfm_state_dict(pr["params"], len(c["fc"]) + 1)
    sd.pop("fm.embedding.weight"), sd.pop("fm.embedding_one.weight")     # the table is born lazily
    m.set_dict(sd)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)
    lo, hi = rank * c["B"], (rank + 1) * c["B"]
    batches = [(t(k[lo:hi]), t(d[lo:hi]), t(l[lo:hi])) for k, d, l in make_batches(world)]
    out = {}
    for step, (ids_t, dense_t, label_t) in enumerate(batches):
        nxt = batches[step + 1][0] if step + 1 < len(batches) else None
        loss, pred = m.train_step(ids_t, dense_t, label_t, lr=c["lr"], next_sparse_inputs=nxt)
        out["loss%d" % step] = loss.cpu().numpy().copy()
    out["rec"] = m.ps.rec.cpu().numpy()
    out["mlp_w0"] = m.dense.p["dnn.linear_0.weight"].cpu().numpy()
    out["status"] = m.status.cpu().numpy()
    out["trace"] = np.asarray(m.comm.trace)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
