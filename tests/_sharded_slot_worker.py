"""One rank of a row-sharded BenchmarkDNNLayer run (spawned by tests/test_sharded_slot_dnn.py).

    python tests/_sharded_slot_worker.py <rank> <world> <port> <cpu|gpu> <outdir>

The reference's own multi-value lines (first lines of slot_dnn/data/demo_10): rank r trains on lines
[r*B, (r+1)*B) of every global batch; uint64 feasigns hashed on the "device"; the PS accessor table sharded row-wise.
cpu: gloo + the oracle-backed operator stand-in (host orchestration only); gpu: all ranks on cuda:0, HIP kernels."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = dict(B=2, N=4099, D=9, S=300, layers=(16, 8), steps=3, lr=1e-6,     # tiny dense lr: see the test
          
           accessor=dict(lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0), initial_range=1e-2, embedx_threshold=0.5,
                         nonclk_coeff=0.1, click_coeff=1.0, seed=11))


def global_batches(world):
    """steps x (values, lod, base, label) for the GLOBAL batch of world*B lines; the 4 fixture lines are cycled."""
    from conftest import GOLDEN
    from paddlerec_amd import reader
    c = CFG
    lines = [ln for ln in open(os.path.join(GOLDEN, "slot_dnn_demo_4.txt"), "rb").read().split(b"\n") if ln.strip()]
    out = []
    for step in range(c["steps"]):
        pick = [lines[(step * world * c["B"] + i) % len(lines)] for i in range(world * c["B"])]
        out.append(pick)
    return out


def parse(lines, S):
    from paddlerec_amd import reader
    chunk = b"\n".join(lines) + b"\n"
    values, lod, base, n = reader.parse_feasign_slots(chunk, 2, S, 0)
    lv, llod, _, _ = reader.parse_feasign_slots(chunk, 1, 1, 0)
    label = lv[llod[0, :-1]].reshape(n, 1).clamp(0, 1)
    return values, lod, base, label


def main():
    rank, world, port, mode, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from paddlerec_amd.sharded import Comm
    from paddlerec_amd.sharded_slot_dnn import ShardedBenchmarkDNNLayer
    if mode == "cpu":
        import cpu_kernels
        dev, kernels = "cpu", cpu_kernels
    else:
        dev, kernels = "cuda:0", None
    c = CFG
    comm = Comm()
    comm.trace = []
    torch.manual_seed(100 + rank)            # different draws per rank: the constructor must broadcast rank 0's MLP
    m = ShardedBenchmarkDNNLayer(c["N"], c["D"], c["S"], list(c["layers"]), device=dev, kernels=kernels, key_mode=1,
                                 accessor=c["accessor"], comm=comm)
    rng = np.random.default_rng(5)           # the same initial MLP on every rank AND in the oracle replay
    sd = {}
    for i, w in enumerate(m.mlp_w):
        sd["linear_%d.weight" % i] = (rng.standard_normal(tuple(w.shape)) * 0.05).astype(np.float32)
        sd["linear_%d.bias" % i] = (rng.standard_normal(tuple(m.mlp_b[i].shape)) * 0.05).astype(np.float32)
    m.set_dict(sd)
    Batch = m.k.MultislotBatch
    out = {}
    for step, lines in enumerate(global_batches(world)):
        mine = lines[rank * c["B"]:(rank + 1) * c["B"]]
        values, lod, base, label = parse(mine, c["S"])
        t = lambda a: a.to(dev)
        loss, pred = m.train_step(Batch(t(values), t(lod), t(base)), t(label), lr=c["lr"])
        out["loss%d" % step] = loss.cpu().numpy().copy()
        out["pred%d" % step] = pred.cpu().numpy().copy()
        if step == 0:
            out["rec_step0"] = m.table.rec.cpu().numpy().copy()
    out["rec"] = m.table.rec.cpu().numpy()
    out["mlp_w0"] = m.mlp_w[0].cpu().numpy()
    from helpers import layer_moments
    out["m_w0"], out["v_w0"] = layer_moments(m)["linear_0.weight"]
    out["status"] = m.status.cpu().numpy()
    out["trace"] = np.asarray(comm.trace)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
