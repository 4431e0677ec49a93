"""One rank of the collective-mode trainer test (gloo, CPU, oracle-backed operator backend).
usage: _trainer_dist_worker.py <rank> <world> <port> <workdir>"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import cpu_kernels  # noqa: E402


def main():
    rank, world, port, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from paddlerec_amd import trainer
    from paddlerec_amd.sharded import Comm
    comm = Comm()
    comm.trace = []
    cfg = trainer.load_yaml(os.path.join(workdir, "config.yaml"))
    out = {}
    # the initial parameters: the same seed path with zero epochs
    cfg0 = dict(cfg)
    cfg0["runner.epochs"] = 0
    _, m0 = trainer.train(cfg0, "deepfm", "cpu", cpu_kernels, comm)
    W, W1 = m0.gather_global_tables()
    out["init.W"], out["init.W1"] = W.numpy().copy(), W1.numpy().copy()
    for k, v in m0.dense.p.items():
        out["init." + k] = v.detach().numpy().copy()
    del m0
    summaries, model = trainer.train(cfg, "deepfm", "cpu", cpu_kernels, comm)
    out["loss"] = np.asarray([s["loss"] for s in summaries])
    out["auc"] = np.asarray([s["auc"] for s in summaries])
    out["samples"] = np.asarray([s["samples"] for s in summaries])
    out["batches"] = np.asarray([s["batches"] for s in summaries])
    W, W1 = model.gather_global_tables()
    out["W"], out["W1"] = W.numpy().copy(), W1.numpy().copy()
    out["mlp_w0"] = model.dense.p["dnn.linear_0.weight"].detach().numpy().copy()
    out["shard_files"] = np.asarray([os.path.exists(os.path.join(s["model_dir"], "rec.shard%dof%d.pdparams" % (rank, world)))
                                     for s in summaries])
    res = trainer.infer(cfg, "deepfm", "cpu", cpu_kernels, comm)
    out["infer_auc"] = np.asarray([r["auc"] for r in res])
    out["infer_samples"] = np.asarray([r["samples"] for r in res])
    out["trace"] = np.asarray(comm.trace)
    np.savez(os.path.join(workdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
