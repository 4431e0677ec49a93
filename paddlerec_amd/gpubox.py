"""The gpubox pass loop on the engine — the caller of the hashed PS table (SURVEY.md §8(e) "reference counterpart",
BASELINE configs[4]).

Host mirror of /root/reference/tools/static_gpubox_trainer.py (`Main`, :85-260): network(), init_reader(), run_worker(),
dataset_train_loop(epoch) and the `core.PSGPU` object it drives (:153-159 set_slot_vector / set_slot_dim_vector /
init_gpu_ps, :244 begin_pass, :207 end_pass, :219 finalize).  What those calls mean here:

    reader.load_into_memory()   the pass's files parsed on the host (rec_parse_feasign_slots: multi-value `feasign:slot`
                                lines -> slot-major CSR per batch), kept in host memory            :239
    PSGPU.begin_pass()          the pass's batches are moved into HBM (uint64 feasigns stay keys: the table is hashed
                                on the device, rows are born at their first pull — there is no CPU parameter server to
                                build the pass's table from); returns what it staged                  :244
    exe.train_from_dataset()    one train_step per batch over the resident pass                       :256
    PSGPU.end_pass()            the accessor's shrink: show / click decay, rows whose score fell below
                                delete_threshold are deleted (config_online.yaml:80-89)               :207
    PSGPU.finalize()            drops the pass buffers                                                :219
The model is slot_dnn's BenchmarkDNNLayer with the PS accessor table (paddlerec_amd/slot_dnn.py); AUC from the device
buckets at the end of a pass, `ips` as the reference logs it (:181-203).  Paddle's static-graph executor, the CPU
parameter servers behind HeterPS and `fleet.save_inference_model` are not mirrored.
"""
import logging
import os
import time

import torch

from . import reader as rd
from .deepfm import auc_from_buckets
from .slot_dnn import StaticModel

logger = logging.getLogger("paddlerec_amd.gpubox")


class PSGPU:
    """core.PSGPU's pass surface over an ops.PsTable."""

    def __init__(self, kernels):
        self.k = kernels
        self.slots, self.slot_dims, self.gpus = None, None, None
        self.table, self.shrink = None, None
        self.host_pass, self.device_pass = None, None
        self.passes = 0

    def set_slot_vector(self, slots):
        self.slots = [int(s) for s in slots]

    def set_slot_dim_vector(self, dims):
        self.slot_dims = [int(d) for d in dims]

    def init_gpu_ps(self, gpus):
        if self.slots is None or self.slot_dims is None or len(self.slots) != len(self.slot_dims):
            raise ValueError("set_slot_vector / set_slot_dim_vector must be called first, with equal lengths")
        self.gpus = [int(g) for g in gpus]

    def bind(self, table, decay=0.98, delete_threshold=0.8, delete_after_unseen_days=float("inf")):
        """Not a PSGPU method: in the reference the table lives in the parameter servers PSGPU talks to; here the
        layer owns it.  decay / delete_threshold / delete_after_unseen_days: ctr_accessor_param
        (config_online.yaml:85-88)."""
        self.table, self.shrink = table, (float(decay), float(delete_threshold), float(delete_after_unseen_days))

    def load_pass(self, batches):
        """The dataset's in-memory pass (what reader.load_into_memory() holds): host batches."""
        self.host_pass = list(batches)

    def begin_pass(self):
        if self.gpus is None or self.table is None:
            raise RuntimeError("init_gpu_ps() and bind() come before begin_pass()")
        if self.host_pass is None:
            raise RuntimeError("no pass loaded (reader.load_into_memory)")
        dev = self.table.rec.device
        self.device_pass = [tuple(t.to(dev, non_blocking=True) if torch.is_tensor(t) else t for t in b)
                            for b in self.host_pass]
        self.passes += 1
        return dict(batches=len(self.device_pass),
                    feasigns=int(sum(b[0].numel() for b in self.host_pass)))

    def end_pass(self):
        """-> number of rows the shrink deleted."""
        if self.device_pass is None:
            raise RuntimeError("end_pass() without begin_pass()")
        self.device_pass, self.host_pass = None, None
        return self.k.ps_shrink_rows(self.table, *self.shrink)

    def finalize(self):
        self.device_pass = self.host_pass = None


class InMemoryReader:
    """The InMemoryDataset of the gpubox runs over slot_dnn/queuedataset_reader.py's line format: load_into_memory()
    parses the pass's files into host batches (values, lod [S,B+1], slot_base, label [B,1]); drop_last like the
    reference's batching."""

    def __init__(self, file_list, batch_size, slot_num, threads=0):
        self.file_list, self.batch_size, self.slot_num, self.threads = list(file_list), batch_size, slot_num, threads
        self.batches = []

    def load_into_memory(self):
        """The pass's files -> host batches through reader.feasign_batches (every file parsed once, whole, by the C
        parser; batches cut by rec_csr_cut: lines are never split or re-joined in python — a 65536-line batch of this
        format is ~650 MB of text).  Slot "1" = click (label), slots "2".."slot_num+1" = features
        (queuedataset_reader.py:45-56): parsed in one pass, the label slot peeled off here."""
        S, out = self.slot_num, []
        for vals, lod_b, base_b in rd.feasign_batches(self.file_list, self.batch_size, 1, S + 1, 0, self.threads):
            nlab = int(base_b[1])
            label = vals[:nlab][lod_b[0, :-1]].reshape(-1, 1).clamp_(0, 1).contiguous()      # first value of slot "1"
            out.append((vals[nlab:].contiguous(), lod_b[1:].contiguous(), (base_b[1:] - base_b[1]).contiguous(), label))
        self.batches = out
        return len(out)

    def release_memory(self):
        self.batches = []


def _data_files(config, key):
    d = config.get(key)
    if d is None:
        raise ValueError("%s is not set" % key)
    if not os.path.isabs(d):
        d = os.path.join(config.get("config_abs_dir", "."), d)
    if not os.path.isdir(d):
        raise ValueError("%s = %r is not a directory" % (key, d))
    files = sorted(os.path.join(d, x) for x in os.listdir(d) if not x.startswith("."))
    if not files:
        raise ValueError("%s = %r holds no files" % (key, d))
    return files


class Main:
    """tools/static_gpubox_trainer.py:85-260 (sync_mode gpubox)."""

    def __init__(self, config, device="cuda", kernels=None):
        from . import ops
        self.config, self.device = config, torch.device(device)
        self.k = kernels if kernels is not None else ops
        self.train_result_dict = {"speed": [], "auc": [], "loss": [], "deleted": []}
        self.PSGPU = None

    def network(self):
        self.model = StaticModel(self.config)
        self.net = self.model.create_model(self.device, kernels=self.k, sparse_optimizer="ps")
        self.metrics_list, self.metric_names = self.model.create_metrics(self.device)

    def init_reader(self):
        files = _data_files(self.config, "runner.train_data_dir")
        self.reader = InMemoryReader(files, int(self.config.get("runner.train_batch_size")), self.model.slot_num)
        self.file_list = files

    def run_worker(self):
        cfg = self.config
        if not hasattr(self, "net"):
            self.network()
        self.init_reader()
        epochs = int(cfg.get("runner.epochs", 1))
        use_auc = bool(cfg.get("runner.use_auc", True))
        ctr = (cfg.get("table_parameters.embedding.accessor", None) or {}).get("ctr_accessor_param", {})
        self.PSGPU = PSGPU(self.k)
        self.PSGPU.set_slot_vector(range(1, self.model.slot_num + 1))                        # :155-157
        self.PSGPU.set_slot_dim_vector([self.model.emb_dim - 1] * self.model.slot_num)       # :156-158 (embedx dim)
        gpus = os.environ.get("FLAGS_selected_gpus", "0")
        self.PSGPU.init_gpu_ps([int(s) for s in gpus.split(",")])                            # :159
        self.PSGPU.bind(self.net.table, ctr.get("show_click_decay_rate", 0.98), ctr.get("delete_threshold", 0.8),
                        ctr.get("delete_after_unseen_days", float("inf")))
        self.ctr = ctr
        save = cfg.get("runner.model_save_path")
        for epoch in range(epochs):
            t0 = time.time()
            n, loss = self.dataset_train_loop(epoch)
            dt = max(time.time() - t0, 1e-9)
            speed = n / dt / len(self.PSGPU.gpus)
            msg = "Epoch: %d, using time: %.3f second, ips: %.1f example/sec." % (epoch, dt, speed)
            if use_auc:
                pos, neg = self.metrics_list[0]
                a = auc_from_buckets(pos, neg)
                self.train_result_dict["auc"].append(a)
                pos.zero_()
                neg.zero_()
                msg += " auc: %.6f" % a
            logger.info(msg)
            self.train_result_dict["speed"].append(speed)
            self.train_result_dict["loss"].append(loss)
            self.train_result_dict["deleted"].append(self.PSGPU.end_pass())                  # :207
            if save:
                self.save_pass(os.path.join(save, str(epoch)))
            self.reader.release_memory()
            status = int(self.net.status.item())
            if status:
                raise RuntimeError("a lookup flagged an out-of-range row (status %d)" % status)
        self.PSGPU.finalize()                                                                # :219
        return self.train_result_dict

    def save_pass(self, model_dir, mode=0):
        """The pass checkpoint: the dense parameters and the EXISTING values of the table only (index + record) — a
        160-GB shard with a few million live features is a few hundred MB on disk.  mode = the `param` of the
        accessor's Save / UpdateStatAfterSave [EXT ctr_accessor.cc]: 0 every value (default); 1 the delta save (values
        with score >= base_threshold, delta_score >= delta_threshold, unseen_days <= delta_keep_days; their
        delta_score restarts at 0); 2 the base save (the same without the delta_score bar); 3 every value, and a day
        passes (unseen_days += 1: what delete_after_unseen_days counts)."""
        import numpy as np
        os.makedirs(model_dir, exist_ok=True)
        t = self.net.table
        ctr = getattr(self, "ctr", {}) or {}
        sel = self.k.ps_save_select(t, int(mode), ctr.get("base_threshold", 1.5), ctr.get("delta_threshold", 0.25),
                                    ctr.get("delta_keep_days", 16.0))
        born = torch.nonzero(sel).reshape(-1)
        out = {"rows": born.cpu().numpy(), "records": t.rec[born].cpu().numpy(),
               "num_rows": np.int64(t.num_rows), "emb_dim": np.int64(t.emb_dim)}
        for k, v in self.net.state_dict().items():
            if k != "embedding":
                out["dense." + k] = v.detach().cpu().numpy()
        np.savez(os.path.join(model_dir, "rec_gpubox.npz"), **out)
        return model_dir

    def load_pass(self, model_dir):
        import numpy as np
        z = np.load(os.path.join(model_dir, "rec_gpubox.npz"))
        t = self.net.table
        if int(z["num_rows"]) != t.num_rows or int(z["emb_dim"]) != t.emb_dim:
            raise ValueError("checkpoint is for a %d x %d table" % (int(z["num_rows"]), int(z["emb_dim"])))
        t.rec.zero_()
        t.rec[torch.as_tensor(z["rows"]).to(t.rec.device)] = torch.as_tensor(z["records"]).to(t.rec.device)
        self.net.set_dict({k[6:]: z[k] for k in z.files if k.startswith("dense.")})

    def dataset_train_loop(self, epoch):
        """:236-259.  -> (examples trained, mean loss of the pass)"""
        t0 = time.time()
        self.reader.load_into_memory()
        self.PSGPU.load_pass(self.reader.batches)
        logger.info("self.reader.load_into_memory cost :%.3f seconds", time.time() - t0)
        t0 = time.time()
        info = self.PSGPU.begin_pass()
        logger.info("begin_pass cost:%.3f seconds (%d batches, %d feasigns staged)", time.time() - t0,
                    info["batches"], info["feasigns"])
        Batch = self.k.MultislotBatch
        n, losses = 0, []
        for values, lod, base, label in self.PSGPU.device_pass:                              # exe.train_from_dataset
            loss, _, _ = self.model.train_forward(self.net, self.metrics_list, Batch(values, lod, base), label)
            losses.append(loss)
            n += label.shape[0]
        mean = float(torch.stack([l.reshape(()) for l in losses]).mean()) if losses else float("nan")
        return n, mean


def load_config(path, overrides=()):
    """The reference's gpubox / online YAML (runner + hyper_parameters flattened as in paddlerec_amd.trainer, plus the
    nested table_parameters.embedding.accessor block, slot_dnn/config_online.yaml:57-89)."""
    import yaml
    from .trainer import load_yaml
    cfg = load_yaml(path, overrides)
    with open(path, "r") as f:
        doc = yaml.safe_load(f) or {}
    acc = ((doc.get("table_parameters") or {}).get("embedding") or {}).get("accessor")
    if acc:
        cfg["table_parameters.embedding.accessor"] = acc
    cfg.setdefault("runner.epochs", 1)
    return cfg


def main(argv=None):
    """python -m paddlerec_amd.gpubox -m <PaddleRec>/models/rank/slot_dnn/config_online.yaml [-o key=value ...]"""
    import argparse
    ap = argparse.ArgumentParser(description="the gpubox pass loop (tools/static_gpubox_trainer.py) on the recengine")
    ap.add_argument("-m", "--config_yaml", required=True)
    ap.add_argument("-o", "--opt", nargs="*", default=[], help="key=value overrides, e.g. runner.epochs=2")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(message)s", level=logging.INFO)
    res = Main(load_config(args.config_yaml, args.opt), args.device).run_worker()
    print(res)
    return res


if __name__ == "__main__":
    main()
