#!/usr/bin/env python3
"""L-trainer level of SURVEY.md §8(d): end-to-end `ips` of the train loop (paddlerec_amd/trainer.py) INCLUDING the
input pipeline — synthetic Criteo slot-text files on disk -> host parser -> device batches -> DeepFM train steps —
at the BASELINE batch size, next to the device-resident step rate of bench.py.

    python tools/trainer_bench.py [--lines 2000000] [--batch 65536] [--dim 16] [--epochs 2] [--device cuda]
The CPU smoke of the same code path: --device cpu --backend oracle --lines 2000 --batch 256 --rows 5000
(the oracle-backed operator stand-in of tests/cpu_kernels.py; only to check the plumbing, never a measurement)."""
import argparse
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402


def write_slot_text(path, n_lines, rows_per_slot, seed, block=16384):
    """Lines in the format of models/rank/deepfm/data/sample_data/train/sample_train.txt.  One block of distinct
    lines is formatted (Python string formatting is ~20 k lines/s) and repeated to the requested length: the parser
    and the trainer do the same work per line either way."""
    rng = np.random.default_rng(seed)
    n = min(block, n_lines)
    label = (rng.random(n) < 0.25).astype(np.int64)
    dense = rng.random((n, 13)).round(6)
    ids = rng.integers(1, rows_per_slot, (n, 26))
    text = "\n".join("click:%d %s %s" % (label[i], " ".join("dense_feature:%s" % repr(float(v)) for v in dense[i]),
                                          " ".join("%d:%d" % (s + 1, ids[i, s]) for s in range(26)))
                     for i in range(n)) + "\n"
    with open(path, "w") as f:
        for _ in range(n_lines // n):
            f.write(text)
        f.write("".join(text.splitlines(keepends=True)[: n_lines % n]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=2_000_000)
    ap.add_argument("--files", type=int, default=4)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--rows", type=int, default=1_000_001, help="rows of the (shared) table, as the reference config")
    ap.add_argument("--fc", default="400,400,400")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--backend", choices=("hip", "oracle"), default="hip")
    ap.add_argument("--tables", action="store_true",
                    help="26 tables of --rows rows (BASELINE configs[1], bench.py's layout) instead of one shared table")
    ap.add_argument("--no-checkpoint", action="store_true", help="do not write the per-epoch checkpoint")
    args = ap.parse_args()
    from paddlerec_amd import trainer
    kernels = None
    if args.backend == "oracle":
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import cpu_kernels as kernels
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "data"))
        t0 = time.time()
        per = args.lines // args.files
        for i in range(args.files):
            write_slot_text(os.path.join(d, "data", "part-%02d" % i), per, args.rows, 100 + i)
        gen_s = time.time() - t0
        size = sum(os.path.getsize(os.path.join(d, "data", x)) for x in os.listdir(os.path.join(d, "data")))
        cfg = {"config_abs_dir": d, "runner.train_data_dir": "data", "runner.train_batch_size": args.batch,
               "runner.epochs": args.epochs, "runner.print_interval": 1 << 30, "runner.use_auc": True,
               "runner.model_save_path": os.path.join(d, "out"),
               "hyper_parameters.sparse_feature_number": args.rows, "hyper_parameters.sparse_feature_dim": args.dim,
               "hyper_parameters.dense_input_dim": 13, "hyper_parameters.sparse_inputs_slots": 27,
               "hyper_parameters.fc_sizes": [int(x) for x in args.fc.split(",")],
               "hyper_parameters.optimizer.learning_rate": 0.001, "hyper_parameters.optimizer.lazy_mode": True,
               "runner.save_checkpoint": not args.no_checkpoint}
        summaries, _ = trainer.train(cfg, "deepfm", args.device, kernels)
    last = summaries[-1]            # the first epoch carries allocation / first-touch costs
    stages = {k: [round(s[k], 4) for s in summaries] for k in ("epoch_s", "reader_wait_s", "step_issue_s", "final_sync_s",
                                                                "checkpoint_s")}
    print(json.dumps({"metric": "trainer ips (text files -> host parser -> device -> DeepFM train step)",
                      "value": last["ips"], "unit": "samples/s", "epochs": [s["ips"] for s in summaries],
                      "batches_per_epoch": last["batches"], "stages_s_per_epoch": stages,
                      "reader_trace_last_epoch": last.get("reader_trace"),
                      "lines": per * args.files, "text_bytes": size, "batch": args.batch, "dim": args.dim,
                      "backend": args.backend, "loss": last["loss"], "auc": last["auc"],
                      "generate_text_s": round(gen_s, 2),
                      "note": "ips = samples / epoch_s (first batch requested .. device idle); the per-epoch checkpoint is timed "
                              "separately (checkpoint_s); lazy Adam (bench.py's optimizer)"}))


if __name__ == "__main__":
    main()
