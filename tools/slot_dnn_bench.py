#!/usr/bin/env python3
"""Row P measured: the one-launch multi-slot sum-pool on the slot_dnn benchmark shape (408 slots x D 9, batch 65536,
models/rank/slot_dnn/config_queuedataset.yaml:53-56) and a full BenchmarkDNNLayer train step.  One JSON line.

    python tools/slot_dnn_bench.py [--batch 65536 --slots 408 --dim 9 --rows 10000019 --mean-len 1.13 --opt adam|ps]

Synthetic batch: per (slot, sample) a geometric number of feasigns with the demo file's mean (388-536 feasigns per
sample over 408 slots ~ 1.13 per segment, <= 15), an absent slot = the single padding id 0 (queuedataset_reader.py:75-80:
66 % of the slots of a demo line are absent), uint64 feasigns hashed on the device.
Roofline of the pooling kernel (HBM): algorithmic bytes = ids 8 B + segment offsets + 36-B rows of the live ids +
the [B, S*D] output + counts + the backward's seg / rows side outputs; "designed" charges the 64-B half line a
16-float record row really costs."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402
from paddlerec_amd.slot_dnn import BenchmarkDNNLayer  # noqa: E402

DEV = "cuda"


def make_batch(B, S, absent, extra_p, g):
    """lens: 1 (padding id) with prob `absent`; else 1 + Geometric(extra_p) capped at 15."""
    u = torch.rand(S, B, device=DEV, generator=g)
    geo = torch.floor(torch.log(torch.rand(S, B, device=DEV, generator=g).clamp_min(1e-9)) /
                      torch.log(torch.tensor(extra_p, device=DEV))).to(torch.int64).clamp_(0, 14)
    present = u >= absent
    lens = torch.where(present, 1 + geo, torch.ones_like(geo))
    lod = torch.zeros(S, B + 1, dtype=torch.int64, device=DEV)
    lod[:, 1:] = torch.cumsum(lens, dim=1)
    base = torch.zeros(S + 1, dtype=torch.int64, device=DEV)
    base[1:] = torch.cumsum(lod[:, -1], 0)
    nnz = int(base[-1].item())
    values = torch.randint(1, 2 ** 62, (nnz,), device=DEV, generator=g, dtype=torch.int64) * 2 + 1   # uint64 patterns
    # the single id of an absent (slot, sample) is the padding id 0
    first = (base[:-1, None] + lod[:, :-1]).reshape(-1)
    values[first[(~present).reshape(-1)]] = 0
    return ops.MultislotBatch(values, lod, base), int((values != 0).sum().item())


def timeit(fn, R=10, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(R):
            fn(i)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / R)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--slots", type=int, default=408)
    ap.add_argument("--dim", type=int, default=9)
    ap.add_argument("--rows", type=int, default=10_000_019)
    ap.add_argument("--absent", type=float, default=0.66)
    ap.add_argument("--extra-p", type=float, default=0.6)
    ap.add_argument("--opt", default="adam")
    ap.add_argument("--layers", default="512,256,128,128,128")
    ap.add_argument("--pool-bench", type=int, default=0, help="time only the pooling kernel and print its line")
    ap.add_argument("--pool-only", type=int, default=0, help="only loop the pooling kernel this many times (profiling)")
    ap.add_argument("--push-alone", action="store_true", help="also time the accessor push of a step ALONE (no GEMM beside it)")
    args = ap.parse_args()
    B, S, D, N = args.batch, args.slots, args.dim, args.rows
    g = torch.Generator(device=DEV).manual_seed(5)
    torch.manual_seed(5)
    batches = [make_batch(B, S, args.absent, args.extra_p, g) for _ in range(2)]
    mb, live = batches[0]
    model = BenchmarkDNNLayer(N, D, S, [int(x) for x in args.layers.split(",")], device=DEV, sparse_optimizer=args.opt,
                              key_mode=1, accessor=dict(embedx_threshold=0.0) if args.opt == "ps" else None)
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.25).to(torch.int64)
    W = model.embedding
    out = torch.empty(B, S * D, device=DEV)
    lazy = model.table.lazy_init if model.table is not None else None
    if args.pool_only:
        for i in range(args.pool_only):
            ops.multislot_sumpool(batches[i % 2][0], W, N, 0, 1, model.status, out=out, lazy_init=lazy)
        torch.cuda.synchronize()
        print("pool-only done")
        return
    ms_pool = timeit(lambda i: ops.multislot_sumpool(batches[i % 2][0], W, N, 0, 1, model.status, out=out,
                                                     lazy_init=lazy))
    nnz = mb.nnz
    if args.pool_bench:
        print("pool_fwd_ms %.4f  (ids %d, live %d, rows %d)" % (ms_pool, nnz, live, N))
        return
    alg = nnz * 8 + S * (B + 1) * 8 + live * D * 4 + B * S * D * 4 + B * S * 4 + nnz * 4 + nnz * 8
    designed = alg - live * D * 4 + live * 64
    for i in range(2):
        model.train_step(batches[i % 2][0], label, 1e-3)
    model.timers = {}
    ms_step = timeit(lambda i: model.train_step(batches[i % 2][0], label, 1e-3), R=5)
    ev = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in model.timers.items() if not k.endswith("@host")}
    if args.push_alone and model.table is not None:      # the step's own groups / gradient buffers, the push by itself
        dx = torch.randn(B, S * D, device=DEV) * 1e-3
        model.table.accessor.grad_scale = float(B)
        ms_push = timeit(lambda i: ops.ps_push_rows(model.table, model._groups, dx, S, show=None, click=label.reshape(-1)), R=5)
        u = int(model._groups.n_uniq[0])
        by = u * 256 + live * 64 + live * 4 + u * 12
        print("ps_push_rows alone: %.3f ms  (%d features, %d live ids; records r/w + one 64-B gradient sector per id + index "
              "arrays = %.2f GB -> %.2f TB/s)" % (ms_push, u, live, by / 1e9, by / ms_push / 1e9), file=sys.stderr)
    flops = 3 * sum(2 * B * a * b for a, b in zip([S * D] + model.layer_sizes, model.layer_sizes + [1]))
    print(json.dumps({
        "workload": "slot_dnn BenchmarkDNNLayer: %d slots x D %d, batch %d, %d ids (%d live), hashed table %d rows, %s"
                    % (S, D, B, nnz, live, N, args.opt),
        "pool_fwd_ms": ms_pool, "pool_algorithmic_bytes": alg, "pool_designed_bytes": designed,
        "roofline": {"bound": "hbm", "achieved": alg / ms_pool / 1e6, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / ms_pool / 1e6 / 8000.0, "GBs_designed": designed / ms_pool / 1e6,
                     "kernel": "multislot_sumpool_kernel"},
        "train_step_ms": ms_step, "samples_per_s": B / ms_step * 1e3, "kernels_ms": ev,
        "mlp_tflops_in_step": flops / ((ev.get("mlp_fwd", 0) + ev.get("mlp_bwd", 0) + ev.get("mlp_bwd_dw0", 0)) * 1e-3) / 1e12
        if ev.get("mlp_fwd") else None,
        "index_oob_flag": int(model.status.item())}))


if __name__ == "__main__":
    main()
