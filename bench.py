#!/usr/bin/env python3
"""Headline benchmark: Criteo DeepFM training step, bs=65536 per GPU, on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 26 sparse slots x 1M rows x dim 16 (one 26M-row table with
slot offsets), 13 dense fields, top MLP 400-400-400 (config_bigdata.yaml:47), fp32, batch 65536.
One "step" = train_forward + backward + optimizer of the reference graph
(deepfm/dygraph_model.py:76-88, tools/trainer.py:148-152) on a device-resident synthetic batch:
fused embedding+FM forward, MLP GEMMs, loss head, MLP backward, FM backward, SelectedRows merge,
lazy sparse Adam on both tables, dense Adam.  Lazy Adam is the reference's static-graph optimizer
(deepfm/static_model.py:83-84) and the only one that scales to the 10B-row table of configs[4].

N>1 (BASELINE.json configs[4]): weak scaling on the hashed gpubox table — every rank keeps batch 65536 and owns
1.25e9 rows (one 128-B record each, 160 GB) of a table that grows with N: 10^10 rows at 8 GPUs.  uint64 feasigns are
hashed to rows on the device, rows are sharded row-wise (row r on rank r % N) and born at their first pull, ids / rows /
gradients are exchanged by RCCL all-to-all over xGMI, the PS accessor (AdaGrad rule, show / click) updates the touched
rows, dense gradients are all-reduced.  `--table adam` keeps the lazy-Adam record table of N = 1 (26M rows per GPU).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# cpu_baseline mixes an OpenMP C oracle with BLAS threads: spinning idle workers would starve each other
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_MFMA_PEAK_TF = 157.3   # dense f32 MFMA peak
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md); the bf16 x 3 GEMMs spend six bf16 MFMAs per f32 product
X3_ON = os.environ.get("REC_GEMM_BF16X3", "1") != "0"      # csrc/gemm_bf16x3.h (the library's default: on)


def algorithmic_bytes(B, S, Dn, D):
    """DeepFM embedding + FM forward/backward bytes per step."""
    F = S + Dn
    fwd = B * S * 8 + B * S * D * 4 + B * S * 4 + B * Dn * 4 + B * F * D * 4 + B * 8
    bwd = B * S * 8 + B * F * D * 4 + B * 4 + B * S * D * 4 + B * S * D * 4 + B * S * 4
    return fwd, bwd


def designed_bytes(B, S, Dn, D):
    """Bytes the two kernels are DESIGNED to move per launch: one 128-B record line per lookup
    (64 B of it useful), compact feat [B,S+1,D] instead of [B,S+Dn,D] — what the PMC counters should show if
    nothing is re-read.  fwd: ids + lines + dense + feat' + sum_emb + y1,y2;  bwd: feat' + d_feat' + sum_emb + dy +
    dense (reads), row_grad (write)."""
    F = S + 1
    fwd = B * S * 8 + B * S * 128 + B * Dn * 4 + B * F * D * 4 + B * D * 4 + B * 8
    bwd = 2 * B * F * D * 4 + B * D * 4 + B * 8 + B * Dn * 4 + B * S * D * 4
    return fwd, bwd


def pmc_traffic(B, D):
    """HBM bytes per launch of (fm_fwd, fm_bwd) from the COMMITTED rocprofv3 --pmc passes of this same command
    (tools/profile_bench.sh -> profiles/*_pmc_traffic.json: TCC_EA0_RDREQ_{32,64,128}B / WRREQ{,_64B} request
    counters x their sizes, i.e. already in bytes — not the FETCH_SIZE KB figure that needs the x2 gfx950
    correction).  NOT measured in this run (PMC needs rocprofv3 around the process): a cross-reference, returned
    with the file it came from.  Only valid for the profiled shape; None otherwise."""
    if B != 65536 or D != 16:
        return None, None, None
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None, None
    try:
        t = json.load(open(files[-1]))
        return (float(t["fm_fwd_kernel"]["hbm_bytes"]), float(t["fm_bwd_kernel"]["hbm_bytes"]),
                os.path.relpath(files[-1], REPO))
    except Exception:
        return None, None, None


def mlp_flops(B, sizes):
    return sum(2 * B * sizes[i] * sizes[i + 1] for i in range(len(sizes) - 1))


def make_batches(n, B, S, Dn, rows_per_table, device, seed, dist="uniform"):
    """Device-resident synthetic Criteo-shaped batches, 3 % padding (id 0).  ids: (U) uniform in [1,rows) or
    (Z) Zipf(1.05) ranks through a fixed random permutation, clipped to [1,rows-1]: hot rows."""
    g = torch.Generator(device=device).manual_seed(seed)
    rng = np.random.default_rng(seed)
    perm = torch.as_tensor(rng.permutation(rows_per_table)).to(device) if dist == "zipf" else None
    out = []
    for _ in range(n):
        if dist == "zipf":
            rank = np.minimum(rng.zipf(1.05, size=(B, S)), rows_per_table - 1)
            ids = perm[torch.as_tensor(rank).to(device)].clamp_(1, rows_per_table - 1)
        else:
            ids = torch.randint(1, rows_per_table, (B, S), device=device, generator=g, dtype=torch.int64)
        ids[torch.rand(B, S, device=device, generator=g) < 0.03] = 0
        dense = torch.rand(B, Dn, device=device, generator=g)
        label = (torch.rand(B, 1, device=device, generator=g) < 0.25).to(torch.int64)
        out.append((ids, dense, label))
    return out


def gemm_accuracy(dev):
    """The bf16 x 3 GEMM and the exact-f32 MFMA GEMM on ONE MLP-layer problem of the bench's shape (8192 rows of it),
    both against a float64 product of the same f32 inputs — measured in this run so that the line carries the evidence for
    its `dtype`: max |C - C64| / sum_k |a||b| per output (the scale f32 round-off lives on).  Not part of the timed region."""
    from paddlerec_amd import ops
    g = torch.Generator(device=dev).manual_seed(7)
    M, K, N = 8192, 400, 400
    A = torch.rand(M, K, device=dev, generator=g) - 0.5
    W = (torch.rand(K, N, device=dev, generator=g) - 0.5) * 0.1
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    ref = torch.relu(A.double() @ W.double() + bias.double())
    mag = A.double().abs() @ W.double().abs() + bias.double().abs()
    out, keep = {}, os.environ.get("REC_GEMM_BF16X3")
    try:
        for name, v in (("bf16x3", "1"), ("exact_f32", "0")):
            os.environ["REC_GEMM_BF16X3"] = v
            C_ = ops.gemm(A, W, ops.Workspace(dev), epilogue="bias_relu", bias=bias)
            out[name] = float(((C_.double() - ref).abs() / mag).max())
    finally:
        if keep is None:
            os.environ.pop("REC_GEMM_BF16X3", None)
        else:
            os.environ["REC_GEMM_BF16X3"] = keep
    out["problem"] = "relu(A[8192,400] @ W[400,400] + b), uniform operands; max |C - C_float64| / (sum_k |a||w| + |b|)"
    return out


def cpu_baseline(B, S, Dn, D, fc, rows_per_table, budget_s=20.0):
    """Times the ORACLE (C restatement for embedding+FM+sparse Adam, NumPy GEMMs for the MLP) on this
    host's cores, on a bounded sample of the same workload: whole steps of batch B until ~budget_s."""
    import ctypes as C
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import c_adam_rows, c_fm_bwd, c_fm_fwd
    from oracle import deepfm_ref as R
    lib = C.CDLL(os.path.join(REPO, "oracle", "_build", "liboracle.so"))
    lib.oracle_num_threads.restype = C.c_int
    cores = lib.oracle_num_threads()
    rng = np.random.default_rng(20250404)
    N = rows_per_table * S
    std = 0.1 / np.sqrt(D)
    W = (rng.standard_normal((N, D), dtype=np.float32) * std)
    W1 = (rng.standard_normal((N, 1), dtype=np.float32) * std)
    M, V = np.zeros_like(W), np.zeros_like(W)
    M1, V1 = np.zeros_like(W1), np.zeros_like(W1)
    dw = (rng.standard_normal((1, Dn, D), dtype=np.float32) * std)
    dw1 = (rng.standard_normal(Dn, dtype=np.float32) * std)
    sizes = [(S + Dn) * D] + list(fc) + [1]
    mw = [(rng.standard_normal((sizes[i], sizes[i + 1]), dtype=np.float32) / np.sqrt(sizes[i]))
          for i in range(len(sizes) - 1)]
    mb = [np.zeros(sizes[i + 1], np.float32) for i in range(len(sizes) - 1)]
    so = (np.arange(S, dtype=np.int64) * rows_per_table)
    ids = rng.integers(1, rows_per_table, (B, S), dtype=np.int64)
    ids[rng.random((B, S)) < 0.03] = 0
    dense = rng.random((B, Dn), dtype=np.float32)
    label = (rng.random((B, 1)) < 0.25).astype(np.int64)
    steps, t_total, dstate = 0, 0.0, {}
    while True:
        t0 = time.perf_counter()
        y1, y2, feat, sum_emb = c_fm_fwd(lib, ids, dense, W, W1, dw, dw1, 0, so)
        y_dnn, acts = R.dnn_forward(feat, mw, mb, return_acts=True)
        pred = R.sigmoid(y1[:, None] + y2[:, None] + y_dnn)
        dz = R.log_loss_mean_grad_z(pred, label).astype(np.float32)
        dflat, dws, dbs = R.dnn_backward(dz, acts, mw)
        rg, rg1, ddw, ddw1 = c_fm_bwd(lib, S, dense, feat, sum_emb, dflat.reshape(feat.shape), dz, dz)
        rows, valid = R.effective_rows(ids, 0, so)
        spos, uniq, offs = R.group_ids(rows.reshape(-1), valid.reshape(-1))
        c_adam_rows(lib, uniq, offs, spos, rg, W, M, V, steps + 1)
        c_adam_rows(lib, uniq, offs, spos, rg1, W1, M1, V1, steps + 1)
        for arr, gr in [(dw, ddw), (dw1, ddw1)] + list(zip(mw, dws)) + list(zip(mb, dbs)):
            mm, vv = dstate.setdefault(id(arr), (np.zeros_like(arr), np.zeros_like(arr)))
            R.adam_update(arr, mm, vv, np.asarray(gr, np.float32).reshape(arr.shape), steps + 1)
        t_total += time.perf_counter() - t0
        steps += 1
        if t_total > budget_s or steps >= 8:
            break
    return {"value": B * steps / t_total, "unit": "samples/s", "cores": int(cores), "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": "%d full steps of batch %d (oracle: C embedding+FM fwd/bwd + lazy Adam on both tables, "
                      "NumPy/BLAS MLP fwd/bwd + dense Adam), %.1f s" % (steps, B, t_total)}


XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0          # 7 point-to-point links per GPU, ~153 GB/s each way


def link_model_us(tag, remote_bytes, world, calls=1.0):
    """Analytic time of one collective's REMOTE bytes over xGMI (microseconds per step), next to which the measured time
    is printed so that the first real multi-GPU run diagnoses itself:
      all-to-all : every peer is reached over its own link, min(world-1, 7) links busy at once
      all-reduce : ring — per hop one link each way; remote_bytes already holds the 2 (G-1)/G factor
    + ~8 us of launch / rendezvous per collective call (RCCL kernel launch + peer handshake, order of magnitude)."""
    if world <= 1 or remote_bytes <= 0:
        return 8.0 * calls
    links = 1 if tag.startswith("allreduce") else min(world - 1, XGMI_LINKS)
    return remote_bytes / (links * XGMI_LINK_GBS * 1e9) * 1e6 + 8.0 * calls


def _distinct_share(n, rows, zipf_alpha=None, seed=20250404):
    """Expected fraction of a rank's n lookups that are DISTINCT rows (what the deduplicated exchange moves), for uniform
    ids over `rows` rows (closed form) or Zipf(alpha) ids (one sampled batch; the ids of SURVEY section 8(d): alpha 1.05)."""
    import numpy as np
    if zipf_alpha is None:
        return float(rows * (1.0 - np.exp(-n / rows)) / n) if rows < 1e12 else 1.0
    r = np.random.default_rng(seed).zipf(zipf_alpha, size=min(n, 1 << 21))
    return float(len(np.unique(np.minimum(r, int(rows) - 1))) / len(r))


def dry_links(B, S, D, table, worlds=(2, 4, 8), rows=None, dedup_cap=1.25):
    """bench.py --dry-links: the per-GPU, per-step exchange volumes of the row-sharded step (SURVEY §8(d) "all-to-all
    bytes") and the link model's time for each collective, WITHOUT touching a GPU — what `exchange.per_collective`
    should look like on real links.  Uniform ids: a fraction (G-1)/G of a rank's lookups is owned by a peer.
    Both exchanges are priced: the plain one (every lookup crosses, exact sizes one step ahead through pinned memory) and
    the DEDUPLICATED one (REC_SHARD_DEDUP=1: distinct rows only, fixed capacity dedup_cap x n / G slots per owner — the
    slots travel whether filled or not, so uniform ids pay the slack and Zipf ids save what they repeat)."""
    out = []
    c1 = 2 if table == "ps" else 1                  # PS: the click label rides with dz
    c1d = 3 if table == "ps" else 1                 # deduplicated PS: dz | occurrences | clicks of the distinct row
    dense_params = 13 * (D + 1) + (S + 1) * D * 400 + 400 + 2 * (400 * 400 + 400) + 400 + 1 + 1
    rows = float(rows or 26e6)
    for G in worlds:
        n, f = B * S, (G - 1) / G
        per = {"a2a_counts": (2 * G * 8, 2 * G * 8 * f, 2), "a2a_ids": (n * 8, n * 8 * f, 1),
               "a2a_rows": (n * (D + 1) * 4, n * (D + 1) * 4 * f, 2),
               "a2a_grads": (n * (D + c1) * 4, n * (D + c1) * 4 * f, 2),
               "allreduce_dense": (dense_params * 4, 2 * dense_params * 4 * f, 1)}
        slots = G * (int(dedup_cap * n / G) + 64)              # ShardedDeepFMLayer.dedup_capacity
        ded = {"a2a_ids": (slots * 8, slots * 8 * f, 1), "a2a_rows": (slots * (D + 1) * 4, slots * (D + 1) * 4 * f, 2),
               "a2a_grads": (slots * (D + c1d) * 4, slots * (D + c1d) * 4 * f, 2),
               "allreduce_dense": per["allreduce_dense"]}
        fmt = lambda d_: {k: {"bytes_per_step": b, "remote_bytes_per_step": r, "calls_per_step": c,
                              "predicted_us": round(link_model_us(k, r, G, c), 1)} for k, (b, r, c) in d_.items()}
        tot = lambda d_: round(sum(link_model_us(k, r, G, c) for k, (b, r, c) in d_.items()), 1)
        out.append({"world": G, "per_collective": fmt(per), "predicted_exchange_us_total": tot(per),
                    "dedup": {"slots_per_step": slots, "per_collective": fmt(ded), "predicted_exchange_us_total": tot(ded),
                              "distinct_share_uniform": round(_distinct_share(n, rows), 4),
                              "distinct_share_zipf_1p05": round(_distinct_share(n, rows, 1.05), 4),
                              "note": "slots = G x (%.2f n / G + 64): the fixed capacity travels; with exact sizes the "
                                      "payload would be distinct_share x the plain exchange's bytes" % dedup_cap}})
    return {"workload": "row-sharded DeepFM step, batch %d per GPU, %d slots, dim %d, table %s, %.3g rows" % (B, S, D, table, rows),
            "model": "remote bytes / (links x %.0f GB/s) + 8 us per call; all-to-all uses min(G-1, %d) links, ring "
                     "all-reduce one" % (XGMI_LINK_GBS, XGMI_LINKS), "worlds": out}


def empty_bracket_us(dev, n=50):
    """Median elapsed time between two torch timing events recorded back to back on the current stream (microseconds)."""
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


def other_configs(timeout_s=120):
    """The other BASELINE configs measured in THIS run on THIS box (subprocesses of the repository's own tools, each under
    its own timeout; a failure is reported, never hidden, and never touches the main line): configs[2] DCN-v2 and
    configs[3] DIN train steps + the sibling nets (tools/bench_models.py), row P = the multi-slot pool kernel and the
    gpubox model's train step on the PS accessor table (tools/slot_dnn_bench.py), configs[4] = one GPU's 1.25e9-row share
    of the hashed table through the row-sharded code path (this script, world 1)."""
    import subprocess
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    jobs = [("tools/bench_models.py", [sys.executable, os.path.join(REPO, "tools", "bench_models.py")]),
            ("tools/slot_dnn_bench.py --opt ps", [sys.executable, os.path.join(REPO, "tools", "slot_dnn_bench.py"),
                                                  "--opt", "ps"]),
            # L-trainer through the reference's own entry point: unpatched net.py / patched (custom op) / the engine's loop
            ("tools/ref_entry_bench.py", [sys.executable, os.path.join(REPO, "tools", "ref_entry_bench.py")]),
            ("bench.py --force-sharded --table ps --hashed-rows 1250000000",
             [sys.executable, os.path.abspath(__file__), "--force-sharded", "--table", "ps", "--hashed-rows",
              "1250000000", "--no-cpu-baseline", "--steps", "20", "--warmup", "5"]),
            # BASELINE.md §2 config 2b — the reference's own layout: ONE shared [1 000 001, D] table, D 9
            # (deepfm/config.yaml:48-50) and D 10 (benchmark.yaml:21); and the Zipf(1.05) ids of SURVEY §8(d) (hot rows)
            ("bench.py --shared-table --dim 9", [sys.executable, os.path.abspath(__file__), "--shared-table", "--dim", "9",
                                                 "--no-cpu-baseline", "--steps", "20", "--warmup", "5"]),
            ("bench.py --shared-table --dim 10", [sys.executable, os.path.abspath(__file__), "--shared-table", "--dim", "10",
                                                  "--no-cpu-baseline", "--steps", "20", "--warmup", "5"]),
            ("bench.py --ids zipf", [sys.executable, os.path.abspath(__file__), "--ids", "zipf", "--no-cpu-baseline",
                                     "--steps", "20", "--warmup", "5"]),
            # the reference's dygraph-default optimizer on configs[1]: Adam lazy_mode=False, the whole table every step
            # (~10 GB per step on 26 tables) next to the lazy headline
            ("bench.py --non-lazy-adam", [sys.executable, os.path.abspath(__file__), "--non-lazy-adam", "--no-cpu-baseline",
                                          "--steps", "10", "--warmup", "3"]),
            # the headline workload with every GEMM on the exact-f32 MFMA kernels (v_mfma_f32_16x16x4_f32): what the
            # bf16 x 3 split of the tall MLP GEMMs buys, same run, same box
            ("REC_GEMM_BF16X3=0 bench.py", [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--steps", "20",
                                            "--warmup", "5"]),
            # configs[4] share with the exchanges of an 8-GPU step emulated by paced link kernels (rec_link_emulate:
            # profiles/r06_links.txt) — the one-GPU evidence for "the exchange fits beside the dW GEMMs"
            ("REC_EMULATE_LINKS=8 bench.py --force-sharded --table ps --hashed-rows 1250000000",
             [sys.executable, os.path.abspath(__file__), "--force-sharded", "--table", "ps", "--hashed-rows",
              "1250000000", "--no-cpu-baseline", "--steps", "20", "--warmup", "5"]),
            # the reference's OWN batch size (deepfm/config_bigdata.yaml:23: 512): launch-bound.  On its own table layout (one
            # shared table, D 9) and on 26 slot tables (D 16) as round 6 leaves it — one-launch GEMMs (csrc/gemm_direct.h),
            # merge + update by row buckets, folds + dense Adam as roles of that launch (csrc/tail_roles.h), issued by
            # rec_deepfm_train_step — and with those three switched off (the round-5 step: tiled GEMMs + split-K reduces,
            # wave-per-lookup merge, every fold a launch, the recorded call list)
            ("bench.py --batch 512 --shared-table --dim 9",
             [sys.executable, os.path.abspath(__file__), "--batch", "512", "--shared-table", "--dim", "9", "--no-cpu-baseline",
              "--steps", "300", "--warmup", "30", "--no-other-configs"]),
            ("bench.py --batch 512",
             [sys.executable, os.path.abspath(__file__), "--batch", "512", "--no-cpu-baseline",
              "--steps", "300", "--warmup", "30", "--no-other-configs"]),
            ("REC_GEMM_DIRECT=0 REC_SMALL_BUCKET=0 REC_SMALL_C_STEP=0 bench.py --batch 512 --shared-table --dim 9",
             [sys.executable, os.path.abspath(__file__), "--batch", "512", "--shared-table", "--dim", "9", "--no-cpu-baseline",
              "--steps", "300", "--warmup", "30", "--no-other-configs"]),
            # the pipelined step schedule (DeepFMLayer.pipelined: fm_bwd -> update -> next lookup on one stream)
            ("REC_DEEPFM_PIPELINED=1 bench.py", [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--steps",
                                                 "20", "--warmup", "5", "--no-other-configs"])]
    keep = ("config", "workload", "ms", "ms_per_step", "samples_per_s", "value", "unit", "roofline", "pool_fwd_ms",
            "train_step_ms", "kernels_ms", "entry", "reader_ms", "batch_ms", "error")
    out = []
    for name, cmd in jobs:
        t0 = time.time()
        try:
            job_env = dict(env, REC_GEMM_BF16X3="0") if name.startswith("REC_GEMM_BF16X3=0") else env
            if name.startswith("REC_EMULATE_LINKS=8"):
                job_env = dict(env, REC_EMULATE_LINKS="8")
            if name.startswith("REC_DEEPFM_PIPELINED=1"):
                job_env = dict(env, REC_DEEPFM_PIPELINED="1")
            if name.startswith("REC_GEMM_DIRECT=0"):      # every leading VAR=value of the name
                job_env = dict(env, **dict(t.split("=", 1) for t in name.split(" bench.py")[0].split()))
            r = subprocess.run(cmd + (["--no-other-configs"] if name.startswith("REC_GEMM_BF16X3=0") else []), cwd=REPO,
                               env=job_env, capture_output=True, text=True, timeout=timeout_s)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out.append({"command": name, "error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])})
                continue
            for ln in lines:
                d = json.loads(ln)
                e = {k: d[k] for k in keep if k in d}
                if "config" in d and isinstance(d["config"], dict):      # a bench.py line: its config object names the workload
                    e["workload"] = d["config"].get("workload")
                    e["config"] = ("configs[4] share + emulated links of an 8-GPU step (rec_link_emulate: 8 workgroups per "
                                   "collective for remote bytes / (7 x 153 GB/s) + 8 us)" if name.startswith("REC_EMULATE_LINKS")
                                   else "configs[1], pipelined step schedule" if name.startswith("REC_DEEPFM_PIPELINED") else
                                   "configs[0] shape at the reference's batch size 512 (launch-bound), the round-5 step: tiled "
                                   "GEMMs + split-K reduce launches, wave-per-lookup merge, every fold a launch of its own"
                                   if name.startswith("REC_GEMM_DIRECT=0") else
                                   "configs[0] shape at the reference's batch size 512 (launch-bound)" if "512" in cmd else
                                   "configs[4] (one GPU's share, row-sharded path at world 1)" if "--table" in cmd else
                                   "configs[1] layout 2b (one shared table)" if "--shared-table" in cmd else
                                   "configs[1] with the dygraph-default NON-lazy Adam" if "--non-lazy-adam" in cmd else
                                   "configs[1] with every GEMM on the exact-f32 MFMA kernels (REC_GEMM_BF16X3=0)"
                                   if name.startswith("REC_GEMM_BF16X3=0") else "configs[1] with Zipf(1.05) ids")
                    if name.startswith("REC_GEMM_BF16X3=0"):
                        e["dtype"] = d.get("dtype")
                    e["roofline"] = {k: d["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac",
                                                                       "in_step_frac")}
                e["command"] = name
                out.append(e)
        except Exception as e:      # timeout, missing tool, bad JSON
            out.append({"command": name, "error": repr(e)[:300]})
        out[-1]["wall_s"] = round(time.time() - t0, 1)
    return out


def reference_trainer_bs512(budget_s=90.0, lines=4096):
    """the reference's unmodified tools/trainer.py once more at the batch size of its OWN
    full-data config (models/rank/deepfm/config_bigdata.yaml: bs 512, D 10? fc 400x3) on a synthetic slot file of `lines`
    lines in the reference's text format, so that `reference_trainer` is not only a bs-2 number.  Same caveat: the
    reference's loop + reader over the NumPy oracle backend, not Paddle's kernels."""
    import re
    import subprocess
    import tempfile
    ref = next((d for d in (os.path.join(REPO, "oracle", "_ref", "PaddleRec"), "/root/reference")
                if os.path.isdir(os.path.join(d, "tools"))), None)
    cfg = os.path.join(ref or "", "models", "rank", "deepfm", "config_bigdata.yaml")
    if ref is None or not os.path.isfile(cfg):
        return {"error": "no staged config_bigdata.yaml (oracle/make_ref_tree.py)"}
    threads = min(os.cpu_count() or 1, 32)
    env = dict(os.environ, REC_COMPAT_KERNELS="cpu_kernels", OMP_NUM_THREADS=str(threads), PYTHONDONTWRITEBYTECODE="1")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "tests"), REPO, env.get("PYTHONPATH", "")])
    rng = np.random.default_rng(7)
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "train")
        os.makedirs(data)
        with open(os.path.join(data, "part-0"), "w") as f:
            for _ in range(lines):
                ids = rng.integers(1, 1000000, 26)
                dense = rng.random(13)
                f.write("click:%d " % (rng.random() < 0.25) + " ".join("dense_feature:%.6f" % x for x in dense) + " " +
                        " ".join("%d:%d" % (i + 1, v) for i, v in enumerate(ids)) + "\n")
        cmd = [sys.executable, "-m", "paddlerec_amd.run_reference", os.path.join(ref, "tools", "trainer.py"), "-m", cfg,
               "-o", "runner.epochs=1", "runner.print_interval=2", "runner.use_gpu=False", "runner.train_data_dir=%s" % data,
               "runner.model_save_path=%s" % os.path.join(tmp, "o")]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd=ref, env=env, capture_output=True, text=True, timeout=budget_s)
        except subprocess.TimeoutExpired:
            return {"error": "did not finish in %.0f s" % budget_s}
        dt = time.perf_counter() - t0
    log = r.stdout + r.stderr
    ips = [float(x) for x in re.findall(r"ips: ([0-9.]+) ins/s", log)]
    bs = re.search(r"train_batch_size: (\d+)", log)
    if r.returncode != 0 or len(ips) < 2:
        return {"error": "failed (rc %d): %s" % (r.returncode, log[-300:])}
    return {"value": sum(ips[1:]) / len(ips[1:]), "unit": "samples/s", "cores": threads, "host_cores": os.cpu_count(),
            "kind": "reference-trainer-over-shim",
            "sample": "the reference's unmodified tools/trainer.py, 1 epoch of config_bigdata.yaml (bs %s) on %d synthetic "
                      "slot-text lines, NumPy oracle backend; its own ips lines (%d intervals), %.1f s wall"
                      % (bs.group(1) if bs else "?", lines, len(ips) - 1, dt)}


def reference_trainer_baseline(budget_s=60.0):
    """north_star / SURVEY §8(d) "CPU baseline (1)": the reference's OWN tools/trainer.py, unmodified, on its own
    models/rank/deepfm/config.yaml (BASELINE configs[0]: sample data, bs 2, D 9, 1 000 001-row table, dygraph Adam
    lazy_mode=False), run as a subprocess through paddlerec_amd.run_reference over the `paddle` compat namespace with
    the ORACLE operator backend (tests/cpu_kernels.py), on this box's host cores, in this same bench run.  The value
    is the reference's own `ips` log line (tools/trainer.py:179-186), mean over the printed intervals after the first.
    Real PaddlePaddle CPU numbers cannot be produced (no Paddle wheel): this measures the reference's Python loop +
    reader + the NumPy oracle.  The tree is oracle/_ref/PaddleRec (byte copies staged by oracle/make_ref_tree.py) or
    /root/reference where that exists."""
    import re
    import subprocess
    import tempfile
    ref = next((d for d in (os.path.join(REPO, "oracle", "_ref", "PaddleRec"), "/root/reference")
                if os.path.isdir(os.path.join(d, "tools"))), None)
    if ref is None:
        return {"error": "no reference tree (oracle/make_ref_tree.py stages it in the build container)"}
    threads = min(os.cpu_count() or 1, 16)        # the NumPy oracle at bs 2 does not scale past a few threads
    env = dict(os.environ, REC_COMPAT_KERNELS="cpu_kernels", OMP_NUM_THREADS=str(threads), PYTHONDONTWRITEBYTECODE="1")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "tests"), REPO, env.get("PYTHONPATH", "")])
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, "-m", "paddlerec_amd.run_reference", os.path.join(ref, "tools", "trainer.py"),
               "-m", os.path.join(ref, "models", "rank", "deepfm", "config.yaml"), "-o", "runner.epochs=1",
               "runner.print_interval=5", "runner.use_gpu=False", "runner.model_save_path=%s" % os.path.join(tmp, "o")]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd=ref, env=env, capture_output=True, text=True, timeout=budget_s)
        except subprocess.TimeoutExpired:
            return {"error": "reference trainer did not finish in %.0f s" % budget_s}
        dt = time.perf_counter() - t0
    log = r.stdout + r.stderr
    ips = [float(x) for x in re.findall(r"ips: ([0-9.]+) ins/s", log)]
    if r.returncode != 0 or len(ips) < 2:
        return {"error": "reference trainer failed (rc %d): %s" % (r.returncode, log[-300:])}
    return {"value": sum(ips[1:]) / len(ips[1:]), "unit": "samples/s", "cores": threads, "host_cores": os.cpu_count(),
            "kind": "reference-trainer-over-shim",
            "sample": "the reference's unmodified tools/trainer.py, 1 epoch of models/rank/deepfm/config.yaml "
                      "(configs[0]: 80 sample lines, bs 2, D 9, N 1000001, non-lazy Adam) over the compat namespace "
                      "with the NumPy oracle backend; its own ips lines (%d intervals), %.1f s wall" % (len(ips) - 1, dt)}


class _QuietStdout:
    """fd 1 -> fd 2 for a region, C stdio flushed before it is restored."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *a):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def self_spawn(n):
    """`python bench.py --gpus N` called plainly (no WORLD_SIZE in the environment): re-execute this same command
    line as N ranks, one per GPU, under torch.distributed.run on 127.0.0.1 with a free port — the counterpart of the
    reference's one-command 8-GPU launch (tools/run_gpubox.sh:21, tools/static_gpubox_trainer.py:152-160).  The
    children inherit stdout: rank 0 prints the ONE JSON line.  Returns the launcher's exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env["REC_BENCH_SELF_SPAWNED"] = "1"
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--rows-per-table", type=int, default=1_000_000)
    ap.add_argument("--fc", type=str, default="400,400,400")
    ap.add_argument("--ids", choices=("uniform", "zipf"), default="uniform",
                    help="id distribution of SURVEY.md §8(d): (U) uniform [default] or (Z) Zipf 1.05 (hot rows)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU baselines AND the other-configs section (quick runs, profiling)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the BASELINE configs[2]/[3]/[4] + row-P measurements appended to the line at N = 1")
    ap.add_argument("--c-step", action="store_true",
                    help="issue every step through the one-call C entry point rec_deepfm_train_step (one stream; what a "
                         "non-Python binder gets) instead of the Python mirror's step")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the row-sharded path (RCCL all-to-all) even with one rank")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--non-lazy-adam", action="store_true",
                    help="single GPU: the reference's DYGRAPH optimizer (deepfm/dygraph_model.py:61-65, Adam lazy_mode=False"
                         " — every row of both tables decays every step: 6 N (D+1) 4 B of extra traffic) instead of lazy Adam")
    ap.add_argument("--dry-links", action="store_true",
                    help="print the analytic per-collective xGMI volumes / times of the row-sharded step for 2, 4, 8 GPUs "
                         "(no GPU needed) and exit")
    ap.add_argument("--shared-table", action="store_true",
                    help="layout (2b) of SURVEY §8(d), the reference's own: ONE table of rows-per-table (+1) rows shared "
                         "by the 26 slots (deepfm/config.yaml:48-50 with --dim 9, benchmark.yaml:21 with --dim 10)")
    ap.add_argument("--table", choices=("auto", "adam", "ps"), default="auto",
                    help="ps = BASELINE configs[4]: the hashed gpubox table (uint64 feasigns -> mix64 %% N rows on the "
                         "device, AdaGrad accessor record, rows born lazily), row-sharded; implies the sharded path.  "
                         "auto (default): adam on one GPU (configs[1]), ps on N > 1 GPUs — 1.25e9 rows per GPU, i.e. "
                         "the 10^10-row table of configs[4] at 8 GPUs (weak scaling: the table grows with N)")
    ap.add_argument("--hashed-rows", type=int, default=1_250_000_000,
                    help="--table ps: table rows PER GPU (10^10 / 8 = 1.25e9 = 160 GB of 128-B records)")
    args = ap.parse_args()

    if args.dry_links:
        print(json.dumps(dry_links(args.batch, 26, args.dim, "ps" if args.table in ("auto", "ps") else "adam")))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.table == "auto":
        args.table = "ps" if world > 1 else "adam"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rows_req = args.hashed_rows
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch as `python bench.py --gpus N` (spawns its own ranks) or "
                         "under torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    # diagnostics on a one-GPU box: REC_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and REC_BENCH_BACKEND=gloo swaps the
    # transport (RCCL refuses two ranks on one device) — the N > 1 code path of this file with the real kernels; the
    # numbers of such a run mean nothing
    if os.environ.get("REC_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("REC_BENCH_BACKEND", "nccl")
    # REC_BENCH_STANDIN=1 (tests/test_bench_cli.py only): this file's launch / rank / collective logic on a box
    # without a GPU — CPU tensors, gloo, and the tests' oracle-backed operator stand-in injected through the layer's
    # `kernels=` argument.  The line it prints says so ("data": "cpu-standin ...") and is not a measurement.
    standin = os.environ.get("REC_BENCH_STANDIN") == "1"
    kernels = None
    if standin:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import cpu_kernels as kernels
        backend = "gloo"
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("rank %d: no GPU %d on this box (%d visible)" % (rank, local_rank,
                                                                             torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    dist = None
    if world > 1 or args.force_sharded or args.table == "ps":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints a version banner on the C-level stdout when a communicator is created (flushed at exit, i.e.
        # AFTER the result): stdout is pointed at stderr until the warm-up has created every communicator, so that
        # this program's stdout stays the ONE JSON line of the contract
        quiet = _QuietStdout()
        quiet.__enter__()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if not standin:
        from paddlerec_amd import _lib
        _lib.lib()                               # fail loudly if the HIP library is not built
    B, S, Dn, D = args.batch, 26, 13, args.dim
    fc = [int(x) for x in args.fc.split(",")]
    # weak scaling: every GPU holds 26 x rows_per_table rows; the global table grows with the world
    N = args.rows_per_table * S * world
    so = torch.arange(S, dtype=torch.int64, device=dev) * (args.rows_per_table * world)
    if args.shared_table:
        N, so = args.rows_per_table * world, None
    torch.manual_seed(20250404)                  # same random-init dense weights on every run (and rank)
    if dist is None:
        from paddlerec_amd.deepfm import DeepFMLayer
        model = DeepFMLayer(N, D, Dn, S, fc, device=dev, slot_offset=so, kernels=kernels)
        if args.non_lazy_adam:
            model.lazy_mode = False
        parallelism = "single"
    elif args.table == "ps":
        from paddlerec_amd.sharded import ShardedDeepFMLayer
        # Memory pre-flight (the first real --gpus 8 run must not die of OOM): one 128-B record per row + ~6 GB of
        # activations / exchange buffers / workspaces of a batch-65536 step must fit what the device reports free.  The
        # ranks agree on the smallest answer; a shrunk table is SAID in the line (config.table_rows_requested).
        rows_req = args.hashed_rows
        if not standin:
            free_b, _tot = torch.cuda.mem_get_info(dev)
            fit = int((free_b - 6 * 2 ** 30) * 0.92) // 128
            fit_t = torch.tensor([max(fit, 1000)], dtype=torch.int64, device=dev)
            if world > 1:
                dist.all_reduce(fit_t, op=dist.ReduceOp.MIN)
            if int(fit_t.item()) < args.hashed_rows:
                args.hashed_rows = int(fit_t.item())
                print("[bench] rank %d: %d rows per GPU do not fit (%.1f GB free): table shrunk to %d rows per GPU" %
                      (rank, rows_req, free_b / 2 ** 30, args.hashed_rows), file=sys.stderr, flush=True)
        N = args.hashed_rows * world
        # embedx_threshold 0: a feature is created whole at its first pull (the full-work regime; the shipped
        # config_online.yaml gates embedx behind 10 shows)
        model = ShardedDeepFMLayer(N, D, Dn, S, fc, device=dev, group=dist.group.WORLD, table="ps",
                                   accessor=dict(embedx_threshold=0.0), hash_keys=True, kernels=kernels)
        parallelism = "rowshard%d+dp%d (PS accessor table, hashed uint64 keys)" % (world, world)
    else:
        from paddlerec_amd.sharded import ShardedDeepFMLayer
        model = ShardedDeepFMLayer(N, D, Dn, S, fc, device=dev, slot_offset=so, group=dist.group.WORLD,
                                   kernels=kernels)
        parallelism = "rowshard%d+dp%d" % (world, world)
    batches = make_batches(4, B, S, Dn, args.rows_per_table * world, dev, 20250404 + rank, args.ids)
    if args.table == "ps":      # raw uint64 feasigns (as int64 bit patterns), 3 % padding id 0
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        for i, (ids, dense, label) in enumerate(batches):
            keys = torch.randint(-2 ** 63, 2 ** 63 - 1, ids.shape, device=dev, generator=g, dtype=torch.int64)
            keys[ids == 0] = 0
            batches[i] = (keys, dense, label)

    def step(i):
        ids, dense, label = batches[i % len(batches)]
        if dist is not None:      # the sharded layer routes the next batch's ids a step ahead
            return model.train_step(ids, dense, label, lr=1e-3, next_sparse_inputs=batches[(i + 1) % len(batches)][0])
        if args.c_step:
            return model.train_step_c(ids, dense, label, lr=1e-3)
        return model.train_step(ids, dense, label, lr=1e-3)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    for i in range(args.warmup):
        step(i)
    # A fresh process needs ~10 steps before a step costs what it costs from then on (allocator pools, the probed side
    # stream, workspaces, split-K plans: tools/warmup_probe.py, profiles/r02c_warmup_probe.txt).  The requested W
    # warm-up steps are topped up to 12 untimed steps; the timed region below is EXACTLY --steps steps.
    settle = max(0, 12 - args.warmup)
    for i in range(settle):
        step(args.warmup + i)
    barrier()
    if dist is not None:
        quiet.__exit__()
    first = args.warmup + settle
    t0 = time.perf_counter()
    host_issue = 0.0
    step_host = []
    for i in range(args.steps):
        h0 = time.perf_counter()
        loss, _ = step(first + i)
        step_host.append(time.perf_counter() - h0)
        host_issue += step_host[-1]
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("REC_BENCH_STEP_TIMES"):       # diagnosis: host time of every step (ms) on stderr
        print("host ms per step: " + " ".join("%.2f" % (1e3 * x) for x in step_host), file=sys.stderr)
    # per-region HIP-event timings (kernels_ms / host_issue_ms / in_step_event) come from a few MORE steps with the
    # event brackets switched on — outside the timed region, which therefore carries no measurement markers
    model.timers = None if standin else {}
    n_brk = 0 if standin else min(args.steps, 10)
    if dist is not None:
        model.comm.stats = {}
        if standin:
            n_brk = 1
    for i in range(n_brk):
        step(first + args.steps + i)
    sync()
    exch = None
    if dist is not None:
        exch = model.comm.stats_summary(max(n_brk, 1))
        model.comm.stats = None
        for tag, e in exch.items():      # the link model next to the measurement (self-diagnosing first multi-GPU run)
            e["measured_us"] = e.get("us_per_step")
            e["predicted_us"] = round(link_model_us(tag, e["remote_bytes_per_step"], world, e["calls_per_step"]), 1)
    if standin:
        model.timers = {}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = 1e3 * dt / args.steps
    loss_v = float(loss.item())
    oob = int(model.status.item())

    def med(v):         # median over the bracketed steps: the first of them can carry a one-off host stall of tens of ms
        v = sorted(v)   # (event pools / allocator growth when the brackets switch on) that lands in whichever region
        return v[len(v) // 2] if v else 0.0   # it happens to hit and would dominate a mean over ten steps

    def avg_ms(name):
        return med([a.elapsed_time(b) for a, b in model.timers.get(name, [])])

    k_ms = {k: avg_ms(k) for k in model.timers if not k.endswith("@host")}
    host_ms = {k[:-5]: 1e3 * med(v) for k, v in model.timers.items() if k.endswith("@host")}
    host_ms["step_issue_total"] = 1e3 * host_issue / args.steps     # host time inside train_step (no device sync)
    # The roofline kernels once more, the way rocprofv3's kernel trace sees them: back-to-back launches between
    # one pair of HIP events (no event markers / stream joins between the launches), on the step's own buffers.
    pair_us = None
    if dist is None and hasattr(model, "time_fm_pair") and not standin:
        model.timers = None
        pair_us = model.time_fm_pair([(b[0], b[1]) for b in batches], repeats=20, rounds=3)
    fwd_b, bwd_b = algorithmic_bytes(B, S, Dn, D)
    dfwd_b, dbwd_b = designed_bytes(B, S, Dn, D)
    t_step_pair = (k_ms.get("fm_fwd", 0) + k_ms.get("fm_bwd", 0)) * 1e-3
    in_step = (fwd_b + bwd_b) / t_step_pair / 1e9 if t_step_pair > 0 else 0.0
    if pair_us is not None:
        fwd_s, bwd_s = pair_us[0] * 1e-6, pair_us[1] * 1e-6
        timing = "%d back-to-back launches per HIP-event pair, median of 3 (= the back-to-back population of rocprofv3's kernel trace; the in-step launches are roofline.in_step_event)" % 20
    else:       # sharded path: only the in-step brackets exist
        fwd_s, bwd_s = k_ms.get("fm_fwd", 0) * 1e-3, k_ms.get("fm_bwd", 0) * 1e-3
        timing = "in-step HIP-event brackets"
    achieved = (fwd_b + bwd_b) / (fwd_s + bwd_s) / 1e9 if fwd_s + bwd_s > 0 else 0.0
    # flops EXECUTED: layer 0 runs on the folded [B,(S+1)D] input, not on the [B,(S+Dn)D] graph
    model_compact = getattr(model, "compact", False)
    sizes = [((S + 1) if model_compact else (S + Dn)) * D] + fc + [1]
    t_gemm = (k_ms.get("mlp_fwd", 0) + k_ms.get("mlp_bwd", 0) + k_ms.get("mlp_bwd_dw0", 0)) * 1e-3
    gemm_tf = 3 * mlp_flops(B, sizes) / t_gemm / 1e12 if k_ms.get("mlp_fwd") else 0.0
    tr_f, tr_b, tr_src = pmc_traffic(B, D)
    out = {
        "metric": "CTR samples/sec, Criteo DeepFM bs=65536 (train step: fwd+bwd+optimizer)",
        "value": world * B * args.steps / dt, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        # f32 storage, f32 accumulation everywhere.  The tall MLP GEMMs (forward, dX, dW of the 400-wide layers) multiply
        # on the bf16 matrix pipe: each f32 operand is split into three bf16 terms (x = x0 + x1 + x2 to 2^-25 |x|) and six
        # of the nine term products are accumulated in f32 — error against float64 of the order of the exact-f32 MFMA
        # kernels' (1-4e-7 of sum |a||b| either way: mlp_gemm.error_vs_float64 of this line, tests/test_gemm_gpu.py::
        # test_gemm_bf16x3*); REC_GEMM_BF16X3=0 = those kernels (other_configs)
        "dtype": "f32 (MLP GEMMs: f32 operands as 3 bf16 terms, 6 bf16 MFMAs per product, f32 accumulate)" if X3_ON
        else "f32",
        "data": "synthetic" if not standin else "cpu-standin (REC_BENCH_STANDIN=1: host-logic test of this file's "
                                                "launch path on the tests' operator stand-in; NOT a measurement)",
        "config": {"workload": "DeepFM full Criteo: 26 sparse slots x %d rows x dim %d, 13 dense, "
                               "MLP %s, batch %d per GPU, %s Adam, %s ids" % (args.rows_per_table, D, args.fc, B,
                                                                              "NON-lazy (dygraph default)" if args.non_lazy_adam
                                                                              else "lazy", args.ids)
                   + (" [ONE shared table: the reference's layout]" if args.shared_table else "")
                   if args.table != "ps" else
                   "DeepFM on the hashed gpubox table (configs[4]): %d rows per GPU x %d GPUs x dim %d (one 128-B "
                   "accessor record per row, born lazily), 26 slots of uint64 feasigns hashed on the device, 13 dense, "
                   "MLP %s, batch %d per GPU, AdaGrad accessor push" % (args.hashed_rows, world, D, args.fc, B),
                   "global_batch": world * B, "parallelism": parallelism, "untimed_settle_steps": settle,
                   **({"exchange": "rec_alltoall_exchange (C-ABI, RCCL)" if model.comm.native is not None
                       else "torch.distributed (%s)" % ("RCCL" if backend == "nccl" else backend),
                       "rccl_ranks": model.comm.native_ranks} if dist is not None else {}),
                   "table_rows_total": N, "loss": loss_v, "index_oob_flag": oob,
                   **({"step_entry": "rec_deepfm_train_step (one C-ABI call per step, one stream)"}
                      if args.c_step or getattr(model, "_ws_c", None) is not None else {}),
                   **({"table_rows_requested": rows_req * world} if args.table == "ps" and dist is not None
                      and rows_req != args.hashed_rows else {}),
                   **({"native_init_timed_out": True} if dist is not None and getattr(model.comm, "native_init_timed_out", False)
                      else {}),
                   # the N = 1 headline line measures configs[1] (plain Adam table, no exchange); THIS workload at world 1
                   # is the N = 1 line's other_configs entry named below — the like-for-like base of a scaling ratio
                   **({"scaling_base": "other_configs entry \"configs[4] (one GPU's share, row-sharded path at world 1)\" of "
                                       "the --gpus 1 line (same table, same step, exchange degenerated to copies); the "
                                       "--gpus 1 headline value is configs[1] on the plain table"}
                      if dist is not None and args.table == "ps" else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (tr_f + tr_b) if tr_f is not None else None,
                     "traffic_source": tr_src,       # committed PMC passes of this command, NOT this run's counters
                     "algorithmic_bytes": fwd_b + bwd_b,
                     "kernel": "fm_fwd_kernel + fm_bwd_kernel (embedding+FM fwd+bwd, SURVEY §8(d) bytes: "
                               "%d B/sample)" % ((fwd_b + bwd_b) // B),
                     "timing": timing,
                     "per_kernel": {
                         "fm_fwd": {"us": fwd_s * 1e6, "algorithmic_bytes": fwd_b, "designed_bytes": dfwd_b,
                                    "GBs_algorithmic": fwd_b / fwd_s / 1e9 if fwd_s else None,
                                    "GBs_designed": dfwd_b / fwd_s / 1e9 if fwd_s else None, "pmc_bytes": tr_f},
                         "fm_bwd": {"us": bwd_s * 1e6, "algorithmic_bytes": bwd_b, "designed_bytes": dbwd_b,
                                    "GBs_algorithmic": bwd_b / bwd_s / 1e9 if bwd_s else None,
                                    "GBs_designed": dbwd_b / bwd_s / 1e9 if bwd_s else None, "pmc_bytes": tr_b}},
                     # the same pair as bracketed INSIDE a step (event marker + join overhead included)
                     "in_step_event": {"fm_fwd_ms": k_ms.get("fm_fwd"), "fm_bwd_ms": k_ms.get("fm_bwd"),
                                       "achieved": in_step, "frac": in_step / HBM_PEAK_GBS,
                                       # what a bracket measures around NOTHING (two event records back to back on the
                                       # stream): the marker's own share of every bracketed region above — reported,
                                       # not subtracted (profiles/*_bench_populations.txt holds rocprofv3's kernel clock)
                                       "empty_bracket_us": empty_bracket_us(dev) if not standin else None},
                     "frac_designed_bytes": ((dfwd_b + dbwd_b) / (fwd_s + bwd_s) / 1e9 / HBM_PEAK_GBS)
                     if fwd_s + bwd_s > 0 else None,
                     # the honest in-step numbers as top-level scalars of this object (the driver's record keeps scalars,
                     # not nested objects): the pair bracketed inside a step, and the MLP GEMMs of the same step
                     "in_step_frac": in_step / HBM_PEAK_GBS, "in_step_us": t_step_pair * 1e6,
                     "back_to_back_us": (fwd_s + bwd_s) * 1e6,
                     "mlp_gemm_frac": gemm_tf / FP32_MFMA_PEAK_TF, "mlp_gemm_tflops": gemm_tf},
        "kernels_ms": k_ms, "host_issue_ms": host_ms,
        **({"exchange": {"rccl_ranks": model.comm.native_ranks, "world": world,
                         "xgmi_links_per_gpu": 7, "xgmi_link_GBs": 153.0,
                         # per step and per GPU (rank 0's view): bytes handed to each collective, bytes that leave the
                         # GPU, microseconds on the stream, egress GB/s = remote bytes / time (all 7 links together)
                         "per_collective": exch,
                         "predicted_us": round(sum(e["predicted_us"] for e in exch.values()), 1),
                         "measured_us": round(sum((e["measured_us"] or 0.0) for e in exch.values()), 1),
                         "model": "remote bytes / (links x 153 GB/s) + 8 us per call: all-to-all on min(G-1, 7) links, "
                                  "ring all-reduce on one (bench.py --dry-links prints it for 2 / 4 / 8 GPUs)"}}
           if exch is not None else {}),
        # f32-equivalent flops of the MLP / their time in the step.  With the bf16 x 3 kernels the pipe that bounds them is
        # the bf16 MFMA at six instructions per f32 product: peak = 2500 / 6; the exact-f32 figure stays beside it
        "mlp_gemm": {"bound": "mfma", "achieved": gemm_tf,
                     "peak": BF16_MFMA_PEAK_TF / 6 if X3_ON else FP32_MFMA_PEAK_TF, "unit": "TFLOP/s (f32-equivalent)",
                     "frac": gemm_tf / (BF16_MFMA_PEAK_TF / 6 if X3_ON else FP32_MFMA_PEAK_TF),
                     "arithmetic": "bf16x3 (forward, dX and dW GEMMs of the tower; REC_GEMM_BF16X3=0: exact f32 MFMA)"
                     if X3_ON else "exact f32 MFMA",
                     "f32_mfma_peak": FP32_MFMA_PEAK_TF, "vs_f32_mfma_peak": gemm_tf / FP32_MFMA_PEAK_TF,
                     "flops_executed_per_step": 3 * mlp_flops(B, sizes)},
    }
    if rank == 0 and not standin and dev.type == "cuda":
        try:
            out["mlp_gemm"]["error_vs_float64"] = gemm_accuracy(dev)
        except Exception as e:
            out["mlp_gemm"]["error_vs_float64"] = {"error": repr(e)[:200]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(B, S, Dn, D, fc, args.rows_per_table, args.cpu_budget)
            except Exception as e:  # the oracle .so is test infrastructure; report, do not hide
                out["cpu_baseline"] = {"error": repr(e)}
            try:    # the reference's own trainer loop on this box's cores, same run (its CPU-runnable config)
                out["cpu_baseline"]["reference_trainer"] = reference_trainer_baseline()
            except Exception as e:
                out["cpu_baseline"]["reference_trainer"] = {"error": repr(e)}
            try:    # ... and at the batch size of its full-data config (bs 512) on a synthetic slot file
                out["cpu_baseline"]["reference_trainer_bs512"] = reference_trainer_bs512()
            except Exception as e:
                out["cpu_baseline"]["reference_trainer_bs512"] = {"error": repr(e)}
            if not args.no_other_configs and B == 65536 and not standin:
                del model, batches          # the sub-benchmarks get the whole GPU
                torch.cuda.empty_cache()
                out["other_configs"] = other_configs()
        print(json.dumps(out), flush=True)
    if dist is not None:
        # leave together: a rank that tears its communicator down while a peer is still inside its last collective
        # takes that peer down with it (seen with 8 gloo ranks: "terminate called without an active exception")
        try:
            dist.barrier()
        except Exception as e:      # noqa: BLE001 - the line is out; a failed good-bye must not turn into a non-zero exit
            print("[bench] rank %d: final barrier failed: %r" % (rank, e), file=sys.stderr)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
