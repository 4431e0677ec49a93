"""bench.py's command line as the DRIVER calls it: `python bench.py --gpus N` with no
WORLD_SIZE in the environment must spawn its own N ranks (the reference launches its 8-GPU run with one command,
/root/reference/tools/run_gpubox.sh:21) and rank 0 must print exactly ONE JSON line with n_gpus = N.

No GPU here: REC_BENCH_STANDIN=1 runs the same file on CPU tensors over gloo with the tests' operator stand-in — the
launch, rendezvous, rank / table selection (configs[4]: the hashed PS table, row-sharded), collective order and JSON
assembly are the code under test, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env=None, timeout=600):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(REC_BENCH_STANDIN="1", OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must be ONE JSON line, got:\n" + r.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--steps", "2", "--warmup", "1", "--batch", "64", "--fc", "16,8", "--no-cpu-baseline"]


def test_plain_command_line_spawns_two_ranks_on_the_ps_table():
    d = _run(["--gpus", "2"] + SMALL + ["--hashed-rows", "5000"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 128
    assert d["config"]["table_rows_total"] == 10000          # weak scaling: rows per GPU x world
    assert "configs[4]" in d["config"]["workload"] and "rowshard2+dp2" in d["config"]["parallelism"]
    assert d["config"]["exchange"].startswith("torch.distributed")      # gloo here; RCCL C-ABI on the GPU
    assert d["config"]["index_oob_flag"] == 0
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "samples/s"
    assert d["value"] > 0 and d["data"].startswith("cpu-standin")
    assert 0.0 < d["config"]["loss"] < 5.0


def test_plain_command_line_adam_table_two_ranks():
    d = _run(["--gpus", "2", "--table", "adam", "--rows-per-table", "300"] + SMALL)
    assert d["n_gpus"] == 2 and d["config"]["table_rows_total"] == 300 * 26 * 2
    assert d["config"]["parallelism"] == "rowshard2+dp2"


def test_single_rank_line_and_contract_keys():
    d = _run(["--rows-per-table", "300"] + SMALL)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"].startswith("f32")


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", REC_BENCH_STANDIN="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_eight_ranks_on_the_ps_table_print_the_exchange_fields():
    """nothing had run with more than 2 ranks.  `python bench.py --gpus 8` as the driver starts it
    (gloo stand-in): eight ranks rendezvous, shard the hashed table 8 ways, and rank 0's line carries the per-collective
    exchange fields — the three all-to-alls (ids, rows, gradients) and the dense all-reduce with bytes per step."""
    d = _run(["--gpus", "8"] + SMALL + ["--hashed-rows", "2000"], timeout=900)
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 * 64 and d["config"]["table_rows_total"] == 16000
    assert "rowshard8+dp8" in d["config"]["parallelism"] and d["config"]["index_oob_flag"] == 0
    ex = d["exchange"]
    assert ex["world"] == 8 and ex["rccl_ranks"] == 0                 # gloo here: the C-ABI communicator has no ranks
    pc = ex["per_collective"]
    for tag in ("a2a_ids", "a2a_rows", "a2a_grads", "allreduce_dense"):
        assert tag in pc and pc[tag]["bytes_per_step"] > 0, tag
    # 7/8 of the routed lookups leave the rank (uniform hash): ids 8 B, rows 16 + 1 floats, gradients 16 + 2 floats
    n_lookups = 64 * 26 * 0.97
    assert 0.7 * n_lookups * 8 < pc["a2a_ids"]["remote_bytes_per_step"] < n_lookups * 8
    # (a rank sends its n_send ids / gradient rows out and the n_recv rows it OWNS back: equal only on average)
    assert abs(pc["a2a_rows"]["bytes_per_step"] / pc["a2a_ids"]["bytes_per_step"] / ((17 * 4) / 8) - 1) < 0.2
    assert abs(pc["a2a_grads"]["bytes_per_step"] / pc["a2a_ids"]["bytes_per_step"] / ((18 * 4) / 8) - 1) < 0.2
    assert 0.0 < d["config"]["loss"] < 5.0
    # the link model next to every collective: predicted from the remote bytes, 7 links at 8 ranks
    for tag, e in pc.items():
        want = e["remote_bytes_per_step"] / ((1 if tag.startswith("allreduce") else 7) * 153e9) * 1e6 + 8.0 * e["calls_per_step"]
        assert abs(e["predicted_us"] - want) < 0.06 and "measured_us" in e, tag
    assert abs(ex["predicted_us"] - sum(e["predicted_us"] for e in pc.values())) < 0.3 and "measured_us" in ex



def test_dry_links_prints_the_analytic_exchange_table():
    """bench.py --dry-links: per-GPU, per-step exchange volumes and link-model times for 2 / 4 / 8 GPUs, no GPU, no ranks
    — SURVEY §8(d): at G = 8 the rows coming back are B S D 4 (G-1)/G = 95.4 MB (+ the first-order column)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-links"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert [w["world"] for w in d["worlds"]] == [2, 4, 8]
    w8 = d["worlds"][2]["per_collective"]
    assert abs(w8["a2a_ids"]["remote_bytes_per_step"] - 65536 * 26 * 8 * 7 / 8) < 1
    assert abs(w8["a2a_rows"]["remote_bytes_per_step"] - 65536 * 26 * 17 * 4 * 7 / 8) < 1
    assert 80 < w8["a2a_rows"]["predicted_us"] < 140          # ~101 MB over 7 x 153 GB/s
    assert d["worlds"][0]["predicted_exchange_us_total"] > d["worlds"][2]["predicted_exchange_us_total"]
