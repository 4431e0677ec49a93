"""Run one of the reference's own entry points (tools/trainer.py, tools/infer.py) UNMODIFIED over the engine:

    python -m paddlerec_amd.run_reference /path/to/PaddleRec/tools/trainer.py -m models/rank/deepfm/config.yaml [-o k=v ...]

`import paddle` inside the reference then resolves to paddlerec_amd/compat/paddle (the HIP kernels behind the C-ABI for
the lookup, the GEMMs, the SelectedRows merge and the optimizers).  The working directory is the reference root, as
its scripts expect."""
import os
import runpy
import sys


def _selected_gpus():
    return [g for g in os.environ.get("FLAGS_selected_gpus", "").split(",") if g.strip() != ""]


def _precreate_save_dir(argv):
    """The script's `if not os.path.exists(p): os.makedirs(p)` on runner.model_save_path (static_gpubox_trainer.py:143-146)
    is written for ONE process driving all GPUs; with one process per GPU two ranks can both find the directory missing and
    the second makedirs raises.  The launcher creates it before any rank starts (-m <config.yaml>, -o overrides)."""
    try:
        import yaml
        cfg_path, over = None, []
        for i, a in enumerate(argv):
            if a == "-m" and i + 1 < len(argv):
                cfg_path = argv[i + 1]
            if a == "-o":                                   # -o k=v [k=v ...]
                j = i + 1
                while j < len(argv) and not argv[j].startswith("-"):
                    over.append(argv[j])
                    j += 1
        path = None
        if cfg_path and os.path.exists(cfg_path):
            with open(cfg_path) as f:
                cfg = yaml.safe_load(f) or {}
            path = (cfg.get("runner") or {}).get("model_save_path")
        for o in over:
            if o.startswith("runner.model_save_path="):
                path = o.split("=", 1)[1]
        if path:
            os.makedirs(path, exist_ok=True)
    except Exception:        # best effort: the script reports a bad config itself
        pass


def _spawn_ranks(argv):
    """tools/static_gpubox_trainer.py with FLAGS_selected_gpus naming N > 1 GPUs (tools/run_gpubox.sh:21): the reference
    drives them from one process; the engine runs one process per GPU — re-execute under torch.distributed.run, one rank
    per GPU (what bench.py --gpus N does).  -> exit code, or None when this process should run the script itself."""
    n = len(_selected_gpus())
    if (n <= 1 or "WORLD_SIZE" in os.environ or os.environ.get("TRAINING_ROLE", "TRAINER") == "PSERVER"
            or os.path.basename(argv[0]) != "static_gpubox_trainer.py" or os.environ.get("REC_COMPAT_SPAWN", "1") == "0"):
        return None
    import socket
    import subprocess
    _precreate_save_dir(argv)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "paddlerec_amd.run_reference"] + list(argv)
    env = dict(os.environ)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = os.pathsep.join([here, env.get("PYTHONPATH", "")])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    rc = _spawn_ranks(argv)
    if rc is not None:
        raise SystemExit(rc)
    script = os.path.abspath(argv[0])
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))                 # paddlerec_amd itself
    sys.path.insert(0, os.path.join(here, "compat"))          # `import paddle` -> the compat namespace
    sys.path.insert(0, os.path.dirname(script))               # what `python script.py` puts first
    sys.argv = [script] + argv[1:]
    # a net.py patched by integration/*.patch names the custom-op source through this variable
    os.environ.setdefault("REC_PADDLE_OPS_CC", os.path.join(here, "paddle_ops", "rec_paddle_ops.cc"))
    seed = os.environ.get("REC_COMPAT_SEED")        # tests: the same initial parameters in two runs of one script
    if seed:
        import torch
        torch.manual_seed(int(seed))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
