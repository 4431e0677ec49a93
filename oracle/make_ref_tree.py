#!/usr/bin/env python3
"""Stages the few reference files its own entry points need into oracle/_ref/PaddleRec/ (test infrastructure).

    python oracle/make_ref_tree.py [--ref /root/reference]

/root/reference does not exist on the GPU box; oracle/_ref/ is git-ignored (never part of this repo's history) but
travels with the gpurun snapshot like the built .so files.  With the staged tree the box can run
  * the reference's UNMODIFIED tools/trainer.py on its own models/rank/deepfm/config.yaml (BASELINE configs[0]) over
    the compat namespace with the oracle operator backend — bench.py's `cpu_baseline.reference_trainer`: the reference's
    own CPU trainer loop timed on the GPU box's host cores in the same run (north_star; SURVEY §8(d) "CPU baseline (1)");
  * the same trainer over the HIP kernels (tests/test_reference_entrypoint.py, -m gpu);
  * tools/static_gpubox_trainer.py + its helpers over the compat namespace (row N1).
Nothing here is compiled, edited or imported by the product; files are copied byte for byte (cmp-checked in
tests/test_reference_entrypoint.py::test_staged_tree_is_a_byte_copy)."""
import argparse
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref", "PaddleRec")

FILES = [
    "tools/trainer.py", "tools/infer.py", "tools/static_gpubox_trainer.py", "tools/run_gpubox.sh", "tools/profiler.py",
    "tools/utils/__init__.py", "tools/utils/utils_single.py", "tools/utils/save_load.py", "tools/utils/envs.py",
    "tools/utils/static_ps/__init__.py", "tools/utils/static_ps/reader_helper.py",
    "tools/utils/static_ps/program_helper.py", "tools/utils/static_ps/common_ps.py",
    "tools/utils/static_ps/config_fleet.py", "tools/utils/static_ps/flow_helper.py",
    "tools/utils/static_ps/time_helper.py", "tools/utils/static_ps/metric_helper.py",
    "tools/utils/static_ps/infer_args.py",
    "models/rank/deepfm/config.yaml", "models/rank/deepfm/config_bigdata.yaml", "models/rank/deepfm/net.py", "models/rank/deepfm/dygraph_model.py",
    "models/rank/deepfm/criteo_reader.py", "models/rank/deepfm/data/sample_data/train/sample_train.txt",
    "models/rank/dnn/config_gpubox.yaml", "models/rank/dnn/config.yaml", "models/rank/dnn/net.py",
    "models/rank/dnn/static_model.py", "models/rank/dnn/dygraph_model.py", "models/rank/dnn/criteo_reader.py",
    "models/rank/dnn/queuedataset_reader.py", "models/rank/dnn/data/sample_data/train/sample_train.txt",
    "models/rank/wide_deep/config_gpups.yaml", "models/rank/wide_deep/net.py", "models/rank/wide_deep/static_model.py",
    "models/rank/wide_deep/queuedataset_reader.py", "models/rank/wide_deep/criteo_reader.py",
    "models/rank/wide_deep/data/sample_data/train/sample_train.txt",
    "models/rank/slot_dnn/config_online.yaml", "models/rank/slot_dnn/net.py", "models/rank/slot_dnn/static_model.py",
    "models/rank/slot_dnn/queuedataset_reader.py", "models/rank/slot_dnn/data/demo_10",
    "models/rank/dcn_v2/config.yaml", "models/rank/dcn_v2/net.py", "models/rank/dcn_v2/dygraph_model.py",
    "models/rank/dcn_v2/reader.py", "models/rank/dcn_v2/data/sample_data/sample_train.txt",
    "models/rank/din/config.yaml", "models/rank/din/net.py", "models/rank/din/dygraph_model.py",
    "models/rank/din/dinReader.py", "models/rank/din/data/train_data/sample_data.txt",
]


def stage(ref, dest=DEST, verbose=True):
    n = 0
    for rel in FILES:
        src = os.path.join(ref, rel)
        if not os.path.exists(src):          # optional helpers differ between reference versions
            if verbose:
                print("[make_ref_tree] not in the reference, skipped:", rel)
            continue
        dst = os.path.join(dest, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        n += 1
    with open(os.path.join(dest, "STAGED_FROM"), "w") as f:
        f.write("byte copies of %d files of %s, made by oracle/make_ref_tree.py (test infrastructure, git-ignored)\n"
                % (n, ref))
    return n


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    if not os.path.isdir(os.path.join(a.ref, "tools")):
        sys.exit("reference tree not found at %s" % a.ref)
    print("[make_ref_tree] staged %d files into %s" % (stage(a.ref), DEST))
