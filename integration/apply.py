#!/usr/bin/env python3
"""Applies integration/*.patch to a PaddleRec tree.

    python integration/apply.py                       # build container: oracle/_ref/PaddleRec -> oracle/_ref/PaddleRec_rec_ops
    python integration/apply.py --tree /path/to/PaddleRec --in-place      # a maintainer's own checkout

The patches are the whole reference-side change that puts the engine's fused kernels behind the reference's own entry
points: each edits one `models/rank/<model>/net.py` (<= 15 added lines) so that the hot block of its `forward` calls a
custom operator of `rec_ops` (paddlerec_amd/paddle_ops/rec_paddle_ops.cc, loaded with paddle.utils.cpp_extension.load)
instead of the ~15 Paddle ops it is written in; YAML configs, readers, dygraph_model.py and tools/trainer.py stay as they
are.  No reference file is tracked in this repository: the default mode copies the staged byte copies
(oracle/make_ref_tree.py, git-ignored) to a sibling directory and patches the copy (test infrastructure: the GPU box
has no /root/reference)."""
import argparse
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
STAGED = os.path.join(REPO, "oracle", "_ref", "PaddleRec")
PATCHED = os.path.join(REPO, "oracle", "_ref", "PaddleRec_rec_ops")


def patches():
    return sorted(glob.glob(os.path.join(HERE, "*.patch")))


def apply(tree):
    for p in patches():
        r = subprocess.run(["patch", "-p1", "--forward", "--no-backup-if-mismatch", "-i", p], cwd=tree, capture_output=True,
                           text=True)
        if r.returncode != 0:
            raise RuntimeError("patch %s did not apply to %s:\n%s%s" % (os.path.basename(p), tree, r.stdout, r.stderr))


def stage_patched(src=STAGED, dst=PATCHED):
    if not os.path.isdir(os.path.join(src, "tools")):
        raise RuntimeError("no staged reference tree at %s (run oracle/make_ref_tree.py)" % src)
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns("__pycache__", "output_model*", "*.pyc"))
    apply(dst)
    with open(os.path.join(dst, "PATCHED_WITH"), "w") as f:
        f.write("\n".join(os.path.basename(p) for p in patches()) + "\n")
    return dst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", default=None)
    ap.add_argument("--in-place", action="store_true")
    a = ap.parse_args()
    if a.tree and a.in_place:
        apply(a.tree)
        print("[integration] patched", a.tree)
    elif a.tree:
        print("[integration] patched copy:", stage_patched(a.tree, PATCHED))
    else:
        if not os.path.isdir(STAGED):
            sys.exit("no staged tree; run oracle/make_ref_tree.py first")
        print("[integration] patched copy:", stage_patched())
