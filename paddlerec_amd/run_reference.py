"""Run one of the reference's own entry points (tools/trainer.py, tools/infer.py) UNMODIFIED over the engine:

    python -m paddlerec_amd.run_reference /path/to/PaddleRec/tools/trainer.py -m models/rank/deepfm/config.yaml [-o k=v ...]

`import paddle` inside the reference then resolves to paddlerec_amd/compat/paddle (the HIP kernels behind the C-ABI for
the lookup, the GEMMs, the SelectedRows merge and the optimizers).  The working directory is the reference root, as
its scripts expect."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))                 # paddlerec_amd itself
    sys.path.insert(0, os.path.join(here, "compat"))          # `import paddle` -> the compat namespace
    sys.path.insert(0, os.path.dirname(script))               # what `python script.py` puts first
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
