#!/usr/bin/env python3
"""Where does the HOST spend a train step?  Wraps torch.empty / torch.zeros / Event.synchronize, every public function
of paddlerec_amd.ops and the Comm collectives with wall-clock timers, runs bench.py's own loop with the given
arguments, and prints calls / total / max per function (ms per step).

    python tools/host_trace.py --force-sharded --no-cpu-baseline --steps 40 --warmup 10"""
import collections
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from paddlerec_amd import ops, sharded  # noqa: E402

STATS = collections.defaultdict(lambda: [0, 0.0, 0.0])
DEPTH = [0]


def wrap(owner, name, label):
    f = getattr(owner, name)

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t
            s = STATS[label]
            s[0] += 1
            s[1] += dt
            s[2] = max(s[2], dt)
    setattr(owner, name, w)


for nm in ("empty", "zeros", "empty_like", "zeros_like"):
    wrap(torch, nm, "torch." + nm)
wrap(torch.cuda.Event, "synchronize", "Event.synchronize")
wrap(torch.cuda.Event, "record", "Event.record")
wrap(torch.cuda.Stream, "wait_stream", "Stream.wait_stream")
for nm, f in list(vars(ops).items()):
    if isinstance(f, types.FunctionType) and not nm.startswith("_"):
        wrap(ops, nm, "ops." + nm)
for nm in ("all_to_all", "all_reduce_sum", "exchange_counts_device", "exchange_counts"):
    if hasattr(sharded.Comm, nm):
        wrap(sharded.Comm, nm, "Comm." + nm)
for nm in ("_lookup", "_route_async", "train_step"):
    wrap(sharded.ShardedDeepFMLayer, nm, "Sharded." + nm)

sys.argv = ["bench.py"] + sys.argv[1:]
sys.path.insert(0, REPO)
import bench  # noqa: E402

steps = 1
for i, a in enumerate(sys.argv):
    if a in ("--steps", "--warmup"):
        steps += int(sys.argv[i + 1]) if a == "--steps" else int(sys.argv[i + 1])
bench.main()
print("%-34s %8s %12s %10s   (nested calls are counted in their callers too)" % ("function", "calls", "ms/step", "max ms"))
for k, (n, tot, mx) in sorted(STATS.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-34s %8d %12.3f %10.3f" % (k, n, 1e3 * tot / max(steps - 1, 1), 1e3 * mx))
