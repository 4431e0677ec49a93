#!/usr/bin/env python3
"""xDeepFM train step at B 65536 (xdeepfm/config.yaml: D 9, CIN 128-32, DNN 512-256-128) against the size of the CIN's
outer-product scratch chunk (REC_CIN_CHUNK_MB): one line per size."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from paddlerec_amd.xdeepfm import xDeepFMLayer
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
B = 65536
m = xDeepFMLayer(1000001, 9, 13, 26, [128, 32], [512, 256, 128], device=DEV)
ids = torch.randint(0, 1000001, (B, 26), device=DEV, generator=g)
dense = torch.rand(B, 13, device=DEV, generator=g)
label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
for _ in range(3): m.train_step(ids, dense, label)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): m.train_step(ids, dense, label)
b.record(); torch.cuda.synchronize()
print("%%.2f" %% (a.elapsed_time(b) / 5))
''' % ROOT
for mb in sys.argv[1:] or ["1024", "512", "256", "128", "96", "64", "32"]:
    env = dict(os.environ, REC_CIN_CHUNK_MB=mb)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("REC_CIN_CHUNK_MB=%-5s xDeepFM train step B 65536: %s ms" % (mb, r.stdout.strip().split("\n")[-1] if r.returncode == 0 else "FAILED " + r.stderr[-300:]))
