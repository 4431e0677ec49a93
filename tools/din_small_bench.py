#!/usr/bin/env python3
"""DIN train step at the shipped batch size (din/config.yaml: 32 samples, history up to 152): eager against the recorded
call list (REC_STEP_PLAN) against the hipGraph replay against ONE C-ABI call per step (rec_din_train_step).  Prints one
line per mode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd.din import DINLayer  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
B, T = 32, 152


def timeit(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def problem():
    hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
    hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
    ti = torch.randint(0, 63001, (B, 1), device=DEV, generator=g)
    tc = torch.randint(0, 801, (B, 1), device=DEV, generator=g)
    lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
    mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.5).float()
    return hi, hc, ti, tc, label, mask, ti.expand(B, T).contiguous(), tc.expand(B, T).contiguous()


args = problem()
for mode in ("plan", "eager", "graph", "c-abi", "plan", "c-abi"):   # twice: the first measurement of a process carries its warm-up
    os.environ["REC_STEP_PLAN"] = "0" if mode == "eager" else "1"
    m = DINLayer(64, 64, "sigmoid", False, True, 63001, 801, device=DEV)
    fn = ((lambda: m.train_step_graphed(*args)) if mode == "graph" else
          (lambda: m.train_step_c(*args)) if mode == "c-abi" else (lambda: m.train_step(*args)))   # c-abi: rec_din_train_step
    t = timeit(fn)
    print("DIN train step B=%d T=%d %-5s: %.3f ms  (%.1f k samples/s)" % (B, T, mode, t, B / t))
    del m
