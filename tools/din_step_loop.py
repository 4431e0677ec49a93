#!/usr/bin/env python3
"""DIN train steps at the shipped batch size (din/config.yaml: 32) for a kernel trace: python tools/din_step_loop.py [B T steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd.din import DINLayer
DEV = "cuda"
B, T, steps = (int(x) for x in (sys.argv[1:4] + ["32", "152", "20"][len(sys.argv) - 1:]))
g = torch.Generator(device=DEV).manual_seed(3)
m = DINLayer(64, 64, "sigmoid", False, True, 63001, 801, device=DEV)
hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
ti = torch.randint(0, 63001, (B, 1), device=DEV, generator=g)
tc = torch.randint(0, 801, (B, 1), device=DEV, generator=g)
lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
label = (torch.rand(B, 1, device=DEV, generator=g) < 0.5).float()
tis, tcs = ti.expand(B, T).contiguous(), tc.expand(B, T).contiguous()
for _ in range(steps):
    m.train_step(hi, hc, ti, tc, label, mask, tis, tcs)
torch.cuda.synchronize()
print("done")
