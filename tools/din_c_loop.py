#!/usr/bin/env python3
"""DIN train step at batch 32 through rec_din_train_step, in a loop (for rocprofv3 timelines)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd.din import DINLayer
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
B, T = 32, 152
hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
ti = torch.randint(0, 63001, (B, 1), device=DEV, generator=g)
tc = torch.randint(0, 801, (B, 1), device=DEV, generator=g)
lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
label = (torch.rand(B, 1, device=DEV, generator=g) < 0.5).float()
args = (hi, hc, ti, tc, label, mask, ti.expand(B, T).contiguous(), tc.expand(B, T).contiguous())
m = DINLayer(64, 64, "sigmoid", False, True, 63001, 801, device=DEV)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    m.train_step_c(*args)
torch.cuda.synchronize()
