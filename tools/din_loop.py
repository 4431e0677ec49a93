#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import ops
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(3)
B, T = 4096, 512
tabs = [torch.randn(n, 64, device=DEV, generator=g) * 0.05 for n in (63001, 801, 63001, 801)]
hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
aw = [torch.randn(s, device=DEV, generator=g) * 0.05 for s in ((512, 80), (80, 40), (40, 1))]
ab = [torch.zeros(s, device=DEV) for s in (80, 40, 1)]
st = ops.new_status(DEV)
for _ in range(3):
    ops.din_attention_pool(hi, hc, hi, hc, mask, *tabs, aw, ab, st)
torch.cuda.synchronize()
print("done")
