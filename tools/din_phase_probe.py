#!/usr/bin/env python3
"""Where a block of din_attention_fwd_ct_kernel spends its cycles, phase by phase (wave 0's shader clock).  Needs a
measurement build:  REC_HIPCC_DEFINES=-DREC_DIN_PHASE_TIMING=1 (forward; =2: backward, run with DIN_PROBE=bwd) python -m paddlerec_amd.build  (touch csrc/din_attention.hip
first); the shipped library has no such symbol.  python tools/din_phase_probe.py [B T]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import _lib  # noqa: E402
from paddlerec_amd.din import DINLayer  # noqa: E402

DEV = "cuda"
B, T = (int(x) for x in (sys.argv[1:3] + ["4096", "100"][len(sys.argv) - 1:]))
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
args = (hi, hc, ti, tc, label, mask, tis, tcs)
lib = _lib.lib()
if not hasattr(lib, "rec_din_debug_phases"):
    sys.exit("not a -DREC_DIN_PHASE_TIMING build")
buf = (C.c_ulonglong * 16)()
for _ in range(3):
    m.train_step(*args)
torch.cuda.synchronize()
lib.rec_din_debug_phases(buf, 1)
N = 10
for _ in range(N):
    m.train_step(*args)
torch.cuda.synchronize()
lib.rec_din_debug_phases(buf, 1)
KERNEL = os.environ.get("DIN_PROBE", "fwd")      # the measurement build instruments ONE kernel: -DREC_DIN_PHASE_TIMING=1 fwd, =2 bwd
names = ["dl (dout . h) + barrier", "a2 recompute + dz2 + barrier", "dz1 + barrier", "dx MFMA + dh / dq stores",
         "next tile all-zero test (2 barriers)", "sample load + tile load (gather, act1)", "barrier", "-", "-",
         "loop top"] if KERNEL == "bwd" else ["ids issue + layer-1 MFMA", "ids store + partial-sum tree (2 barriers)", "sigmoid + act1 store + barrier",
         "layer 2/3 + barrier", "softmax (wave 0) + barrier", "pool + end of sample + barrier", "row gather issue",
         "row gather wait + LDS store + barrier", "-", "loop top"]
tot = sum(buf[i] for i in range(10))
walked = float(((lens + 31) // 32).sum())
print("B %d T %d: %d launches, %.0f walked tiles per launch; wave-0 cycles per walked tile:" % (B, T, N, walked))
for i in ((9, 0, 1, 2, 3, 4, 5, 6) if KERNEL == "bwd" else (9, 0, 1, 2, 3, 4, 5, 6, 7)):
    print("  %-45s %8.0f cycles  %5.1f %%" % (names[i], buf[i] / N / walked, 100.0 * buf[i] / tot))
print("  total %.0f cycles per tile (two blocks per CU interleave)" % (tot / N / walked))
