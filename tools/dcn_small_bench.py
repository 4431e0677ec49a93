#!/usr/bin/env python3
"""DCN-v2 (CrossNetV2 depth 3, d 1560, DNN 768-768, train-mode dropout 0.5 + L2Decay + clip) train step at small batch
sizes: recorded call list (REC_STEP_PLAN=1) against the eager step against ONE C-ABI call per step
(rec_dcn_v2_train_step).  One line per (batch, mode)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd.dcn_v2 import DCN_V2Layer  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
for B in (32, 128, 512):
    ids = torch.randint(1, 1100001, (B, 26), device=DEV, generator=g)
    dense = torch.rand(B, 13, device=DEV, generator=g)
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
    for mode in ("plan", "eager", "c-abi"):
        os.environ["REC_STEP_PLAN"] = "0" if mode == "eager" else "1"
        m = DCN_V2Layer(1100001, 40, 13, 26, [768, 768], 3, device=DEV, dropout_rate=0.5, l2_dnn=1e-7)
        step = m.train_step_c if mode == "c-abi" else m.train_step
        for _ in range(20):
            step(ids, dense, label, lr=1e-3)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            step(ids, dense, label, lr=1e-3)
        b.record()
        torch.cuda.synchronize()
        print("DCN-v2 train step B=%-4d %-5s: %.3f ms" % (B, mode, a.elapsed_time(b) / 200))
        del m
        torch.cuda.empty_cache()
