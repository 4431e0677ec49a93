#!/usr/bin/env python3
"""Sustained f32 MFMA rate of the board (register-only loop, tools/mfma_probe/mfma_probe.hip) against launch length,
occupancy and operand data — the ceiling the f32 GEMM of the hot path should be read against.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_probe/mfma_probe.hip -o tools/mfma_probe/_build/libmfmaprobe.so
    python tools/mfma_probe/run.py"""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "_build", "libmfmaprobe.so"))
lib.mfma_probe.restype = C.c_float
lib.mfma_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
big = torch.randn(1 << 28, device="cuda")          # 1 GiB: beyond L2 and the Infinity Cache
out = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
NACC = 8
FLOP_PER_TRIP = 4 * NACC * 2 * 16 * 16 * 4       # per wave and loop trip


def run(blocks, iters, mode, reps=3, lds=0):
    best = 1e9
    for _ in range(reps):
        best = min(best, lib.mfma_probe(blocks, iters, mode, C.c_void_p(out.data_ptr()), C.c_void_p(st), lds,
                                        C.c_void_p(big.data_ptr()), big.numel() // 4))
    tf = blocks * 4 * iters * FLOP_PER_TRIP / (best * 1e-3) / 1e12
    return best, tf


names = {0: "zeros", 1: "random dense", 2: "ReLU-sparse A"}
print("peak by the datasheet: 157.3 TF (256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz)")
for waves_per_simd in (1, 2):
    blocks = 256 * waves_per_simd
    for mode in (0, 1, 2):
        row = []
        for iters in (2000, 20000, 100000):
            ms, tf = run(blocks, iters, mode)
            row.append("%7.3f ms %6.1f TF" % (ms, tf))
        print("%d wave/SIMD  %-14s  %s" % (waves_per_simd, names[mode], "   ".join(row)))
print("fragments re-read from LDS every 32 MFMAs (a GEMM's feeding pattern):")
for waves_per_simd in (1, 2, 4):
    blocks = 256 * waves_per_simd
    for mode in (0, 1, 2):
        row = []
        for iters in (2000, 20000):
            ms, tf = run(blocks, iters, mode, lds=1)
            row.append("%7.3f ms %6.1f TF" % (ms, tf))
        print("%d wave/SIMD  %-14s  %s" % (waves_per_simd, names[mode], "   ".join(row)))
print("the same plus 16 B per lane and trip streamed from a 1 GiB buffer (HBM traffic beside MFMA + LDS):")
for waves_per_simd in (1, 2, 4):
    blocks = 256 * waves_per_simd
    for mode in (0, 1):
        row = []
        for iters in (500, 2000, 20000):
            ms, tf = run(blocks, iters, mode, lds=2)
            gbs = blocks * 4 * iters * 1024 / (ms * 1e-3) / 1e9
            row.append("%7.3f ms %6.1f TF %5.0f GB/s" % (ms, tf, gbs))
        print("%d wave/SIMD  %-14s  %s" % (waves_per_simd, names[mode], "   ".join(row)))
# one CU busy, the rest idle: the clock an unloaded chip gives a single block
ms, tf = run(1, 100000, 1)
print("ONE block (4 waves on one CU), random dense: %.3f ms -> %.1f TF if all 256 CUs ran at this rate" % (ms, tf * 256))
