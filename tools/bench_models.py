#!/usr/bin/env python3
"""L-step / L-kernel numbers for the DCN-v2 and DIN rows of SURVEY.md §8 (BASELINE configs[2], configs[3]).
Not part of bench.py's contract; results are copied into profiles/.  Besides the readable lines, ONE JSON line per
configuration with a `roofline` object (MFMA for the cross / attention GEMM work, HBM for the DIN gather part)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402
from paddlerec_amd.dcn_v2 import DCN_V2Layer  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def timeit_stream(fn, iters=200, warm=40):
    """Launch-bound steps (DIN at batch 32: ~0.2 ms of dependent launches): `iters` steps back to back inside ONE event
    pair after a warm-up long enough for the recorded call list to exist (the third step of a signature records it) —
    a per-step bracket would time the host's gaps, not the step."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


JSON = []
PEAK_TF, PEAK_GBS = 157.3, 8000.0


def record(config, workload, ms, flops, samples, extra=None, hbm_bytes=None, bytes_formula=None):
    """hbm_bytes: the line is NOT MFMA-bound (few rows per GEMM, BatchNorm / optimizer sweeps, a chain of small launches):
    its roofline object is then the HBM one — algorithmic bytes of the step (bytes_formula says which) over the time —
    and the GEMM rate is kept beside it as `mfma_tflops` for reference (a blanket "mfma" said nothing for
    DLRM and for DIN at batch 32)."""
    tf = flops / ms / 1e9
    if hbm_bytes is None:
        roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_TF, "unit": "TFLOP/s", "frac": tf / PEAK_TF, "flops": flops}
        if os.environ.get("REC_GEMM_BF16X3", "1") != "0":     # f32-equivalent flops against the exact-f32 MFMA peak
            roof["note"] = ("tall GEMMs with N >= 368 run as bf16 x 3 (csrc/gemm_bf16x3.h: 6 bf16 MFMAs per f32 product, "
                            "own peak 2500 / 6 = 417 TF-equivalent); peak here is the exact-f32 MFMA figure")
    else:
        gbs = hbm_bytes / ms / 1e6
        roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_GBS, "unit": "GB/s", "frac": gbs / PEAK_GBS,
                "bytes": hbm_bytes, "bytes_formula": bytes_formula, "mfma_tflops": tf}
    d = {"config": config, "workload": workload, "ms": ms, "samples_per_s": samples / ms * 1e3, "dtype": "f32",
         "roofline": roof}
    if extra:
        d.update(extra)
    JSON.append(d)


g = torch.Generator(device=DEV).manual_seed(3)
# ---- DCN-v2, datasets/criteo_dcn_v2 shapes: N 1 100 001, D 40, d 1560, fc [768,768] (dcn_v2/config.yaml:42-58)
for B, mix, depth in ((65536, False, 3), (512, False, 3), (65536, True, 2)):
    m = DCN_V2Layer(1100001, 40, 13, 26, [768, 768], depth, is_Stacked=True, use_low_rank_mixture=mix,
                    low_rank=256, num_experts=4, device=DEV)
    ids = torch.randint(1, 1100001, (B, 26), device=DEV, generator=g)
    dense = torch.rand(B, 13, device=DEV, generator=g)
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.25).long()
    t_f = timeit(lambda: m.forward(ids, dense))
    d = 1560
    if mix:
        fl = depth * (2.0 * B * d * 256 * 4 * 2 + 2.0 * B * 256 * 256 * 4 + 2.0 * B * d * 4) + 2.0 * B * (d * 768 + 768 * 768 + 768)
        print("DCN-v2 CrossNetMix r256 E4 depth%d B=%d: forward %.2f ms  (%.1f TF, %.2f M samples/s)" %
              (depth, B, t_f, fl / t_f / 1e9, B / t_f / 1e3))
        record("configs[2] (shipped CrossNetMix)", "DCN-v2 CrossNetMix r256 E4 depth %d, d 1560, B %d: forward" % (depth, B),
               t_f, fl, B)
    else:
        fl_f = depth * 2.0 * B * d * d + 2.0 * B * (d * 768 + 768 * 768 + 768) + 2.0 * B * 13 * 520
        t_s = timeit(lambda: m.train_step(ids, dense, label, lr=1e-3))
        print("DCN-v2 CrossNetV2 depth%d B=%d: forward %.2f ms (%.1f TF)  train step %.2f ms (%.1f TF, %.2f M samples/s)" %
              (depth, B, t_f, fl_f / t_f / 1e9, t_s, 3 * fl_f / t_s / 1e9, B / t_s / 1e3))
        n_par = depth * (d * d + d) + d * 768 + 768 * 768 + 768 + 768 + 13 * 520 + 520 + 2 * 768
        small = B < 4096      # the reference's own batch: every GEMM has 512 rows — the step streams parameters, not rows
        record("configs[2]", "DCN-v2 CrossNetV2 depth %d, d 1560, DNN 768-768, B %d: train step (fwd + bwd + clip + "
               "Adam)" % (depth, B), t_s, 3 * fl_f, B, {"forward_ms": t_f, "forward_tflops": fl_f / t_f / 1e9},
               hbm_bytes=(n_par * 4 * (7 + 2 + 3) if small else None),
               bytes_formula=("dense parameters P = %d: Adam 7 x 4P (read g, p, m, v; write p, m, v) + clip norm and L2 "
                              "passes 2 x 4P + weights read by the forward / dX / dW GEMMs 3 x 4P" % n_par if small else None))
    del m
    torch.cuda.empty_cache()

# ---- DIN attention-pool, amazonElec shapes: item 63001 x 64, cat 801 x 64 (din/config.yaml:38-44)
for B, T in ((32, 152), (4096, 100), (4096, 512)):
    tabs = [torch.randn(n, 64, device=DEV, generator=g) * 0.05 for n in (63001, 801, 63001, 801)]
    hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
    hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
    lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
    mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
    aw = [torch.randn(s, device=DEV, generator=g) * 0.05 for s in ((512, 80), (80, 40), (40, 1))]
    ab = [torch.zeros(s, device=DEV) for s in (80, 40, 1)]
    st = ops.new_status(DEV)
    att_ws = ops.Workspace(DEV)            # as DINLayer.forward calls it: tile split (B 32) / sample tickets (B 4096)
    t = timeit(lambda: ops.din_attention_pool(hi, hc, hi, hc, mask, *tabs, aw, ab, st, ws=att_ws))
    per_pos = 2.0 * (512 * 80 + 80 * 40 + 40 + 128)
    fl_dense = B * T * per_pos                  # what the reference computes: all B x T positions (padding included)
    # what the KERNEL computes: the 32-position tiles that hold at least one valid position (the padded
    # tail is not walked, so a rate on all B x T positions is a dense-equivalent figure, not a fraction of the pipe)
    pos_exec = int(((lens.reshape(-1) + 31) // 32 * 32).clamp(max=(T + 31) // 32 * 32).sum().item())
    fl = pos_exec * per_pos
    by = B * T * (4 * 8 + 8 + 4 * 256)          # ids + mask + 4 rows of 256 B (algorithmic, rows hit L2)
    print("DIN attention-pool B=%d T=%d: %.3f ms  (%.1f M positions/s dense-equivalent, %.2f TF executed = %.3f of the f32 "
          "MFMA peak on %d of %d positions; %.2f TF dense-equivalent; %.0f GB/s algorithmic)" %
          (B, T, t, B * T / t / 1e3, fl / t / 1e9, fl / t / 1e9 / PEAK_TF, pos_exec, B * T, fl_dense / t / 1e9, by / t / 1e6))
    record("configs[3]", "DIN attention-pool forward (4 gathers + 512-80-40-1 MLP + masked softmax + pool), B %d, "
           "T %d, history lengths uniform in [1, T] (padded tail tiles are not walked: roofline on EXECUTED flops)" % (B, T),
           t, fl, B,
           {"positions_per_s": B * T / t * 1e3, "positions_executed": pos_exec, "positions_dense": B * T,
            "frac_executed": fl / t / 1e9 / PEAK_TF, "dense_equivalent_tflops": fl_dense / t / 1e9,
            "gather_roofline": {"bound": "hbm", "achieved": by / t / 1e6, "peak": PEAK_GBS, "unit": "GB/s",
                                "frac": by / t / 1e6 / PEAK_GBS, "bytes": by}},
           hbm_bytes=(by if B * T < 65536 else None),
           bytes_formula=("B T (4 ids x 8 + mask 8 + 4 rows x 256 B): 32 samples = 32 blocks on 256 CUs, a dependent "
                          "chain per block — neither roofline binds, the bytes are what the kernel must move"
                          if B * T < 65536 else None))

# ---- DIN full train step (din/dygraph_model.py:85-100): attention-pool fwd + bwd, the concat MLP, 7 row-merged SGDs
from paddlerec_amd.din import DINLayer  # noqa: E402

for B, T in ((32, 152), (4096, 100), (4096, 512)):
    m = DINLayer(64, 64, "sigmoid", False, True, 63001, 801, device=DEV)
    hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
    hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
    ti = torch.randint(0, 63001, (B, 1), device=DEV, generator=g)
    tc = torch.randint(0, 801, (B, 1), device=DEV, generator=g)
    lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
    mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.5).float()
    tis, tcs = ti.expand(B, T).contiguous(), tc.expand(B, T).contiguous()
    tm = timeit_stream if B * T <= 15360 else timeit
    t = tm(lambda: m.train_step(hi, hc, ti, tc, label, mask, tis, tcs))
    per_pos = 2.0 * (512 * 80 + 80 * 40 + 40 + 128) + 2.0 * (512 * 80 + 2 * 80 * 40 + 128)
    fl_dense = B * T * per_pos
    pos_exec = int(((lens.reshape(-1) + 31) // 32 * 32).clamp(max=(T + 31) // 32 * 32).sum().item())
    fl = pos_exec * per_pos                     # the 32-position tiles the forward and the backward walk
    print("DIN train step B=%d T=%d: %.3f ms  (%.1f k samples/s, %.1f M positions/s dense-equivalent, attention fwd+bwd "
          "%.1f TF executed = %.3f of the f32 MFMA peak; %.1f TF dense-equivalent)"
          % (B, T, t, B / t, B * T / t / 1e3, fl / t / 1e9, fl / t / 1e9 / PEAK_TF, fl_dense / t / 1e9))
    # bytes of a step: history rows gathered by forward and backward (4 x 256 B per position each), act1 saved and re-read
    # (80 floats), dh / dq written (2 x 128 floats), row gradients merged into 7 tables (read + write 128 floats), ids + mask
    by_step = B * T * (2 * 4 * 256 + 2 * 80 * 4 + 2 * 128 * 4 + 2 * 128 * 4 + 5 * 8)
    record("configs[3]", "DIN train step (attention-pool fwd + bwd on saved activations, concat MLP, row-merged SGD on "
           "7 tables), B %d, T %d (roofline on EXECUTED flops)" % (B, T), t, fl, B,
           {"positions_per_s": B * T / t * 1e3, "positions_executed": pos_exec, "positions_dense": B * T,
            "frac_executed": fl / t / 1e9 / PEAK_TF, "dense_equivalent_tflops": fl_dense / t / 1e9},
           hbm_bytes=(by_step if B * T < 65536 else None),
           bytes_formula=("B T (gathers fwd + bwd 2 x 4 x 256 + act1 2 x 320 + dh / dq 2 x 512 + row-gradient merge 2 x 512 "
                          "+ ids / mask 40) B: ~25 dependent launches at the ~5 us floor each are the step time"
                          if B * T < 65536 else None))
    if B == 32:     # the same step replayed from a hipGraph (paddlerec_amd/graph.py): the launch-bound shape
        tg = timeit_stream(lambda: m.train_step_graphed(hi, hc, ti, tc, label, mask, tis, tcs))
        print("DIN train step B=%d T=%d, hipGraph replay: %.3f ms  (%.1f k samples/s; eager %.3f ms)"
              % (B, T, tg, B / tg, t))
        record("configs[3]", "DIN train step replayed from a hipGraph (same launches), B %d, T %d" % (B, T), tg, fl, B,
               {"positions_per_s": B * T / tg * 1e3, "eager_ms": t}, hbm_bytes=by_step,
               bytes_formula="as the eager step above")
    del m
    torch.cuda.empty_cache()

# ---- sibling nets: xDeepFM (config.yaml: D 9, CIN 128-32, DNN 512-256-128) and DLRM (D 16, bot 512-256-64-16, top 512-256-2)
from paddlerec_amd.dlrm import DLRMLayer  # noqa: E402
from paddlerec_amd.xdeepfm import xDeepFMLayer  # noqa: E402

for B in (4096, 65536):
    ids = torch.randint(0, 1000001, (B, 26), device=DEV, generator=g)
    dense = torch.rand(B, 13, device=DEV, generator=g)
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.25).long()
    m = xDeepFMLayer(1000001, 9, 13, 26, [128, 32], [512, 256, 128], device=DEV)
    t = timeit(lambda: m.train_step(ids, dense, label, lr=1e-3), iters=3, warm=1)
    fl_cin = 2.0 * B * 9 * (39 * 39 * 128 + 39 * 128 * 32)
    fl_dnn = 2.0 * B * (351 * 512 + 512 * 256 + 256 * 128 + 128)
    print("xDeepFM train step B=%d: %.2f ms  (%.2f M samples/s; CIN + DNN GEMMs %.1f TF executed, fwd + 2x bwd)"
          % (B, t, B / t / 1e3, 3 * (fl_cin + fl_dnn) / t / 1e9))
    record("sibling net", "xDeepFM (CIN 128-32 over 39 fields x D 9, DNN 512-256-128) train step, B %d; the CIN's "
           "outer-product rows go through HBM (REC_CIN_CHUNK_MB scratch chunks)" % B, t, 3 * (fl_cin + fl_dnn), B)
    del m
    torch.cuda.empty_cache()
    m = DLRMLayer(13, [512, 256, 64, 16], 1000001, 16, [512, 256, 2], 26, device=DEV)
    t = timeit(lambda: m.train_step(ids, dense, label, lr=1e-3), iters=3, warm=1)
    fl = 2.0 * B * (13 * 512 + 512 * 256 + 256 * 64 + 64 * 16 + 367 * 512 + 512 * 256 + 256 * 2 + 27 * 27 * 16)
    print("DLRM train step B=%d: %.2f ms  (%.2f M samples/s; non-lazy Adam sweeps the 1M x 16 table every step)"
          % (B, t, B / t / 1e3))
    widths = 512 + 256 + 64 + 16 + 512 + 256 + 2
    by_dlrm = 6 * 1000001 * 16 * 4 + 7 * B * widths * 4 + 3 * B * 26 * 16 * 4
    record("sibling net", "DLRM (bot 512-256-64-16, top 512-256-2, BatchNorm after every layer, 27 x 27 dot "
           "interaction) train step with non-lazy Adam, B %d" % B, t, 3 * fl, B, hbm_bytes=by_dlrm,
           bytes_formula="non-lazy Adam sweeps the table 6 N D 4 B (N 1000001, D 16) + BatchNorm after every Linear: 7 passes "
                         "(3 forward, 4 backward) over B x %d activations x 4 B + lookup rows 3 B 26 D 4" % widths)
    del m
    torch.cuda.empty_cache()

for d in JSON:
    print(json.dumps(d))
