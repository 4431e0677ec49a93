#!/usr/bin/env python3
"""DIN attention-pool kernel timing (forward and backward) on amazonElec shapes (din/config.yaml:38-44), with a float64
torch restatement of net.py:141-173 on the first samples as a sanity check (not the parity test: tests/test_din_gpu.py).

    python tools/din_bench.py [--cases 4096x512,4096x100,32x152] [--bwd] [--iters 5]
Kernel variants are chosen by the environment (REC_DIN_FWD_VARIANT, REC_DIN_FWD_GENERIC, REC_DIN_BWD_GENERIC)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def torch_ref(n, hi, hc, ti, tc, mask, tabs, aw, ab, d_out=None):
    """float64 forward (and gradient w.r.t. the gathered rows) of the first n samples."""
    hi, hc, ti, tc, mask = (x[:n] for x in (hi, hc, ti, tc, mask))
    h = torch.cat([tabs[0][hi], tabs[1][hc]], -1).double().requires_grad_(True)
    q = torch.cat([tabs[2][ti], tabs[3][tc]], -1).double().requires_grad_(True)
    x = torch.cat([h, q, h - q, h * q], -1)
    a1 = torch.sigmoid(x @ aw[0].double() + ab[0].double())
    a2 = torch.sigmoid(a1 @ aw[1].double() + ab[1].double())
    s = (a2 @ aw[2].double() + ab[2].double()).squeeze(-1)
    w = torch.softmax((s + mask.double()) * h.shape[-1] ** -0.5, -1)
    out = (w.unsqueeze(-1) * h).sum(1)
    if d_out is None:
        return out.detach(), w.detach()
    dh, dq = torch.autograd.grad(out, (h, q), d_out[:n].double())
    return out.detach(), w.detach(), dh, dq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="4096x512,4096x100,32x152")
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    g = torch.Generator(device=DEV).manual_seed(3)
    tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("REC_DIN"))
    for case in args.cases.split(","):
        B, T = (int(x) for x in case.split("x"))
        tabs = [torch.randn(n, 64, device=DEV, generator=g) * 0.3 for n in (63001, 801, 63001, 801)]
        hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
        hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
        ti = torch.randint(0, 63001, (B, 1), device=DEV, generator=g).expand(B, T).contiguous()
        tc = torch.randint(0, 801, (B, 1), device=DEV, generator=g).expand(B, T).contiguous()
        lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
        mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
        aw = [torch.randn(s, device=DEV, generator=g) * 0.2 for s in ((512, 80), (80, 40), (40, 1))]
        ab = [torch.randn(s, device=DEV, generator=g) * 0.1 for s in (80, 40, 1)]
        st = ops.new_status(DEV)
        saved = {}
        out, attw, _ = ops.din_attention_pool(hi, hc, ti, tc, mask, *tabs, aw, ab, st, saved=saved)
        n = min(B, 16)
        d_out = torch.randn(B, 128, device=DEV, generator=g)
        ref = torch_ref(n, hi, hc, ti, tc, mask, tabs, aw, ab, d_out if args.bwd else None)
        e_out = float((out[:n].double() - ref[0]).abs().max() / ref[0].abs().max())
        e_w = float((attw[:n].double() - ref[1]).abs().max())
        assert int(st.item()) == 0, "status flags %d" % int(st.item())
        t = timeit(lambda: ops.din_attention_pool(hi, hc, ti, tc, mask, *tabs, aw, ab, st), args.iters)
        t_s = timeit(lambda: ops.din_attention_pool(hi, hc, ti, tc, mask, *tabs, aw, ab, st, saved=saved), args.iters)
        fl = 2.0 * B * T * (512 * 80 + 80 * 40 + 40 + 128)
        print("[%s] fwd B=%d T=%d: %.3f ms  %.2f TF  %.1f M positions/s  (saving act1: %.3f ms)  (err out %.1e, weights %.1e)"
              % (tag, B, T, t, fl / t / 1e9, B * T / t / 1e3, t_s, e_out, e_w), flush=True)
        if args.bwd:
            dh, dq = ops.din_attention_pool_bwd(hi, hc, ti, tc, *tabs, aw, ab, attw, d_out, saved=saved)
            e_h = float((dh[:n].double() - ref[2]).abs().max() / ref[2].abs().max())
            e_q = float((dq[:n].double() - ref[3]).abs().max() / ref[3].abs().max())
            t = timeit(lambda: ops.din_attention_pool_bwd(hi, hc, ti, tc, *tabs, aw, ab, attw, d_out, saved=saved),
                       args.iters)
            on_saved = saved.get("act1") is not None and "REC_DIN_BWD_GENERIC" not in os.environ
            # executed: dz1 W1^T, a2 recompute, dz2 W2^T, dout . h  (+ the layer-1 recompute and the first pass over the
            # history when the activations were not saved)
            flb = 2.0 * B * T * (512 * 80 + 2 * 80 * 40 + 128 + (0 if on_saved else 512 * 80 + 128))
            print("[%s] bwd B=%d T=%d: %.3f ms  %.2f TF executed  (err dh %.1e, dq %.1e)"
                  % (tag, B, T, t, flb / t / 1e9, e_h, e_q), flush=True)


if __name__ == "__main__":
    main()
