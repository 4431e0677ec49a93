#!/usr/bin/env python3
"""Where do the DeepFM GEMMs lose their 30 % against the MFMA peak?  Shape probes of C = relu(A B + b), N = 400:
rounds of tiles (M 65536 = 2.5 rounds of 512 resident 256x80 blocks, M 131072 = 5 whole rounds), K depth (25 vs 100
k-steps: prologue / epilogue share), for the LDS-DMA ring and the register-staged kernel (child processes)."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from paddlerec_amd import ops
    DEV = "cuda"
    g = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: torch.rand(*s, device=DEV, generator=g) - 0.5
    ws = ops.Workspace(DEV)
    out = {}
    for M, N, K in [(65536, 400, 400), (131072, 400, 400), (65536, 400, 1600), (131072, 400, 1600), (32768, 400, 400),
                    (65536, 80, 400), (65536, 400, 80)]:
        nset = max(1, min(3, int(3e8 // (M * (K + N) * 4)) + 1))
        sets = [(rnd(M, K), rnd(K, N), rnd(N), torch.empty(M, N, device=DEV)) for _ in range(nset)]

        def run(i):
            A, B, bias, C = sets[i % nset]
            ops.gemm(A, B, ws, epilogue="bias_relu", bias=bias, out=C)
        for i in range(4):
            run(i)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(8):
                run(i)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 8)
        t = sorted(ts)[2]
        out["%dx%dx%d" % (M, N, K)] = (t * 1e3, 2.0 * M * N * K / (t * 1e-3) / 1e12)
        del sets
    print(json.dumps(out))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        res = {}
        variants = [("reg-staged", {"REC_GEMM_GLDS": "0"}), ("LDS-DMA ring", {})]
        for a in sys.argv[1:]:          # extra variants: NAME:VAR=VAL,VAR=VAL
            n, kv = a.split(":")
            variants.append((n, dict(x.split("=") for x in kv.split(","))))
        for tag, env in variants:
            env = dict(os.environ, **env)
            r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, "FAILED", r.stderr[-800:])
                continue
            res[tag] = json.loads(line[-1])
        for case in next(iter(res.values())):
            print("%-20s" % case, "   ".join("%s: %7.1f us %6.1f TF" % (k, v[case][0], v[case][1]) for k, v in res.items()))
