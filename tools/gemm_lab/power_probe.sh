#!/bin/bash
# clocks / power of the GPU while one GEMM shape runs in a loop (random vs constant operands): is the f32 GEMM clock-limited?
R=${GRAFT_REPO_ROOT:-$(pwd)}
for d in rand zeros; do
  echo "== data $d"
  python - "$d" <<'PY' &
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from paddlerec_amd import ops
d = sys.argv[1]
M, N, K = 65536, 400, 400
mk = (lambda *s: torch.rand(*s, device="cuda") - 0.5) if d == "rand" else (lambda *s: torch.zeros(*s, device="cuda"))
A, B, bias, C = mk(M, K), mk(K, N), mk(N), torch.empty(M, N, device="cuda")
ws = ops.Workspace("cuda")
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    for _ in range(200):
        ops.gemm(A, B, ws, epilogue="bias_relu", bias=bias, out=C)
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
print("   %d GEMMs in %.2f s: %.1f us each, %.1f TF" % (n, dt, dt / n * 1e6, 2.0 * M * N * K * n / dt / 1e12), flush=True)
PY
  pid=$!
  sleep 4
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power \(W\)|Average Graphics|Socket" | tr -s '\t ' ' ' | paste -sd'|' ; sleep 1; done
  wait $pid
done
