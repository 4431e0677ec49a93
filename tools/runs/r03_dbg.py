import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from oracle import slot_dnn_ref as M
from paddlerec_amd import ops
import test_slot_dnn as t
D, stride, B, S, N = 9, 16, 130, 21, 4001
rng = np.random.default_rng(D * 1000 + B)
samples = t._random_problem(rng, B, S, N)
values, lod, base = M.csr_from_samples(samples, S)
Wfull = rng.standard_normal((N, stride)).astype(np.float32)
T = lambda a: torch.as_tensor(a).cuda()
mb = ops.MultislotBatch(T(values), T(lod), T(base))
out = ops.multislot_sumpool(mb, T(Wfull)[:, :D], N, 0, 0)[0].cpu().numpy()
want = M.multislot_sumpool(values, lod, base, Wfull[:, :D], 0, 0)[0]
bad = np.argwhere(np.abs(out - want).reshape(B, S, D).max(-1) > 1e-4)
print(len(bad), "bad cells")
for b, s in bad[:40]:
    ids = samples[b][s]
    print("b", b, "s", s, "ids", ids, "got0 %.4f want0 %.4f" % (out[b, s * D], want[b, s * D]),
          "prefix sums", [round(float(sum(Wfull[i, 0] for i in ids[:k] if i)), 4) for k in range(1, len(ids) + 1)])
