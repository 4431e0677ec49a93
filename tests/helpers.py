"""Shared test helpers: golden loading, seeded synthetic DeepFM problems, C-oracle wrappers."""
import ctypes as C
import os

import numpy as np

from conftest import GOLDEN


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def deepfm_params_from_golden(g):
    n = int(g["n_mlp"])
    return dict(W=g["W"], W1=g["W1"], dense_w=g["dense_w"], dense_w_one=g["dense_w_one"],
                mlp_w=[g["mlp_w%d" % i] for i in range(n)], mlp_b=[g["mlp_b%d" % i] for i in range(n)])


def make_deepfm_problem(B, S=26, Dn=13, D=16, N=5000, fc=(64, 32), seed=0, pad_frac=0.03,
                        zipf=False, tables=False, dtype=np.float32):
    """Synthetic batch + weights, SURVEY.md §8(d) recipe (uniform or Zipf ids, 3 % padding)."""
    rng = np.random.default_rng(seed)
    if zipf:
        ranks = np.minimum(rng.zipf(1.05, size=(B, S)), N - 1)
        perm = rng.permutation(N)
        ids = np.clip(perm[ranks], 1, N - 1).astype(np.int64)
    else:
        ids = rng.integers(1, N, size=(B, S), dtype=np.int64)
    ids[rng.random((B, S)) < pad_frac] = 0
    slot_offsets = (np.arange(S, dtype=np.int64) * N) if tables else None
    rows_total = N * S if tables else N
    dense = rng.random((B, Dn), dtype=np.float32)
    label = (rng.random((B, 1)) < 0.25).astype(np.int64)
    std = 0.1 / np.sqrt(D)
    tn = lambda *s: np.clip(rng.standard_normal(s) * std, -2 * std, 2 * std).astype(dtype)
    sizes = [(S + Dn) * D] + list(fc) + [1]
    params = dict(
        W=tn(rows_total, D), W1=tn(rows_total, 1), dense_w=tn(1, Dn, D), dense_w_one=tn(Dn),
        mlp_w=[(rng.standard_normal((sizes[i], sizes[i + 1])) / np.sqrt(sizes[i])).astype(dtype)
               for i in range(len(sizes) - 1)],
        mlp_b=[(rng.standard_normal(sizes[i + 1]) * 0.01).astype(dtype) for i in range(len(sizes) - 1)])
    return dict(ids=ids, dense=dense, label=label, params=params, slot_offsets=slot_offsets,
                N=rows_total, S=S, Dn=Dn, D=D, fc=list(fc))


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


def c_fm_fwd(lib, ids, dense, W, W1, dense_w, dense_w_one, pad=0, slot_off=None):
    B, S = ids.shape
    Dn = dense.shape[1]
    D = W.shape[1]
    y1 = np.empty(B, np.float32)
    y2 = np.empty(B, np.float32)
    feat = np.empty((B, S + Dn, D), np.float32)
    sum_emb = np.empty((B, D), np.float32)
    lib.oracle_fm_fwd.restype = None
    lib.oracle_fm_fwd(C.c_int64(B), C.c_int(S), C.c_int(Dn), C.c_int(D), fptr(ids), fptr(dense),
                      fptr(W), fptr(np.ascontiguousarray(W1.reshape(-1))),
                      fptr(np.ascontiguousarray(dense_w.reshape(Dn, D))), fptr(dense_w_one),
                      C.c_int64(pad), fptr(slot_off) if slot_off is not None else None,
                      fptr(y1), fptr(y2), fptr(feat), fptr(sum_emb))
    return y1, y2, feat, sum_emb


def c_fm_bwd(lib, S, dense, feat, sum_emb, d_feat, dy1, dy2):
    B, F, D = feat.shape
    Dn = F - S
    row_grad = np.empty((B * S, D), np.float32)
    row_grad1 = np.empty(B * S, np.float32)
    ddw = np.empty((Dn, D), np.float32)
    ddw1 = np.empty(Dn, np.float32)
    lib.oracle_fm_bwd.restype = None
    lib.oracle_fm_bwd(C.c_int64(B), C.c_int(S), C.c_int(Dn), C.c_int(D), fptr(dense), fptr(feat),
                      fptr(sum_emb), fptr(np.ascontiguousarray(d_feat)),
                      fptr(np.ascontiguousarray(dy1.reshape(-1))),
                      fptr(np.ascontiguousarray(dy2.reshape(-1))), fptr(row_grad), fptr(row_grad1),
                      fptr(ddw), fptr(ddw1))
    return row_grad, row_grad1, ddw, ddw1


def c_adam_rows(lib, uniq, seg_off, spos, row_grad, P, M, V, step, lr=1e-3, b1=0.9, b2=0.999,
                eps=1e-8):
    lib.oracle_adam_rows.restype = None
    D = P.shape[1]
    lib.oracle_adam_rows(C.c_int64(len(uniq)), C.c_int(D), fptr(uniq), fptr(seg_off), fptr(spos),
                         fptr(row_grad), fptr(P), fptr(M), fptr(V), C.c_float(lr), C.c_float(b1),
                         C.c_float(b2), C.c_float(eps), C.c_int64(step))


def deepfm_state_dict(p, n_mlp):
    sd = {"fm.embedding.weight": p["W"], "fm.embedding_one.weight": p["W1"],
          "fm.dense_w": p["dense_w"], "fm.dense_w_one": p["dense_w_one"]}
    for i in range(n_mlp):
        sd["dnn.linear_%d.weight" % i] = p["mlp_w"][i]
        sd["dnn.linear_%d.bias" % i] = p["mlp_b"][i]
    return sd


class OracleTrainer:
    """Unsharded DeepFM training on the NumPy oracle: fwd + bwd + lazy Adam (tables) + Adam (dense)."""

    def __init__(self, params, slot_offsets=None, lr=1e-2):
        from oracle import deepfm_ref as R
        self.R = R
        self.p = {k: (v.copy() if not isinstance(v, list) else [x.copy() for x in v])
                  for k, v in params.items()}
        self.so, self.lr, self.step = slot_offsets, lr, 0
        self.st = {k: np.zeros_like(params[t]) for k, t in
                   (("mW", "W"), ("vW", "W"), ("mW1", "W1"), ("vW1", "W1"))}
        self.dstate = {}

    def train_step(self, ids, dense, label):
        R, op, st = self.R, self.p, self.st
        self.step += 1
        o = R.deepfm_loss_and_grads(ids, dense, label, op, slot_offsets=self.so)
        uniq, merged, _ = R.merge_rows(o["rows"], o["row_valid"], o["row_grad"])
        R.adam_update_rows(op["W"], st["mW"], st["vW"], uniq, merged, self.step, lr=self.lr)
        uniq1, merged1, _ = R.merge_rows(o["rows"], o["row_valid"], o["row_grad1"])
        R.adam_update_rows(op["W1"], st["mW1"], st["vW1"], uniq1, merged1, self.step, lr=self.lr)
        pairs = [("dense_w", o["d_dense_w"]), ("dense_w_one", o["d_dense_w_one"])]
        for i in range(len(op["mlp_w"])):
            pairs += [(("mlp_w", i), o["mlp_dw"][i]), (("mlp_b", i), o["mlp_db"][i])]
        for key, gr in pairs:
            arr = op[key] if not isinstance(key, tuple) else op[key[0]][key[1]]
            mm, vv = self.dstate.setdefault(key, (np.zeros_like(arr), np.zeros_like(arr)))
            R.adam_update(arr, mm, vv, gr.reshape(arr.shape).astype(arr.dtype), self.step, lr=self.lr)
        return o["loss"], o["pred"]


class OracleDCNTrainer:
    """DCN-v2 training on the NumPy oracle: fwd + bwd + ClipGradByGlobalNorm + lazy Adam (table) +
    Adam (dense), as dcn_v2/dygraph_model.py:73-88 configures it (L2Decay 1e-7 omitted, see dcn_v2.py)."""

    def __init__(self, params, lr=1e-3, clip_norm=10.0, dropout=None, l2_dnn=0.0):
        """dropout = (p, seed): train-mode Dropout of the DNN tower with the engine's counter-based masks; l2_dnn:
        L2Decay on the DNN weights, appended after the clip ([EXT] Optimizer._apply_optimize order)."""
        from oracle import dcn_v2_ref as X
        from oracle import deepfm_ref as R
        self.X, self.R = X, R
        self.dropout, self.l2_dnn = dropout, float(l2_dnn)
        self.p = {k: v.copy() for k, v in params.items()}
        self.lr, self.clip, self.step = lr, clip_norm, 0
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}

    def train_step(self, ids, dense, label):
        X, R, p = self.X, self.R, self.p
        self.step += 1
        drop = None if self.dropout is None else (self.dropout[0], self.dropout[1], self.step)
        pred, saved = X.forward(ids, dense, p, return_saved=True, dropout=drop)
        loss = R.log_loss_mean(pred, label)
        t = label.astype(np.float32)
        eps = np.float32(1e-4)
        dpred = (-t / (pred + eps) + (1 - t) / (1 - pred + eps)) / np.float32(pred.shape[0])
        g = X.backward(ids, dense, p, saved, dpred)
        rows = ids.reshape(-1)
        uniq, merged, _ = R.merge_rows(rows, rows != 0, g["_row_grad"])
        scale = np.float32(1.0)
        if self.clip:
            ss = sum(float((np.asarray(v, np.float64) ** 2).sum()) for k, v in g.items()
                     if k not in ("_row_grad", "embedding.weight"))
            ss += float((merged.astype(np.float64) ** 2).sum())
            scale = np.float32(self.clip / max(np.sqrt(ss), self.clip))
        R.adam_update_rows(p["embedding.weight"], self.m["embedding.weight"], self.v["embedding.weight"],
                           uniq, merged * scale, self.step, lr=self.lr)
        for k, gv in g.items():
            if k in ("_row_grad", "embedding.weight"):
                continue
            gk = (np.asarray(gv).reshape(p[k].shape) * scale).astype(np.float32)
            if self.l2_dnn and k.startswith("DNN_.linear_") and k.endswith(".weight"):
                gk = (gk + np.float32(self.l2_dnn) * p[k]).astype(np.float32)
            R.adam_update(p[k], self.m[k], self.v[k], gk, self.step, lr=self.lr)
        return loss, pred, g


# ------------------------------------------------------------------------------------------------------------------
# The parity bar of the fp32 layers: 1e-5 relative to the TENSOR's scale.
def assert_close_scaled(got, want, rel=1e-5, err_msg=""):
    """|got - want| <= rel * (|want| + max|want|): elements that are sums of opposite-sign terms carry the fp32
    summation noise of the terms, not of the (cancelled) result."""
    got, want = np.asarray(got), np.asarray(want)
    np.testing.assert_allclose(got.reshape(want.shape), want, rtol=rel, atol=rel * float(np.abs(want).max()),
                               err_msg=err_msg)


def layer_moments(layer):
    """{dense parameter name: (m, v)} numpy views of a host mirror's flat Adam moments (paddlerec_amd.deepfm._FlatParams
    keeps every dense parameter, its gradient and both moments in flat buffers; .pm / .pv are the per-tensor views)."""
    d = layer.dense
    return {n: (d.pm[n].detach().cpu().numpy(), d.pv[n].detach().cpu().numpy()) for n in d.names}


def assert_moments_close(layer, want_m, want_v, names=None, rel=1e-5, rename=None):
    """Adam's moments after a few steps against the oracle trainer's: m is linear and v quadratic in the gradients, so
    both are held to the 1e-5-of-scale bar.  (Weights are NOT: a step is lr * m / (sqrt(v) + eps) — an entry whose
    gradient is ~eps-sized moves by lr * sign(noise), which turns fp32 summation-order noise into lr-sized differences;
    a wrong update rule shows up in the loss / prediction of the following steps, which the tests hold to 2e-5.)"""
    got = layer_moments(layer)
    n = 0
    for k, (gm, gv) in got.items():
        ok = (rename or {}).get(k, k)
        if names is not None and ok not in names and k not in names:
            continue
        if ok not in want_m:
            continue
        assert_close_scaled(gm, want_m[ok], rel, err_msg="m of " + k)
        assert_close_scaled(gv, want_v[ok], rel, err_msg="v of " + k)
        n += 1
    return n


def assert_close_floor(got, want64, want32, rel=1e-5, err_msg="", scale=None):
    """Against a float64 reference, with the fp32 noise floor MEASURED next to it: `want32` is the same computation
    carried out in float32 (the oracle itself); the kernel may be off by 1e-5 of the tensor's scale or by 4x what fp32
    round-off alone costs the oracle, whichever is larger — the floor is printed when it is the binding one."""
    got, want64, want32 = np.asarray(got, np.float64), np.asarray(want64, np.float64), np.asarray(want32, np.float64)
    scale = float(np.abs(want64).max()) if scale is None else float(scale)   # scale: of the terms the result cancels
    floor = float(np.abs(want32 - want64).max())
    bound = max(rel * scale, 4.0 * floor)
    err = float(np.abs(got.reshape(want64.shape) - want64).max())
    assert err <= bound, "%s max err %.3e > bound %.3e (1e-5 of scale %.3e, measured fp32 floor %.3e)" % (
        err_msg, err, bound, rel * scale, floor)


def assert_sibling_moments(layer, st, rel=1e-5):
    """Adam moments of the sibling nets (dnn / wide_deep / fm / xdeepfm mirrors) against their test-local oracle
    trainers, whose state is {key: (m, v)} with keys ("w", i) / ("b", i) for the MLP, "W" / "W1" for the tables and the
    parameter's own name otherwise.  -> number of tensors compared."""
    import re
    want_m, want_v = {}, {}
    for n in layer.dense.names:
        mm = re.search(r"linear_(\d+)\.(weight|bias)$", n)
        key = None
        if mm and (("w" if mm.group(2) == "weight" else "b"), int(mm.group(1))) in st:
            key = ("w" if mm.group(2) == "weight" else "b", int(mm.group(1)))
        else:
            for cand in (n, n.split(".", 1)[-1], n.replace(".", "_"), n.split(".", 1)[-1].replace(".", "_")):
                if cand in st:
                    key = cand
                    break
        if key is not None:
            want_m[n], want_v[n] = st[key]
    n = assert_moments_close(layer, want_m, want_v, rel=rel)
    sp = getattr(layer, "sparse_state", None) or {}
    for mk, vk, key in (("m", "v", "W"), ("m1", "v1", "W1")):
        if mk in sp and key in st:
            assert_close_scaled(sp[mk].detach().cpu().numpy(), st[key][0], rel, err_msg="m of " + key)
            assert_close_scaled(sp[vk].detach().cpu().numpy(), st[key][1], rel, err_msg="v of " + key)
            n += 1
    return n


def assert_adam_weights_close(got, want, lr, steps, rel=1e-5, frac=0.98, err_msg=""):
    """Weights after `steps` Adam steps against the oracle's.  Adam divides the gradient by its own magnitude, so an
    element whose gradient is ~eps-sized (fp32 noise decides its sign) can end a whole step (~lr) away on the two sides
    while every moment agrees to 1e-5 of its scale (assert_moments_close / assert_sibling_moments check those).  The
    statement here has two parts instead of one loose rtol:
      * at least `frac` of the elements are within `rel` of the tensor's scale — the stated bar;
      * NO element is further away than Adam can move it: steps * lr * max(1, (1 - b1) / sqrt(1 - b2)) on each side."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64).reshape(np.asarray(got).shape)
    err = np.abs(got - want)
    scale = max(float(np.abs(want).max()), 1e-30)
    tight = float(np.mean(err <= rel * scale))
    assert tight >= frac, "%s only %.4f of the elements within %.0e of the scale %.3e (max err %.3e)" % (
        err_msg, tight, rel, scale, float(err.max()))
    cap = 2.0 * steps * lr * 3.17
    assert float(err.max()) <= cap, "%s max err %.3e exceeds what %d Adam steps of lr %.3g can move (%.3e)" % (
        err_msg, float(err.max()), steps, lr, cap)
    return tight
