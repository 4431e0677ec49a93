"""The Paddle custom-operator shim (paddlerec_amd/paddle_ops/rec_paddle_ops.cc) and the reference-side patches that call
it (integration/*.patch) — SURVEY.md §8(b), VERDICT r04 item 1.

not gpu:
  * the shim builds against the stand-in paddle/extension.h, loads, registers the operators the patches call, and its
    InferShape / InferDtype functions describe the outputs (host functions: they run here);
  * the patches apply to the staged reference tree, touch nothing but the five net.py files, and add <= 15 lines each
    per edited block;
  * the reference's UNMODIFIED tools/trainer.py on the PATCHED net.py files produces the trajectory of the unpatched run
    (deepfm, dcn_v2 with CrossNetMix and with CrossNetV2, din) — operator stand-in backend (REC_COMPAT_KERNELS), so this
    checks the loader's autograd plumbing, the SelectedRows hand-over and the patches, not the kernels.
-m gpu:
  * every operator called THROUGH the shim (registered kernel function -> C-ABI -> HIP kernel) equals the same entry point
    called through paddlerec_amd.ops bit for bit, forward and gradient, and the oracle within 1e-5;
  * the patched trainer on the HIP kernels equals the oracle's trajectory and the unpatched GPU run."""
import os
import pickle
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

STAGED = os.path.join(REPO, "oracle", "_ref", "PaddleRec")
PATCHED = os.path.join(REPO, "oracle", "_ref", "PaddleRec_rec_ops")
needs_trees = pytest.mark.skipif(not (os.path.isdir(os.path.join(STAGED, "tools")) and
                                      os.path.isdir(os.path.join(PATCHED, "tools"))),
                                 reason="staged / patched reference trees not present (python __graft_entry__.py)")

EXPECTED = {
    "rec_deepfm_fm": (["Ids", "Dense", "W", "W1", "DenseW", "DenseWOne"], ["Y1", "Y2", "FeatEmb", "SumEmb", "Status"],
                      [("padding_idx", "int64_t")]),
    "rec_crossnet_v2_layer": (["X0", "Xl", "W", "B"], ["Out", "U"], []),
    "rec_crossnet_mix_layer": (["X0", "Xl", "U", "V", "C", "Bias", "GateW", "GateB"], ["Out", "T1", "T2", "Prob"], []),
    "rec_din_attention_pool": (["HistItem", "HistCat", "TgtItemSeq", "TgtCatSeq", "Mask", "WHistItem", "WHistCat",
                                "WTgtItemSeq", "WTgtCatSeq", "AttW1", "AttB1", "AttW2", "AttB2", "AttW3", "AttB3"],
                               ["Out", "AttWeight", "Act1", "Status"], []),
    # VERDICT r05 missing 2: the lookup / sum-pool and the GPU-PS pull / push pair
    "rec_multislot_sumpool": (["Values", "Offsets", "W"], ["Out", "Counts", "SegOfValue", "Rows", "Status"],
                              [("emb_dim", "int"), ("padding_idx", "int64_t"), ("key_mode", "int")]),
    "rec_ps_pull": (["Keys", "Rec", "ShowClick", "Anchor"], ["Out", "Rows", "Status"],
                    [("emb_dim", "int"), ("accessor", "std::vector<float>")]),
}


def _shim():
    sys.path.insert(0, os.path.join(REPO, "paddlerec_amd", "compat"))
    try:
        from paddlerec_amd import build
        build.build(verbose=False)              # incremental: librecengine.so first (the shim links it), then the shim
        from paddlerec_amd.compat.paddle.utils import cpp_extension as X
        return X, X.shim()
    finally:
        sys.path.pop(0)


def test_shim_registers_the_operators_the_patches_call():
    X, s = _shim()
    for name, (ins, outs, attrs) in EXPECTED.items():
        f, g = s.fwd[name], s.grad[name]
        assert f.inputs == ins and f.outputs == outs and f.attrs == attrs, name
        assert f.has_kernel and g.has_kernel and f.has_shape, name
        # a gradient op reads forward inputs / outputs / output gradients and writes gradients of forward inputs only
        for n in g.inputs:
            assert n in ins or n in outs or (n.endswith("@GRAD") and n[:-5] in outs), (name, n)
        for n in g.outputs:
            assert n.endswith("@GRAD") and n[:-5] in ins, (name, n)
        for table, ids in g.selected_rows.items():      # the SelectedRows rows: an id input, or the Rows the forward emits
            assert table in ins and (ids in ins or ids in outs) and table + "@GRAD" in g.outputs, (name, table)
    assert s.grad["rec_deepfm_fm"].selected_rows == {"W": "Ids", "W1": "Ids"}
    assert len(s.grad["rec_din_attention_pool"].selected_rows) == 4
    assert s.grad["rec_multislot_sumpool"].selected_rows == {"W": "Rows"}
    # the push is the gradient operator of the pull; the record table is updated in place and listed with the program
    assert s.fwd["rec_ps_pull"].ps_tables == ["Rec"] and s.grad["rec_ps_pull"].outputs == ["Anchor@GRAD"]
    assert s.grad["rec_ps_pull"].attrs == s.fwd["rec_ps_pull"].attrs and s.grad["rec_ps_pull"].has_shape
    # every operator a patch calls is registered
    called = set()
    for p in sorted(os.listdir(os.path.join(REPO, "integration"))):
        if p.endswith(".patch"):
            called |= set(re.findall(r"^\+.*rec_ops\.(rec_\w+)\(", open(os.path.join(REPO, "integration", p)).read(), re.M))
    assert called == set(EXPECTED)


def test_infer_shape_and_dtype_functions():
    import torch
    X, s = _shim()
    B, S, Dn, D, N = 6, 26, 13, 9, 50
    e = lambda *shape, dt=torch.float32: torch.empty(*shape, dtype=dt)
    ids = e(B, S, dt=torch.int64)
    got = s.infer(s.fwd["rec_deepfm_fm"], [ids, e(B, Dn), e(N, D), e(N, 1), e(1, Dn, D), e(Dn)], {"padding_idx": 0})
    assert got == [((B, 1), torch.float32), ((B, 1), torch.float32), ((B, S + Dn, D), torch.float32),
                   ((B, D), torch.float32), ((1,), torch.int32)]
    got = s.infer(s.grad["rec_deepfm_fm"], [ids, e(B, Dn), e(B, S + Dn, D), e(B, D), e(1, Dn, D), e(B, S + Dn, D), e(B, 1),
                                            e(B, 1)], {})
    assert [g[0] for g in got] == [(B * S, D), (B, 1), (1, Dn, D), (Dn,)]
    d = 40
    got = s.infer(s.fwd["rec_crossnet_v2_layer"], [e(B, d), e(B, d), e(d, d), e(d)], {})
    assert got == [((B, d), torch.float32)] * 2
    E, r = 4, 8
    got = s.infer(s.fwd["rec_crossnet_mix_layer"], [e(B, d), e(B, d), e(E, d, r), e(E, d, r), e(E, r, r), e(d, 1), e(d, E),
                                                    e(E)], {})
    assert [g[0] for g in got] == [(B, d), (B, E * r), (B, E * r), (B, E)]
    T = 7
    tid = e(B, T, dt=torch.int64)
    args = [tid, tid, tid, tid, tid, e(100, 64), e(30, 64), e(100, 64), e(30, 64), e(512, 80), e(80), e(80, 40), e(40),
            e(40, 1), e(1)]
    got = s.infer(s.fwd["rec_din_attention_pool"], args, {})
    assert got[0][0] == (B, 128) and got[1][0] == (B, T) and got[3][0] == (1,)
    assert got[2][0] in ((B, T, 80), (1,))          # layer-1 activations are saved for the reference's shape only
    with pytest.raises(TypeError):
        s.infer(s.fwd["rec_crossnet_v2_layer"], [e(B, d)], {})
    nnz, Sm = 40, 5
    a = {"emb_dim": 9, "padding_idx": 0, "key_mode": 1}
    got = s.infer(s.fwd["rec_multislot_sumpool"], [e(nnz, 1, dt=torch.int64), e(Sm, B + 1, dt=torch.int64), e(N, 16)], a)
    assert got == [((B, Sm * 9), torch.float32), ((B, Sm), torch.int32), ((nnz,), torch.int32), ((nnz,), torch.int64),
                   ((1,), torch.int32)]
    got = s.infer(s.grad["rec_multislot_sumpool"], [e(nnz, dt=torch.int32), e(B, Sm * 9)], {"emb_dim": 9})
    assert got[0][0] == (nnz, 9)
    acc = [0.05, 3.0, -10.0, 10.0, 1e-4] * 2 + [10.0, 0.1, 1.0, 2025.0]
    got = s.infer(s.fwd["rec_ps_pull"], [ids, e(N, 16), e(B, 2), e(1)], {"emb_dim": 9, "accessor": acc})
    assert got == [((B, S, 9), torch.float32), ((B * S,), torch.int64), ((1,), torch.int32)]
    got = s.infer(s.grad["rec_ps_pull"], [e(B * S, dt=torch.int64), e(N, 16), e(B, 2), e(B, S, 9)],
                  {"emb_dim": 9, "accessor": acc})
    assert got[0][0] == (1,)


def test_kernel_errors_surface_as_python_exceptions():
    """PD_CHECK inside a kernel function -> std::runtime_error -> caught at the C boundary -> RuntimeError here (no C++
    exception crosses into the interpreter)."""
    import torch
    X, s = _shim()
    B, d = 4, 8
    with pytest.raises(RuntimeError, match="CrossNetV2 shapes"):
        s.run(s.fwd["rec_crossnet_v2_layer"], [torch.zeros(B, d), torch.zeros(B, d), torch.zeros(d, d + 1), torch.zeros(d)], {})
    with pytest.raises(RuntimeError, match="wrong dtype"):
        s.run(s.fwd["rec_deepfm_fm"], [torch.zeros(B, 3), torch.zeros(B, 2), torch.zeros(5, 4), torch.zeros(5, 1),
                                       torch.zeros(1, 2, 4), torch.zeros(2)], {"padding_idx": 0})
    # a list-of-float attribute reaches the kernel function as std::vector<float> (13 instead of 14 entries: PD_CHECK)
    with pytest.raises(RuntimeError, match="accessor: 14 floats"):
        s.run(s.grad["rec_ps_pull"], [torch.zeros(6, dtype=torch.int64), torch.zeros(5, 16), torch.zeros(3, 2),
                                      torch.zeros(3, 2, 9)], {"emb_dim": 9, "accessor": [0.0] * 13})


@needs_trees
def test_patches_touch_only_the_net_files_and_are_small():
    changed = []
    for root, _, files in os.walk(STAGED):
        if "__pycache__" in root or "output_model" in root:
            continue
        for f in files:
            rel = os.path.relpath(os.path.join(root, f), STAGED)
            other = os.path.join(PATCHED, rel)
            if f.endswith(".py") and os.path.exists(other) and open(os.path.join(root, f), "rb").read() != open(other, "rb").read():
                changed.append(rel)
    assert sorted(changed) == ["models/rank/dcn_v2/net.py", "models/rank/deepfm/net.py", "models/rank/din/net.py",
                               "models/rank/dnn/net.py", "models/rank/slot_dnn/net.py"]
    for p in sorted(os.listdir(os.path.join(REPO, "integration"))):
        if not p.endswith(".patch"):
            continue
        text = open(os.path.join(REPO, "integration", p)).read()
        hunks = re.split(r"^@@.*@@.*$", text, flags=re.M)[1:]
        for h in hunks:
            added = [ln for ln in h.split("\n") if ln.startswith("+")]
            assert len(added) <= 15, (p, len(added))
    # the drivers are byte-identical in both trees
    for rel in ("tools/trainer.py", "tools/utils/utils_single.py", "models/rank/deepfm/dygraph_model.py",
                "models/rank/deepfm/config.yaml", "models/rank/din/dygraph_model.py", "models/rank/dcn_v2/dygraph_model.py",
                "tools/static_gpubox_trainer.py", "models/rank/dnn/static_model.py", "models/rank/dnn/config_gpubox.yaml",
                "models/rank/slot_dnn/static_model.py"):
        assert open(os.path.join(STAGED, rel), "rb").read() == open(os.path.join(PATCHED, rel), "rb").read(), rel


def _env(gpu):
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONDONTWRITEBYTECODE="1", REC_COMPAT_SEED="11")
    if gpu:
        env.pop("REC_COMPAT_KERNELS", None)
    else:
        env["REC_COMPAT_KERNELS"] = "cpu_kernels"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "tests"), REPO, env.get("PYTHONPATH", "")])
    return env


CASES = {      # model -> (config, extra -o overrides)
    "deepfm": ("models/rank/deepfm/config.yaml", ["runner.train_batch_size=8"]),
    "dcn_mix": ("models/rank/dcn_v2/config.yaml", ["runner.train_batch_size=16"]),
    "dcn_v2": ("models/rank/dcn_v2/config.yaml", ["runner.train_batch_size=16", "hyper_parameters.use_low_rank_mixture=False",
                                                  "hyper_parameters.cross_num=3"]),
    "din": ("models/rank/din/config.yaml", []),
}


def _launch(tree, case, out_dir, gpu):
    cfg, extra = CASES[case]
    cmd = [sys.executable, "-m", "paddlerec_amd.run_reference", os.path.join(tree, "tools", "trainer.py"), "-m",
           os.path.join(tree, cfg), "-o", "runner.epochs=1", "runner.print_interval=1",
           "runner.model_save_path=%s" % out_dir, "runner.use_gpu=%s" % ("True" if gpu else "False")] + extra
    return subprocess.Popen(cmd, cwd=tree, env=_env(gpu), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _losses(log):
    return [float(m.group(1)) for m in re.finditer(r"loss:\s*([0-9.eE+-]+),", log)]


def _patched_equals_unpatched(tmp_path, cases, gpu, tol):
    """Both trees, same seed (REC_COMPAT_SEED: identical initial parameters — the patches change no constructor), same
    data: printed losses and every saved parameter agree.  DCN-v2's Dropout(0.5) draws from torch's generator in both
    runs; the patched forward consumes the generator identically (the operators draw nothing)."""
    procs = {}
    for c in cases:
        for tag, tree in (("plain", STAGED), ("patched", PATCHED)):
            procs[(c, tag)] = _launch(tree, c, tmp_path / ("%s_%s" % (c, tag)), gpu)
    logs = {}
    for k, p in procs.items():
        out, _ = p.communicate(timeout=1500)
        assert p.returncode == 0, (k, out[-3000:])
        logs[k] = out
    for c in cases:
        la, lb = _losses(logs[(c, "plain")]), _losses(logs[(c, "patched")])
        assert len(la) >= 3 and len(la) == len(lb), (c, len(la), len(lb))
        np.testing.assert_allclose(lb, la, rtol=tol, atol=tol, err_msg=c)
        a = pickle.load(open(tmp_path / ("%s_plain" % c) / "0" / "rec.pdparams", "rb"))
        b = pickle.load(open(tmp_path / ("%s_patched" % c) / "0" / "rec.pdparams", "rb"))
        assert set(a) == set(b)
        for k in a:
            scale = max(1.0, float(np.abs(a[k]).max()))
            np.testing.assert_allclose(b[k], a[k], rtol=0, atol=tol * scale, err_msg="%s %s" % (c, k))


@needs_trees
def test_patched_net_files_train_like_the_unpatched_ones_cpu_backend(tmp_path):
    _patched_equals_unpatched(tmp_path, ["deepfm", "dcn_mix", "dcn_v2", "din"], gpu=False, tol=2e-6)


# ------------------------------------------------------------------------------------------------------------ GPU
GPU_SCRIPT = r"""
import os, sys
import numpy as np, torch
REPO = %(repo)r
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "paddlerec_amd", "compat")); sys.path.insert(0, os.path.join(REPO, "tests"))
import paddle
paddle.set_device("gpu")
from paddle.utils.cpp_extension import load
from paddlerec_amd import ops
rec_ops = load(name="rec_ops", sources=[os.path.join(REPO, "paddlerec_amd", "paddle_ops", "rec_paddle_ops.cc")])
dev = "cuda:0"
g = torch.Generator().manual_seed(5)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
eq = lambda a, b, what: (_ for _ in ()).throw(AssertionError(what)) if not torch.equal(a, b) else None

# ---- rec_deepfm_fm: shim == ops.py bit for bit (same C-ABI entry point, same inputs), oracle within 1e-5
from oracle import deepfm_ref as R
B, S, Dn, D, N = 300, 26, 13, 9, 1000
ids = torch.randint(0, N, (B, S), generator=g).to(dev); ids[::7, 3] = 0
dense = torch.rand(B, Dn, generator=g).to(dev)
W = torch.nn.Parameter(rnd(N, D, scale=0.05)); W1 = torch.nn.Parameter(rnd(N, 1, scale=0.05))
W._is_embedding = W1._is_embedding = True
dw = torch.nn.Parameter(rnd(1, Dn, D, scale=0.05)); dw1 = torch.nn.Parameter(rnd(Dn, scale=0.05))
y1, y2, feat, sum_emb, status = rec_ops.rec_deepfm_fm(ids, dense, W, W1, dw, dw1, padding_idx=0)
o = ops.deepfm_fm_fwd(ids, dense, W.detach(), W1.detach(), dw.detach(), dw1.detach(), 0, None, None)
eq(y1.reshape(-1), o[0].reshape(-1), "y1"); eq(y2.reshape(-1), o[1].reshape(-1), "y2"); eq(feat, o[2], "feat")
assert int(status.item()) == 0 and not status.requires_grad
ry1, ry2, rfeat = R.fm_forward(ids.cpu().numpy(), dense.cpu().numpy(), W1.detach().cpu().numpy(), W.detach().cpu().numpy(),
                               dw1.detach().cpu().numpy(), dw.detach().cpu().numpy(), 0, None)
np.testing.assert_allclose(y2.detach().cpu().numpy().reshape(-1), np.asarray(ry2).reshape(-1), rtol=1e-5, atol=1e-6)
np.testing.assert_array_equal(feat.detach().cpu().numpy()[:, :S], np.asarray(rfeat)[:, :S])       # gathered rows: bit-exact
gy1, gy2, gf = rnd(B, 1), rnd(B, 1), rnd(B, S + Dn, D)
(y1 * gy1).sum().add((y2 * gy2).sum()).add((feat * gf).sum()).backward()
ws = ops.Workspace(dev)
rg, ddw, ddw1 = ops.deepfm_fm_bwd(dense, o[2], o[3], gf, gy1.reshape(-1), gy2.reshape(-1), S, ws, dense_w=dw.detach())
(sid, val, pad, div), = W._sparse_grads
eq(val, rg, "W rows-form gradient"); eq(sid, ids.reshape(-1), "SelectedRows rows"); assert pad == 0 and div == 1
(sid1, val1, pad1, div1), = W1._sparse_grads
eq(val1.reshape(-1), gy1.reshape(-1), "W1 gradient"); assert div1 == S
eq(dw.grad.reshape(-1), ddw.reshape(-1), "d dense_w"); eq(dw1.grad, ddw1, "d dense_w_one")
assert W.grad is None and W1.grad is None
print("deepfm_fm ok")

# ---- rec_crossnet_v2_layer / rec_crossnet_mix_layer
B, d = 256, 120
x0 = rnd(B, d).requires_grad_(True); xl = rnd(B, d).requires_grad_(True)
Wc = torch.nn.Parameter(rnd(d, d, scale=0.1)); bc = torch.nn.Parameter(rnd(d, scale=0.1))
out, u = rec_ops.rec_crossnet_v2_layer(x0, xl, Wc, bc)
ref = xl.detach().double() + x0.detach().double() * (xl.detach().double() @ Wc.detach().double() + bc.detach().double())
np.testing.assert_allclose(out.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
o_out = ops.crossnet_v2_layer_fwd(x0.detach(), xl.detach(), Wc.detach(), bc.detach(), ws)
eq(out, o_out if isinstance(o_out, torch.Tensor) else o_out[0], "crossnet_v2 fwd")
gd = rnd(B, d)
(out * gd).sum().backward()
x0d, xld, Wd, bd = (t.detach().double().requires_grad_(True) for t in (x0, xl, Wc, bc))
((xld + x0d * (xld @ Wd + bd)) * gd.double()).sum().backward()
for a, b_, n in ((x0.grad, x0d.grad, "dx0"), (xl.grad, xld.grad, "dxl"), (Wc.grad, Wd.grad, "dW"), (bc.grad, bd.grad, "db")):
    np.testing.assert_allclose(a.cpu().numpy(), b_.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(b_.abs().max()), err_msg=n)
print("crossnet_v2_layer ok")

E, r = 4, 16
U, V = torch.nn.Parameter(rnd(E, d, r, scale=0.1)), torch.nn.Parameter(rnd(E, d, r, scale=0.1))
Cm = torch.nn.Parameter(rnd(E, r, r, scale=0.3)); bias = torch.nn.Parameter(rnd(d, 1, scale=0.1))
gw, gb = torch.nn.Parameter(rnd(d, E, scale=0.1)), torch.nn.Parameter(rnd(E, scale=0.1))
x0 = rnd(B, d).requires_grad_(True); xl = rnd(B, d).requires_grad_(True)
out = rec_ops.rec_crossnet_mix_layer(x0, xl, U, V, Cm, bias, gw, gb)[0]
(out * gd).sum().backward()
dd = lambda t: t.detach().double().requires_grad_(True)
x0d, xld, Ud, Vd, Cd, bsd, gwd, gbd = map(dd, (x0, xl, U, V, Cm, bias, gw, gb))
p = torch.softmax(xld @ gwd + gbd, dim=1)
acc = xld
for e in range(E):
    acc = acc + p[:, e:e + 1] * x0d * (torch.tanh(torch.tanh(xld @ Vd[e]) @ Cd[e].t()) @ Ud[e].t() + bsd.reshape(1, -1))
np.testing.assert_allclose(out.detach().cpu().numpy(), acc.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
(acc * gd.double()).sum().backward()
for a, b_, n in ((x0.grad, x0d.grad, "dx0"), (xl.grad, xld.grad, "dxl"), (U.grad, Ud.grad, "gU"), (V.grad, Vd.grad, "gV"),
                 (Cm.grad, Cd.grad, "gC"), (bias.grad, bsd.grad, "gbias"), (gw.grad, gwd.grad, "ggw"), (gb.grad, gbd.grad, "ggb")):
    np.testing.assert_allclose(a.cpu().numpy(), b_.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(b_.abs().max()), err_msg=n)
print("crossnet_mix_layer ok")

# ---- rec_din_attention_pool: shim == ops.py bit for bit, forward and the four SelectedRows values
B, T, Ei, Ec, H1, H2, NI, NC = 48, 37, 64, 64, 80, 40, 500, 60
E_ = Ei + Ec
hi = torch.randint(0, NI, (B, T), generator=g).to(dev); hc = torch.randint(0, NC, (B, T), generator=g).to(dev)
ti = torch.randint(0, NI, (B, 1), generator=g).expand(B, T).contiguous().to(dev)
tc = torch.randint(0, NC, (B, 1), generator=g).expand(B, T).contiguous().to(dev)
lens = torch.randint(1, T + 1, (B,), generator=g)
mask = torch.where(torch.arange(T)[None, :] < lens[:, None], 0, int(-1e9)).to(torch.int64).to(dev)
tabs = [torch.nn.Parameter(rnd(n_, e_, scale=0.3)) for n_, e_ in ((NI, Ei), (NC, Ec), (NI, Ei), (NC, Ec))]
for t_ in tabs: t_._is_embedding = True
aw = [rnd(4 * E_, H1, scale=0.05), rnd(H1, H2, scale=0.1), rnd(H2, 1, scale=0.1)]
ab = [rnd(H1, scale=0.1), rnd(H2, scale=0.1), rnd(1, scale=0.1)]
out, att, act1, st = rec_ops.rec_din_attention_pool(hi, hc, ti, tc, mask, *tabs, aw[0], ab[0], aw[1], ab[1], aw[2], ab[2])
saved = {}
o_out, o_att, _ = ops.din_attention_pool(hi, hc, ti, tc, mask, *[t_.detach() for t_ in tabs], aw, ab, saved=saved, ws=ws)
eq(out, o_out, "din out"); eq(att, o_att, "din att weights")
gd = rnd(B, E_)
(out * gd).sum().backward()
dh, dq = ops.din_attention_pool_bwd(hi, hc, ti, tc, *[t_.detach() for t_ in tabs], aw, ab, o_att, gd, saved=saved)
dh, dq = dh.reshape(B * T, E_), dq.reshape(B * T, E_)
want = [dh[:, :Ei], dh[:, Ei:], dq[:, :Ei], dq[:, Ei:]]
for t_, w_, idt in zip(tabs, want, (hi, hc, ti, tc)):
    (sid, val, pad, div), = t_._sparse_grads
    eq(val, w_.contiguous(), "din rows-form gradient"); eq(sid, idt.reshape(-1), "din rows"); assert pad is None and div == 1
from oracle import din_ref
h = np.concatenate([tabs[0].detach().cpu().numpy()[hi.cpu().numpy()], tabs[1].detach().cpu().numpy()[hc.cpu().numpy()]], 2)
q = np.concatenate([tabs[2].detach().cpu().numpy()[ti.cpu().numpy()], tabs[3].detach().cpu().numpy()[tc.cpu().numpy()]], 2)
r_out = din_ref.attention_pool(h, q, mask.cpu().numpy().astype(np.float32), [a.cpu().numpy() for a in aw], [a.cpu().numpy() for a in ab])
np.testing.assert_allclose(out.detach().cpu().numpy(), r_out, rtol=1e-5, atol=1e-5)
print("din_attention_pool ok")
# ---- rec_multislot_sumpool: shim == ops.py bit for bit (forward: pooled sums, counts, segments, rows; rows-form gradient)
from oracle import slot_dnn_ref as SM
B, S, D, NT = 37, 12, 9, 5003
rng = np.random.default_rng(3)
vals, offs = [], np.zeros((S, B + 1), np.int64)
for s_ in range(S):
    for b_ in range(B):
        k_ = 0 if rng.random() < 0.2 else int(rng.integers(1, 6))
        v_ = rng.integers(1, 2 ** 63, size=k_, dtype=np.int64)
        v_[rng.random(k_) < 0.1] = 0                                    # explicit padding ids inside a segment
        vals.extend(v_.tolist()); offs[s_, b_ + 1] = len(vals)
    if s_ + 1 < S: offs[s_ + 1, 0] = len(vals)
offs[:, 0] = np.concatenate([[0], offs[:-1, -1]])
values = torch.as_tensor(np.asarray(vals, np.int64)).to(dev); offsets = torch.as_tensor(offs).to(dev)
rec_t = torch.zeros(NT, 16, device=dev); rec_t[:, :D] = rnd(NT, D, scale=0.3)
Wp = torch.nn.Parameter(rec_t)
out, counts, seg, rows, st = rec_ops.rec_multislot_sumpool(values, offsets, Wp, emb_dim=D, padding_idx=0, key_mode=1)
mb = ops.MultislotBatch(values, offsets, torch.zeros(S + 1, dtype=torch.int64, device=dev))
o_out, o_cnt, o_seg, o_rows, _ = ops.multislot_sumpool(mb, rec_t[:, :D], NT, 0, 1)
eq(out, o_out, "multislot out"); eq(counts, o_cnt, "multislot counts"); eq(seg, o_seg[:values.numel()], "seg"); eq(rows, o_rows[:values.numel()], "rows")
r_out, r_cnt, r_seg, r_rows = SM.multislot_sumpool(np.asarray(vals, np.int64), offs, np.zeros(S + 1, np.int64), rec_t[:, :D].cpu().numpy(), 0, 1, NT)
assert np.array_equal(counts.cpu().numpy(), r_cnt) and np.array_equal(rows.cpu().numpy(), r_rows) and np.array_equal(seg.cpu().numpy(), r_seg)
np.testing.assert_allclose(out.detach().cpu().numpy(), r_out, rtol=1e-5, atol=1e-6)
gd = rnd(B, S * D)
(out * gd).sum().backward()
(sid, val, pad, div), = Wp._sparse_grads
eq(sid, rows, "SelectedRows rows = the operator's Rows"); assert pad == 0 and div == 1 and Wp.grad is None
eq(val, gd.reshape(B * S, D)[seg.long()], "rows-form gradient")
eq(val, ops.multislot_sumpool_bwd(seg, gd, S, D), "rec_multislot_sumpool_bwd through ops.py")
print("multislot_sumpool ok")

# ---- rec_ps_pull: pull == feasign_rows + emb_gather; its gradient operator == ids_group + ps_push_rows on a twin table
B, S, D, NT = 64, 26, 9, 20011
acc = [0.05, 3.0, -10.0, 10.0, 1e-4] * 2 + [10.0, 0.1, 1.0, 2025.0]
keys = torch.randint(1, 400, (B, S), generator=g).to(dev); keys[::5, 2] = 0
twin = ops.PsTable(NT, D, dev, kind="slot")
recv = torch.zeros(NT, 16, device=dev)
label = (torch.rand(B, generator=g) < 0.3).to(torch.int64).to(dev)
show_click = torch.stack([torch.ones(B, device=dev), label.float()], 1).contiguous()
anchor = torch.nn.Parameter(torch.zeros(1, device=dev))
for step_ in range(3):                                                  # step 0 creates keys, later ones train / extend them
    emb, prow, st = rec_ops.rec_ps_pull(keys, recv, show_click, anchor, emb_dim=D, accessor=acc)
    rows_ = ops.feasign_rows(keys.reshape(-1).contiguous(), NT)
    w_, _ = ops.emb_gather(rows_, twin.W, None, ops.new_status(dev))
    eq(prow, rows_, "pull rows"); eq(emb.reshape(B * S, D), w_, "pulled values")
    gd = rnd(B, S, D, scale=0.01)
    (emb * gd).sum().backward()
    grp = ops.IdGroups(B * S, dev)
    ops.ids_group(rows_, NT, 0, ws, None, ops.new_status(dev), grp)
    twin.accessor.grad_scale = float(B)
    ops.ps_push_rows(twin, grp, gd.reshape(B * S, D).contiguous(), S, click=label)
    eq(recv, twin.rec, "record table after push " + str(step_))
assert float(recv[:, D].sum()) == 3 * int((keys != 0).sum()) and float(recv[:, D + 4].max()) >= 1
print("ps_pull / push ok")
print("custom ops through the shim: ALL OK")
"""


@pytest.mark.gpu
def test_custom_ops_through_the_shim_gpu(engine_lib):
    r = subprocess.run([sys.executable, "-c", GPU_SCRIPT % dict(repo=REPO)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0 and "ALL OK" in r.stdout, (r.stdout + r.stderr)[-4000:]


@needs_trees
@pytest.mark.gpu
def test_reference_trainer_on_patched_deepfm_matches_oracle_gpu(tmp_path, engine_lib):
    """tools/trainer.py (unmodified) + the patched deepfm/net.py on the HIP kernels through the shim == the oracle's
    trajectory from the same initial parameters (the check the unpatched entry-point test applies)."""
    from test_reference_entrypoint import _run_trainer_and_check
    _run_trainer_and_check(tmp_path, gpu=True, tree=PATCHED)


@needs_trees
@pytest.mark.gpu
def test_patched_net_files_train_like_the_unpatched_ones_gpu(tmp_path, engine_lib):
    # fp32 on the device: the fused kernels sum in a different order than the op-by-op graph
    _patched_equals_unpatched(tmp_path, ["deepfm", "dcn_v2", "dcn_mix", "din"], gpu=True, tol=2e-5)
