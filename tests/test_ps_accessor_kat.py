"""Known-answer tests of the PS accessor restatement (oracle/ps_ref.py), computed BY HAND from the published Paddle
source quoted in that file (SparseAdaGradSGDRule::UpdateValueWork, CtrCommonAccessor::Update / Shrink / Save,
MemorySparseTable::PushSparse) — VERDICT r02: "add a known-answer vector computed by hand for k duplicate pushes".
The expected numbers below are literals worked out on paper (shown in the comments), not produced by the code under
test.  The same vectors are pushed through the HIP kernels in tests/test_ps_gpu.py."""
import numpy as np
import pytest

from oracle import ps_ref

ACC = dict(lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0), initial_range=1e-2, embedx_threshold=1.0,
           nonclk_coeff=0.1, click_coeff=1.0, seed=7)
# slot layout, D = 3: [embed_w, embedx0, embedx1 | show click g2w g2x state delta unseen]
LAY = dict(embed_off=0, embedx_off=1, embedx_dim=2, stat_off=3)
S = LAY["stat_off"]


def kat_existing_value():
    rec = np.zeros((4, 10), np.float32)
    rec[2, :3] = [0.5, 0.1, -0.2]
    rec[2, S:S + 7] = [5, 1, 0.0, 1.0, 2, 0.3, 4]      # show click g2w g2x state delta unseen
    return rec


def test_existing_value_three_duplicate_occurrences():
    """k = 3 occurrences of the key in the batch: merged gradient (0.6 | 0.3, -0.9), pushed show 3, click 1.
    scale = 3:  embed  scaled = 0.2,  ratio = sqrt(3/3) = 1          w = 0.5 - 0.05*0.2       = 0.49
                                                                     g2sum_w = 0 + 0.2^2      = 0.04
                embedx scaled = (0.1, -0.3), ratio = sqrt(3/4) = 0.8660254
                       w0 = 0.1 - 0.05*0.1*0.8660254 = 0.09566987,  w1 = -0.2 + 0.05*0.3*0.8660254 = -0.18700962
                       g2sum_x = 1 + (0.01 + 0.09)/2 = 1.05
    show 5+3 = 8, click 1+1 = 2, delta_score 0.3 + (3-1)*0.1 + 1*1.0 = 1.5, unseen_days 0."""
    rec = kat_existing_value()
    ps_ref.push_rows(rec, LAY, [2], np.float32([0.6]), np.float32([[0.3, -0.9]]), [3], [1], ACC)
    np.testing.assert_allclose(rec[2, :3], [0.49, 0.09566987, -0.18700962], rtol=0, atol=2e-8)
    np.testing.assert_allclose(rec[2, S:S + 7], [8, 2, 0.04, 1.05, 2, 1.5, 0], rtol=0, atol=1e-7)
    assert not rec[[0, 1, 3]].any()


def test_k_duplicates_move_like_one_occurrence():
    """The point of the show scaling: a key hit k times with the same per-occurrence gradient g moves exactly as far
    as a key hit once with g (not k times as far)."""
    g_w, g_x = np.float32(0.25), np.float32([0.125, -0.5])
    one, four = kat_existing_value(), kat_existing_value()
    ps_ref.push_rows(one, LAY, [2], [g_w], [g_x], [1], [0], ACC)
    ps_ref.push_rows(four, LAY, [2], [4 * g_w], [4 * g_x], [4], [0], ACC)
    assert np.array_equal(one[2, :3], four[2, :3])
    assert np.array_equal(one[2, S + 2:S + 4], four[2, S + 2:S + 4])        # g2sums
    assert one[2, S] == 6 and four[2, S] == 9
    off = kat_existing_value()                                              # show_scale off: k times as far
    ps_ref.push_rows(off, LAY, [2], [4 * g_w], [4 * g_x], [4], [0], dict(ACC, show_scale=False))
    np.testing.assert_allclose(0.5 - off[2, 0], 4 * (0.5 - one[2, 0]), rtol=1e-5)   # differences of rounded floats


def test_new_key_then_embedx_creation_then_update():
    """threshold 1.0, nonclk 0.1, clk 1.
    push 1 (show 2, click 0, g_embed -0.4, g_embedx ignored): key created with embed_w = 0 (zero_init);
        scaled = -0.2, ratio 1: embed_w = 0 + 0.05*0.2 = 0.01, g2sum_w = 0.04; score (2-0)*0.1 = 0.2 < 1: no embedx,
        its gradient is dropped; delta_score = 0.2
    push 2 (show 1, click 1, g_embed 0.3): ratio = sqrt(3/3.04) = 0.99339927
        embed_w = 0.01 - 0.05*0.3*0.99339927 = -0.00490099, g2sum_w = 0.04 + 0.09 = 0.13
        counters show 3, click 1: score (3-1)*0.1 + 1 = 1.2 >= 1 -> embedx CREATED at the end of this push with its
        creation values, g2sum_x = 0, this push's embedx gradient dropped; delta_score = 0.2 + 1.0 = 1.2
    push 3 (show 1, click 0, g_embedx (0.2, 0.4)): ratio_x = sqrt(3/3) = 1
        embedx = init - 0.05*(0.2, 0.4),  g2sum_x = (0.04 + 0.16)/2 = 0.1"""
    rec = np.zeros((3, 10), np.float32)
    ps_ref.push_rows(rec, LAY, [1], np.float32([-0.4]), np.float32([[9.0, 9.0]]), [2], [0], ACC)
    np.testing.assert_allclose(rec[1, :3], [0.01, 0, 0], atol=1e-9)
    np.testing.assert_allclose(rec[1, S:S + 7], [2, 0, 0.04, 0, 1, 0.2, 0], atol=1e-8)
    ps_ref.push_rows(rec, LAY, [1], np.float32([0.3]), np.float32([[9.0, 9.0]]), [1], [1], ACC)
    init = [ps_ref.init_value(7, 1, 1, 1e-2), ps_ref.init_value(7, 1, 2, 1e-2)]
    assert all(abs(v) <= 1e-2 for v in init) and init[0] != init[1]
    np.testing.assert_allclose(rec[1, 0], -0.00490099, atol=2e-9)
    assert np.array_equal(rec[1, 1:3], np.float32(init))
    np.testing.assert_allclose(rec[1, S:S + 7], [3, 1, 0.13, 0, 2, 1.2, 0], atol=1e-7)
    ps_ref.push_rows(rec, LAY, [1], np.float32([0.0]), np.float32([[0.2, 0.4]]), [1], [0], ACC)
    np.testing.assert_allclose(rec[1, 1:3], np.float32(init) - np.float32([0.01, 0.02]), atol=2e-9)
    np.testing.assert_allclose(rec[1, S + 3], 0.1, atol=1e-8)
    assert rec[1, S] == 4 and rec[1, S + 4] == 2


def test_missing_key_reads_as_zeros_and_grad_scale_bounds():
    rec = np.zeros((3, 10), np.float32)
    assert not ps_ref.pull_value(rec, LAY, 1, ACC, 3).any()                 # PullSparse of a missing key
    # grad_scale = batch size 4: pushed gradient = 4 * 0.05 = 0.2, show 1 -> w = 0 - 0.05*0.2 = -0.01
    ps_ref.push_rows(rec, LAY, [1], np.float32([0.05]), np.float32([[0, 0]]), [1], [0], dict(ACC, grad_scale=4.0))
    np.testing.assert_allclose(rec[1, 0], -0.01, atol=1e-9)
    np.testing.assert_allclose(rec[1, S + 2], 0.04, atol=1e-8)
    # bounds: a huge gradient stops at the bound
    ps_ref.push_rows(rec, LAY, [1], np.float32([-1e6]), np.float32([[0, 0]]), [1], [0], dict(ACC, bounds=(-10, 0.25)))
    assert rec[1, 0] == np.float32(0.25)
    # embed_zero_init off: the key reads as, and is created with, its creation value
    acc = dict(ACC, embed_zero_init=False)
    v = ps_ref.init_value(7, 2, 0, 1e-2)
    assert ps_ref.pull_value(rec, LAY, 2, acc, 3)[0] == v
    ps_ref.push_rows(rec, LAY, [2], np.float32([0.0]), np.float32([[0, 0]]), [1], [0], acc)
    assert rec[2, 0] == v


def test_shrink_unseen_days_and_save_kinds():
    """Shrink: decay 0.98, delete_threshold 0.8, delete_after_unseen_days 30.
       row 0: show 10 click 0 -> score 9.8*0.1 = 0.98 >= 0.8, kept (show 9.8)
       row 1: show 5 click 0  -> 4.9*0.1 = 0.49 < 0.8, deleted
       row 2: show 100, unseen_days 31 > 30, deleted although its score is high
       row 3: missing key, untouched
    Save(param 1, delta): base_threshold 1.5, delta_threshold 0.25, keep_days 16:
       row 0: score 0.98 < 1.5 -> not saved.   Add row 4: show 20 click 1 (score 1.9+1 = 2.9), delta 0.3, unseen 2 ->
       saved, delta_score reset;  row 5: same but delta 0.1 -> not saved by a delta save, saved by a base save (param 2)
    Save(param 3): every existing row, unseen_days += 1."""
    rec = np.zeros((6, 10), np.float32)
    rec[0, S:S + 7] = [10, 0, 0, 0, 1, 0, 0]
    rec[1, S:S + 7] = [5, 0, 0, 0, 1, 0, 0]
    rec[2, S:S + 7] = [100, 0, 0, 0, 2, 0, 31]
    rec[4, S:S + 7] = [20, 1, 0, 0, 2, 0.3, 2]
    rec[5, S:S + 7] = [20, 1, 0, 0, 2, 0.1, 2]
    rec[:, 0] = 1.0
    n = ps_ref.shrink_rows(rec, LAY, ACC, 0.98, 0.8, 30.0)
    assert n == 2 and not rec[1].any() and not rec[2].any()
    np.testing.assert_allclose(rec[0, S], 9.8, rtol=1e-7)
    assert rec[3, 0] == 1.0 and not rec[3, S:S + 7].any()
    m1 = ps_ref.save_select(rec, LAY, ACC, 1)
    assert m1.tolist() == [False, False, False, False, True, False]
    assert rec[4, S + 5] == 0 and rec[5, S + 5] == np.float32(0.1)
    m2 = ps_ref.save_select(rec, LAY, ACC, 2)
    assert m2.tolist() == [False, False, False, False, True, True] and rec[5, S + 5] == 0
    m3 = ps_ref.save_select(rec, LAY, ACC, 3)
    assert m3.tolist() == [True, False, False, False, True, True]
    assert rec[0, S + 6] == 1 and rec[4, S + 6] == 3
    m0 = ps_ref.save_select(rec, LAY, ACC, 0)
    assert m0.tolist() == m3.tolist() and rec[0, S + 6] == 1


def test_float_division_of_the_float_pushed_gradient():
    """`double scaled_grad = grad[i] / scale;` divides a float by a float (round 3 divided in double and
    carried g x grad_scale in double).  A vector on which the two typings end on different floats:
      merged gradient g = 0.82f (= 0.819999992847...), grad_scale (batch size) 4, pushed show 6, embed_w 0.5, g2sum 0
      pushed = float(0.82f * 4)       = 3.2799999713897705            (exact in float)
      float division pushed / 6       = 0.5466666618982...  -> float 0.54666668176651      (the nearer neighbour)
      w = float(0.5 - 0.05f * 0.54666668176651 * 1)  = 0.47266665     [the double quotient gives 0.47266668]
      g2sum = float(0 + 0.54666668176651^2 / 1)      = 0.29884446     [the double quotient gives 0.29884443]"""
    rec = np.zeros((3, 10), np.float32)
    rec[1, 0] = 0.5
    rec[1, S:S + 7] = [6, 0, 0.0, 0.0, 1, 0.0, 0]                            # a key without embedx
    acc = dict(ACC, embedx_threshold=1e9, grad_scale=4.0)
    ps_ref.push_rows(rec, LAY, [1], np.float32([0.82]), np.float32([[0.0, 0.0]]), [6], [0], acc)
    assert rec[1, 0] == np.float32(0.47266665) and rec[1, 0] != np.float32(0.47266668)
    assert rec[1, S + 2] == np.float32(0.29884446) and rec[1, S + 2] != np.float32(0.29884443)
    # the same statement on the typed helper alone
    w = np.float32([0.5])
    g2 = ps_ref.update_value_work(w, np.float32(0), [np.float32(3.2799999713897705)], np.float32(6), np.float32(0.05),
                                  np.float32(3.0), np.float32(-10), np.float32(10))
    assert w[0] == np.float32(0.47266665) and g2 == np.float32(0.29884446)
