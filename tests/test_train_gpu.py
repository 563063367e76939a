"""GPU parity tests for the training path (BPR fwd/bwd, dense Adam, CSR SpMM, MF / LightGCN steps)
through the C ABI, against the oracle and the reference's golden vectors.  Floating point:
1e-5 relative on losses and embedding norms (north_star), looser elementwise where Adam's
normalisation amplifies rounding."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as orc
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(DEV)


def _bpr_gpu(U, V, ui, pi, ni, reg):
    from coldrec_amd import ops
    tU, tV = t(U, torch.float32), t(V, torch.float32)
    gU, gV = torch.zeros_like(tU), torch.zeros_like(tV)
    loss = ops.bpr_fwd_bwd(tU, tV, tV, t(ui, torch.int32), t(pi, torch.int32), t(ni, torch.int32), reg, gU, gV, gV)
    torch.cuda.synchronize()
    return loss.cpu().numpy(), gU.cpu().numpy(), gV.cpu().numpy()


@pytest.mark.parametrize("d,B,rows", [(4, 1, 3), (8, 5, 4), (16, 64, 40), (64, 1000, 300), (128, 4096, 5000),
                                      (200, 513, 100), (256, 2048, 900)])
def test_bpr_fwd_bwd_vs_oracle(d, B, rows):
    rng = np.random.default_rng(d * 7 + B)
    U = (rng.standard_normal((rows, d)) * 0.3).astype(np.float32)
    V = (rng.standard_normal((rows + 11, d)) * 0.3).astype(np.float32)
    ui, pi, ni = rng.integers(0, rows, B), rng.integers(0, rows + 11, B), rng.integers(0, rows + 11, B)
    reg = 0.01
    loss, gU, gV = _bpr_gpu(U, V, ui, pi, ni, reg)
    bpr, l2, wU, wV, _ = orc.bpr_l2_fwd_bwd(U, V, ui, pi, ni, reg)
    np.testing.assert_allclose(loss[0], bpr, rtol=1e-5)
    np.testing.assert_allclose(loss[1], l2, rtol=1e-5)
    sc = max(np.abs(wU).max(), np.abs(wV).max())
    np.testing.assert_allclose(gU, wU, rtol=1e-4, atol=2e-6 * sc)
    np.testing.assert_allclose(gV, wV, rtol=1e-4, atol=2e-6 * sc)


def _plan_views(plan):
    nu, ni, L = int(plan[0]), int(plan[1]), int(plan[2])
    urow, uptr, ulist = plan[3:3 + L], plan[3 + L:4 + 2 * L], plan[4 + 2 * L:4 + 3 * L]
    o = 4 + 3 * L
    irow, iptr, ilist = plan[o:o + 2 * L], plan[o + 2 * L:o + 4 * L + 1], plan[o + 4 * L + 1:o + 6 * L + 1]
    ho = 3 + (3 * L + 1) + (6 * L + 1)                       # heavy rows: [count, slots ascending...]
    heavy = plan[ho + 1: ho + 1 + int(plan[ho])]
    cnt = np.concatenate([np.diff(uptr[:nu + 1]), np.diff(iptr[:ni + 1])])
    from coldrec_amd import _lib
    T = int(_lib.lib().crh_bpr_heavy_threshold())
    assert np.array_equal(heavy, np.nonzero(cnt > T)[0]), "heavy list != rows with more than T entries"
    return dict(urow=urow[:nu], uptr=uptr[:nu + 1], ulist=ulist[:uptr[nu]], irow=irow[:ni], iptr=iptr[:ni + 1],
                ilist=ilist[:iptr[ni]], heavy=heavy)


@pytest.mark.parametrize("B,L", [(1, 1), (37, 64), (4096, 4096)])
def test_plan_backward_is_deterministic_and_exact(B, L):
    """Reverse-index ("plan") backward: host and device builders agree, gradients equal the oracle,
    two runs are bit-identical (no atomics), rows not touched stay zero."""
    from coldrec_amd import ops
    rng = np.random.default_rng(B)
    rows, d = 300, 64
    U = (rng.standard_normal((rows, d)) * 0.3).astype(np.float32)
    V = (rng.standard_normal((rows + 50, d)) * 0.3).astype(np.float32)
    ui = rng.integers(0, rows, B).astype(np.int32)
    pi = rng.zipf(1.5, B).clip(1, rows + 50).astype(np.int32) - 1           # heavy duplicates
    ni = rng.integers(0, rows + 50, B).astype(np.int32)
    host = ops.build_plans(ui, pi, ni, L)[0]
    devp = ops.build_plans_device(t(ui), t(pi), t(ni), L)
    devt = ops.build_plans_device(t(ui), t(pi), t(ni), L, lds_max_batch=1)      # the multi-block builder on the same batch
    for k, v in _plan_views(host).items():
        assert np.array_equal(v, _plan_views(devp[0].cpu().numpy())[k]), k
        assert np.array_equal(v, _plan_views(devt[0].cpu().numpy())[k]), k
    pv = _plan_views(host)
    assert np.array_equal(pv["urow"], np.unique(ui)) and len(pv["ulist"]) == B and len(pv["ilist"]) == 2 * B
    tU, tV = t(U), t(V)
    outs = []
    for _ in range(2):
        gU, gV = torch.zeros_like(tU), torch.zeros_like(tV)
        loss = ops.bpr_fwd_bwd(tU, tV, tV, t(ui), t(pi), t(ni), 0.01, gU, gV, gV, plan=devp[0])
        outs.append((loss.clone(), gU, gV))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    bpr, l2, wU, wV, _ = orc.bpr_l2_fwd_bwd(U, V, ui, pi, ni, 0.01)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), [bpr, l2], rtol=1e-5)
    sc = max(np.abs(wU).max(), np.abs(wV).max())
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), wU, rtol=1e-4, atol=2e-6 * sc)
    np.testing.assert_allclose(outs[0][2].cpu().numpy(), wV, rtol=1e-4, atol=2e-6 * sc)
    untouched = np.setdiff1d(np.arange(rows), ui)
    assert (outs[0][1].cpu().numpy()[untouched] == 0).all()


@pytest.mark.parametrize("n_rec,L,id_max,forced", [(10_000, 4096, 700, True), (20_000, 8193, 5_000, False),
                                                   (150_000, 65_536, 11_000_000, False), (70_000, 20_000, 300, False),
                                                   (9, 4, 50, True)])
def test_large_plan_builder_equals_host_builder(n_rec, L, id_max, forced):
    """crh_bpr_plan_build_large (batches beyond one workgroup's LDS: chunk sorts + merge-path passes + segment emission,
    several workgroups per batch) against crh_bpr_plan_build_host on whole epochs with a short last batch: every view of
    every plan equal, heavy lists included; `forced` sends LDS-sized batches through it as well; two runs bit-identical;
    the plan backward built on it equals the atomics-free oracle gradients."""
    from coldrec_amd import ops
    rng = np.random.default_rng(n_rec + L)
    u, p, n = (rng.integers(0, id_max, n_rec).astype(np.int32) for _ in range(3))
    u[:3], p[:3], n[:3] = id_max - 1, id_max - 1, id_max - 1
    p[100:min(400, n_rec)] = 5                                                  # a heavy row
    dev = ops.build_plans_device(t(u), t(p), t(n), L, lds_max_batch=1 if forced else 8192)
    dev2 = ops.build_plans_device(t(u), t(p), t(n), L, lds_max_batch=1 if forced else 8192)
    assert torch.equal(dev, dev2)
    dev = dev.cpu().numpy()
    host = ops.build_plans(u, p, n, L)
    assert dev.shape == host.shape
    for b in range(host.shape[0]):
        hv, dv = _plan_views(host[b]), _plan_views(dev[b])
        for k, v in hv.items():
            assert np.array_equal(v, dv[k]), (b, k)
    if L <= 8192:                                                               # same bits as the LDS builder where both apply
        lds = ops.build_plans_device(t(u), t(p), t(n), L).cpu().numpy()
        for b in range(host.shape[0]):
            for k, v in _plan_views(lds[b]).items():
                assert np.array_equal(v, _plan_views(dev[b])[k]), (b, k)


@pytest.mark.parametrize("id_max", [700, 131_071, 131_073, 30_000_000])     # 32-bit sort keys below 2^17 rows, else 64-bit
def test_plan_kernel_whole_epoch_with_short_last_batch(id_max):
    from coldrec_amd import ops
    rng = np.random.default_rng(5)
    n_rec, L = 10_000, 4096
    u, p, n = (rng.integers(0, id_max, n_rec).astype(np.int32) for _ in range(3))
    u[:3], p[:3], n[:3] = id_max - 1, id_max - 1, id_max - 1                   # the boundary ids are present
    p[100:400] = 5                                                              # a heavy row
    dev = ops.build_plans_device(t(u), t(p), t(n), L).cpu().numpy()
    host = ops.build_plans(u, p, n, L)
    from coldrec_amd import _lib
    T = int(_lib.lib().crh_bpr_heavy_threshold())
    assert dev.shape == host.shape == (3, 9 * L + 5 + 1 + (3 * L // T + 2))   # + [n_heavy, slots...]
    for b in range(3):
        for k, v in _plan_views(host[b]).items():
            assert np.array_equal(v, _plan_views(dev[b])[k]), (b, k)
    assert _plan_views(dev[2])["uptr"][-1] == n_rec - 2 * L


@pytest.mark.parametrize("case", ["rand", "reg", "sat"])
def test_bpr_golden_g2_gathered_tensors(case):
    """bpr_loss / l2_reg_loss on already gathered (B,d) tensors: identity indices (NULL)."""
    from coldrec_amd import ops
    g = load_golden("g2_loss.npz")
    u, p, n = (t(g[f"{case}_{k}"]) for k in "upn")
    gu, gp, gn = torch.zeros_like(u), torch.zeros_like(p), torch.zeros_like(n)
    loss = ops.bpr_fwd_bwd(u, p, n, None, None, None, float(g[f"{case}_reg"]), gu, gp, gn).cpu().numpy()
    np.testing.assert_allclose(loss[0], g[f"{case}_bpr"], rtol=1e-5)
    np.testing.assert_allclose(loss[1], g[f"{case}_l2"], rtol=1e-5)
    sc = np.abs(g[f"{case}_gu"]).max()
    for got, k in ((gu, "gu"), (gp, "gp"), (gn, "gn")):
        np.testing.assert_allclose(got.cpu().numpy(), g[f"{case}_{k}"], rtol=1e-4, atol=2e-6 * sc)


def test_bpr_golden_g2_duplicate_rows():
    g = load_golden("g2_loss.npz")
    loss, gU, gV = _bpr_gpu(g["dup_U"], g["dup_V"], g["dup_ui"], g["dup_pi"], g["dup_ni"], float(g["dup_reg"]))
    np.testing.assert_allclose(loss.sum(), g["dup_loss"], rtol=1e-5)
    np.testing.assert_allclose(gU, g["dup_gU"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(gV, g["dup_gV"], rtol=1e-4, atol=1e-7)


def test_forward_only_and_zero_norm():
    from coldrec_amd import ops
    u = torch.zeros((8, 16), device=DEV)
    loss = ops.bpr_fwd_bwd(u, u, u, None, None, None, 0.1).cpu().numpy()
    np.testing.assert_allclose(loss[0], -np.log(1e-5 + 0.5), rtol=1e-6)
    assert loss[1] == 0.0
    g = torch.zeros_like(u)
    ops.bpr_fwd_bwd(u, u, u, None, None, None, 0.1, g, g, g)
    assert torch.isfinite(g).all() and (g == 0).all()      # grad of |0|_F is 0, as autograd returns


@pytest.mark.parametrize("n", [4, 1000, 6040 * 128 + 3706 * 128])
def test_adam_dense_vs_oracle(n):
    from coldrec_amd import ops
    rng = np.random.default_rng(n)
    p = rng.standard_normal(n).astype(np.float32) * 0.05
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    tp, tm, tv = t(p), t(m), t(v)
    for step in range(1, 6):
        g = (rng.standard_normal(n) * 1e-3).astype(np.float32)
        g[rng.random(n) < 0.6] = 0.0                    # rows without gradient still move (dense Adam)
        tg = t(g)
        ops.adam_dense(tp, tg, tm, tv, step, lr=1e-3)
        p, m, v = orc.adam_dense(p, g, m, v, step, lr=1e-3)
        assert (tg == 0).all()                          # zero_grad fused
    np.testing.assert_allclose(tp.cpu().numpy(), p, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(tm.cpu().numpy(), m, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(tv.cpu().numpy(), v, rtol=1e-5, atol=1e-13)


def test_adam_two_tensors_equals_two_calls():
    from coldrec_amd import ops
    rng = np.random.default_rng(1)
    a = [t(rng.standard_normal(400).astype(np.float32)) for _ in range(4)]
    b = [t(rng.standard_normal(1200).astype(np.float32)) for _ in range(4)]
    for x in (a, b):
        x[3].abs_()
    a2, b2 = [x.clone() for x in a], [x.clone() for x in b]
    ops.adam_dense(*a, 3, second=b, zero_grad=False)
    ops.adam_dense(*a2, 3, zero_grad=False)
    ops.adam_dense(*b2, 3, zero_grad=False)
    for x, y in zip(a + b, a2 + b2):
        assert torch.equal(x, y)


@pytest.mark.parametrize("d", [16, 64])
def test_mf_training_golden_g3(d):
    """50 optimiser steps on the reference's own triples: per-step loss and table norms within 1e-5."""
    from coldrec_amd.train import MFEngine
    g = load_golden("g3_mf.npz")
    eng = MFEngine(g[f"d{d}_U0"], g[f"d{d}_V0"], float(g["lr"]), float(g["reg"]), DEV)
    off = np.concatenate([[0], np.cumsum(g["sizes"])])
    tu, ti, tj = t(g["u"], torch.int32), t(g["i"], torch.int32), t(g["j"], torch.int32)
    from coldrec_amd import ops
    for s in range(50):
        sl = slice(int(off[s]), int(off[s + 1]))
        # odd steps through the deterministic plan path, even steps through the atomic path
        plan = ops.build_plans_device(tu[sl], ti[sl], tj[sl], int(off[s + 1] - off[s]))[0] if s % 2 else None
        eng.step(tu[sl], ti[sl], tj[sl], plan)
        np.testing.assert_allclose(eng.last_loss(), g[f"d{d}_loss"][s], rtol=1e-5)
        if s + 1 in (1, 10, 50):
            for got, want in ((eng.user_emb, g[f"d{d}_U_step{s+1}"]), (eng.item_emb, g[f"d{d}_V_step{s+1}"])):
                got = got.cpu().numpy()
                np.testing.assert_allclose(np.linalg.norm(got), np.linalg.norm(want), rtol=1e-5)
                assert np.abs(got - want).max() < 2e-4 * np.abs(want).max()


def _graph():
    g4 = load_golden("g4_graph.npz")
    return g4["indptr"], g4["indices"].astype(np.int32), g4["data"]


@pytest.mark.parametrize("d", [4, 32, 128, 200])
def test_spmm_bit_exact_vs_oracle(d):
    from coldrec_amd import ops
    rowptr, col, val = _graph()
    n = len(rowptr) - 1
    rng = np.random.default_rng(d)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Z = rng.standard_normal((n, d)).astype(np.float32)
    tX, tZ = t(X), t(Z)
    Y, A = torch.empty_like(tX), torch.empty_like(tX)
    ops.spmm_csr(t(rowptr), t(col), t(val), tX, y=Y, acc_in=tZ, s_in=0.5, acc_out=A, s_out=0.25)
    want = orc.spmm(rowptr, col, val, X)
    np.testing.assert_array_equal(Y.cpu().numpy(), want)            # same fmaf order as the oracle
    np.testing.assert_allclose(A.cpu().numpy(), (Z * np.float32(0.5) + want) * np.float32(0.25), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("giant", [1024, 200, 0])
def test_spmm_segment_schedule_on_skewed_graph(monkeypatch, giant):
    """Zipf catalogue: rows above the segment length are given to a workgroup each (lane groups split the list, partial
    sums combined in a fixed order), the longest first at the head of the grid; rows above `giant` edges are cut into 2
    (4 above 4 * giant) column ranges with a workgroup each (giant = 200 exercises both cuts, 0 switches them off); the
    other rows are work items ordered by descending length."""
    from coldrec_amd import ops
    monkeypatch.setattr(ops.SpmmSchedule, "GIANT", giant)
    rng = np.random.default_rng(11)
    n_u, n_i, d = 3000, 500, 64
    w = 1.0 / np.arange(1, n_i + 1) ** 1.1
    items = rng.choice(n_i, 120_000, p=w / w.sum())
    key = np.unique(rng.integers(0, n_u, 120_000) * n_i + items)
    rowptr, col, val = orc.norm_adj_csr(key // n_i, key % n_i, n_u, n_i)
    deg = np.diff(rowptr)
    sched = ops.SpmmSchedule(rowptr, DEV)
    T = sched.seg                                                                 # 64, or 256 for dense graphs
    assert T in (64, 256)
    assert deg.max() > 1500 and sched.n_seg == len(deg)
    order, slot = sched.t[0].cpu().numpy(), sched.t[2].cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(len(deg)))                   # every row exactly once
    n_light = int((deg <= T).sum())
    assert (slot[:n_light] == -1).all() and (slot[n_light:] >= 0).all() and (deg[order[n_light:]] > T).all()
    assert (np.diff(deg[order[:n_light]]) <= 0).all()                            # light rows by descending length
    # heavy workgroups: longest rows first, n_sub consecutive entries (sub = 0 .. n_sub - 1) per row
    hrow, hcut = sched.t[3].cpu().numpy(), sched.t[5].cpu().numpy()
    n_sub, sub = hcut & 255, hcut >> 8
    want_sub = np.where(deg[hrow] > 4 * giant, 4, np.where(deg[hrow] > giant, 2, 1)) if giant else np.ones(len(hrow), int)
    assert np.array_equal(n_sub, want_sub) and (np.diff(deg[hrow]) <= 0).all()
    assert len(hrow) == int(want_sub[sub == 0].sum()) and set(hrow.tolist()) == set(np.nonzero(deg > T)[0].tolist())
    starts = np.nonzero(sub == 0)[0]
    for s0 in starts:
        assert (hrow[s0:s0 + n_sub[s0]] == hrow[s0]).all() and np.array_equal(sub[s0:s0 + n_sub[s0]], np.arange(n_sub[s0]))
    if giant == 200:
        assert (n_sub == 4).any() and (n_sub == 2).any() and (n_sub == 1).any()
    X = rng.standard_normal((n_u + n_i, d)).astype(np.float32)
    Z = rng.standard_normal((n_u + n_i, d)).astype(np.float32)
    tX, tZ = t(X), t(Z)
    Y0, Y1, A1 = torch.empty_like(tX), torch.empty_like(tX), torch.empty_like(tX)
    ops.spmm_csr(t(rowptr), t(col), t(val), tX, y=Y0)                              # row per lane group
    ops.spmm_csr(t(rowptr), t(col), t(val), tX, y=Y1, acc_in=tZ, s_in=2.0, acc_out=A1, s_out=0.5, sched=sched)
    want = orc.spmm(rowptr, col, val, X)
    np.testing.assert_array_equal(Y0.cpu().numpy(), want)
    one = deg <= T
    np.testing.assert_array_equal(Y1.cpu().numpy()[one], want[one])                # one lane group: same chain
    np.testing.assert_allclose(Y1.cpu().numpy()[~one], want[~one], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(A1.cpu().numpy(), (Z * 2 + want) * 0.5, rtol=1e-5, atol=1e-6)
    Y2 = torch.empty_like(tX)
    ops.spmm_csr(t(rowptr), t(col), t(val), tX, y=Y2, sched=sched)
    assert torch.equal(Y1, Y2)                                                      # deterministic
    # narrow tables: fewer lanes per lane group than column cuts (d = 8: two lanes) -- every cut falls back to the whole
    # slice and the duplicate workgroups write identical rows
    Xn = np.ascontiguousarray(X[:, :8])
    Yn = torch.empty((n_u + n_i, 8), dtype=torch.float32, device=DEV)
    ops.spmm_csr(t(rowptr), t(col), t(val), t(Xn), y=Yn, sched=sched)
    wn = orc.spmm(rowptr, col, val, Xn)
    np.testing.assert_array_equal(Yn.cpu().numpy()[one], wn[one])
    np.testing.assert_allclose(Yn.cpu().numpy()[~one], wn[~one], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,d", [(3000, 32), (16000, 128), (3000, 64), (3000, 200)])
def test_spmm_record_stream_path_equals_descriptor_path_and_oracle(monkeypatch, n, d):
    """Round 3: with the edge arrays at hand the schedule lays the light rows out as a record stream
    (crh_spmm_sched::slab: header + edges in whole units of G pairs, buckets by unit count, arithmetic record addresses)
    for launches whose lane groups have G = 8 lanes -- d = 32 unsliced, d = 128 in four XCD-pinned column slices (the
    CiteULike-sized LightGCN); other widths (d = 64: G = 16, d = 200: a padded group of 64) keep the descriptor path.
    Same bits as the descriptor path and as the oracle's edge-order chain on every light row -- rows of 0, G - 2, G - 1,
    G, 2G - 1 and 64 edges included -- with the fused layer sum and with the optimiser epilogue."""
    from coldrec_amd import _lib, ops
    rng = np.random.default_rng(d)
    want_deg = np.concatenate([[0, 1, 6, 7, 8, 9, 15, 16, 63, 64, 65, 300, 1500], rng.integers(0, 25, n - 13)])
    rowptr = np.concatenate([[0], np.cumsum(want_deg)]).astype(np.int64)
    G = int(_lib.lib().crh_spmm_lane_group(n, d, int(rowptr[-1])))
    assert G == {32: 8, 128: 8, 64: 16, 200: 64}[d]
    col = np.concatenate([np.sort(rng.choice(n, k, replace=False)) for k in want_deg]).astype(np.int32)
    val = rng.standard_normal(len(col)).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Z = rng.standard_normal((n, d)).astype(np.float32)
    tX, tZ, rp, cl, vl = t(X), t(Z), t(rowptr), t(col), t(val)
    plain = ops.SpmmSchedule(rowptr, DEV)                               # no edge arrays: descriptor path
    slab = ops.SpmmSchedule(rowptr, DEV, col=col, val=val)
    assert not plain.for_launch(n, d).slab
    if G == 8:
        assert slab.for_launch(n, d).slab and slab.for_launch(n, d).slab_lanes == 8
    else:
        assert not slab.for_launch(n, d).slab
    outs = []
    for sc in (plain, slab):
        Y, A = torch.empty_like(tX), torch.empty_like(tX)
        ops.spmm_csr(rp, cl, vl, tX, y=Y, acc_in=tZ, s_in=0.5, acc_out=A, s_out=0.25, sched=sc)
        P, M, V = tX.clone(), torch.zeros_like(tX), torch.zeros_like(tX)
        Zc = tZ.clone()
        ops.spmm_csr_adam(rp, cl, vl, tX, Zc, 1.0, None, 0.25, sc, P, M, V, 3, lr=1e-2, zero_acc_in=True)
        outs.append((Y, A, P, M, V, Zc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    want = orc.spmm(rowptr, col, val, X)
    light = want_deg <= plain.seg
    np.testing.assert_array_equal(outs[1][0].cpu().numpy()[light], want[light])
    np.testing.assert_allclose(outs[1][0].cpu().numpy()[~light], want[~light], rtol=1e-5, atol=2e-4)   # sums of up to 1500 N(0,1) products
    # a schedule is bound to the (col, val) its record stream was built from.  A launch that brings OTHER values
    # (a rescaled matrix over a reused schedule) must not mix the stream's baked-in light rows with the launch's heavy rows:
    # it takes the descriptor path and every row follows the launch's arrays
    vl2 = (vl * 2.0).contiguous()
    assert slab.for_launch(n, d, cl, vl).slab == (slab.for_launch(n, d).slab if G == 8 else None)
    assert not slab.for_launch(n, d, cl, vl2).slab
    Ys = torch.empty_like(tX)
    ops.spmm_csr(rp, cl, vl2, tX, y=Ys, sched=slab)
    want2 = orc.spmm(rowptr, col, (val * np.float32(2.0)).astype(np.float32), X)
    np.testing.assert_array_equal(Ys.cpu().numpy()[light], want2[light])
    np.testing.assert_allclose(Ys.cpu().numpy()[~light], want2[~light], rtol=1e-5, atol=4e-4)
    vl.mul_(1.0)                                                        # an in-place write bumps the version: re-checked, same sums
    assert bool(slab.for_launch(n, d, cl, vl).slab) == (G == 8)
    # the same hole with the schedule built FROM DEVICE TENSORS (LGCNEngine.from_device, xl_spmm_probe): rescaling
    # ``val`` in place after the stream was baked keeps pointer and length, so only the version counter / the bake-time
    # checksums can tell -- the launch must take the descriptor path and every row must follow the live values
    cl_d, vl_d = cl.clone(), vl.clone()
    dev_sched = ops.SpmmSchedule(rowptr, DEV, col=cl_d, val=vl_d)
    if G == 8:
        assert dev_sched.for_launch(n, d, cl_d, vl_d).slab                 # untouched since the bake: pointer fast path
        vl_d.mul_(3.0)
        assert not dev_sched.for_launch(n, d, cl_d, vl_d).slab             # same pointers, other contents
        Yd = torch.empty_like(tX)
        ops.spmm_csr(rp, cl_d, vl_d, tX, y=Yd, sched=dev_sched)
        want3 = orc.spmm(rowptr, col, (val * np.float32(3.0)).astype(np.float32), X)
        np.testing.assert_array_equal(Yd.cpu().numpy()[light], want3[light])
        np.testing.assert_allclose(Yd.cpu().numpy()[~light], want3[~light], rtol=1e-5, atol=6e-4)
        vl_d.mul_(1.0 / 3.0)                                               # (x 3 / 3 is not the identity in fp32 for every value:
        same = torch.equal(vl_d, vl)                                       #  whatever it gave, the stream is used iff the bits are back)
        assert bool(dev_sched.for_launch(n, d, cl_d, vl_d).slab) == same
        # fresh value tensors every step are recycled at ONE address -- a verdict must never be remembered by address.
        # A fresh tensor with other contents right after a matching one (same address, version 0) takes the descriptor path
        for step in range(20):
            fresh = vl_d.clone() if step % 2 == 0 else (vl_d * 2.0)
            assert bool(dev_sched.for_launch(n, d, cl_d, fresh).slab) == (same and step % 2 == 0), step
            del fresh
        perm = torch.arange(cl_d.numel() - 1, -1, -1, device=cl_d.device)    # the same multiset of edges in another order
        assert not dev_sched.for_launch(n, d, cl_d[perm].contiguous(), vl_d[perm].contiguous()).slab
    # a schedule built for another struct layout is refused by name, not decoded
    slab.c.version = 3
    with pytest.raises(RuntimeError, match="version"):
        ops.spmm_csr(rp, cl, vl, tX, y=Ys, sched=slab)
    slab.c.version = _lib.SPMM_SCHED_VERSION
    monkeypatch.setattr(ops.SpmmSchedule, "GIANT", 0)                   # and under another heavy-row layout
    again = ops.SpmmSchedule(rowptr, DEV, col=col, val=val)
    Y2 = torch.empty_like(tX)
    ops.spmm_csr(rp, cl, vl, tX, y=Y2, sched=again)
    np.testing.assert_array_equal(Y2.cpu().numpy()[light], want[light])


def test_lgcn_forward_golden_g5_and_training():
    from coldrec_amd.train import LGCNEngine
    g5 = load_golden("g5_lgcn.npz")
    rowptr, col, val = _graph()
    for L in (1, 2, 3):
        eng = LGCNEngine(g5["U0"], g5["V0"], rowptr, col, val, L, 1e-3, 1e-4, DEV)
        uo, io = eng.forward()
        np.testing.assert_allclose(uo.cpu().numpy(), g5[f"L{L}_user_out"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(io.cpu().numpy(), g5[f"L{L}_item_out"], rtol=1e-5, atol=1e-7)
    eng = LGCNEngine(g5["U0"], g5["V0"], rowptr, col, val, 3, float(g5["lr"]), float(g5["reg"]), DEV)
    eng.keep_grad = True                     # dE0 is otherwise consumed inside the last SpMM's epilogue
    off = np.concatenate([[0], np.cumsum(g5["train_sizes"])])
    tu, ti, tj = (t(g5[k], torch.int32) for k in ("train_u", "train_i", "train_j"))
    for s in range(20):
        sl = slice(int(off[s]), int(off[s + 1]))
        eng.step(tu[sl], ti[sl], tj[sl])
        if s == 0:
            sc = np.abs(g5["train_gU_step1"]).max()
            np.testing.assert_allclose(eng.G[: eng.user_num].cpu().numpy(), g5["train_gU_step1"], rtol=1e-4, atol=2e-6 * sc)
            np.testing.assert_allclose(eng.G[eng.user_num:].cpu().numpy(), g5["train_gV_step1"], rtol=1e-4, atol=2e-6 * sc)
        np.testing.assert_allclose(eng.last_loss(), g5["train_loss"][s], rtol=1e-5)
    for got, want in ((eng.user_emb, g5["train_U_end"]), (eng.item_emb, g5["train_V_end"])):
        np.testing.assert_allclose(np.linalg.norm(got.cpu().numpy()), np.linalg.norm(want), rtol=1e-5)


def test_full_size_mf_and_lgcn_steps_vs_oracle():
    """BASELINE configs 2 and 3 at their real sizes (S-ML 6040x3706 d=128 B=4096; S-CUL 5551x16980,
    L=3, d=128): one full step against the numpy/C oracle, plus linearity of the propagation."""
    from coldrec_amd.train import MFEngine, LGCNEngine
    rng = np.random.default_rng(0)
    U, I, d, B = 6040, 3706, 128, 4096
    U0 = (rng.uniform(-1, 1, (U, d)) * (6 / (U + d)) ** 0.5).astype(np.float32)
    V0 = (rng.uniform(-1, 1, (I, d)) * (6 / (I + d)) ** 0.5).astype(np.float32)
    ui, pi, ni = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    eng = MFEngine(U0, V0, 1e-3, 1e-4, DEV)
    eng.step(t(ui, torch.int32), t(pi, torch.int32), t(ni, torch.int32))
    bpr, l2, gU, gV, _ = orc.bpr_l2_fwd_bwd(U0, V0, ui, pi, ni, 1e-4)
    np.testing.assert_allclose(eng.last_loss(), bpr + l2, rtol=1e-5)
    z = np.zeros_like
    wU, _, _ = orc.adam_dense(U0, gU, z(U0), z(U0), 1)
    wV, _, _ = orc.adam_dense(V0, gV, z(V0), z(V0), 1)
    np.testing.assert_allclose(np.linalg.norm(eng.user_emb.cpu().numpy()), np.linalg.norm(wU), rtol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(eng.item_emb.cpu().numpy()), np.linalg.norm(wV), rtol=1e-5)
    assert (eng.G == 0).all()
    # S-CUL graph
    U, I, nnz = 5551, 16980, 130_000
    key = np.unique(rng.integers(0, U * I, nnz))
    ru, ri = key // I, key % I
    rowptr, col, val = orc.norm_adj_csr(ru, ri, U, I)
    U0 = (rng.uniform(-1, 1, (U, d)) * 0.03).astype(np.float32)
    V0 = (rng.uniform(-1, 1, (I, d)) * 0.03).astype(np.float32)
    eng = LGCNEngine(U0, V0, rowptr, col, val, 3, 1e-3, 1e-4, DEV)
    eng.keep_grad = True
    uo, io = eng.forward()
    wu, wi = orc.lgcn_forward(rowptr, col, val, U0, V0, 3)
    np.testing.assert_allclose(uo.cpu().numpy(), wu, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(io.cpu().numpy(), wi, rtol=1e-5, atol=1e-8)
    ui, pi, ni = ru[:B], ri[:B], rng.integers(0, I, B)
    eng.step(t(ui, torch.int32), t(pi, torch.int32), t(ni, torch.int32))
    bpr, l2, gOu, gOi, _ = orc.bpr_l2_fwd_bwd(wu, wi, ui, pi, ni, 1e-4)
    np.testing.assert_allclose(eng.last_loss(), bpr + l2, rtol=1e-5)
    gU, gV = orc.lgcn_backward(rowptr, col, val, gOu.astype(np.float32), gOi.astype(np.float32), 3)
    sc = max(np.abs(gU).max(), np.abs(gV).max())
    np.testing.assert_allclose(eng.G[:U].cpu().numpy(), gU, rtol=1e-4, atol=2e-6 * sc)
    np.testing.assert_allclose(eng.G[U:].cpu().numpy(), gV, rtol=1e-4, atol=2e-6 * sc)
    # linearity (size-independent property): A(2x + y) == 2 A x + A y up to rounding
    from coldrec_amd import ops
    x, y = torch.randn_like(eng.E), torch.randn_like(eng.E)
    ax, ay, axy = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    ops.spmm_csr(eng.rowptr, eng.col, eng.val, x, y=ax)
    ops.spmm_csr(eng.rowptr, eng.col, eng.val, y, y=ay)
    ops.spmm_csr(eng.rowptr, eng.col, eng.val, 2 * x + y, y=axy)
    torch.testing.assert_close(axy, 2 * ax + ay, rtol=1e-4, atol=1e-5)


def test_epoch_runner_graph_replay_equals_eager():
    """Four epochs through EpochRunner (eager, capture, replay, replay) == the same steps issued one by one."""
    from coldrec_amd.train import EpochRunner, MFEngine, LGCNEngine
    g = load_golden("g3_mf.npz")
    rowptr, col, val = _graph()
    g5 = load_golden("g5_lgcn.npz")
    rng = np.random.default_rng(0)
    n, B = 1500, 512
    for make in (lambda: MFEngine(g["d16_U0"], g["d16_V0"], 1e-3, 1e-4, DEV),
                 lambda: LGCNEngine(g5["U0"], g5["V0"], rowptr, col, val, 2, 1e-3, 1e-4, DEV)):
        a, b = make(), make()
        runner = EpochRunner(a, n, B, fused=False)       # the one-launch MF step has its own test (not bit-equal)
        for epoch in range(4):
            u = rng.integers(0, a.user_num, n).astype(np.int32)
            i = rng.integers(0, a.item_num, n).astype(np.int32)
            j = rng.integers(0, a.item_num, n).astype(np.int32)
            losses = runner.run(u, i, j).clone()
            tu, ti, tj = t(u), t(i), t(j)
            from coldrec_amd import ops
            plans = ops.build_plans_device(tu, ti, tj, B)
            for s, lo in enumerate(range(0, n, B)):
                b.step(tu[lo:lo + B], ti[lo:lo + B], tj[lo:lo + B], plans[s])
                assert torch.equal(losses[s], b.loss), (epoch, s)
        assert runner.graph is not None and a.step_count == b.step_count == 12
        assert torch.equal(a.E, b.E) and torch.equal(a.M, b.M) and torch.equal(a.V, b.V)


@pytest.mark.parametrize("d,B,cuts", [(128, 4096, (0, 2048, 4096)), (64, 1001, (0, 125, 250, 500, 501, 1001)),
                                      (16, 7, (0, 3, 7))])
def test_bpr_split_fwd_bwd_equals_fused(d, B, cuts):
    """Data-parallel split (crh_bpr_fwd_f32 / crh_bpr_bwd_f32): per-slice forwards, summed batch sums (what the
    RCCL all-reduce produces), per-slice backwards with the global batch size == the fused single-call step."""
    from coldrec_amd import ops
    rng = np.random.default_rng(B + d)
    rows = 500
    U = (rng.standard_normal((rows, d)) * 0.3).astype(np.float32)
    V = (rng.standard_normal((rows + 9, d)) * 0.3).astype(np.float32)
    ui, pi, ni = (rng.integers(0, n, B).astype(np.int32) for n in (rows, rows + 9, rows + 9))
    reg = 0.02
    tU, tV, tu, tp, tn = t(U), t(V), t(ui), t(pi), t(ni)
    want_loss, wU, wV = _bpr_gpu(U, V, ui, pi, ni, reg)
    slices = list(zip(cuts[:-1], cuts[1:]))
    wss = [ops.bpr_workspace(hi - lo, DEV) for lo, hi in slices]
    sums = [torch.zeros(4, device=DEV) for _ in slices]
    for (lo, hi), ws, s in zip(slices, wss, sums):
        ops.bpr_fwd(tU, tV, tV, tu[lo:hi], tp[lo:hi], tn[lo:hi], s, ws)
    total = torch.stack(sums).sum(0)
    gU, gV = torch.zeros_like(tU), torch.zeros_like(tV)
    loss = torch.zeros(2, device=DEV)
    for (lo, hi), ws in zip(slices, wss):
        ops.bpr_bwd(tU, tV, tV, tu[lo:hi], tp[lo:hi], tn[lo:hi], B, reg, total, gU, gV, gV, loss, ws)
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.cpu().numpy(), want_loss, rtol=1e-5)
    sc = max(np.abs(wU).max(), np.abs(wV).max())
    np.testing.assert_allclose(gU.cpu().numpy(), wU, rtol=1e-4, atol=2e-6 * sc)
    np.testing.assert_allclose(gV.cpu().numpy(), wV, rtol=1e-4, atol=2e-6 * sc)
    bpr, l2, oU, oV, _ = orc.bpr_l2_fwd_bwd(U, V, ui, pi, ni, reg)
    np.testing.assert_allclose(loss.cpu().numpy(), [bpr, l2], rtol=1e-5)
    np.testing.assert_allclose(gU.cpu().numpy(), oU, rtol=1e-4, atol=2e-6 * sc)


def test_dp_engines_world1_match_plain_engines():
    """The data-parallel step (slice fwd -> all-reduce -> slice bwd -> all-reduce(grad) -> Adam) at world size 1
    runs the split HIP entry points; it must match the fused step (1e-5 on losses and table norms)."""
    from coldrec_amd.train import DPContext, LGCNEngine, MFEngine
    from coldrec_amd.util.databuilder import bipartite_norm_adj_csr
    rng = np.random.default_rng(11)
    n_u, n_i, d, B = 300, 500, 64, 512
    U0 = (rng.standard_normal((n_u, d)) * 0.1).astype(np.float32)
    V0 = (rng.standard_normal((n_i, d)) * 0.1).astype(np.float32)
    pairs = np.unique(np.stack([rng.integers(0, n_u, 4000), rng.integers(0, n_i - 20, 4000)], 1), axis=0)
    rowptr, col, val = bipartite_norm_adj_csr(pairs[:, 0], pairs[:, 1], n_u, n_i)
    tri = [tuple(t(rng.integers(0, n, B).astype(np.int32)) for n in (n_u, n_i, n_i)) for _ in range(5)]
    for make in (lambda: MFEngine(U0, V0, 1e-2, 1e-3, DEV),
                 lambda: LGCNEngine(U0, V0, rowptr, col, val, 3, 1e-2, 1e-3, DEV)):
        a, b = make(), make()
        b.enable_data_parallel(DPContext(1, 0))
        for u, i, j in tri:
            a.step(u, i, j)
            b.step(u, i, j)
            np.testing.assert_allclose(b.last_loss(), a.last_loss(), rtol=1e-5)
        na, nb = float(a.E.norm()), float(b.E.norm())
        assert abs(na - nb) <= 1e-5 * na
        assert float((a.E - b.E).norm()) <= 1e-4 * na


@pytest.mark.parametrize("n_u,n_i,d", [(9000, 3000, 128), (15000, 7000, 128), (15000, 7000, 64), (30000, 2000, 32)])
def test_spmm_xcd_column_slices(n_u, n_i, d):
    """Operand sizes where crh_spmm_csr_f32 cuts the feature columns into 2 or 4 XCD-resident slices (and one
    where it does not): single-segment rows stay bit-identical to the oracle's chain, heavy rows (one wave each,
    shuffle combine) agree to rounding, with and without the schedule."""
    from coldrec_amd import ops
    rng = np.random.default_rng(n_u + d)
    w = 1.0 / np.arange(1, n_i + 1) ** 0.9
    nnz = 12 * n_u
    key = np.unique(rng.integers(0, n_u, nnz) * n_i + rng.choice(n_i, nnz, p=w / w.sum()))
    rowptr, col, val = orc.norm_adj_csr(key // n_i, key % n_i, n_u, n_i)
    deg = np.diff(rowptr)
    assert deg.max() > 64
    X = rng.standard_normal((n_u + n_i, d)).astype(np.float32)
    tX = t(X)
    want = orc.spmm(rowptr, col, val, X)
    Y0, Y1 = torch.empty_like(tX), torch.empty_like(tX)
    ops.spmm_csr(t(rowptr), t(col), t(val), tX, y=Y0)
    np.testing.assert_array_equal(Y0.cpu().numpy(), want)
    sched = ops.SpmmSchedule(rowptr, DEV)
    ops.spmm_csr(t(rowptr), t(col), t(val), tX, y=Y1, sched=sched)
    one = deg <= sched.seg
    np.testing.assert_array_equal(Y1.cpu().numpy()[one], want[one])
    np.testing.assert_allclose(Y1.cpu().numpy()[~one], want[~one], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("d,B", [(64, 300), (128, 2048)])
def test_lazy_adam_replay_is_bitwise_dense_adam(d, B):
    """Touched-rows replay of dense Adam (crh_adam_rows_f32) vs the dense kernel on the same deterministic
    gradients: after every step the ROWS OF THE NEXT BATCH, and after the flush the WHOLE tables (p, m, v), must be
    bit-identical -- including rows never touched (pure zero-gradient decay) and rows touched on consecutive steps."""
    from coldrec_amd import ops
    from coldrec_amd.train import MFEngine
    rng = np.random.default_rng(d)
    n_u, n_i, steps = 3000, 5000, 25
    U0 = (rng.standard_normal((n_u, d)) * 0.1).astype(np.float32)
    V0 = (rng.standard_normal((n_i, d)) * 0.1).astype(np.float32)
    dense, lazy = MFEngine(U0, V0, 1e-2, 1e-3, DEV), MFEngine(U0, V0, 1e-2, 1e-3, DEV)
    lazy.enable_lazy_adam()
    hot = rng.integers(0, n_i, 40)                                 # a few items recur in most batches
    for s in range(steps):
        u = rng.integers(0, n_u, B).astype(np.int32)
        i = np.where(rng.random(B) < 0.3, hot[rng.integers(0, 40, B)], rng.integers(0, n_i, B)).astype(np.int32)
        j = rng.integers(0, n_i, B).astype(np.int32)
        tu, ti, tj = t(u), t(i), t(j)
        plan = ops.build_plans_device(tu, ti, tj, B)[0]
        dense.step(tu, ti, tj, plan=plan)
        lazy.step(tu, ti, tj, plan=plan)
        assert torch.equal(dense.loss, lazy.loss), s               # same forward inputs -> same bits
    torch.cuda.synchronize()
    assert not torch.equal(dense.E, lazy.E)                        # untouched rows are still behind ...
    ue, ie = lazy.forward()                                        # ... until the flush
    assert torch.equal(dense.E.view(torch.int32), lazy.E.view(torch.int32))
    assert torch.equal(dense.M.view(torch.int32), lazy.M.view(torch.int32))
    assert torch.equal(dense.V.view(torch.int32), lazy.V.view(torch.int32))
    assert int(lazy.last_step.min()) == steps and float(lazy.G.abs().max()) == 0.0
    # training goes on after a flush
    tu, ti, tj = t(rng.integers(0, n_u, B).astype(np.int32)), t(rng.integers(0, n_i, B).astype(np.int32)), t(rng.integers(0, n_i, B).astype(np.int32))
    dense.step(tu, ti, tj, plan=ops.build_plans_device(tu, ti, tj, B)[0])
    lazy.step(tu, ti, tj)
    lazy.sync_tables()
    assert torch.equal(dense.E.view(torch.int32), lazy.E.view(torch.int32))


@pytest.mark.parametrize("d,n_rec,B,n_u,n_i", [(128, 3 * 512 + 77, 512, 300, 500), (64, 4 * 300, 300, 300, 500),
                                               (200, 700, 256, 300, 500), (8, 5000, 4096, 300, 500),
                                               (128, 2 * 2048 + 9, 2048, 12000, 9000)])   # rows > 2048 blocks x 8: stride loop
def test_fused_mf_step_matches_three_kernel_step(d, n_rec, B, n_u, n_i):
    """crh_mf_step_f32 (one launch per step: recomputed score differences, Adam in registers, norms of the next
    batch from the updated rows) against forward + plan backward + dense Adam: same losses and tables up to the fp32
    summation order of the three norms; bit-reproducible; hot items exercise the heavy-row paths; odd and even step
    counts exercise the ping-pong copy-back; the last batch is short."""
    from coldrec_amd.train import EpochRunner, MFEngine
    rng = np.random.default_rng(d + B)
    U0 = (rng.standard_normal((n_u, d)) * 0.1).astype(np.float32)
    V0 = (rng.standard_normal((n_i, d)) * 0.1).astype(np.float32)
    epochs = []
    for _ in range(3):
        u = rng.integers(0, n_u, n_rec).astype(np.int32)
        hot = rng.random(n_rec) < 0.3
        i = np.where(hot, rng.integers(0, 3, n_rec), rng.integers(0, n_i, n_rec)).astype(np.int32)
        j = rng.integers(0, n_i, n_rec).astype(np.int32)
        j = np.where(j == i, (j + 1) % n_i, j).astype(np.int32)
        epochs.append((u, i, j))
    runs = {}
    for tag, fused in (("fused", True), ("fused2", True), ("plain", False)):
        eng = MFEngine(U0, V0, 1e-2, 1e-3, DEV)
        runner = EpochRunner(eng, n_rec, B, fused=fused)
        assert eng.fused == fused
        losses = [runner.run(*ep).clone() for ep in epochs]           # eager, captured + replayed, replayed
        torch.cuda.synchronize()
        runs[tag] = (torch.cat(losses).cpu().numpy(), eng.E.cpu().numpy(), eng.M.cpu().numpy(), eng.V.cpu().numpy(),
                     eng.step_count)
    for a, b in zip(runs["fused"][:4], runs["fused2"][:4]):
        assert np.array_equal(a, b)                                   # deterministic
    assert runs["fused"][4] == runs["plain"][4] == 3 * ((n_rec + B - 1) // B)
    np.testing.assert_allclose(runs["fused"][0], runs["plain"][0], rtol=2e-6, atol=1e-9)
    for a, b in zip(runs["fused"][1:4], runs["plain"][1:4]):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-5 * np.abs(b).max())   # sums with cancellation


def test_fused_mf_step_first_step_against_oracle():
    """One fused step from the C ABI against the fp64 closed form + the oracle's Adam."""
    from coldrec_amd import ops
    rng = np.random.default_rng(77)
    n_u, n_i, d, B = 90, 140, 64, 333
    U0 = (rng.standard_normal((n_u, d)) * 0.2).astype(np.float32)
    V0 = (rng.standard_normal((n_i, d)) * 0.2).astype(np.float32)
    ui = rng.integers(0, n_u, B).astype(np.int32)
    pi = np.where(rng.random(B) < 0.4, 7, rng.integers(0, n_i, B)).astype(np.int32)      # item 7 is a heavy row
    ni = rng.integers(0, n_i, B).astype(np.int32)
    ni = np.where(ni == pi, (ni + 1) % n_i, ni).astype(np.int32)
    reg, lr = 0.05, 1e-2
    bpr, l2, gU, gV, _ = orc.bpr_l2_fwd_bwd(U0, V0, ui, pi, ni, reg)
    E0 = np.concatenate([U0, V0])
    z = np.zeros_like(E0)
    want_p, want_m, want_v = orc.adam_dense(E0, np.concatenate([gU, gV]).astype(np.float32), z, z, 1, lr=lr)
    E, E2 = t(E0), torch.empty_like(t(E0))
    M, V = torch.zeros_like(E), torch.zeros_like(E)
    tu, tp, tn = t(ui), t(pi), t(ni)
    plans = ops.build_plans_device(tu, tp, tn, B)
    rng_t, mult_t, ent_t = ops.mf_step_tables(plans, tu, tp, tn, B, n_u, n_i)
    touched = (rng_t[0, :, 1] > rng_t[0, :, 0]).cpu().numpy()
    assert touched.sum() == len(np.unique(ui)) + len(np.unique(np.concatenate([pi, ni])))
    mu = mult_t[0].cpu().numpy()
    assert np.array_equal(mu[:n_u], np.bincount(ui, minlength=n_u))
    assert np.array_equal(mu[n_u:] & 0xffff, np.bincount(pi, minlength=n_i))
    assert np.array_equal(mu[n_u:] >> 16, np.bincount(ni, minlength=n_i))
    ws = ops.bpr_workspace(B, DEV)
    ops.bpr_fwd(E[:n_u], E[n_u:], E[n_u:], tu, tp, tn, torch.zeros(4, device=DEV), ws)
    parts = torch.zeros(4 * ops.mf_step_parts(n_u + n_i, d), device=DEV)
    loss = torch.zeros(2, device=DEV)
    sc = torch.from_numpy(ops.adam_step_scalars(1, 1, lr)).to(DEV)
    ops.mf_step(E, E2, M, V, n_u, B, reg, plans[0], rng_t[0], ent_t[0], None, ws.view(torch.float32),
                ops.bpr_fwd_parts(B, d), parts, None, 0, loss, sc[0])
    ops.mf_step_finish(parts, ops.mf_step_parts(n_u + n_i, d), B, loss)
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.cpu().numpy(), [bpr, l2], rtol=1e-5)
    np.testing.assert_allclose(M.cpu().numpy(), want_m, rtol=1e-4, atol=1e-9)
    # the first Adam step moves an element by lr * g / (|g| + 1e-8): ill-conditioned where |g| ~ 1e-8, hence lr-scaled
    np.testing.assert_allclose(E2.cpu().numpy(), want_p, rtol=1e-5, atol=1e-3 * lr)
    assert torch.equal(E, t(E0))                                       # the input buffer is left alone


@pytest.mark.parametrize("L", [1, 2, 3])
def test_lightgcn_adam_in_spmm_epilogue_is_bit_identical(L, monkeypatch):
    """crh_spmm_csr_adam_f32 (the optimiser in the last backward SpMM's epilogue, dOUT cleared there for L >= 2)
    == separate gradient table + crh_adam_dense_f32 + dOUT.zero_(): same bits on E, M, V and the losses."""
    from coldrec_amd.train import LGCNEngine
    g5 = load_golden("g5_lgcn.npz")
    rowptr, col, val = _graph()
    rng = np.random.default_rng(L)
    engines = []
    for fused in ("1", "0"):
        monkeypatch.setenv("CRH_LGCN_FUSED", fused)
        eng = LGCNEngine(g5["U0"], g5["V0"], rowptr, col, val, L, 1e-2, 1e-3, DEV)
        assert eng.fuse_adam == (fused == "1")
        engines.append(eng)
    from coldrec_amd import ops
    for step in range(5):
        B = 700
        u = t(rng.integers(0, engines[0].user_num, B).astype(np.int32))
        i = t(rng.integers(0, engines[0].item_num, B).astype(np.int32))
        j = t(rng.integers(0, engines[0].item_num, B).astype(np.int32))
        plan = ops.build_plans_device(u, i, j, B)[0]
        for eng in engines:
            eng.step(u, i, j, plan)
        a, b = engines
        assert torch.equal(a.loss, b.loss)
        assert torch.equal(a.E, b.E) and torch.equal(a.M, b.M) and torch.equal(a.V, b.V), (L, step)
        if L >= 2:
            assert a._dout_clean and not bool(a.dOUT.any())


def test_spmm_adam_epilogue_ops_level():
    """crh_spmm_csr_adam_f32 == crh_spmm_csr_f32 (gradient into a table) + crh_adam_dense_f32, bit for bit, on a
    NON-symmetric weighted matrix with and without the schedule; zero_acc_in clears the consumed operand; the
    gradient can still be stored (acc_out)."""
    from coldrec_amd import ops
    import scipy.sparse as sp
    rng = np.random.default_rng(21)
    n, d = 3000, 64
    deg = np.minimum(rng.zipf(1.6, n), 400)
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, rows.shape[0])
    A = sp.csr_matrix((rng.random(rows.shape[0]).astype(np.float32), (rows, cols)), shape=(n, n))
    A.sum_duplicates(); A.sort_indices()
    rp, cl, vl = t(A.indptr.astype(np.int64)), t(A.indices.astype(np.int32)), t(A.data.astype(np.float32))
    x, z = t(rng.standard_normal((n, d)).astype(np.float32)), t(rng.standard_normal((n, d)).astype(np.float32))
    p0 = t((rng.standard_normal((n, d)) * 0.1).astype(np.float32))
    m0 = t((rng.standard_normal((n, d)) * 1e-3).astype(np.float32))
    v0 = t((rng.random((n, d)) * 1e-5).astype(np.float32))
    for sched in (None, ops.SpmmSchedule(A.indptr, DEV)):
        g = torch.empty_like(x)
        ops.spmm_csr(rp, cl, vl, x, acc_in=z, s_in=0.5, acc_out=g, s_out=0.25, sched=sched)
        p1, m1, v1 = p0.clone(), m0.clone(), v0.clone()
        ops.adam_dense(p1, g.clone(), m1, v1, 7, lr=1e-2, zero_grad=False)
        p2, m2, v2, z2, g2 = p0.clone(), m0.clone(), v0.clone(), z.clone(), torch.empty_like(x)
        ops.spmm_csr_adam(rp, cl, vl, x, z2, 0.5, g2, 0.25, sched, p2, m2, v2, 7, lr=1e-2, zero_acc_in=True)
        assert torch.equal(g, g2) and torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)
        assert not bool(z2.any())
        sc = torch.from_numpy(ops.adam_step_scalars(7, 1, 1e-2)).to(DEV)            # graph-replay form: factors from memory
        p3, m3, v3 = p0.clone(), m0.clone(), v0.clone()
        ops.spmm_csr_adam(rp, cl, vl, x, z.clone(), 0.5, None, 0.25, sched, p3, m3, v3, 0, lr=1e-2, step_scalars=sc[0])
        assert torch.equal(p1, p3) and torch.equal(m1, m3) and torch.equal(v1, v3)
