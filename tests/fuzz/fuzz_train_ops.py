#!/usr/bin/env python3
"""Randomised parity fuzzing of the training kernels against the oracle (tests-style tool: it imports oracle/).

Per case: random table sizes / widths / batch sizes / duplicate patterns, then
  * crh_bpr_fwd_bwd_f32 with the atomics backward and with the reverse-index ("plan") backward vs the fp64
    closed form (1e-5 on the losses, 5e-4 / scaled absolute on gradients); the plan backward twice -> identical bits;
  * crh_spmm_csr_f32 with and without the schedule on a random Zipf graph vs the C oracle: rows in one piece
    bit-exact, heavy rows to rounding;
  * a few optimiser steps with the dense Adam and with the touched-rows replay -> identical bits after the flush;
  * whole epochs with the one-launch MF step (crh_mf_step_f32) against the three-kernel step -> same losses and tables
    up to the summation order of the norms, and bit-identical when repeated.

    python tests/fuzz/fuzz_train_ops.py --minutes 5 [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coldrec_amd import ops  # noqa: E402
from coldrec_amd.train import MFEngine  # noqa: E402
from oracle import oracle_np as orc  # noqa: E402

DEV = torch.device("cuda:0")


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return (x if dtype is None else x.to(dtype)).to(DEV)


def fail(what, **kw):
    print("MISMATCH", what, kw, flush=True)
    sys.exit(1)


def case_bpr(rng):
    d = int(rng.choice([4, 8, 16, 64, 128, 200, 256]))
    n_u, n_i = int(rng.integers(1, 3000)), int(rng.integers(2, 4000))
    B = int(rng.choice([1, 7, 64, 513, 4096]))
    U = (rng.standard_normal((n_u, d)) * 0.3).astype(np.float32)
    V = (rng.standard_normal((n_i, d)) * 0.3).astype(np.float32)
    ui = rng.integers(0, n_u, B).astype(np.int32)
    hot = rng.integers(0, n_i, 3)
    pi = np.where(rng.random(B) < rng.choice([0.0, 0.3, 0.9]), hot[rng.integers(0, 3, B)], rng.integers(0, n_i, B)).astype(np.int32)
    ni = rng.integers(0, n_i, B).astype(np.int32)
    ni = np.where(ni == pi, (ni + 1) % n_i, ni).astype(np.int32)   # the sampler never returns neg == pos (the g*u terms
    reg = float(rng.choice([0.0, 1e-4, 0.05]))                      # of such a triple cancel catastrophically in fp32)
    bpr, l2, wU, wV, _ = orc.bpr_l2_fwd_bwd(U, V, ui, pi, ni, reg)
    sc = max(np.abs(wU).max(), np.abs(wV).max(), 1e-30)
    # fp32 evaluates (1 - sigmoid(x)) with ~6e-8 absolute error (the reference's fp32 autograd does the same):
    # the relative error of a saturated triple's gradient is 6e-8 / (1 - sigmoid)
    x64 = (U.astype(np.float64)[ui] * (V.astype(np.float64)[pi] - V.astype(np.float64)[ni])).sum(1)
    sat = 1.0e-6 / max(1e-12, float((1.0 - 1.0 / (1.0 + np.exp(-x64))).min()))
    tU, tV = t(U), t(V)
    outs = []
    plan = ops.build_plans_device(t(ui), t(pi), t(ni), B)[0] if B <= 8192 else None
    for mode in ("atomic", "plan", "plan"):
        gU, gV = torch.zeros_like(tU), torch.zeros_like(tV)
        loss = ops.bpr_fwd_bwd(tU, tV, tV, t(ui), t(pi), t(ni), reg, gU, gV, gV, plan=None if mode == "atomic" else plan)
        torch.cuda.synchronize()
        ln = loss.cpu().numpy()
        # -log(1e-5 + sigmoid) of a saturated triple is ~6e-8 absolute in fp32: 1e-5 relative for a real batch mean,
        # a little more for the single-triple batches drawn here
        if not (np.allclose(ln[0], bpr, rtol=1e-5 if B >= 64 else 1e-4, atol=1e-7) and
                np.allclose(ln[1], l2, rtol=1e-5, atol=1e-12)):
            fail("bpr loss", d=d, B=B, mode=mode, got=ln.tolist(), want=[float(bpr), float(l2)])
        for g, w in ((gU, wU), (gV, wV)):
            # fp32 sigmoid / (1 - sigmoid) loses relative precision when saturated (as the reference's own fp32
            # autograd does): 5e-4 relative, plus a floor scaled by the largest gradient entry
            if not np.allclose(g.cpu().numpy(), w, rtol=5e-4, atol=(1e-4 + sat) * sc):
                fail("bpr grad", d=d, B=B, mode=mode, err=float(np.abs(g.cpu().numpy() - w).max()), scale=float(sc),
                     sat=sat, reg=reg, x=x64[:4].tolist())
        outs.append((gU, gV))
    if not (torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])):
        fail("plan backward not deterministic", d=d, B=B)


def case_spmm(rng):
    d = int(rng.choice([4, 32, 64, 128, 256]))
    n_u, n_i = int(rng.integers(50, 30000)), int(rng.integers(20, 9000))
    w = 1.0 / np.arange(1, n_i + 1) ** float(rng.choice([0.0, 0.8, 1.1]))
    nnz = int(rng.integers(n_u, 12 * n_u))
    key = np.unique(rng.integers(0, n_u, nnz) * n_i + rng.choice(n_i, nnz, p=w / w.sum()))
    rowptr, col, val = orc.norm_adj_csr(key // n_i, key % n_i, n_u, n_i)
    deg = np.diff(rowptr)
    X = rng.standard_normal((n_u + n_i, d)).astype(np.float32)
    Z = rng.standard_normal((n_u + n_i, d)).astype(np.float32)
    want = orc.spmm(rowptr, col, val, X)
    tX, tZ = t(X), t(Z)
    rp, cl, vl = t(rowptr), t(col), t(val)
    Y0 = torch.empty_like(tX)
    ops.spmm_csr(rp, cl, vl, tX, y=Y0)
    if not np.array_equal(Y0.cpu().numpy(), want):
        fail("spmm rows", d=d, n=n_u + n_i)
    sched = ops.SpmmSchedule(rowptr, DEV)
    Y1, A1 = torch.empty_like(tX), torch.empty_like(tX)
    ops.spmm_csr(rp, cl, vl, tX, y=Y1, acc_in=tZ, s_in=0.5, acc_out=A1, s_out=2.0, sched=sched)
    y1 = Y1.cpu().numpy()
    one = deg <= sched.seg
    if not np.array_equal(y1[one], want[one]):
        fail("spmm sched light rows", d=d, n=n_u + n_i)
    # heavy rows are summed in a different (fixed) association: bound the difference by the row's condition,
    # 4 ulp-ish of sum |a_ij x_j| (fp64 reference via scipy)
    import scipy.sparse as sp
    A = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(n_u + n_i, n_u + n_i))
    p64 = A @ X.astype(np.float64)
    bound = (abs(A) @ np.abs(X).astype(np.float64)) * 4e-7 + 1e-9
    if not np.all(np.abs(y1[~one] - p64[~one]) <= bound[~one] + 1e-6 * np.abs(p64[~one])):
        fail("spmm sched heavy rows", d=d, n=n_u + n_i)
    a_want = (Z.astype(np.float64) * 0.5 + p64) * 2.0
    if not np.all(np.abs(A1.cpu().numpy() - a_want) <= 2 * bound + 1e-6 * (np.abs(a_want) + np.abs(Z))):
        fail("spmm epilogue", d=d)
    # round 3: the same launch with the light rows taken from the schedule's record stream (built when the edge arrays are
    # at hand; used for 8-lane groups, i.e. d = 32 or a sliced d = 128): every bit must agree with the descriptor path
    slab = ops.SpmmSchedule(rowptr, DEV, col=col, val=val)
    Y2, A2 = torch.empty_like(tX), torch.empty_like(tX)
    ops.spmm_csr(rp, cl, vl, tX, y=Y2, acc_in=tZ, s_in=0.5, acc_out=A2, s_out=2.0, sched=slab)
    if not (torch.equal(Y1, Y2) and torch.equal(A1, A2)):
        fail("spmm record stream != descriptor path", d=d, n=n_u + n_i, slab=bool(slab.for_launch(n_u + n_i, d).slab))


def case_adam(rng):
    d = int(rng.choice([8, 64, 128]))
    n_u, n_i = int(rng.integers(10, 4000)), int(rng.integers(10, 6000))
    B = int(rng.choice([5, 300, 2048]))
    U0 = (rng.standard_normal((n_u, d)) * 0.1).astype(np.float32)
    V0 = (rng.standard_normal((n_i, d)) * 0.1).astype(np.float32)
    lr = float(rng.choice([1e-3, 1e-2]))
    dense, lazy = MFEngine(U0, V0, lr, 1e-3, DEV), MFEngine(U0, V0, lr, 1e-3, DEV)
    lazy.enable_lazy_adam()
    for s in range(int(rng.integers(1, 12))):
        tri = [t(rng.integers(0, n, B).astype(np.int32)) for n in (n_u, n_i, n_i)]
        plan = ops.build_plans_device(*tri, B)[0]
        dense.step(*tri, plan=plan)
        lazy.step(*tri, plan=plan)
        if rng.random() < 0.2:
            lazy.sync_tables()
    lazy.sync_tables()
    torch.cuda.synchronize()
    for name in ("E", "M", "V"):
        if not torch.equal(getattr(dense, name).view(torch.int32), getattr(lazy, name).view(torch.int32)):
            fail("lazy adam", what=name, d=d, B=B)


def grad_abs_terms(E, n_u, eu, ei, ej, reg):
    """A[row, col] = sum of the ABSOLUTE values of every product that enters d(loss)/d(E[row, col]) for one batch, at the
    level of the operands: |g_b| (|p_c| + |n_c|) + |cu u_c| per triple for a user row, |g_b u_c| + |cp p_c| (or cn n_c) for an
    item row.  An fp32 evaluation of the element -- in ANY order -- is only good to ~eps32 x A, and so is the element after
    the ulp-level differences two valid fp32 forms of the PREVIOUS steps leave in p, n, u.  (Round 4, seed 92: a user
    row's element with p_c - n_c cancelling to 4e-4 and the BPR term cancelling against the L2 term to 2e-4 -- |g| = 1.5e-9,
    below Adam's eps -- is invisible to a sum over the per-triple totals, whose single term IS g.)"""
    E64 = E.astype(np.float64)
    u, p, n = E64[eu], E64[n_u + ei.astype(np.int64)], E64[n_u + ej.astype(np.int64)]
    B = u.shape[0]
    x = (u * p).sum(1) - (u * n).sum(1)
    with np.errstate(over="ignore"):
        sig = 1.0 / (1.0 + np.exp(-x))
    g = np.abs((1.0 / B) * sig * (1.0 - sig) / (1e-5 + sig))[:, None]
    nu, npp, nn = (np.sqrt((t * t).sum()) for t in (u, p, n))
    cu, cp, cn = (reg / (B * t) if t > 0 else 0.0 for t in (nu, npp, nn))
    A = np.zeros_like(E64)
    np.add.at(A, eu, g * (np.abs(p) + np.abs(n)) + cu * np.abs(u))
    np.add.at(A, n_u + ei.astype(np.int64), g * np.abs(u) + cp * np.abs(p))
    np.add.at(A, n_u + ej.astype(np.int64), g * np.abs(u) + cn * np.abs(n))
    return A


STATS = None    # --stats: list of per-case error records of the one-launch and the three-kernel step vs the fp64 replay
FORCE = {}      # --force '{"d": 200, "B": 1000, ...}': pins case_fused's shape (reproducing a reported case's class)


def case_fused(rng):
    from coldrec_amd.train import EpochRunner
    d = int(rng.choice([4, 16, 64, 128, 200, 256]))
    # (a handful of items shared by every triple lets ONE ill-conditioned Adam element -- |g| ~ 1e-8 in the first step --
    # reach dozens of user rows one step later: not a property of either kernel, so the catalogue has >= 64 items)
    n_u, n_i = int(rng.integers(2, 900)), int(rng.integers(64, 1500))
    B = int(rng.choice([3, 64, 1000, 4096]))
    n_rec = int(rng.integers(1, 4 * B + 2))
    U0 = (rng.standard_normal((n_u, d)) * 0.1).astype(np.float32)
    V0 = (rng.standard_normal((n_i, d)) * 0.1).astype(np.float32)
    hot_frac = float(rng.choice([0.0, 0.3, 0.95]))
    n_epochs = int(rng.integers(1, 4))
    if FORCE:
        d, n_u, n_i, B, n_rec = (int(FORCE.get(k, v)) for k, v in (("d", d), ("n_u", n_u), ("n_i", n_i), ("B", B), ("n_rec", n_rec)))
        hot_frac, n_epochs = float(FORCE.get("hot", hot_frac)), int(FORCE.get("epochs", n_epochs))
        U0 = (rng.standard_normal((n_u, d)) * 0.1).astype(np.float32)
        V0 = (rng.standard_normal((n_i, d)) * 0.1).astype(np.float32)
    epochs = []
    for _ in range(n_epochs):
        u = rng.integers(0, n_u, n_rec).astype(np.int32)
        i = np.where(rng.random(n_rec) < hot_frac, rng.integers(0, min(3, n_i), n_rec), rng.integers(0, n_i, n_rec)).astype(np.int32)
        j = rng.integers(0, n_i, n_rec).astype(np.int32)
        j = np.where(j == i, (j + 1) % n_i, j).astype(np.int32)
        epochs.append((u, i, j))
    res = []
    for fused in (True, True, False):
        eng = MFEngine(U0, V0, 1e-2, 1e-3, DEV)
        runner = EpochRunner(eng, n_rec, B, fused=fused)
        losses = torch.cat([runner.run(*ep).clone() for ep in epochs])
        torch.cuda.synchronize()
        res.append((losses.cpu().numpy(), eng.E.cpu().numpy(), eng.M.cpu().numpy(), eng.V.cpu().numpy()))
    for a, b in zip(res[0], res[1]):
        if not np.array_equal(a, b):
            fail("fused step not deterministic", d=d, B=B, n_rec=n_rec)
    close = np.isclose(res[0][0], res[2][0], rtol=1e-5, atol=1e-9).all(axis=1)
    if not close.all():
        # Same inputs give the same loss to 1e-5 (the two forms differ in summation order only); losses that differ are a
        # defect UNLESS the tables already differ -- which happens when an element's gradient sat on its noise floor in an
        # EARLIER step (Adam turns a rounding-noise sign into +-lr, see the table rules below; with B = 3 one such element
        # is a visible share of the next batch's scores).  Replay in fp64 up to the first differing step and look.
        first_bad = int(np.nonzero(~close)[0][0])
        E, M, V = np.concatenate([U0, V0]), np.zeros((n_u + n_i, d), np.float32), np.zeros((n_u + n_i, d), np.float32)
        eps32, step, noisy_at = float(np.finfo(np.float32).eps), 0, None
        for (eu, ei, ej) in epochs:
            for lo in range(0, n_rec, B):
                if step >= first_bad or noisy_at is not None:
                    break
                sl = slice(lo, min(lo + B, n_rec))
                _, _, gU, gV, _ = orc.bpr_l2_fwd_bwd(E[:n_u], E[n_u:], eu[sl], ei[sl], ej[sl], 1e-3)
                A = grad_abs_terms(E, n_u, eu[sl], ei[sl], ej[sl], 1e-3)
                g = np.concatenate([gU, gV])
                if ((A > 0) & (np.abs(g) <= 4.0 * d * eps32 * A)).any():
                    noisy_at = step
                step += 1
                E, M, V = orc.adam_dense(E, g.astype(np.float32), M, V, step, lr=1e-2)
        err = float(np.abs(res[0][0] - res[2][0]).max())
        if noisy_at is None:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)       # keep the case for tools/fuzz_case_replay.py
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fuzz_fail_case.npz"), U0=U0, V0=V0, B=B, n_rec=n_rec,
                                **{"e%d_%s" % (q, nm): arr for q, ep in enumerate(epochs) for nm, arr in zip("uij", ep)})
            fail("fused losses", d=d, B=B, n_rec=n_rec, n_u=n_u, n_i=n_i, first_bad_step=first_bad, err=err)
        print("fused losses differ from step %d on (max %.3g) AFTER a noise-floor gradient element in step %d: excused "
              "(d=%d B=%d n_rec=%d n_u=%d n_i=%d)" % (first_bad, err, noisy_at, d, B, n_rec, n_u, n_i), flush=True)
        if err > 2 * 1e-2 * (first_bad + 1):          # ... but never by more than the steps could move a score
            fail("fused losses beyond what noise-floor elements can move", d=d, B=B, n_rec=n_rec, err=err)
        return
    n_steps = len(epochs) * ((n_rec + B - 1) // B)
    # north_star criterion, no excuses: per-step losses (above) AND the Frobenius norms of both tables within 1e-5 relative
    # between the one-launch step and the three-kernel step -- whatever single elements on their noise floor do
    for lo_, hi_ in ((0, n_u), (n_u, n_u + n_i)):
        na, nb = np.linalg.norm(res[0][1][lo_:hi_].astype(np.float64)), np.linalg.norm(res[2][1][lo_:hi_].astype(np.float64))
        if abs(na - nb) > 1e-5 * nb:
            fail("fused table norm (north_star 1e-5)", d=d, B=B, n_rec=n_rec, got=na, want=nb)
    if STATS is not None:
        # --stats: which of the two forms is closer to the closed form?  fp64 gradients + the oracle's Adam step by step
        E, M, V = np.concatenate([U0, V0]), np.zeros((n_u + n_i, d), np.float32), np.zeros((n_u + n_i, d), np.float32)
        step = 0
        ref_loss = []
        for (eu, ei, ej) in epochs:
            for lo in range(0, n_rec, B):
                sl = slice(lo, min(lo + B, n_rec))
                bpr, l2, gU, gV, _ = orc.bpr_l2_fwd_bwd(E[:n_u], E[n_u:], eu[sl], ei[sl], ej[sl], 1e-3)
                ref_loss.append(bpr + l2)
                step += 1
                E, M, V = orc.adam_dense(E, np.concatenate([gU, gV]).astype(np.float32), M, V, step, lr=1e-2)
        ref_loss = np.array(ref_loss)
        rec = {}
        for tag, r in (("fused", res[0]), ("plain", res[2])):
            err = np.abs(r[1].astype(np.float64) - E)
            rec[tag] = dict(max=float(err.max()), p999=float(np.quantile(err, 0.999)), rms=float(np.sqrt((err ** 2).mean())),
                            loss=float(np.max(np.abs(r[0].sum(1) - ref_loss) / np.abs(ref_loss))),
                            norm=float(abs(np.linalg.norm(r[1].astype(np.float64)) - np.linalg.norm(E)) / np.linalg.norm(E)))
        STATS.append(rec)
    for name, a, b in zip("EMV", res[0][1:], res[2][1:]):
        # gradient elements are sums with cancellation (floor 1e-5 of the largest entry).  An element whose gradient is
        # ~1e-8 (Adam's eps) is ill-conditioned: a step moves it by lr * g / (|g| + eps), anywhere in [-lr, lr] -- allow
        # a handful of such elements, bounded by what the steps taken can move them
        bad = ~np.isclose(a, b, rtol=5e-4, atol=1e-5 * np.abs(b).max())
        if bad.sum() > max(2, 1e-4 * bad.size) or (name == "E" and np.abs(a - b).max() > 2 * 1e-2 * n_steps):
            # Which of the two left the closed form?  fp64 gradients + the oracle's Adam, step by step, and beside each
            # gradient element the sum of the ABSOLUTE values of its terms, A: an fp32 sum of those terms -- in any
            # order -- is only good to ~d * eps32 * A, so an element with |g| below that floor carries a gradient whose
            # sign is rounding noise, and Adam's m / (sqrt(v) + eps) turns that into anything in [-lr, lr].
            E, M, V = np.concatenate([U0, V0]), np.zeros((n_u + n_i, d), np.float32), np.zeros((n_u + n_i, d), np.float32)
            step, eps32 = 0, float(np.finfo(np.float32).eps)
            a_max = np.zeros((n_u + n_i, d))
            noisy = np.zeros((n_u + n_i, d), bool)
            cond = np.full((n_u + n_i, d), np.inf)        # smallest |g| / (eps32 * A) an element saw in any step
            tainted = np.zeros((n_u + n_i, d), bool)      # noisy elements + same-column elements downstream of them
            cnt_frac = np.zeros(n_u + n_i)                # largest share of a batch's triples that touch the row
            for (eu, ei, ej) in epochs:
                for lo in range(0, n_rec, B):
                    sl = slice(lo, min(lo + B, n_rec))
                    _, _, gU, gV, _ = orc.bpr_l2_fwd_bwd(E[:n_u], E[n_u:], eu[sl], ei[sl], ej[sl], 1e-3)
                    A = grad_abs_terms(E, n_u, eu[sl], ei[sl], ej[sl], 1e-3)
                    g = np.concatenate([gU, gV])
                    # information flow: this step's gradient of u_c is sum g_b (p_c - n_c) (+ reg), that of p_c is g_b u_c: an
                    # element that was noise in an EARLIER step has, by now, moved its row by something in [-lr, lr] per
                    # step, and the same column of every row it shares a triple with inherits a (much smaller) share of that
                    pu, pp, pn = eu[sl].astype(np.int64), n_u + ei[sl].astype(np.int64), n_u + ej[sl].astype(np.int64)
                    t_u, t_p, t_n = tainted[pu], tainted[pp], tainted[pn]
                    np.logical_or.at(tainted, pu, t_p | t_n)
                    np.logical_or.at(tainted, pp, t_u)
                    np.logical_or.at(tainted, pn, t_u)
                    nb_ = sl.stop - sl.start
                    cnt_frac = np.maximum(cnt_frac, np.bincount(np.concatenate([pu, pp, pn]), minlength=n_u + n_i) / nb_)
                    noisy |= (A > 0) & (np.abs(g) <= 4.0 * d * eps32 * A)
                    tainted |= noisy
                    with np.errstate(divide="ignore", invalid="ignore"):
                        cond = np.minimum(cond, np.where(A > 0, np.abs(g) / (eps32 * A), np.inf))
                    a_max = np.maximum(a_max, A)
                    step += 1
                    E, M, V = orc.adam_dense(E, g.astype(np.float32), M, V, step, lr=1e-2)
            ref = {"E": E, "M": M, "V": V}[name]
            rows = np.unique(np.nonzero(bad)[0])
            ea, eb = float(np.abs(a - ref).max()), float(np.abs(b - ref).max())
            print("rows with differences:", rows[:10], "of", n_u, "+", n_i, "| fused vs oracle", ea, "| plain vs oracle", eb,
                  "| noise-floor elements among them:", int((bad & noisy).sum()), "of", int(bad.sum()), flush=True)
            idx = np.argwhere(bad & ~tainted)[:16]
            print("  downstream (tainted) elements among them:", int((bad & tainted & ~noisy).sum()), flush=True)
            print("  elements beyond the noise rule (row, col, smallest |g|/(eps A) seen, |fused-ref|, |plain-ref|, |fused-plain|):",
                  [(int(r), int(c), float("%.3g" % cond[r, c]), float("%.2g" % abs(a[r, c] - ref[r, c])),
                    float("%.2g" % abs(b[r, c] - ref[r, c])), float("%.2g" % abs(a[r, c] - b[r, c]))) for r, c in idx], flush=True)
            if name == "M":
                # m is linear in the gradients: an ABSOLUTE bound for each form on its own (a convex combination of the
                # steps' gradients, each good to ~d * eps32 * A) -- the one-launch step gets no credit for the
                # three-kernel step being worse
                # ... for elements whose inputs were the same in both runs.  Once a noise-floor element of some row has moved
                # differently (by up to lr per step, see E below), the same column of every row that shares a triple with it
                # sees a different gradient, |dg| <= sum_b |g_b| |dp| <= (entries of the row / B) * 2 lr steps (|g_b| < 1 / B),
                # and m is a convex combination of those (round 4, seed 52 forced class: 12 downstream elements 1.4e-7 apart)
                tol = 8.0 * d * eps32 * a_max + 1e-12
                tol = np.where(tainted, tol + cnt_frac[:, None] * 2.0 * 1e-2 * n_steps, tol)
                if (np.abs(a - ref) <= tol).all() and (np.abs(b - ref) <= tol).all():
                    continue
            elif not (bad & ~tainted).any() and np.abs(a - b)[tainted & ~noisy].max(initial=0.0) <= 0.02 * 1e-2 * n_steps:
                # E and V are not linear in g; the only elements excused are those whose gradient sat on its own noise
                # floor in some step (seed 42 of round 2: 9 of 85 504 elements on a 50 x 284 table) and, to 2 % of what the
                # steps can move an element, the same column of the rows such an element reaches through later triples
                # (seed 52 of round 3: ONE noise-floor element of a hot item's row -- 950 of 1 000 positives on 3 items --
                # and column 99 of 13 users who rated that item, 1e-5 apart; per-step launch and three-kernel step)
                continue
            fail("fused tables", table=name, d=d, B=B, n_rec=n_rec, n_u=n_u, n_i=n_i, hot=hot_frac, epochs=len(epochs),
                 err=float(np.abs(a - b).max()), scale=float(np.abs(b).max()), nbad=int(bad.sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="", help="run only this kind of case (bpr / spmm / adam / fused)")
    ap.add_argument("--force", default="", help="JSON: pin the shape of the fused case (d, n_u, n_i, B, n_rec, hot, epochs)")
    ap.add_argument("--stats", action="store_true",
                    help="fused cases: also replay every case in fp64 and report the error distributions of the one-launch "
                         "step and of the three-kernel step against it (is one of them systematically noisier?)")
    args = ap.parse_args()
    if args.stats:
        global STATS
        STATS = []
    if args.force:
        import json
        FORCE.update(json.loads(args.force))
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.minutes * 60
    counts = {"bpr": 0, "spmm": 0, "adam": 0, "fused": 0}
    while time.time() < t_end:
        which = args.only or str(rng.choice(["bpr", "spmm", "adam", "fused"]))
        {"bpr": case_bpr, "spmm": case_spmm, "adam": case_adam, "fused": case_fused}[which](rng)
        counts[which] += 1
    print(f"fuzz ok: {counts} random cases within parity, seed {args.seed}")
    if STATS:
        print("error of the final table E against the fp64 replay (oracle gradients + oracle Adam), %d fused cases:" % len(STATS))
        for key in ("max", "p999", "rms", "loss", "norm"):
            f = np.array([r["fused"][key] for r in STATS])
            p = np.array([r["plain"][key] for r in STATS])
            ratio = f / np.maximum(p, 1e-30)
            print("  %-5s one-launch: median %.3g p90 %.3g max %.3g | three-kernel: median %.3g p90 %.3g max %.3g | "
                  "ratio one-launch / three-kernel: median %.2f, p10 %.2f, p90 %.2f; one-launch worse in %d of %d cases"
                  % (key, np.median(f), np.quantile(f, 0.9), f.max(), np.median(p), np.quantile(p, 0.9), p.max(),
                     np.median(ratio), np.quantile(ratio, 0.1), np.quantile(ratio, 0.9), int((f > p).sum()), len(f)))


if __name__ == "__main__":
    main()
