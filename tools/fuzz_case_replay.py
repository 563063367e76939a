#!/usr/bin/env python3
"""Replay ONE saved case of tests/fuzz/fuzz_train_ops.py (fused BPR-MF epochs): the one-launch step, the three-kernel step
and the fp64 closed form (oracle gradients + oracle Adam) side by side, step by step -- losses, and after every step the
largest table difference of each GPU form against the replay with the element it sits on.

    python tools/fuzz_case_replay.py tests/fuzz/cases/<case>.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coldrec_amd import ops  # noqa: E402
from coldrec_amd.train import EpochRunner, MFEngine  # noqa: E402
from oracle import oracle_np as orc  # noqa: E402

DEV = torch.device("cuda:0")
c = np.load(sys.argv[1])
U0, V0, B, n_rec = c["U0"], c["V0"], int(c["B"]), int(c["n_rec"])
n_u, n_i, d = U0.shape[0], V0.shape[0], U0.shape[1]
epochs = []
q = 0
while "e%d_u" % q in c.files:
    epochs.append((c["e%d_u" % q], c["e%d_i" % q], c["e%d_j" % q]))
    q += 1
print("case: d=%d B=%d n_rec=%d n_u=%d n_i=%d epochs=%d" % (d, B, n_rec, n_u, n_i, len(epochs)))
steps = [(e, lo, min(lo + B, n_rec)) for e in range(len(epochs)) for lo in range(0, n_rec, B)]

# GPU forms, stepped ONE step at a time so that the tables after every step are visible: the three-kernel step eagerly, the
# one-launch step through a runner over a one-step "epoch" is not the same code path as a whole epoch, so the fused form is
# run as whole epochs (its per-step losses) and its tables are taken at the epoch ends only.
plain = MFEngine(U0, V0, 1e-2, 1e-3, DEV)
fused = MFEngine(U0, V0, 1e-2, 1e-3, DEV)
fr = EpochRunner(fused, n_rec, B, fused=True)
E, M, V = np.concatenate([U0, V0]), np.zeros((n_u + n_i, d), np.float32), np.zeros((n_u + n_i, d), np.float32)
eps32 = float(np.finfo(np.float32).eps)
k = 0
for e, (eu, ei, ej) in enumerate(epochs):
    fl = fr.run(eu, ei, ej).clone().cpu().numpy()
    tu, ti, tj = (torch.from_numpy(x).to(DEV) for x in (eu, ei, ej))
    plans = ops.build_plans_device(tu, ti, tj, B)
    for s, lo in enumerate(range(0, n_rec, B)):
        sl = slice(lo, min(lo + B, n_rec))
        plain.step(tu[sl], ti[sl], tj[sl], plan=plans[s])
        pl = plain.loss.cpu().numpy().astype(np.float64)
        bpr, l2, gU, gV, (a_u, a_p, a_n) = orc.bpr_l2_fwd_bwd(E[:n_u], E[n_u:], eu[sl], ei[sl], ej[sl], 1e-3)
        A = np.zeros((n_u + n_i, d))
        np.add.at(A, eu[sl], np.abs(a_u)); np.add.at(A, n_u + ei[sl].astype(np.int64), np.abs(a_p)); np.add.at(A, n_u + ej[sl].astype(np.int64), np.abs(a_n))
        g = np.concatenate([gU, gV])
        with np.errstate(divide="ignore", invalid="ignore"):
            cond = np.where(A > 0, np.abs(g) / (eps32 * A), np.inf)
        k += 1
        E, M, V = orc.adam_dense(E, g.astype(np.float32), M, V, k, lr=1e-2)
        dp = np.abs(plain.E.cpu().numpy().astype(np.float64) - E)
        r, cc = np.unravel_index(np.argmax(dp), dp.shape)
        print("step %2d (epoch %d, triples %s %s %s): bpr fp64 %.9f | three-kernel %.9f (%+.2e) | one-launch %.9f (%+.2e) | "
              "smallest |g|/(eps A) %.3g | three-kernel table vs replay: max %.2e at (%d,%d)" % (
                  k - 1, e, eu[sl].tolist(), ei[sl].tolist(), ej[sl].tolist(), bpr, pl[0], pl[0] - bpr, fl[s, 0], fl[s, 0] - bpr,
                  float(cond.min()), dp.max(), r, cc))
    df = np.abs(fused.E.cpu().numpy().astype(np.float64) - E)
    r, cc = np.unravel_index(np.argmax(df), df.shape)
    print("  end of epoch %d: one-launch table vs replay: max %.2e at (%d,%d); three-kernel vs replay %.2e; one-launch vs three-kernel %.2e"
          % (e, df.max(), r, cc, dp.max(), float((fused.E - plain.E).abs().max())))
