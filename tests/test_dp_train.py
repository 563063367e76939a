"""Data-parallel training plumbing on 2 CPU ranks over gloo.

The HIP kernels are stood in by numpy/torch-CPU restatements with the SAME call signatures as
coldrec_amd.ops (legitimate in tests only); what is under test is the engine's slice arithmetic, the
two all-reduces and that the replicas stay identical and match the reference-call port (oracle/ref_port.py)
of model/MF.py:12-29 / model/LightGCN.py:14-29 run on ONE process with the whole batch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["CR_ROOT"])
import numpy as np, torch
import torch.distributed as dist
import coldrec_amd.train as train
from coldrec_amd.train import DPContext, MFEngine, LGCNEngine
from coldrec_amd.util.databuilder import bipartite_norm_adj_csr
from oracle import oracle_np as orc, ref_port


class NpKernels:
    """CPU stand-ins with the signatures of coldrec_amd.ops (tests only), assigned to coldrec_amd.train.ops below."""

    class SpmmSchedule:
        def __init__(self, rowptr, device, seg=None, col=None, val=None):
            self.seg = 64 if seg is None else seg

    @staticmethod
    def bpr_workspace(batch, device):
        return {}

    @staticmethod
    def _x(tu, tp, tn, u, p, n):
        ue, pe, ne = tu[u.long()].double(), tp[p.long()].double(), tn[n.long()].double()
        return ue, pe, ne, (ue * pe).sum(1) - (ue * ne).sum(1)

    @staticmethod
    def bpr_fwd(tu, tp, tn, u, p, n, sums, ws):
        ue, pe, ne, x = NpKernels._x(tu, tp, tn, u, p, n)
        sig = torch.sigmoid(x)
        sums.copy_(torch.stack([(ue * ue).sum(), (pe * pe).sum(), (ne * ne).sum(),
                                (-torch.log(1e-5 + sig)).sum()]).float())
        ws["x"] = x
        return sums

    @staticmethod
    def bpr_bwd(tu, tp, tn, u, p, n, B, reg, sums, gu, gp, gn, loss_out, ws, plan=None):
        ue, pe, ne, x = NpKernels._x(tu, tp, tn, u, p, n)
        assert torch.equal(x, ws["x"])                      # same slice as the forward
        sig = torch.sigmoid(x)
        g = (-(1.0 / B) * sig * (1 - sig) / (1e-5 + sig))[:, None]
        nu, npp, nn = (float(np.sqrt(float(sums[q]))) for q in range(3))
        gu.index_add_(0, u.long(), (g * (pe - ne) + reg / (B * nu) * ue).float())
        gp.index_add_(0, p.long(), (g * ue + reg / (B * npp) * pe).float())
        gn.index_add_(0, n.long(), (-g * ue + reg / (B * nn) * ne).float())
        if loss_out is not None:
            loss_out[0] = float(sums[3]) / B
            loss_out[1] = reg * (nu + npp + nn) / B
        return loss_out

    @staticmethod
    def bpr_fwd_bwd(tu, tp, tn, u, p, n, reg, gu, gp, gn, loss_out, plan=None, workspace=None):
        ws, sums = {}, torch.zeros(4)
        NpKernels.bpr_fwd(tu, tp, tn, u, p, n, sums, ws)
        return NpKernels.bpr_bwd(tu, tp, tn, u, p, n, u.shape[0], reg, sums, gu, gp, gn, loss_out, ws)

    @staticmethod
    def adam_dense(p, g, m, v, step, lr=1e-3, zero_grad=True, step_scalars=None):
        pn, mn, vn = orc.adam_dense(p.numpy(), g.numpy(), m.numpy(), v.numpy(), step, lr=lr)
        p.copy_(torch.from_numpy(pn)); m.copy_(torch.from_numpy(mn)); v.copy_(torch.from_numpy(vn))
        if zero_grad:
            g.zero_()

    # ---- the touched-rows step's calls (MFEngine._lazy_step_dp): plan = the batch's touched rows in slot order
    LR = 1e-2

    @staticmethod
    def build_plans_device(u, p, n, B):
        uu = torch.unique(u.long())
        ii = torch.unique(torch.cat([p.long(), n.long()]))
        return [{"users": uu, "items": ii}]

    @staticmethod
    def _slots(plan, user_rows):
        return torch.cat([plan["users"], plan["items"] + user_rows])

    @staticmethod
    def adam_step_scalars(first, n, lr):
        return np.zeros((n, 2), np.float32)

    @staticmethod
    def adam_rows(p, g, m, v, last_step, plan, batch, user_rows, step, scalar_table, mode):
        rows = torch.arange(p.shape[0]) if plan is None else NpKernels._slots(plan, user_rows)
        for r in rows.tolist():
            first, last = int(last_step[r]) + 1, (step - 1 if mode == 0 else step)
            for t in range(first, last + 1):
                grad = g[r:r + 1].numpy() if (mode == 1 and t == step) else np.zeros((1, p.shape[1]), np.float32)
                pn, mn, vn = orc.adam_dense(p[r:r + 1].numpy(), grad, m[r:r + 1].numpy(), v[r:r + 1].numpy(), t, lr=NpKernels.LR)
                p[r] = torch.from_numpy(pn[0]); m[r] = torch.from_numpy(mn[0]); v[r] = torch.from_numpy(vn[0])
            if last >= first:
                last_step[r] = last
            if mode == 1:
                g[r] = 0

    @staticmethod
    def bpr_bwd_owned(tu, ti, u, p, n, reg, sums, gu, gi, loss_out, ws, plan, own_mod, own_rem):
        fu, fi = torch.zeros_like(gu), torch.zeros_like(gi)
        NpKernels.bpr_bwd(tu, ti, ti, u, p, n, u.shape[0], reg, sums, fu, fi, fi, loss_out, ws)
        slots = NpKernels._slots(plan, tu.shape[0])
        mine = slots[own_rem::own_mod]
        G = torch.cat([fu, fi])
        for r in mine.tolist():
            (gu if r < tu.shape[0] else gi)[r if r < tu.shape[0] else r - tu.shape[0]] = G[r]
        return loss_out

    @staticmethod
    def rows_pack_cap(batch, own_mod):
        return -(-3 * batch // own_mod)

    @staticmethod
    def rows_pack(table, plan, batch, user_rows, own_mod, own_rem, out_ids, out_rows):
        mine = NpKernels._slots(plan, user_rows)[own_rem::own_mod]
        out_ids.fill_(-1)
        out_ids[: len(mine)] = mine.to(torch.int32)
        out_rows[: len(mine)] = table[mine]

    @staticmethod
    def rows_unpack(table, ids, rows):
        ok = ids >= 0
        assert len(torch.unique(ids[ok])) == int(ok.sum())            # every row has exactly one owner
        table[ids[ok].long()] = rows[ok]

    @staticmethod
    def spmm_csr(rowptr, col, val, x, y=None, acc_in=None, s_in=1.0, acc_out=None, s_out=1.0, sched=None):
        P = torch.from_numpy(orc.spmm(rowptr.numpy(), col.numpy(), val.numpy(), x.numpy()))
        if y is not None:
            y.copy_(P)
        if acc_out is not None:
            base = acc_in * s_in if acc_in is not None else 0.0
            acc_out.copy_((base + P) * s_out)


train.ops = NpKernels                                # the seam: the engines call whatever train.ops is (tests only)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(7)                      # same stream on every rank: replicated sampler
n_u, n_i, d, B, steps = 40, 70, 16, 101, 4         # odd batch: uneven slices
pairs = np.unique(np.stack([rng.integers(0, n_u, 600), rng.integers(0, n_i - 5, 600)], 1), axis=0)
U0 = (rng.standard_normal((n_u, d)) * 0.1).astype(np.float32)
V0 = (rng.standard_normal((n_i, d)) * 0.1).astype(np.float32)
tri = [(rng.integers(0, n_u, B).astype(np.int32), rng.integers(0, n_i, B).astype(np.int32),
        rng.integers(0, n_i, B).astype(np.int32)) for _ in range(steps)]
rowptr, col, val = bipartite_norm_adj_csr(pairs[:, 0], pairs[:, 1], n_u, n_i)
torch.set_num_threads(1)

for name in ("mf", "lgcn"):
    if name == "mf":
        eng = MFEngine(U0, V0, 1e-2, 1e-3, "cpu")
        port = ref_port.MFPort(U0, V0, 1e-2, 1e-3)
    else:
        eng = LGCNEngine(U0, V0, rowptr, col, val, 2, 1e-2, 1e-3, "cpu")
        port = ref_port.LGCNPort(U0, V0, ref_port.coo_adj(rowptr, col, val), 2, 1e-2, 1e-3)
    eng.enable_data_parallel(DPContext(world, rank))
    for (u, i, j) in tri:
        eng.step(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j))
        want = port.step(u, i, j)
        got = eng.last_loss()
        assert abs(got - want) <= 1e-5 * abs(want), (name, got, want)
    ref = torch.cat([port.U.detach(), port.V.detach()], 0)
    err = float((eng.E - ref).norm() / ref.norm())
    assert err < 1e-5, (name, err)                   # the tolerance north_star states for losses / norms
    # replicas must be bitwise identical after the same all-reduced update
    gathered = [torch.empty_like(eng.E) for _ in range(world)]
    dist.all_gather(gathered, eng.E)
    assert all(torch.equal(gathered[0], t) for t in gathered), name
    lo, hi = eng.dp.slice(B)
    assert 0 <= lo < hi <= B and hi - lo in (B // world, B // world + 1)
# the data-parallel TOUCHED-ROWS step (VERDICT r5 #4): row-ownership split of the backward, ONE all-gather of (row id, row)
# slots, touched-rows Adam over the whole plan on every replica -- against the dense-Adam reference port after the flush
eng = MFEngine(U0, V0, 1e-2, 1e-3, "cpu")
eng.enable_data_parallel(DPContext(world, rank))
eng.enable_lazy_adam()
port = ref_port.MFPort(U0, V0, 1e-2, 1e-3)
for (u, i, j) in tri:
    eng.step(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j))
    want = port.step(u, i, j)
    assert abs(eng.last_loss() - want) <= 1e-5 * abs(want), ("touched-rows dp", eng.last_loss(), want)
assert eng.exchange_bytes_per_step == world * (-(-3 * B // world)) * (d + 4) * 4
assert int(eng.last_step.min()) < steps                     # some rows really are behind before the flush
eng.sync_tables()
ref = torch.cat([port.U.detach(), port.V.detach()], 0)
err = float((eng.E - ref).norm() / ref.norm())
assert err < 1e-5, ("touched-rows dp", err)
gathered = [torch.empty_like(eng.E) for _ in range(world)]
dist.all_gather(gathered, eng.E)
assert all(torch.equal(gathered[0], t) for t in gathered), "touched-rows dp replicas differ"
# row-sharded LightGCN propagation (SURVEY.md 8(e), scalable variant): own row block per rank, all-gathered layer states
for L in (1, 3):
    eng = LGCNEngine(U0, V0, rowptr, col, val, L, 1e-2, 1e-3, "cpu")
    eng.enable_row_sharding(DPContext(world, rank))
    port = ref_port.LGCNPort(U0, V0, ref_port.coo_adj(rowptr, col, val), L, 1e-2, 1e-3)
    for (u, i, j) in tri:
        eng.step(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j))
        want = port.step(u, i, j)
        got = eng.last_loss()
        assert abs(got - want) <= 1e-5 * abs(want), ("row-sharded", L, got, want)
    ref = torch.cat([port.U.detach(), port.V.detach()], 0)
    err = float((eng.E - ref).norm() / ref.norm())
    assert err < 1e-5, ("row-sharded", L, err)
    gathered = [torch.empty_like(eng.E) for _ in range(world)]
    dist.all_gather(gathered, eng.E.contiguous())
    assert all(torch.equal(gathered[0], t) for t in gathered), ("row-sharded replicas", L)
    rows, r0, r1 = eng.rs
    assert (r0, r1) == (rank * rows, min(n_u + n_i, (rank + 1) * rows))
# replica-consistency check of the epoch's triples: identical on every rank passes, one differing element raises
ctx = DPContext(world, rank)
t = torch.from_numpy(np.stack(tri[0]))
ctx.check_replicated(t, "triples")
t2 = t.clone()
t2[1, 3] += rank                                    # rank 0 unchanged, rank 1 off by one
try:
    ctx.check_replicated(t2, "triples")
    raise SystemExit("diverged triples were not detected")
except RuntimeError as e:
    assert "differs between ranks" in str(e)
dist.barrier()
if rank == 0:
    print("DP_TRAIN_OK", world)
dist.destroy_process_group()
'''


def test_dp_train_world2_gloo(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, CR_ROOT=ROOT, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29621", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "DP_TRAIN_OK 2" in out.stdout


def test_dp_slices_cover_batch():
    from coldrec_amd.train import DPContext
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 4096, 4097):
            cuts = [DPContext(world, r).slice(n) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
