"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 python tests/golden/make_golden.py

The reference (YuanchenBei/ColdRec, pure Python) is imported in place -- nothing of
its source is copied; only inputs and the outputs it computes are stored, as small
``.npz`` files.  ``model/__init__.py`` of the reference imports faiss (absent here), so
``model`` is registered as a bare namespace package and ``model.MF`` /
``model.LightGCN`` are imported directly.

Fixtures (ids refer to SURVEY.md section 8(c)):
  toy_item.npz / toy_user.npz  the toy split fed to the reference + its id tables
  g1_sampler.npz   util/utils.py:123-157   next_batch_pairwise triples, 3 epochs
  g2_loss.npz      util/utils.py:25-29,44-48  bpr_loss / l2_reg_loss + autograd grads
  g3_mf.npz        model/MF.py:12-29       per-step losses, tables after 1/10/50 Adam steps
  g4_graph.npz     util/databuilder.py:220-254  normalised bipartite adjacency (CSR)
  g5_lgcn.npz      model/LightGCN.py:86-96  encoder outputs L=1..3, 20 training steps
  g6_eval_*.npz    model/BaseRecommender.py:109-188  _evaluate top-k for all/warm/cold
  g7_metrics.npz   util/evaluator.py:153-187  ranking_evaluation on the g6 lists
  g8_e2e.json      model/BaseRecommender.py:353-370  MF.run() d=64, 3 epochs, test metrics
  g9_dropoutnet.*  model/DropoutNet.py:12-72  DropoutNet.run() on the g8 tables, 2 epochs: losses, tables, metrics
  g10_samplers.npz util/utils.py:160-336   next_batch_pairwise_LARA / _CLCRec / _CCFCRec / next_batch_cgrc, 2 epochs each
  g11_lgcn_e2e.*   model/LightGCN.py:14-51 + BaseRecommender.py:353-370  LightGCN.run() L=3, d=64, 3 epochs: losses,
                   metrics, best-epoch snapshot tables, final top-20 lists of the three test settings
  g12_*_real_size.npz  model/MF.py:12-46 / model/LightGCN.py:14-47  trainer.train() at the BASELINE config sizes
                   (MovieLens shape d=128 B=4096; CiteULike shape L=3 d=128), 2 whole epochs each: every batch's loss terms,
                   table norms every 10 steps, sampled rows per epoch, the per-epoch validation metrics
  g15_dropoutnet_real_size.npz  model/DropoutNet.py:13-236 through run() on the CiteULike-shaped split with content, d=128
  g14_graph_real_size.npz  util/databuilder.py:220-254  checksums of the reference's normalised adjacency + id tables at the
                   MovieLens / CiteULike shapes
  g13_eval_100k.npz  model/BaseRecommender.py:109-188  _evaluate for 512 users x 100 000 items, d=128, all / warm / cold:
                   the reference's top-20 ids and scores (tables and split regenerated from seeds on the other side)
  g16_*            cold_object=user through run() (MF toy + CiteULike shape, DropoutNet with user content) and main.py --runs 2
  g17_ngcf.npz     model/NGCF.py:15-104  NGCF.run() on the toy split: a second consumer of the SpMM / BPR / ranking hooks
  g8_lists.npz / g9_lists.npz  model/BaseRecommender.py:153-188  the final top-20 lists of the g8 / g9 runs (re-run,
                   tables asserted equal to the stored ones) with the eval inputs needed to re-derive rank margins
"""
import json
import os
import sys
import types
from argparse import Namespace

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np
import torch

torch.set_num_threads(1)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
_m = types.ModuleType("model")
_m.__path__ = [os.path.join(REF, "model")]
sys.modules["model"] = _m

from util.utils import next_batch_pairwise, bpr_loss, l2_reg_loss, set_seed  # noqa: E402  (reference)
from util.utils import (next_batch_pairwise_LARA, next_batch_pairwise_CLCRec,  # noqa: E402  (reference)
                        next_batch_pairwise_CCFCRec, next_batch_cgrc)
from util.databuilder import ColdStartDataBuilder, TorchGraphInterface  # noqa: E402  (reference)
from util.evaluator import ranking_evaluation  # noqa: E402  (reference)
from model.MF import MF, Matrix_Factorization  # noqa: E402  (reference)
from model.LightGCN import LightGCN, LGCN_Encoder  # noqa: E402  (reference)

from coldrec_amd.data.synth import make_dataset  # noqa: E402  (ours: input generator only)


def out(name):
    return os.path.join(HERE, name)


def ref_builder(split):
    info = split.info
    return ColdStartDataBuilder(
        split.as_lists("warm_train"), split.as_lists("warm_val"), split.as_lists("cold_val"),
        split.as_lists("overall_val"), split.as_lists("warm_test"), split.as_lists("cold_test"),
        split.as_lists("overall_test"), info["user_num"], info["item_num"],
        info["warm_user"], info["warm_item"], info["cold_user"], info["cold_item"],
        split.content if split.cold_object == "user" else None,
        split.content if split.cold_object == "item" else None)


def ref_args(**kw):
    a = dict(dataset="toy", model="MF", epochs=3, layers=2, topN="10,20", bs=512, emb_size=64,
             lr=0.001, reg=0.0001, runs=1, seed=2024, use_gpu=False, save_emb=False, gpu_id=0,
             cold_object="item", backbone="MF", early_stop=10, eval_every=1)
    a.update(kw)
    return Namespace(**a)


def ref_config(data, **kw):
    return types.SimpleNamespace(args=ref_args(**kw), data=data, device=torch.device("cpu"))


def dump_split(split, data, fname):
    """Inputs + the id tables / sets the reference builder derived from them."""
    n_u, n_i = len(data.user), len(data.item)
    user_keys = np.array([data.id2user[k] for k in range(n_u)], dtype=np.int64)
    item_keys = np.array([data.id2item[k] for k in range(n_i)], dtype=np.int64)
    np.savez_compressed(
        out(fname), cold_object=split.cold_object, user_num=split.user_num, item_num=split.item_num,
        warm_train=split.warm_train, warm_val=split.warm_val, warm_test=split.warm_test,
        cold_val=split.cold_val, cold_test=split.cold_test, overall_val=split.overall_val,
        overall_test=split.overall_test, warm_user=split.info["warm_user"],
        warm_item=split.info["warm_item"], cold_user=split.info["cold_user"],
        cold_item=split.info["cold_item"],
        content=split.content if split.content is not None else np.zeros((0, 0), np.float32),
        user_keys=user_keys, item_keys=item_keys,
        mapped_warm_user_idx=np.asarray(data.mapped_warm_user_idx, dtype=np.int64),
        mapped_warm_item_idx=np.asarray(data.mapped_warm_item_idx, dtype=np.int64),
        mapped_cold_user_idx=np.asarray(data.mapped_cold_user_idx, dtype=np.int64),
        mapped_cold_item_idx=np.asarray(data.mapped_cold_item_idx, dtype=np.int64))


def g1_sampler(split):
    data = ref_builder(split)          # fresh builder: training_data order untouched
    np.random.seed(2024)
    u, i, j, sizes = [], [], [], []
    for _epoch in range(3):
        for bu, bi, bj in next_batch_pairwise(data, 1024):
            u += bu; i += bi; j += bj; sizes.append(len(bu))
    tail = np.random.randint(0, 1 << 30, size=4)   # RNG state probe after the 3 epochs
    np.savez_compressed(out("g1_sampler.npz"), seed=2024, batch_size=1024, epochs=3,
                        u=np.array(u, np.int64), i=np.array(i, np.int64), j=np.array(j, np.int64),
                        sizes=np.array(sizes, np.int64), rng_tail=tail)


def g2_loss():
    g = torch.Generator().manual_seed(7)
    res = {}
    cases = {
        "rand": (64, 16, 1.0, 1e-4),
        "reg": (48, 32, 0.3, 0.05),
        "sat": (32, 8, 40.0, 1e-4),     # saturated sigmoid on both sides
    }
    for name, (B, d, scale, reg) in cases.items():
        u = (torch.randn(B, d, generator=g) * scale).requires_grad_()
        p = (torch.randn(B, d, generator=g) * scale).requires_grad_()
        n = (torch.randn(B, d, generator=g) * scale).requires_grad_()
        lb = bpr_loss(u, p, n)
        lr_ = l2_reg_loss(reg, u, p, n)
        (lb + lr_).backward()
        res.update({f"{name}_u": u.detach().numpy(), f"{name}_p": p.detach().numpy(),
                    f"{name}_n": n.detach().numpy(), f"{name}_reg": np.float64(reg),
                    f"{name}_bpr": lb.detach().numpy(), f"{name}_l2": lr_.detach().numpy(),
                    f"{name}_gu": u.grad.numpy(), f"{name}_gp": p.grad.numpy(),
                    f"{name}_gn": n.grad.numpy()})
    # gather from tables with duplicate rows: dense table gradients (index backward accumulates)
    U = (torch.randn(10, 16, generator=g) * 0.5).requires_grad_()
    V = (torch.randn(12, 16, generator=g) * 0.5).requires_grad_()
    ui = [0, 3, 3, 9, 1, 3, 0, 7]
    pi = [2, 2, 5, 11, 0, 2, 4, 4]
    ni = [5, 1, 1, 0, 11, 5, 5, 2]
    ue, pe, ne = U[ui], V[pi], V[ni]
    loss = bpr_loss(ue, pe, ne) + l2_reg_loss(0.01, ue, pe, ne)
    loss.backward()
    res.update(dup_U=U.detach().numpy(), dup_V=V.detach().numpy(), dup_ui=np.array(ui),
               dup_pi=np.array(pi), dup_ni=np.array(ni), dup_reg=np.float64(0.01),
               dup_loss=loss.detach().numpy(), dup_gU=U.grad.numpy(), dup_gV=V.grad.numpy())
    np.savez_compressed(out("g2_loss.npz"), **res)


def g3_mf(split):
    res = {}
    steps_keep = (1, 10, 50)
    for d in (16, 64):
        data = ref_builder(split)
        set_seed(2024, False)
        model = Matrix_Factorization(data, d)
        U0 = model.embedding_dict["user_emb"].detach().clone().numpy()
        V0 = model.embedding_dict["item_emb"].detach().clone().numpy()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        losses, tu, ti, tj, sizes = [], [], [], [], []
        step = 0
        done = False
        while not done:
            for bu, bi, bj in next_batch_pairwise(data, 512):
                ue_all, ie_all = model()
                ue, pe, ne = ue_all[bu], ie_all[bi], ie_all[bj]
                loss = bpr_loss(ue, pe, ne) + l2_reg_loss(1e-4, ue, pe, ne)
                opt.zero_grad(); loss.backward(); opt.step()
                step += 1
                losses.append(loss.item())
                tu += bu; ti += bi; tj += bj; sizes.append(len(bu))
                if step in steps_keep:
                    res[f"d{d}_U_step{step}"] = model.embedding_dict["user_emb"].detach().clone().numpy()
                    res[f"d{d}_V_step{step}"] = model.embedding_dict["item_emb"].detach().clone().numpy()
                if step == max(steps_keep):
                    done = True
                    break
        res.update({f"d{d}_U0": U0, f"d{d}_V0": V0, f"d{d}_loss": np.array(losses, np.float64)})
        if d == 16:
            res.update(u=np.array(tu, np.int32), i=np.array(ti, np.int32), j=np.array(tj, np.int32),
                       sizes=np.array(sizes, np.int64))
        else:   # the triples do not depend on d (separate numpy stream, same seed)
            assert np.array_equal(res["u"], np.array(tu, np.int32))
    res.update(lr=1e-3, reg=1e-4, batch_size=512, seed=2024)
    np.savez_compressed(out("g3_mf.npz"), **res)


def g4_graph(split):
    data = ref_builder(split)
    adj = data.norm_adj.tocsr()
    adj.sort_indices()
    coo = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj)
    np.savez_compressed(out("g4_graph.npz"), n=adj.shape[0], indptr=adj.indptr.astype(np.int64),
                        indices=adj.indices.astype(np.int64), data=adj.data.astype(np.float32),
                        coo_rows=coo.indices()[0].numpy(), coo_cols=coo.indices()[1].numpy(),
                        coo_vals=coo.values().numpy())


def g5_lgcn(split):
    res = {}
    data = ref_builder(split)
    for L in (1, 2, 3):
        set_seed(2024, False)
        enc = LGCN_Encoder(data, 32, L, torch.device("cpu"))
        with torch.no_grad():
            uo, io = enc()
        if L == 1:
            res["U0"] = enc.embedding_dict["user_emb"].detach().clone().numpy()
            res["V0"] = enc.embedding_dict["item_emb"].detach().clone().numpy()
        res[f"L{L}_user_out"] = uo.numpy().copy()
        res[f"L{L}_item_out"] = io.numpy().copy()
    # 20 training steps, L=3, d=32 (LightGCN.py:14-29)
    data = ref_builder(split)
    set_seed(2024, False)
    enc = LGCN_Encoder(data, 32, 3, torch.device("cpu"))
    opt = torch.optim.Adam(enc.parameters(), lr=1e-3)
    losses, tu, ti, tj, sizes = [], [], [], [], []
    gU1 = gV1 = None
    step = 0
    while step < 20:
        for bu, bi, bj in next_batch_pairwise(data, 512):
            ua, ia = enc()
            ue, pe, ne = ua[bu], ia[bi], ia[bj]
            loss = bpr_loss(ue, pe, ne) + l2_reg_loss(1e-4, ue, pe, ne)
            opt.zero_grad(); loss.backward()
            if step == 0:
                gU1 = enc.embedding_dict["user_emb"].grad.detach().clone().numpy()
                gV1 = enc.embedding_dict["item_emb"].grad.detach().clone().numpy()
            opt.step()
            step += 1
            losses.append(loss.item())
            tu += bu; ti += bi; tj += bj; sizes.append(len(bu))
            if step == 20:
                break
    res.update(train_loss=np.array(losses, np.float64), train_u=np.array(tu, np.int32),
               train_i=np.array(ti, np.int32), train_j=np.array(tj, np.int32),
               train_sizes=np.array(sizes, np.int64), train_gU_step1=gU1, train_gV_step1=gV1,
               train_U_end=enc.embedding_dict["user_emb"].detach().numpy().copy(),
               train_V_end=enc.embedding_dict["item_emb"].detach().numpy().copy(),
               lr=1e-3, reg=1e-4, batch_size=512)
    np.savez_compressed(out("g5_lgcn.npz"), **res)


def _rank_gap_ok(U, V, users_int, S_masked, k, d):
    """Rigorous margin: any fp32 summation order of a length-d dot product errs by at most
    gamma_d * sum_k |u_k v_k| (gamma_d = d*2^-24/(1-d*2^-24)); the top-(k+1) adjacent gaps of
    every row must exceed twice that, so the ranking is the same for MKL, fmaf chains and fp64."""
    gam = d * 2.0 ** -24 / (1 - d * 2.0 ** -24)
    err = gam * (np.abs(U[users_int]).astype(np.float64) @ np.abs(V).T.astype(np.float64)).max(axis=1)
    top = -np.sort(-S_masked, axis=1)[:, :k + 1]
    gaps = top[:, :-1] - top[:, 1:]
    real = top[:, 1:] > -1e8          # gaps between two masked (-1e9) entries are ties by design
    gaps = np.where(real, gaps, np.inf)
    return bool((gaps.min(axis=1) > 2 * err).all()), float(gaps.min()), float(err.max())


def g6_eval(split, tag, mode):
    """mode: 'cont' continuous N(0,.3) with verified rank margin (seed searched);
    'fine' multiples of 2^-10 (all dot products exact in fp32, ties absent -- verified);
    'quant' multiples of 2^-4 (exact, many ties: compare per-score multisets)."""
    data = ref_builder(split)
    d = 16 if mode == "cont" else 32
    seed = {"cont": 100, "fine": 200, "quant": 300}[mode]
    while True:
        cfg = ref_config(data, cold_object=split.cold_object, emb_size=d, bs=100)
        trainer = MF(cfg)
        rng = np.random.default_rng(seed)
        if mode == "quant":
            U = rng.integers(-16, 17, size=(data.user_num, d)).astype(np.float32) / 16
            V = rng.integers(-16, 17, size=(data.item_num, d)).astype(np.float32) / 16
        elif mode == "fine":
            U = np.round(rng.standard_normal((data.user_num, d)) * 0.3 * 1024).clip(-1024, 1024).astype(np.float32) / 1024
            V = np.round(rng.standard_normal((data.item_num, d)) * 0.3 * 1024).clip(-1024, 1024).astype(np.float32) / 1024
        else:
            U = rng.standard_normal((data.user_num, d)).astype(np.float32) * 0.3
            V = rng.standard_normal((data.item_num, d)).astype(np.float32) * 0.3
        trainer.user_emb, trainer.item_emb = torch.from_numpy(U), torch.from_numpy(V)
        res = dict(U=U, V=V, k=trainer.max_N, cold_object=split.cold_object, seed=seed, mode=mode)
        lists = {}
        ok_all = True
        for t in ("all", "warm", "cold"):
            test_set = {"all": data.overall_test_set, "warm": data.warm_test_set, "cold": data.cold_test_set}[t]
            rec = trainer.test(t)
            users = list(test_set.keys())
            assert list(rec.keys()) == users
            idx = np.array([[data.item[it] for it, _ in rec[u]] for u in users], np.int64)
            sc = np.array([[s for _, s in rec[u]] for u in users], np.float32)
            cache = trainer._get_eval_cache(test_set, t)
            rated = [r.numpy() if r is not None else np.zeros(0, np.int64) for r in cache["rated_item_ids"]]
            rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rated])]).astype(np.int64)
            cand = cache["candidate_mask"].numpy() if cache["candidate_mask"] is not None else np.zeros(0, np.int64)
            users_int = np.array([data.user[u] for u in users], np.int64)
            res.update({f"{t}_users": np.array(users, np.int64), f"{t}_users_int": users_int,
                        f"{t}_rated_rowptr": rowptr,
                        f"{t}_rated_col": np.concatenate(rated).astype(np.int64) if rated else np.zeros(0, np.int64),
                        f"{t}_cand": cand.astype(np.int64), f"{t}_idx": idx, f"{t}_score": sc})
            lists[t] = (test_set, rec)
            if mode in ("cont", "fine"):
                S = U[users_int].astype(np.float64) @ V.T.astype(np.float64)
                for r in range(len(users)):
                    S[r, rated[r]] = -1e9
                if cand.size:
                    S[:, cand] = -1e9
                ok, gap, err = _rank_gap_ok(U, V, users_int, S, trainer.max_N, d)
                if mode == "fine":   # exact arithmetic: only ties could reorder
                    ok = gap > 0
                res[f"{t}_min_gap"] = gap
                res[f"{t}_max_err"] = err
                ok_all &= ok
        if ok_all:
            break
        seed += 1
    np.savez_compressed(out(f"g6_eval_{tag}.npz"), **res)
    return lists


def g6_small():
    """A user with fewer than k unmasked candidates: masked items are returned (-1e9)."""
    rng = np.random.default_rng(5)
    n_user, n_item = 6, 26
    train = [[u, i, 1.0] for u in range(n_user) for i in rng.choice(n_item - 4, size=4 + 3 * u, replace=False)]
    cold_items = list(range(n_item - 4, n_item))
    test = [[u, cold_items[u % 4], 1.0] for u in range(n_user)]
    warm_items = sorted({t[1] for t in train})
    data = ColdStartDataBuilder(train, test, test, test, test, test, test, n_user, n_item,
                                list(range(n_user)), warm_items, [], cold_items, None,
                                np.zeros((n_item, 2), np.float32))
    cfg = ref_config(data, cold_object="item", emb_size=8, bs=4)
    trainer = MF(cfg)
    U = rng.standard_normal((data.user_num, 8)).astype(np.float32)
    V = rng.standard_normal((data.item_num, 8)).astype(np.float32)
    trainer.user_emb, trainer.item_emb = torch.from_numpy(U), torch.from_numpy(V)
    res = dict(U=U, V=V, k=trainer.max_N)
    for t in ("all", "warm", "cold"):
        rec = trainer.test(t)
        users = list(data.overall_test_set.keys())
        cache = trainer._get_eval_cache(data.overall_test_set, t)
        rated = [r.numpy() if r is not None else np.zeros(0, np.int64) for r in cache["rated_item_ids"]]
        rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rated])]).astype(np.int64)
        cand = cache["candidate_mask"].numpy() if cache["candidate_mask"] is not None else np.zeros(0, np.int64)
        res.update({f"{t}_users_int": np.array([data.user[u] for u in users], np.int64),
                    f"{t}_rated_rowptr": rowptr, f"{t}_rated_col": np.concatenate(rated).astype(np.int64),
                    f"{t}_cand": cand.astype(np.int64),
                    f"{t}_idx": np.array([[data.item[it] for it, _ in rec[u]] for u in users], np.int64),
                    f"{t}_score": np.array([[s for _, s in rec[u]] for u in users], np.float32)})
    np.savez_compressed(out("g6_eval_small.npz"), **res)


def g7_metrics(lists):
    res = {}
    for t, (test_set, rec) in lists.items():
        users = list(test_set.keys())
        measure, perf = ranking_evaluation(test_set, rec, [10, 20])
        gt_rowptr = np.concatenate([[0], np.cumsum([len(test_set[u]) for u in users])]).astype(np.int64)
        gt_items = np.array([it for u in users for it in test_set[u].keys()], np.int64)
        pred = np.array([[it for it, _ in rec[u]] for u in users], np.int64)
        res.update({f"{t}_gt_rowptr": gt_rowptr, f"{t}_gt_items": gt_items, f"{t}_pred": pred,
                    f"{t}_perf": np.array(perf, np.float64), f"{t}_measure": np.array(measure)})
    np.savez_compressed(out("g7_metrics.npz"), **res)


def g8_e2e(split):
    data = ref_builder(split)
    cfg = ref_config(data, emb_size=64, epochs=3, bs=512)
    set_seed(2024, False)
    trainer = MF(cfg)
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trainer.run()
    log = buf.getvalue()
    loss_lines = [ln for ln in log.splitlines() if ln.startswith("training:")]
    payload = dict(
        args=vars(cfg.args), overall=trainer.overall_test_results, cold=trainer.cold_test_results,
        warm=trainer.warm_test_results, best=[trainer.bestPerformance[0], trainer.bestPerformance[1]],
        epochs_ran=trainer.epochs_ran, loss_lines=loss_lines,
        user_emb_norm=float(trainer.user_emb.norm()), item_emb_norm=float(trainer.item_emb.norm()))
    with open(out("g8_e2e.json"), "w") as f:
        json.dump(payload, f, indent=1)
    np.savez_compressed(out("g8_e2e_emb.npz"), U=trainer.user_emb.detach().numpy(),
                        V=trainer.item_emb.detach().numpy())


def g9_dropoutnet(split):
    """The reference's DropoutNet on the toy split, backbone tables = the g8 MF tables."""
    import contextlib
    import io
    import tempfile
    from model.DropoutNet import DropoutNet  # noqa: E402  (reference)
    data = ref_builder(split)
    emb = np.load(out("g8_e2e_emb.npz"))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "emb"))
        torch.save(torch.nn.Parameter(torch.from_numpy(emb["U"])), os.path.join(tmp, "emb", "toy_cold_item_MF_user_emb.pt"))
        torch.save(torch.nn.Parameter(torch.from_numpy(emb["V"])), os.path.join(tmp, "emb", "toy_cold_item_MF_item_emb.pt"))
        os.chdir(tmp)
        try:
            cfg = ref_config(data, model="DropoutNet", emb_size=64, epochs=3, bs=128, n_dropout=0.5,
                             dropoutnet_hidden1=200, dropoutnet_hidden2=100)
            set_seed(2024, False)
            trainer = DropoutNet(cfg)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                trainer.run()
        finally:
            os.chdir(cwd)
    loss_lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("training:")]
    payload = dict(args=vars(cfg.args), overall=trainer.overall_test_results, cold=trainer.cold_test_results,
                   warm=trainer.warm_test_results, best=[trainer.bestPerformance[0], trainer.bestPerformance[1]],
                   epochs_ran=trainer.epochs_ran, loss_lines=loss_lines,
                   user_emb_norm=float(trainer.user_emb.norm()), item_emb_norm=float(trainer.item_emb.norm()))
    with open(out("g9_dropoutnet.json"), "w") as f:
        json.dump(payload, f, indent=1)
    np.savez_compressed(out("g9_dropoutnet_emb.npz"), U=trainer.user_emb.detach().numpy(),
                        V=trainer.item_emb.detach().numpy())


def g10_samplers(split):
    """The other samplers: 2 epochs each from random.seed(2024) / np.random.seed(2024) on a fresh
    builder, batch 1000 (short last batch), plus a probe of both generators afterwards."""
    import random
    res = dict(seed=2024, batch_size=1000, epochs=2)

    def run(tag, fn, *a):
        data = ref_builder(split)
        random.seed(2024)
        np.random.seed(2024)
        batches = [b for _ in range(2) for b in fn(data, 1000, *a)]
        res[tag + "_tail"] = np.array([random.getrandbits(30), int(np.random.randint(0, 1 << 30))], np.int64)
        res[tag + "_sizes"] = np.array([len(b[0]) for b in batches], np.int64)
        return batches

    cat = lambda bs, k: np.concatenate([np.asarray(b[k], np.int16) for b in bs], 0)   # toy ids are < 500
    b = run("lara", next_batch_pairwise_LARA)
    for k, name in enumerate(("u", "i", "nu", "ni")):
        res["lara_" + name] = cat(b, k)
    b = run("lara3", next_batch_pairwise_LARA, 3)
    for k, name in enumerate(("u", "i", "nu", "ni")):
        res["lara3_" + name] = cat(b, k)
    for tag, n_negs in (("clc1", 1), ("clc8", 8)):               # 8 > 5: the enlarged set-size branch of random.sample
        b = run(tag, next_batch_pairwise_CLCRec, n_negs)
        res[tag + "_u"], res[tag + "_i"] = cat(b, 0), cat(b, 1)
    b = run("ccf", next_batch_pairwise_CCFCRec, 3, 4, 5)
    for k, name in enumerate(("u", "i", "nu", "pos", "neg", "sneg")):
        res["ccf_" + name] = cat(b, k)
    for tag, r in (("cgrc32", 32), ("cgrc2", 2)):
        b = run(tag, next_batch_cgrc, r)
        res[tag + "_u"], res[tag + "_i"] = cat(b, 0), cat(b, 1)
        res[tag + "_bset"] = cat(b, 2)
        res[tag + "_bptr"] = np.cumsum([0] + [len(x[2]) for x in b]).astype(np.int64)
    np.savez_compressed(out("g10_samplers.npz"), **res)


def _final_lists(trainer, data):
    """trainer.test(t) for the three settings -> ids (internal), fp32 scores, eval users, rated CSR, candidate mask."""
    res = {}
    for t in ("all", "cold", "warm"):
        test_set = {"all": data.overall_test_set, "warm": data.warm_test_set, "cold": data.cold_test_set}[t]
        rec = trainer.test(t)
        users = list(test_set.keys())
        assert list(rec.keys()) == users
        cache = trainer._get_eval_cache(test_set, t)
        rated = [r.numpy() if r is not None else np.zeros(0, np.int64) for r in cache["rated_item_ids"]]
        cand = cache["candidate_mask"].numpy() if cache["candidate_mask"] is not None else np.zeros(0, np.int64)
        res.update({
            f"{t}_users_int": np.array([data.user[u] for u in users], np.int64),
            f"{t}_rated_rowptr": np.concatenate([[0], np.cumsum([len(r) for r in rated])]).astype(np.int64),
            f"{t}_rated_col": np.concatenate(rated).astype(np.int64) if rated else np.zeros(0, np.int64),
            f"{t}_cand": cand.astype(np.int64),
            f"{t}_idx": np.array([[data.item[it] for it, _ in rec[u]] for u in users], np.int64),
            f"{t}_score": np.array([[sc for _, sc in rec[u]] for u in users], np.float32)})
    return res


def _run_quiet(trainer):
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trainer.run()
    return [ln for ln in buf.getvalue().splitlines() if ln.startswith("training:")]


def g8_lists(split):
    """The g8 run again (same seeds -> same tables, asserted) for the lists its final tables produce."""
    data = ref_builder(split)
    cfg = ref_config(data, emb_size=64, epochs=3, bs=512)
    set_seed(2024, False)
    trainer = MF(cfg)
    _run_quiet(trainer)
    emb = np.load(out("g8_e2e_emb.npz"))
    assert np.array_equal(trainer.user_emb.detach().numpy(), emb["U"]) and np.array_equal(trainer.item_emb.detach().numpy(), emb["V"])
    np.savez_compressed(out("g8_lists.npz"), **_final_lists(trainer, data))


def g9_lists(split):
    import tempfile
    from model.DropoutNet import DropoutNet  # noqa: E402  (reference)
    data = ref_builder(split)
    emb = np.load(out("g8_e2e_emb.npz"))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "emb"))
        torch.save(torch.nn.Parameter(torch.from_numpy(emb["U"])), os.path.join(tmp, "emb", "toy_cold_item_MF_user_emb.pt"))
        torch.save(torch.nn.Parameter(torch.from_numpy(emb["V"])), os.path.join(tmp, "emb", "toy_cold_item_MF_item_emb.pt"))
        os.chdir(tmp)
        try:
            cfg = ref_config(data, model="DropoutNet", emb_size=64, epochs=3, bs=128, n_dropout=0.5,
                             dropoutnet_hidden1=200, dropoutnet_hidden2=100)
            set_seed(2024, False)
            trainer = DropoutNet(cfg)
            _run_quiet(trainer)
        finally:
            os.chdir(cwd)
    want = np.load(out("g9_dropoutnet_emb.npz"))
    assert np.array_equal(trainer.user_emb.detach().numpy(), want["U"]) and np.array_equal(trainer.item_emb.detach().numpy(), want["V"])
    np.savez_compressed(out("g9_lists.npz"), **_final_lists(trainer, data))


def g11_lgcn_e2e(split):
    """model/LightGCN.py:14-51 through BaseRecommender.run(): config 3's trainer (L=3) on the toy split."""
    data = ref_builder(split)
    cfg = ref_config(data, model="LightGCN", layers=3, emb_size=64, epochs=3, bs=512)
    set_seed(2024, False)
    trainer = LightGCN(cfg)
    loss_lines = _run_quiet(trainer)
    payload = dict(
        args=vars(cfg.args), overall=trainer.overall_test_results, cold=trainer.cold_test_results,
        warm=trainer.warm_test_results, best=[trainer.bestPerformance[0], trainer.bestPerformance[1]],
        epochs_ran=trainer.epochs_ran, loss_lines=loss_lines,
        user_emb_norm=float(trainer.user_emb.norm()), item_emb_norm=float(trainer.item_emb.norm()))
    with open(out("g11_lgcn_e2e.json"), "w") as f:
        json.dump(payload, f, indent=1)
    enc = trainer.model
    np.savez_compressed(out("g11_lgcn_e2e_emb.npz"), U=trainer.user_emb.detach().numpy(), V=trainer.item_emb.detach().numpy(),
                        E0_user=enc.embedding_dict["user_emb"].detach().numpy(),
                        E0_item=enc.embedding_dict["item_emb"].detach().numpy(), **_final_lists(trainer, data))


def _crc(*arrays):
    import zlib
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def g12_real_size(which):
    """the reference's OWN training loops at the BASELINE config sizes, whole epochs.
      mf    model/MF.py:12-46        on the MovieLens-shaped split (6 040 x 3 706, ~6.4e5 triples), d=128, B=4096
      lgcn  model/LightGCN.py:14-47  on the CiteULike-shaped split (5 551 x 16 980, ~1.3e5 triples), L=3, d=128, B=4096
      mf64run  BASELINE configs[0] (BPR-MF, cold_object=item, d=64) through ``run()`` (model/BaseRecommender.py:353-370): the
            same recording + the final test metrics of the all / cold / warm settings at MovieLens size
      lgcnrun  BASELINE configs[2]'s trainer (LightGCN L=3, d=128) through ``run()`` at CiteULike size, likewise
    ``trainer.train()`` itself runs (2 epochs: the product's first epoch is eager, its second is captured into a hipGraph and
    replayed -- both are pinned), observed from outside: util.utils.bpr_loss / l2_reg_loss as the trainer's module sees them
    are wrapped to record every batch's two loss terms, next_batch_pairwise to checksum the triples, and a global optimizer
    post-step hook records the Frobenius norms of both tables every 10 steps, at the end of each epoch, and 256 sampled
    rows of each table at the end of each epoch.  The per-epoch validation (fast_evaluation, model/BaseRecommender.py:
    268-351) runs as in the reference and its metric lines are kept.  Stored: small arrays only (< 1 MB per config); the
    triples are NOT stored -- the product's sampler reproduces the NumPy stream bit for bit (G1) and the test checks the
    checksum before it trains."""
    import contextlib
    import importlib
    import io
    import time
    shape, cls_name, layers, seed = {"mf": ("movielens", "MF", 0, 1), "lgcn": ("citeulike", "LightGCN", 3, 2),
                                     "mf64run": ("movielens", "MF", 0, 1), "lgcnrun": ("citeulike", "LightGCN", 3, 2),
                                     # G16: cold_object=user through run() -- no candidate mask in _evaluate
                                     # (model/BaseRecommender.py:130-143), cold USERS ranked over the whole catalogue
                                     "mfuserrun": ("citeulike", "MF", 0, 2), "mfusertoyrun": ("toy", "MF", 0, 2),
                                     "lgcnuserrun": ("citeulike", "LightGCN", 3, 2)}[which]
    cold_object = "user" if "user" in which else "item"
    split = make_dataset(shape, cold_object, seed=seed, with_content=False)
    data = ref_builder(split)
    d, B, epochs = (64 if which in ("mf64run", "mfusertoyrun") else 128), (512 if shape == "toy" else 4096), (3 if shape == "toy" else 2)
    whole_run = which.endswith("run")       # through run(): + the three tests.  mf64run = BASELINE configs[0]: BPR-MF, cold_object=item, d=64 through run(): + the three tests
    cfg = ref_config(data, dataset=shape, model=cls_name, layers=layers or 2, emb_size=d, epochs=epochs, bs=B,
                     cold_object=cold_object)
    set_seed(2024, False)
    mod = importlib.import_module("model." + cls_name)
    trainer = getattr(mod, cls_name)(cfg)
    params = dict(trainer.model.embedding_dict.items())
    U0, V0 = params["user_emb"].detach().clone().numpy(), params["item_emb"].detach().clone().numpy()
    n_steps_epoch = -(-len(data.training_data) // B)
    rows_u = np.sort(np.random.default_rng(12).choice(U0.shape[0], 256, replace=False))
    rows_v = np.sort(np.random.default_rng(13).choice(V0.shape[0], 256, replace=False))
    rec = dict(bpr=[], l2=[], sizes=[], crc=0, norm_step=[], norm_U=[], norm_V=[], end_U=[], end_V=[], end_norm=[])
    real_bpr, real_l2, real_next = mod.bpr_loss, mod.l2_reg_loss, mod.next_batch_pairwise

    def bpr_spy(*a):
        out_ = real_bpr(*a)
        rec["bpr"].append(float(out_.item()))
        return out_

    def l2_spy(*a):
        out_ = real_l2(*a)
        rec["l2"].append(float(out_.item()))
        return out_

    def next_spy(*a, **kw):
        for bu, bi, bj in real_next(*a, **kw):
            rec["crc"] = _crc(np.array(bu, np.int32), np.array(bi, np.int32), np.array(bj, np.int32)) ^ (rec["crc"] * 31 & 0xFFFFFFFF)
            rec["sizes"].append(len(bu))
            yield bu, bi, bj

    def post_step(opt, args, kwargs):
        step = len(rec["sizes"])                       # batches drawn so far == optimiser steps done
        U, V = params["user_emb"].detach(), params["item_emb"].detach()
        if step % 10 == 0 or step % n_steps_epoch == 0:
            rec["norm_step"].append(step)
            rec["norm_U"].append(float(torch.linalg.norm(U.double())))
            rec["norm_V"].append(float(torch.linalg.norm(V.double())))
        if step % n_steps_epoch == 0:
            rec["end_U"].append(U[rows_u].clone().numpy())
            rec["end_V"].append(V[rows_v].clone().numpy())
            rec["end_norm"].append([float(torch.linalg.norm(U.double())), float(torch.linalg.norm(V.double()))])

    from torch.optim.optimizer import register_optimizer_step_post_hook
    mod.bpr_loss, mod.l2_reg_loss, mod.next_batch_pairwise = bpr_spy, l2_spy, next_spy
    handle = register_optimizer_step_post_hook(post_step)
    buf = io.StringIO()
    t0 = time.time()
    try:
        with contextlib.redirect_stdout(buf):
            if whole_run:
                trainer.run()           # model/BaseRecommender.py:353-370: train() + full_evaluation of all / cold / warm
            else:
                trainer.train()
    finally:
        handle.remove()
        mod.bpr_loss, mod.l2_reg_loss, mod.next_batch_pairwise = real_bpr, real_l2, real_next
    secs = time.time() - t0
    log = buf.getvalue().splitlines()
    assert len(rec["bpr"]) == len(rec["l2"]) == len(rec["sizes"]) == epochs * n_steps_epoch, (len(rec["bpr"]), n_steps_epoch)
    if layers:                                         # what the trainer ranks with: the propagated tables of the best epoch
        fin_U, fin_V = trainer.user_emb.detach().numpy(), trainer.item_emb.detach().numpy()
    else:
        fin_U, fin_V = None, None
    res = dict(
        which=which, shape=shape, d=d, batch_size=B, epochs=epochs, layers=layers, lr=cfg.args.lr, reg=cfg.args.reg, seed=2024,
        data_seed=seed, user_num=data.user_num, item_num=data.item_num, n_train=len(data.training_data),
        steps_per_epoch=n_steps_epoch, U0_crc=_crc(U0), V0_crc=_crc(V0), U0_norm=float(np.linalg.norm(U0.astype(np.float64))),
        V0_norm=float(np.linalg.norm(V0.astype(np.float64))), triples_crc=rec["crc"], sizes=np.array(rec["sizes"], np.int32),
        bpr=np.array(rec["bpr"], np.float64), l2=np.array(rec["l2"], np.float64),
        norm_step=np.array(rec["norm_step"], np.int32), norm_U=np.array(rec["norm_U"], np.float64),
        norm_V=np.array(rec["norm_V"], np.float64), rows_u=rows_u, rows_v=rows_v,
        end_U=np.stack(rec["end_U"]), end_V=np.stack(rec["end_V"]), end_norm=np.array(rec["end_norm"], np.float64),
        best_epoch=trainer.bestPerformance[0], best_metrics=json.dumps(trainer.bestPerformance[1]),
        valid_lines=json.dumps([ln for ln in log if "Valid" in ln or "valid" in ln or "NDCG" in ln][:40]),
        torch_version=torch.__version__)       # (no wall-clock field: the file regenerates byte for byte)
    if whole_run:
        res.update(test_overall=np.array(trainer.overall_test_results, np.float64),
                   test_cold=np.array(trainer.cold_test_results, np.float64),
                   test_warm=np.array(trainer.warm_test_results, np.float64), epochs_ran=trainer.epochs_ran)
    if layers:
        res.update(final_out_U=fin_U[rows_u], final_out_V=fin_V[rows_v],
                   final_out_norm=np.array([np.linalg.norm(fin_U.astype(np.float64)), np.linalg.norm(fin_V.astype(np.float64))]))
    if cold_object == "user":
        lists = _final_lists(trainer, data)
        res.update(cold_object="user")
        if shape == "toy":
            res.update(lists)                       # small: the lists themselves
        else:                                       # real size: their checksums + the first 64 users of each setting
            for t in ("all", "cold", "warm"):
                res.update({f"{t}_idx_crc": _crc(lists[f"{t}_idx"].astype(np.int32)), f"{t}_n_users": len(lists[f"{t}_users_int"]),
                            f"{t}_idx_head": lists[f"{t}_idx"][:64].astype(np.int32), f"{t}_score_head": lists[f"{t}_score"][:64]})
        np.savez_compressed(out("g16_%s.npz" % which), **res)
    else:
        np.savez_compressed(out("g12_%s_real_size.npz" % which), **res)
    print("g12 %s: %d steps of the reference in %.1f s; last losses bpr %.6f l2 %.3e; best %s"
          % (which, len(rec["bpr"]), secs, rec["bpr"][-1], rec["l2"][-1], trainer.bestPerformance))


def g13_pairs(n_user=512, n_item=100_000, extra=60_000, seed=13):
    """Interactions for G13 (shared with tests/test_g13_eval_gpu.py, which imports this function's twin): every item occurs at
    least once (an item without a record has no name in the reference's id tables), users uniform."""
    rng = np.random.default_rng(seed)
    u = np.concatenate([rng.integers(0, n_user, n_item), rng.integers(0, n_user, extra)])
    i = np.concatenate([np.arange(n_item), rng.integers(0, n_item, extra)])
    key = np.unique(u.astype(np.int64) * n_item + i)
    return np.stack([key // n_item, key % n_item], axis=1)


def g13_eval_100k():
    """model/BaseRecommender.py:109-188 at a catalogue that is not toy-sized: the reference's ``_evaluate`` (torch.matmul on the
    host -> per-user rated masks -> candidate mask -> torch.topk) for 512 users x 100 000 items, d = 128, the three test
    settings of a cold_object=item split.  Only the OUTPUT travels (ids + scores, 250 KB): tables, split and masks are
    regenerated on the other side from the same seeds (numpy PCG64 / the product's split_cold, both deterministic)."""
    from coldrec_amd.data.synth import split_cold
    import time
    t0 = time.time()
    split = split_cold(g13_pairs(), "item", seed=54)
    data = ref_builder(split)
    d = 128
    cfg = ref_config(data, dataset="g13", emb_size=d, bs=256)
    trainer = MF(cfg)
    rng = np.random.default_rng(1313)
    a_u, a_i = np.sqrt(6.0 / (data.user_num + d)), np.sqrt(6.0 / (data.item_num + d))
    U = ((rng.random((data.user_num, d)) * 2 - 1) * a_u * 8).astype(np.float32)      # xavier-shaped, scaled so that scores are O(0.1)
    V = ((rng.random((data.item_num, d)) * 2 - 1) * a_i * 8).astype(np.float32)
    trainer.user_emb, trainer.item_emb = torch.from_numpy(U), torch.from_numpy(V)
    res = dict(n_user=data.user_num, n_item=data.item_num, d=d, table_seed=1313, split_seed=54, pairs_seed=13,
               U_crc=_crc(U), V_crc=_crc(V), user_keys_crc=_crc(np.array([data.id2user[k] for k in range(len(data.user))], np.int64)),
               item_keys_crc=_crc(np.array([data.id2item[k] for k in range(len(data.item))], np.int64)))
    for t in ("all", "warm", "cold"):
        test_set = {"all": data.overall_test_set, "warm": data.warm_test_set, "cold": data.cold_test_set}[t]
        rec = trainer.test(t)
        users = list(test_set.keys())
        res[t + "_users_int"] = np.array([data.user[u] for u in users], np.int32)
        res[t + "_idx"] = np.array([[data.item[it] for it, _ in rec[u]] for u in users], np.int32)
        res[t + "_score"] = np.array([[sc for _, sc in rec[u]] for u in users], np.float32)
    np.savez_compressed(out("g13_eval_100k.npz"), **res)
    print("g13: %d users x %d items, settings all/warm/cold = %d/%d/%d users, %.1f s" % (
        data.user_num, data.item_num, len(res["all_users_int"]), len(res["warm_users_int"]), len(res["cold_users_int"]),
        time.time() - t0))


def g14_graph_real_size():
    """util/databuilder.py:220-254,953-962 at the BASELINE dataset shapes: checksums of the reference's normalised bipartite
    adjacency (CSR arrays, fp32 values) and of its id tables for the MovieLens- and CiteULike-shaped splits -- the product's
    builder must reproduce them bit for bit (G4 does this on the toy split)."""
    res = {}
    for shape, seed in (("movielens", 1), ("citeulike", 2)):
        split = make_dataset(shape, "item", seed=seed, with_content=False)
        data = ref_builder(split)
        adj = data.norm_adj.tocsr()
        adj.sort_indices()
        res.update({shape + "_n": adj.shape[0], shape + "_nnz": adj.nnz, shape + "_indptr_crc": _crc(adj.indptr.astype(np.int64)),
                    shape + "_indices_crc": _crc(adj.indices.astype(np.int64)), shape + "_data_crc": _crc(adj.data.astype(np.float32)),
                    shape + "_user_keys_crc": _crc(np.array([data.id2user[k] for k in range(len(data.user))], np.int64)),
                    shape + "_item_keys_crc": _crc(np.array([data.id2item[k] for k in range(len(data.item))], np.int64)),
                    shape + "_n_train": len(data.training_data), shape + "_data_seed": seed})
    np.savez_compressed(out("g14_graph_real_size.npz"), **res)
    print("g14:", {k: int(v) for k, v in res.items() if k.endswith("_nnz") or k.endswith("_n")})


def g15_backbone(user_num, item_num, d=128):
    """Stand-in backbone tables for G15 (what ./emb/*_MF_*_emb.pt would hold), drawn from a fixed stream so that neither the
    fixture nor the test has to store 11 MB of them (tests/test_e2e_gpu.py holds this function's twin)."""
    rng = np.random.default_rng(15)
    return (rng.standard_normal((user_num, d), dtype=np.float32) * np.float32(0.1),
            rng.standard_normal((item_num, d), dtype=np.float32) * np.float32(0.1))


def g15_dropoutnet_real_size():
    """SURVEY.md 8(f)3 at a real dataset shape: the reference's DropoutNet (model/DropoutNet.py:13-236) through ``run()`` on
    the CiteULike-shaped cold-item split WITH its 300-wide item content, d=128, batches of 1024, two epochs -- EVERY batch's
    loss (torch.nn.functional.mse_loss wrapped), the generated tables' norms + 256 sampled rows of each, and the test metrics of
    the three settings."""
    import contextlib
    import io
    import tempfile
    import time
    from model.DropoutNet import DropoutNet  # noqa: E402  (reference)
    split = make_dataset("citeulike", "item", seed=2, with_content=True)
    data = ref_builder(split)
    U, V = g15_backbone(data.user_num, data.item_num)
    cwd = os.getcwd()
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "emb"))
        torch.save(torch.nn.Parameter(torch.from_numpy(U)), os.path.join(tmp, "emb", "citeulike_cold_item_MF_user_emb.pt"))
        torch.save(torch.nn.Parameter(torch.from_numpy(V)), os.path.join(tmp, "emb", "citeulike_cold_item_MF_item_emb.pt"))
        os.chdir(tmp)
        try:
            cfg = ref_config(data, dataset="citeulike", model="DropoutNet", emb_size=128, epochs=2, bs=1024, n_dropout=0.5,
                             dropoutnet_hidden1=200, dropoutnet_hidden2=100)
            set_seed(2024, False)
            trainer = DropoutNet(cfg)
            buf = io.StringIO()
            every = []                       # every batch's loss: torch.nn.MSELoss.forward goes through F.mse_loss
            real_mse = torch.nn.functional.mse_loss

            def mse_spy(*a, **kw):
                r = real_mse(*a, **kw)
                every.append(float(r.item()))
                return r

            torch.nn.functional.mse_loss = mse_spy
            with contextlib.redirect_stdout(buf):
                trainer.run()
        finally:
            torch.nn.functional.mse_loss = real_mse
            os.chdir(cwd)
    secs = time.time() - t0
    loss_lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("training:")]
    gu, gv = trainer.user_emb.detach().numpy(), trainer.item_emb.detach().numpy()
    rows_u = np.sort(np.random.default_rng(12).choice(gu.shape[0], 256, replace=False))
    rows_v = np.sort(np.random.default_rng(13).choice(gv.shape[0], 256, replace=False))
    np.savez_compressed(
        out("g15_dropoutnet_real_size.npz"), data_seed=2, d=128, batch_size=1024, epochs=2, every_loss=np.array(every, np.float64), backbone_crc=_crc(U, V),
        user_num=data.user_num, item_num=data.item_num, n_train=len(data.training_data),
        losses=np.array([float(l.split("batch_loss:")[1]) for l in loss_lines], np.float64), loss_lines=json.dumps(loss_lines),
        epochs_ran=trainer.epochs_ran, best_epoch=trainer.bestPerformance[0], best_metrics=json.dumps(trainer.bestPerformance[1]),
        test_overall=np.array(trainer.overall_test_results, np.float64), test_cold=np.array(trainer.cold_test_results, np.float64),
        test_warm=np.array(trainer.warm_test_results, np.float64),
        norm=np.array([np.linalg.norm(gu.astype(np.float64)), np.linalg.norm(gv.astype(np.float64))]),
        rows_u=rows_u, rows_v=rows_v, gen_U=gu[rows_u], gen_V=gv[rows_v], scale=np.array([np.abs(gu).max(), np.abs(gv).max()]),
        torch_version=torch.__version__)       # (no wall-clock field: the file regenerates byte for byte)
    print("g15: the reference's DropoutNet at CiteULike size in %.1f s; %d loss lines, last %s; best %s"
          % (secs, len(loss_lines), loss_lines[-1] if loss_lines else None, trainer.bestPerformance))


def g16_dropoutnet_user():
    """the reference's DropoutNet with cold_object=user (model/DropoutNet.py:82-93,110-134: the USER tower takes
    the content, user rows are the ones dropped out) through run() on the CiteULike-shaped user-cold split with its 300-wide
    user content, d=128, batches of 1024, two epochs.  Recorded exactly like G15."""
    import contextlib
    import io
    import tempfile
    from model.DropoutNet import DropoutNet  # noqa: E402  (reference)
    split = make_dataset("citeulike", "user", seed=2, with_content=True)
    data = ref_builder(split)
    U, V = g15_backbone(data.user_num, data.item_num)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "emb"))
        torch.save(torch.nn.Parameter(torch.from_numpy(U)), os.path.join(tmp, "emb", "citeulike_cold_user_MF_user_emb.pt"))
        torch.save(torch.nn.Parameter(torch.from_numpy(V)), os.path.join(tmp, "emb", "citeulike_cold_user_MF_item_emb.pt"))
        os.chdir(tmp)
        try:
            cfg = ref_config(data, dataset="citeulike", model="DropoutNet", emb_size=128, epochs=2, bs=1024, n_dropout=0.5,
                             dropoutnet_hidden1=200, dropoutnet_hidden2=100, cold_object="user")
            set_seed(2024, False)
            trainer = DropoutNet(cfg)
            buf = io.StringIO()
            every = []
            real_mse = torch.nn.functional.mse_loss

            def mse_spy(*a, **kw):
                r = real_mse(*a, **kw)
                every.append(float(r.item()))
                return r

            torch.nn.functional.mse_loss = mse_spy
            with contextlib.redirect_stdout(buf):
                trainer.run()
        finally:
            torch.nn.functional.mse_loss = real_mse
            os.chdir(cwd)
    gu, gv = trainer.user_emb.detach().numpy(), trainer.item_emb.detach().numpy()
    rows_u = np.sort(np.random.default_rng(12).choice(gu.shape[0], 256, replace=False))
    rows_v = np.sort(np.random.default_rng(13).choice(gv.shape[0], 256, replace=False))
    np.savez_compressed(
        out("g16_dropoutnet_user.npz"), data_seed=2, d=128, batch_size=1024, epochs=2, every_loss=np.array(every, np.float64),
        backbone_crc=_crc(U, V), user_num=data.user_num, item_num=data.item_num, n_train=len(data.training_data),
        epochs_ran=trainer.epochs_ran, best_epoch=trainer.bestPerformance[0], best_metrics=json.dumps(trainer.bestPerformance[1]),
        test_overall=np.array(trainer.overall_test_results, np.float64), test_cold=np.array(trainer.cold_test_results, np.float64),
        test_warm=np.array(trainer.warm_test_results, np.float64),
        norm=np.array([np.linalg.norm(gu.astype(np.float64)), np.linalg.norm(gv.astype(np.float64))]),
        rows_u=rows_u, rows_v=rows_v, gen_U=gu[rows_u], gen_V=gv[rows_v], scale=np.array([np.abs(gu).max(), np.abs(gv).max()]),
        torch_version=torch.__version__)
    print("g16 dropoutnet user: %d batch losses, last %.6f; best %s; overall %s"
          % (len(every), every[-1], trainer.bestPerformance, trainer.overall_test_results))


def g16_runs2():
    """``python main.py --runs 2`` of the reference ITSELF (main.py:149-301 executed by runpy: Config from the
    CSV files our generator wrote, seed = round index, the data object -- and with it the cumulatively shuffled training_data --
    shared by both rounds, mean / std aggregation, one result block).  ``model`` is the bare namespace package of this
    script; its registry is filled with the reference's own classes (model/__init__.py would import faiss).
    Stored: every round's test metrics, the aggregated payload, the result block without its timestamp / timing lines, and the
    checksum of every round's FIRST batch of triples + of all its triples (model.MF.next_batch_pairwise wrapped)."""
    import contextlib
    import io
    import runpy
    import tempfile
    import model.MF as mf_mod
    from coldrec_amd.data.synth import write_dataset
    sys.modules["model"].AVAILABLE_MODELS = {"MF": MF, "LightGCN": LightGCN}
    split = make_dataset("toy", "item", seed=1)
    rec = dict(first=[], allcrc=[], n_batches=[])
    real_next = mf_mod.next_batch_pairwise

    def next_spy(*a, **kw):
        first = True
        for bu, bi, bj in real_next(*a, **kw):
            c = _crc(np.array(bu, np.int32), np.array(bi, np.int32), np.array(bj, np.int32))
            if first and rec["_new_round"]:
                rec["first"].append(c)
                rec["allcrc"].append(0)
                rec["n_batches"].append(0)
                rec["_new_round"] = False
            first = False
            rec["allcrc"][-1] = c ^ (rec["allcrc"][-1] * 31 & 0xFFFFFFFF)
            rec["n_batches"][-1] += 1
            yield bu, bi, bj

    real_run = MF.run

    def run_spy(self):
        rec["_new_round"] = True
        return real_run(self)

    cwd, argv = os.getcwd(), sys.argv
    with tempfile.TemporaryDirectory() as tmp:
        write_dataset(split, os.path.join(tmp, "data"), "toy")
        os.chdir(tmp)
        sys.argv = ["main.py", "--dataset", "toy", "--model", "MF", "--runs", "2", "--epochs", "3", "--bs", "512", "--emb_size",
                    "64", "--use_gpu", "false", "--save_emb", "false", "--cold_object", "item", "--result_file",
                    os.path.join(tmp, "res.txt"), "--result_overwrite"]
        mf_mod.next_batch_pairwise, MF.run = next_spy, run_spy
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                g = runpy.run_path(os.path.join(REF, "main.py"), run_name="__main__")
        finally:
            mf_mod.next_batch_pairwise, MF.run = real_next, real_run
            os.chdir(cwd)
            sys.argv = argv
        block = open(os.path.join(tmp, "res.txt"), encoding="utf-8").read()
    keep = [ln for ln in block.split("--- JSON")[0].splitlines()
            if not ln.startswith(("timestamp:", "seconds_per_completed", "result_file:"))]
    res = g["results"]
    per_run = np.array([[[res[s][m][i] for m in ("hit", "precision", "recall", "ndcg")] for i in range(len(g["top_Ns"]))]
                        for s in ("all", "cold", "warm")], np.float64)           # [setting][topN][metric][run]
    json.dump(dict(argv=["--dataset", "toy", "--model", "MF", "--runs", "2", "--epochs", "3", "--bs", "512", "--emb_size", "64",
                         "--cold_object", "item"],
                   per_run=per_run.tolist(), metrics=g["metrics_payload"], block_lines=keep,
                   first_batch_crc=rec["first"], all_triples_crc=rec["allcrc"], n_batches=rec["n_batches"],
                   console_tail=[ln for ln in buf.getvalue().splitlines()
                                 if ("±" in ln and not ln.startswith("Time:")) or ln.startswith(("Top-", "Start round"))]),
              open(out("g16_runs2.json"), "w"), indent=1, ensure_ascii=False)
    print("g16 runs2: per-run NDCG@20 overall", per_run[0, -1, 3].tolist(), "first-batch crcs", rec["first"])


def g17_ngcf(split):
    """VERDICT r4 #4 -- SURVEY.md 8(f)4 pinned to the reference NUMERICALLY: the reference's own NGCF.run()
    (model/NGCF.py:15-104: torch.sparse.mm over the normalised adjacency + two dense layers per hop + leaky_relu, bpr_loss +
    l2_reg_loss, Adam over tables AND weights, stock batch_predict) on the toy split, L=2, d=64, 3 epochs of 8 batches.
    Stored: the initial state_dict (tables from the xavier stream, the four nn.Linear from kaiming), every batch's two loss
    terms, the propagated tables' norms after every epoch, the weights' norms at the end, best epoch + validation metrics,
    the three test settings' metrics and 64 sampled rows of the final (best-epoch) tables."""
    import contextlib
    import importlib
    import io
    mod = importlib.import_module("model.NGCF")
    data = ref_builder(split)
    cfg = ref_config(data, model="NGCF", layers=2, emb_size=64, epochs=3, bs=512)
    set_seed(2024, False)
    trainer = mod.NGCF(cfg)
    init = {k: v.detach().clone().numpy() for k, v in trainer.model.state_dict().items() if k != "norm_adj"}
    rec = dict(bpr=[], l2=[], crc=0, sizes=[], epoch_norm=[])
    real_bpr, real_l2, real_next = mod.bpr_loss, mod.l2_reg_loss, mod.next_batch_pairwise
    real_fast = mod.NGCF.fast_evaluation

    def bpr_spy(*a):
        r = real_bpr(*a)
        rec["bpr"].append(float(r.item()))
        return r

    def l2_spy(*a):
        r = real_l2(*a)
        rec["l2"].append(float(r.item()))
        return r

    def next_spy(*a, **kw):
        for bu, bi, bj in real_next(*a, **kw):
            rec["crc"] = _crc(np.array(bu, np.int32), np.array(bi, np.int32), np.array(bj, np.int32)) ^ (rec["crc"] * 31 & 0xFFFFFFFF)
            rec["sizes"].append(len(bu))
            yield bu, bi, bj

    def fast_spy(self, *a, **kw):
        rec["epoch_norm"].append([float(torch.linalg.norm(self.user_emb.double())), float(torch.linalg.norm(self.item_emb.double()))])
        return real_fast(self, *a, **kw)

    mod.bpr_loss, mod.l2_reg_loss, mod.next_batch_pairwise, mod.NGCF.fast_evaluation = bpr_spy, l2_spy, next_spy, fast_spy
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            trainer.run()
    finally:
        mod.bpr_loss, mod.l2_reg_loss, mod.next_batch_pairwise, mod.NGCF.fast_evaluation = real_bpr, real_l2, real_next, real_fast
    fin = {k: v.detach().numpy() for k, v in trainer.model.state_dict().items() if k != "norm_adj"}
    U, V = trainer.user_emb.detach().numpy(), trainer.item_emb.detach().numpy()
    rows_u = np.sort(np.random.default_rng(12).choice(U.shape[0], 64, replace=False))
    rows_v = np.sort(np.random.default_rng(13).choice(V.shape[0], 64, replace=False))
    res = {"init__" + k: v for k, v in init.items()}
    res.update(layers=2, d=64, batch_size=512, epochs=3, lr=cfg.args.lr, reg=cfg.args.reg, seed=2024, triples_crc=rec["crc"],
               sizes=np.array(rec["sizes"], np.int32), bpr=np.array(rec["bpr"], np.float64), l2=np.array(rec["l2"], np.float64),
               epoch_norm=np.array(rec["epoch_norm"], np.float64),
               final_param_norm=np.array([np.linalg.norm(fin[k].astype(np.float64)) for k in sorted(fin)]),
               param_names=json.dumps(sorted(fin)), best_epoch=trainer.bestPerformance[0],
               best_metrics=json.dumps(trainer.bestPerformance[1]), epochs_ran=trainer.epochs_ran,
               test_overall=np.array(trainer.overall_test_results, np.float64), test_cold=np.array(trainer.cold_test_results, np.float64),
               test_warm=np.array(trainer.warm_test_results, np.float64), rows_u=rows_u, rows_v=rows_v, final_U=U[rows_u],
               final_V=V[rows_v], final_norm=np.array([np.linalg.norm(U.astype(np.float64)), np.linalg.norm(V.astype(np.float64))]),
               torch_version=torch.__version__)
    np.savez_compressed(out("g17_ngcf.npz"), **res)
    print("g17 ngcf: %d batches, last bpr %.6f l2 %.3e; best %s; overall %s" % (len(rec["bpr"]), rec["bpr"][-1], rec["l2"][-1],
                                                                           trainer.bestPerformance, trainer.overall_test_results))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "g17":
        g17_ngcf(make_dataset("toy", "item", seed=1))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g16":
        for which in ("mfusertoyrun", "mfuserrun"):
            g12_real_size(which)
        g16_dropoutnet_user()
        g16_runs2()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g15":
        g15_dropoutnet_real_size()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g14":
        g14_graph_real_size()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g13":
        g13_eval_100k()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g12":         # round 4: real-size whole-epoch fixtures (minutes of CPU)
        for which in (sys.argv[2:] or ["mf", "lgcn", "mf64run", "lgcnrun"]):
            g12_real_size(which)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g11":         # add the round-3 fixtures without redoing G1-G10
        split = make_dataset("toy", "item", seed=1)
        g8_lists(split)
        g9_lists(split)
        g11_lgcn_e2e(split)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g10":         # add the sampler fixture without redoing G1-G9
        g10_samplers(make_dataset("toy", "item", seed=1))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g9":          # add the DropoutNet fixture without redoing G1-G8
        g9_dropoutnet(make_dataset("toy", "item", seed=1))
        return
    split_i = make_dataset("toy", "item", seed=1)
    split_u = make_dataset("toy", "user", seed=2)
    dump_split(split_i, ref_builder(split_i), "toy_item.npz")
    dump_split(split_u, ref_builder(split_u), "toy_user.npz")
    g1_sampler(split_i)
    g2_loss()
    g3_mf(split_i)
    g4_graph(split_i)
    g5_lgcn(split_i)
    g6_eval(split_i, "item_cont", "cont")
    lists = g6_eval(split_i, "item_fine", "fine")
    g6_eval(split_i, "item_quant", "quant")
    g6_eval(split_u, "user_cont", "cont")
    g6_eval(split_u, "user_fine", "fine")
    g6_small()
    g7_metrics(lists)
    g8_e2e(split_i)
    g9_dropoutnet(split_i)
    g10_samplers(split_i)
    g8_lists(split_i)
    g9_lists(split_i)
    g11_lgcn_e2e(split_i)
    g12_real_size("mf")
    g12_real_size("lgcn")
    g12_real_size("mf64run")
    g12_real_size("lgcnrun")
    g13_eval_100k()
    g14_graph_real_size()
    g15_dropoutnet_real_size()
    for which in ("mfusertoyrun", "mfuserrun"):
        g12_real_size(which)
    g16_dropoutnet_user()
    g16_runs2()
    g17_ngcf(split_i)
    total = sum(os.path.getsize(out(f)) for f in os.listdir(HERE) if f.endswith((".npz", ".json")))
    print("golden vectors written, %.1f KiB" % (total / 1024))


if __name__ == "__main__":
    main()
