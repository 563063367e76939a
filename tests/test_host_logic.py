"""CPU: host-side logic around the hot path -- id tables and graph (golden toy_*, g4), array metrics
(golden g7), CLI / dataset files / result writer, early stopping, sharded-eval plumbing over gloo."""
import argparse
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from coldrec_amd.util.databuilder import ColdStartDataBuilder
from coldrec_amd.util.evaluator import ranking_evaluation, ranking_metrics
from tests.conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def builder(name="toy_item.npz"):
    g = load_golden(name)
    content = g["content"] if g["content"].size else None
    item_side = str(g["cold_object"]) == "item"
    return g, ColdStartDataBuilder(
        g["warm_train"], g["warm_val"], g["cold_val"], g["overall_val"], g["warm_test"], g["cold_test"],
        g["overall_test"], int(g["user_num"]), int(g["item_num"]), g["warm_user"], g["warm_item"],
        g["cold_user"], g["cold_item"], None if item_side else content, content if item_side else None)


@pytest.mark.parametrize("name", ["toy_item.npz", "toy_user.npz"])
def test_id_tables_match_reference(name):
    g, d = builder(name)
    assert np.array_equal(d.user_keys, g["user_keys"]) and np.array_equal(d.item_keys, g["item_keys"])
    for k in ("mapped_warm_user_idx", "mapped_warm_item_idx", "mapped_cold_user_idx", "mapped_cold_item_idx"):
        assert np.array_equal(np.asarray(getattr(d, k)), g[k])
    assert d.id2item[d.item[int(g["item_keys"][7])]] == int(g["item_keys"][7])
    u0 = int(g["warm_train"][0, 0])
    assert set(d.training_set_u[u0]) == d.training_set_uid[d.user[u0]]
    with pytest.raises(Exception, match="not in current id table"):
        d.get_user_id_list([10 ** 9])
    # rated CSR == the training sets, ascending internal ids
    uid = d.user[u0]
    row = d.rated_col[d.rated_rowptr[uid]:d.rated_rowptr[uid + 1]]
    assert (np.diff(row) > 0).all() and set(row.tolist()) == {d.item[i] for i in d.training_set_u[u0]}
    if d.mapped_item_content is not None:
        np.testing.assert_array_equal(d.mapped_item_content[d.item[int(g["item_keys"][3])]],
                                      g["content"][int(g["item_keys"][3])])


def test_normalised_adjacency_bitwise():
    g4 = load_golden("g4_graph.npz")
    _, d = builder()
    rowptr, col, val = d.norm_adj_csr()
    assert np.array_equal(rowptr, g4["indptr"]) and np.array_equal(col, g4["indices"])
    np.testing.assert_array_equal(val, g4["data"])
    from coldrec_amd.util.databuilder import TorchGraphInterface
    adj = TorchGraphInterface.convert_sparse_mat_to_tensor(d.norm_adj)
    assert adj.is_sparse and tuple(adj.shape) == (len(rowptr) - 1,) * 2
    coo = adj._coo
    assert np.array_equal(coo.indices()[1].numpy(), g4["coo_cols"])
    np.testing.assert_array_equal(coo.values().numpy(), g4["coo_vals"])


@pytest.mark.parametrize("t", ["all", "warm", "cold"])
def test_array_metrics_match_reference_g7(t):
    g = load_golden("g7_metrics.npz")
    perf = ranking_metrics(g[f"{t}_gt_rowptr"], g[f"{t}_gt_items"], g[f"{t}_pred"], [10, 20])
    np.testing.assert_array_equal(np.array(perf), g[f"{t}_perf"])
    # dict-based entry point, same strings as the reference returns
    rp, items, pred = g[f"{t}_gt_rowptr"], g[f"{t}_gt_items"], g[f"{t}_pred"]
    origin = {u: {int(i): 1.0 for i in items[rp[u]:rp[u + 1]]} for u in range(len(rp) - 1)}
    res = {u: [(int(i), 0.0) for i in pred[u]] for u in range(len(rp) - 1)}
    measure, perf2 = ranking_evaluation(origin, res, [10, 20])
    assert measure == g[f"{t}_measure"].tolist()
    np.testing.assert_array_equal(np.array(perf2), g[f"{t}_perf"])


def test_metric_edge_cases():
    rp = np.array([0, 0, 2], np.int64)              # first user has no ground truth
    perf = ranking_metrics(rp, np.array([5, 9]), np.array([[1, 2, 3], [9, 7, 5]]), [1, 3])
    assert perf[0] == [0.5, 0.5, 0.5, 1.0]
    assert perf[1][0] == 1.0 and perf[1][2] == 1.0
    assert ranking_metrics(np.array([0, 0]), np.array([], np.int64), np.array([[1, 2]]), [2]) == [[0.0, 0.0, 0.0, 0.0]]


def test_cli_dataset_files_and_result_writer(tmp_path, capsys):
    from coldrec_amd import main as cli
    from coldrec_amd.data.synth import make_dataset, write_dataset
    a = cli.parse_args(["--model", "LightGCN", "--layers", "3", "--use_gpu", "false", "--topN", "5,10"])
    assert a.layers == 3 and a.use_gpu is False and a.bs == 4096 and a.emb_size == 64 and a.seed == 2024
    with pytest.raises(ValueError, match="Invalid model name"):
        cli.parse_args(["--model", "NoSuchModel"])
    write_dataset(make_dataset("toy", "item", seed=1), str(tmp_path), "toy")
    args = cli.parse_args(["--dataset", "toy", "--use_gpu", "false", "--data_root", str(tmp_path),
                           "--result_dir", str(tmp_path / "result")])
    cfg = cli.Config(args, str(tmp_path))
    g = load_golden("toy_item.npz")
    assert np.array_equal(cfg.data.user_keys, g["user_keys"]) and cfg.device.type == "cpu"
    top_ns = ["10", "20"]
    results = {s: {m: [[0.1 * (q + 1), 0.1 * (q + 1)] for _ in top_ns] for q, m in enumerate(cli.METRICS)}
               for _, s in cli.SETTINGS}
    args.runs = 2
    payload = cli.summarise(results, top_ns, [1.5, 2.5], args)
    assert payload["20"]["cold"]["NDCG"] == {"mean": pytest.approx(0.4), "std": pytest.approx(0.0)}
    text = open(tmp_path / "result" / "MF" / "history.txt").read()
    assert text.startswith("=== ColdRec Run Result ===") and "Top-10 Cold-Start: Hit=0.1000±0.0000" in text
    blob = json.loads(text[text.index("--- JSON (machine-readable) ---") + 32:])
    assert blob["efficiency"]["seconds_per_completed_epoch_mean"] == 2.0 and blob["method"] == "MF"
    assert "Time: 2.0000±0.5000 seconds per completed training epoch." in capsys.readouterr().out


def test_rec_list_is_the_references_dict_built_on_demand():
    """``_evaluate`` / ``test()`` return a Mapping over the ranking kernel's arrays that behaves like the
    reference's ``{user: [(item, score), ...]}`` (model/BaseRecommender.py:185-187) -- keys in the data set's order, lists
    materialised on access, ``len`` / iteration / ``items()`` / assignment of an edited list -- and ``full_evaluation`` scores
    an untouched one straight from the arrays (same numbers as the tuple walk)."""
    from coldrec_amd.model.BaseRecommender import BaseColdStartTrainer, RecList
    _, d = builder()
    args = argparse.Namespace(dataset="toy", model="MF", epochs=2, layers=2, topN="10,20", bs=512, emb_size=16,
                              lr=1e-3, reg=1e-4, early_stop=2, eval_every=1, cold_object="item", save_emb=False)
    cfg = types.SimpleNamespace(args=args, data=d, device=torch.device("cpu"))

    class Stub(BaseColdStartTrainer):
        def train(self): ...
        def predict(self, u): ...
        def save(self): ...
        def batch_predict(self, users): ...

    tr = Stub(cfg)
    test_set = d.overall_test_set
    users = list(test_set.keys())
    rng = np.random.default_rng(7)
    idx = np.stack([rng.permutation(d.item_num)[:20] for _ in users]).astype(np.int32)
    for r, u in enumerate(users[:50]):                         # plant real hits so the metrics are not all zero
        truth = [d.item[it] for it in test_set[u]]
        idx[r, : min(3, len(truth))] = truth[:3]
    sc = -np.sort(-rng.random((len(users), 20)).astype(np.float32), axis=1)
    rec = RecList(users, sc, idx, d.item_keys)
    eager = {u: list(zip(d.item_keys[idx[r]].tolist(), sc[r])) for r, u in enumerate(users)}
    assert len(rec) == len(eager) and list(rec) == list(eager) and list(rec.keys()) == users
    assert rec[users[3]] == eager[users[3]] and isinstance(rec[users[3]][0][1], np.float32)
    assert dict(rec) == eager and users[0] in rec and "no such user" not in rec
    with pytest.raises(KeyError):
        rec["no such user"]
    # the array route and the tuple walk give the same metrics ...

    fast = tr._metrics_from_rec_list(test_set, "all", rec, [10, 20])
    slow = tr._metrics_from_rec_list(test_set, "all", eager, [10, 20])
    assert fast == slow and fast[1][0] > 0
    tr.full_evaluation(rec, "all")
    assert tr.overall_test_results == fast
    # ... and an edited list is honoured (the reference's plugins post-process rec_list in place)
    rec[users[0]] = [(d.item_keys[0], np.float32(1.0))] * 20
    eager[users[0]] = rec[users[0]]
    assert not rec.untouched and dict(rec) == eager
    assert tr._metrics_from_rec_list(test_set, "all", rec, [10, 20]) == tr._metrics_from_rec_list(test_set, "all", eager, [10, 20])
    # a list edited IN PLACE is scored as edited (the object handed out is the object kept), on a fresh RecList
    rec2 = RecList(users, sc, idx, d.item_keys)
    eager2 = {u: list(zip(d.item_keys[idx[r]].tolist(), sc[r])) for r, u in enumerate(users)}
    assert isinstance(rec2, dict) and rec2.untouched
    for u in users[:50]:
        rec2[u].reverse()
        eager2[u].reverse()
    got = rec2.get(users[51])
    got.pop(0)
    eager2[users[51]].pop(0)
    rec2[users[52]][:] = []
    eager2[users[52]] = []
    assert rec2[users[0]] is rec2[users[0]] and not rec2.untouched and not rec2._full and len(rec2.handed_out()) == 52
    edited = tr._metrics_from_rec_list(test_set, "all", rec2, [10, 20])
    assert edited == tr._metrics_from_rec_list(test_set, "all", eager2, [10, 20]) and edited != fast
    tr.full_evaluation(rec2, "all")
    assert tr.overall_test_results == edited
    for u, lst in rec2.items():                                      # lists seen while iterating are kept too
        if u == users[60]:
            lst.clear()
    assert rec2[users[60]] == []
    # structural edits: the plain dict it imitates (every list built, keys in the users' order, a new key allowed)
    rec["no such user"] = []
    assert rec._full and list(rec)[:-1] == users and len(rec) == len(users) + 1
    del rec["no such user"]
    assert rec.pop(users[5]) == eager.pop(users[5]) and rec == eager and len(rec) == len(users) - 1
    with pytest.raises(KeyError):
        rec["no such user"]


def test_stock_batch_predict_is_recognised_by_bytecode_in_any_spelling(tmp_path):
    """fused vs dense evaluation must not hang on the source TEXT of batch_predict.  The stock idiom
    (model/MF.py:58-63) is recognised from the function's bytecode: reformatted, with a docstring, other local names, `.T` /
    `@` / torch.mm, behind a functools.wraps decorator, and with the source file gone (.pyc only); a different computation is not."""
    import functools
    import importlib.util
    import py_compile
    from coldrec_amd.model.BaseRecommender import _is_stock_batch_predict
    src = '''
import functools
import torch

def deco(f):
    @functools.wraps(f)
    def inner(*a, **k):
        return f(*a, **k)
    return inner

class A:
    def batch_predict(self, users):
        """the stock one, reformatted"""
        with torch.no_grad():
            users = self.data.get_user_id_list( users )
            users = torch.tensor(users,
                                 device=self.device)   # a comment
            s = torch.matmul(self.user_emb[users],
                             self.item_emb.transpose(0, 1))
            return s

class B:
    def batch_predict(self, users):
        with torch.no_grad():
            users = self.data.get_user_id_list(users)
            users = torch.tensor(users, device=self.device)
            return self.user_emb[users] @ self.item_emb.T

class C:
    @deco
    def batch_predict(self, users):
        users = self.data.get_user_id_list(users)
        users = torch.tensor(users, device=self.device)
        score = torch.mm(self.user_emb[users], self.item_emb.t())
        return score

class Neg:
    def batch_predict(self, users):
        with torch.no_grad():
            users = self.data.get_user_id_list(users)
            users = torch.tensor(users, device=self.device)
            score = -torch.matmul(self.user_emb[users], self.item_emb.transpose(0, 1))
            return score

class Two:
    def batch_predict(self, users):
        with torch.no_grad():
            users = self.data.get_user_id_list(users)
            users = torch.tensor(users, device=self.device)
            score = torch.matmul(self.user_emb[users], self.item_emb.transpose(0, 1))
            score = score + torch.matmul(self.user_aux[users], self.item_aux.transpose(0, 1))
            return score
'''
    path = tmp_path / "plug.py"
    path.write_text(src)
    pyc = tmp_path / "plug_nosrc.pyc"
    py_compile.compile(str(path), cfile=str(pyc))
    for name, file in (("plug", path), ("plug_nosrc", pyc)):
        spec = importlib.util.spec_from_file_location(name, str(file))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if name == "plug_nosrc":
            path.unlink()                                             # no source on disk: inspect.getsource would fail
        assert all(_is_stock_batch_predict(getattr(mod, c).batch_predict) for c in "ABC"), name
        assert not _is_stock_batch_predict(mod.Neg.batch_predict) and not _is_stock_batch_predict(mod.Two.batch_predict)
    assert not _is_stock_batch_predict(len) and not _is_stock_batch_predict(functools.partial(print))


def test_trainer_says_which_route_its_evaluation_takes(capsys):
    """at its first evaluation a trainer prints the route -- the fused kernel the LIBRARY names for the shape, or the
    size of the dense block a non-stock batch_predict materialises per batch -- and warns when that block passes 4 GiB."""
    import warnings
    from coldrec_amd.model.BaseRecommender import BaseColdStartTrainer
    _, d = builder()
    args = argparse.Namespace(dataset="toy", model="MF", epochs=2, layers=2, topN="10,20", bs=512, emb_size=16,
                              lr=1e-3, reg=1e-4, early_stop=2, eval_every=1, cold_object="item", save_emb=False)
    cfg = types.SimpleNamespace(args=args, data=d, device=torch.device("cpu"))

    class Stub(BaseColdStartTrainer):
        def train(self): ...
        def predict(self, u): ...
        def save(self): ...
        def batch_predict(self, users): ...

    tr = Stub(cfg)
    c = {"users": list(range(300)), "bitmap": None}
    tr._tell_route(False, False, c)
    out = capsys.readouterr().out
    assert "Evaluation route: batch_predict -> (512 x %d) fp32 score block" % d.item_num in out and "crh_mask_topk_f32" in out
    assert "not the stock" in out
    tr.batch_size = 4_000_000                                    # (batch x items x 4 B) far beyond 4 GiB
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tr._tell_route(False, True, c)
    assert any("GiB score block" in str(x.message) for x in w)
    assert "not 2-D device tensors" in capsys.readouterr().out
    tr.item_emb = torch.zeros((10_000_000, 128))                 # the fused route's line names the library's kernel for the shape
    tr.max_N, c["users"] = 20, list(range(131072))
    tr._tell_route(True, True, c)
    out = capsys.readouterr().out
    assert "fused HIP scoring + masks + top-20 (score_topk_dma_kernel, fused-dma, barrier form, seeded from a 8192-item prefix)" in out


def test_trainers_refuse_cpu_and_early_stopping_rules():
    from coldrec_amd.model import AVAILABLE_MODELS
    from coldrec_amd.model.BaseRecommender import BaseColdStartTrainer, _is_stock_batch_predict
    _, d = builder()
    args = argparse.Namespace(dataset="toy", model="MF", epochs=2, layers=2, topN="10,20", bs=512, emb_size=16,
                              lr=1e-3, reg=1e-4, early_stop=2, eval_every=1, cold_object="item", save_emb=False)
    cfg = types.SimpleNamespace(args=args, data=d, device=torch.device("cpu"))
    assert sorted(AVAILABLE_MODELS.keys()) == ["DropoutNet", "LightGCN", "MF"]
    with pytest.raises(RuntimeError, match="MI355X only"):
        AVAILABLE_MODELS["MF"](cfg).train()

    class Scripted(BaseColdStartTrainer):
        def train(self): ...
        def predict(self, u): ...
        def save(self): self.saved.append(self.cur)
        def batch_predict(self, users):
            with torch.no_grad():
                users = self.data.get_user_id_list(users)
                users = torch.tensor(users, device=self.device)
                score = torch.matmul(self.user_emb[users], self.item_emb.transpose(0, 1))
                return score
        def _metrics(self, data_set, data_type, topn):
            return [[0.1, 0.1, 0.1, self.cur]]

    assert _is_stock_batch_predict(Scripted.batch_predict)
    assert not _is_stock_batch_predict(BaseColdStartTrainer.fast_evaluation)
    tr = Scripted(cfg)
    tr.saved = []
    script = [0.2, 0.2, float("nan"), 0.3, 0.1, 0.1]
    patience = []
    for e, v in enumerate(script):
        tr.cur = v
        tr.fast_evaluation(e, "all")
        patience.append(tr.early_stop_patience)
    assert tr.saved == [0.2, 0.3]                       # strict improvement only
    assert patience == [2, 1, 0, 2, 1, 0]               # first checkpoint leaves patience untouched
    assert tr.bestPerformance[0] == 4 and tr.bestPerformance[1]["NDCG"] == 0.3


_GLOO_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CR_ROOT"])
import coldrec_amd.eval as ev
from coldrec_amd.eval import ShardedTopK, shard_bounds
from oracle import oracle_np as orc

def local_topk(U, users, V, k, rp, rc, bm, item_base=0):
    s, i = orc.score_topk(U.numpy(), None if users is None else users.numpy(), V.numpy(), k,
                          None if rp is None else rp.numpy(), None if rc is None else rc.numpy(),
                          None if bm is None else bm.numpy().view(np.uint32), item_base=item_base)
    return torch.from_numpy(s), torch.from_numpy(i)

def merge(gs, gi, k):
    s, i = orc.merge_topk(gs.numpy(), gi.numpy(), k)
    return torch.from_numpy(s), torch.from_numpy(i)

class OracleOps:                 # the seam: eval.py calls whatever eval.ops is (tests only)
    score_topk = staticmethod(local_topk)
    merge_topk = staticmethod(merge)
ev.ops = OracleOps
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(0)
n_items, k = 1001, 20
U = torch.from_numpy(rng.standard_normal((37, 16)).astype(np.float32))
V = torch.from_numpy(rng.standard_normal((n_items, 16)).astype(np.float32))
rated = [np.unique(rng.integers(0, n_items, 9)) for _ in range(37)]
rp = torch.from_numpy(np.concatenate([[0], np.cumsum([len(r) for r in rated])]).astype(np.int64))
rc = torch.from_numpy(np.concatenate(rated).astype(np.int32))
bm = torch.from_numpy(orc.make_bitmap(n_items, np.where(rng.random(n_items) < 0.2)[0]).view(np.int32))
lo, hi = shard_bounds(n_items, world, rank)
# shard bounds: contiguous, cover the table once, sizes differ by at most one row (1001 items: uneven for 2 and for 8)
bounds = [shard_bounds(n_items, world, r) for r in range(world)]
assert bounds[0][0] == 0 and bounds[-1][1] == n_items and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
sizes = [b - a for a, b in bounds]
assert max(sizes) - min(sizes) <= 1 and (n_items % world == 0 or max(sizes) != min(sizes))
# data-parallel batch slices (train.DPContext.slice): a 4 096-triple batch and an odd one, cut into `world` slices
from coldrec_amd.train import DPContext
for B in (4096, 1001, 5):
    cuts = [DPContext(world, r).slice(B) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == B and all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
    assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
eng = ShardedTopK(V[lo:hi], lo, n_items, k, world, rank)
s, i = eng.topk(U, None, rp, rc, bm)
ws, wi = local_topk(U, None, V, k, rp, rc, bm)
assert torch.equal(i, wi) and torch.equal(s.view(torch.int32), ws.view(torch.int32)), rank
# user-sharded alternative: replicated items, users cut across the ranks (uneven: 37 users on 2 ranks)
from coldrec_amd.eval import UserShardedTopK
ueng = UserShardedTopK(V, k, world, rank)
users_all = torch.arange(37, dtype=torch.int32)
s2, i2 = ueng.topk(U, users_all, rp, rc, bm)
assert torch.equal(i2, wi) and torch.equal(s2.view(torch.int32), ws.view(torch.int32)), ("user-sharded", rank)
dist.barrier()
if rank == 0:
    print("SHARDED_OK", world)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_eval_plumbing_gloo(tmp_path, world):
    """The N>1 path (contiguous item shards -> packed all_gather -> canonical k * world-way merge) on 2 and on 8 CPU
    ranks (the node size the north_star names), uneven last shard (1001 items), 37 users cut into 8 user slices, the
    data-parallel slices of a 4 096 batch; the compute kernels are stood in by the oracle, which is legitimate in tests."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, CR_ROOT=ROOT, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(29611 + world), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert f"SHARDED_OK {world}" in out.stdout


def test_dense_truth_lookup_equals_sorted_membership():
    """ranking_metrics with the cached (users, items) truth table == the sort-based membership test, including
    predictions outside the item table (padding ids) and users without ground truth."""
    from coldrec_amd.util.evaluator import hit_matrix, ranking_metrics, truth_dense
    rng = np.random.default_rng(3)
    n_u, n_i, k = 400, 250, 20
    lens = rng.integers(0, 12, n_u)
    lens[:5] = 0
    rp = np.concatenate([[0], np.cumsum(lens)])
    gt = np.concatenate([rng.choice(n_i, l, replace=False) for l in lens])
    pred = np.stack([rng.choice(n_i + 10, k, replace=False) for _ in range(n_u)])
    pred[7, :4] = np.iinfo(np.int32).max                      # short lists are padded with INT32_MAX
    dense = truth_dense(rp, gt, n_i)
    assert dense is not None and dense.sum() == len(gt)
    assert np.array_equal(hit_matrix(rp, gt, pred), hit_matrix(rp, gt, pred, dense))
    assert ranking_metrics(rp, gt, pred, [10, 20]) == ranking_metrics(rp, gt, pred, [10, 20], dense=dense)
    assert truth_dense(rp, gt, n_i, max_cells=1000) is None    # too large: callers fall back to the sort


def test_metrics_from_a_precomputed_hit_matrix_and_sequential_sums():
    """ranking_metrics(hit=...) (the trainers test membership on the GPU and bring back one bit per prediction) ==
    the same metrics from the predictions; the per-user averages are summed left to right like the reference's
    Python sum() (np.cumsum), not pairwise."""
    from coldrec_amd.util.evaluator import _seq_sum, hit_matrix, ranking_metrics
    rng = np.random.default_rng(5)
    n_u, n_i, k = 3000, 900, 20
    lens = rng.integers(0, 30, n_u)
    rp = np.concatenate([[0], np.cumsum(lens)])
    gt = np.concatenate([rng.choice(n_i, l, replace=False) for l in lens])
    pred = np.stack([rng.choice(n_i, k, replace=False) for _ in range(n_u)])
    hit = hit_matrix(rp, gt, pred)
    assert ranking_metrics(rp, gt, None, [10, 20], hit=hit) == ranking_metrics(rp, gt, pred, [10, 20])
    x = rng.random(100_000) / 3.0
    assert _seq_sum(x) == float(sum(x.tolist()))
    assert _seq_sum(np.zeros(0)) == 0.0


@pytest.mark.parametrize("name", ["toy_item.npz", "toy_user.npz"])
def test_builder_truth_csr_equals_the_dict_walk(name):
    """The ground-truth CSR the builder precomputes for its six valid / test sets (vectorised, at load time) is what
    evaluator.truth_csr derives from the nested dicts the trainers pass around: same users in the same order, same
    rowptr, same internal item ids in the same order -- including duplicate (user, item) records (counted once)."""
    from coldrec_amd.util.evaluator import truth_csr
    _, d = builder(name)
    for key in ("warm_valid", "warm_test", "cold_valid", "cold_test", "overall_valid", "overall_test"):
        s = getattr(d, key + "_set")
        users, rp, items = truth_csr(s, item_of=d.item)
        cu, crp, ci = d.truth_csr_cached(s)
        assert cu == users and np.array_equal(crp, rp) and np.array_equal(ci, items), key
    assert d.truth_csr_cached({}) is None
    # a dict a plugin changed IN PLACE must not be answered from the arrays of its old contents
    s = d.overall_test_set
    u0 = next(iter(s))
    dropped = s.pop(u0)
    assert d.truth_csr_cached(s) is None                       # a user filtered out
    s[u0] = dropped
    assert d.truth_csr_cached(s) is not None                   # restored: same counts again
    it0 = next(iter(s[u0]))
    s[u0][-12345] = 1.0
    assert d.truth_csr_cached(s) is None                       # an item added to one user's truth
    del s[u0][-12345]
    assert s[u0].get(it0) == 1.0 and d.truth_csr_cached(s) is not None
    # same-count in-place swaps are not detected by the (users, pairs) fingerprint: the documented hook drops the arrays
    d.truth_csr_invalidate()
    assert d.truth_csr_cached(s) is None
    # duplicates and interleaved users
    p = np.array([[5, 9], [3, 9], [5, 7], [5, 9], [3, 1], [5, 7], [4, 9]], np.int64)
    import types
    fake = types.SimpleNamespace(item_keys=np.array([9, 7, 1], np.int64))
    u, rp, it = type(d)._pairs_truth_csr(fake, p)
    assert u == [5, 3, 4] and rp.tolist() == [0, 2, 4, 5] and it.tolist() == [0, 1, 0, 2, 0]


def test_membership_routes_equal_the_host_hit_matrix():
    """BaseColdStartTrainer._membership (the test of every prediction against its user's ground truth that the trainers run on
    the device) -- both routes: the dense truth table of small catalogues and the sorted (row, item) keys with one binary
    search per prediction that S-EVAL-sized evaluations take (in blocks of 2^18 users) -- against evaluator.hit_matrix, on CPU
    tensors (the routes are plain torch ops): padding ids, ids beyond the catalogue, empty truths, duplicate predictions."""
    from coldrec_amd.model.BaseRecommender import BaseColdStartTrainer
    from coldrec_amd.util.evaluator import hit_matrix, ranking_metrics, truth_dense
    rng = np.random.default_rng(8)
    n_users, n_items, k = 700, 5000, 20
    lens = rng.integers(0, 9, n_users)
    gt_rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    gt_items = np.concatenate([rng.choice(n_items, n, replace=False) for n in lens]).astype(np.int64)
    pred = rng.integers(0, n_items, (n_users, k)).astype(np.int32)
    rows = np.repeat(np.arange(n_users), lens)
    take = rng.random(len(rows)) < 0.4                       # plant real hits
    pred[rows[take], rng.integers(0, k, int(take.sum()))] = gt_items[take]
    pred[5, :3] = np.iinfo(np.int32).max                      # padding of a short list
    pred[6, 0] = pred[6, 1]                                   # the same item twice
    want = hit_matrix(gt_rowptr, gt_items, np.where(pred == np.iinfo(np.int32).max, -1, pred).astype(np.int64))

    class T(BaseColdStartTrainer):
        def train(self): ...
        def predict(self, u): ...
        def batch_predict(self, users): ...
        def save(self): ...

    tr = object.__new__(T)
    tr.data = types.SimpleNamespace(item=range(n_items))
    tr.max_N = k
    i = torch.from_numpy(pred)
    base = {"users": list(range(n_users)), "gt_rowptr": gt_rowptr, "gt_items": gt_items}
    dense = tr._membership(dict(base, gt_dense=truth_dense(gt_rowptr, gt_items, n_items)), i)
    keys = tr._membership(dict(base, gt_dense=None), i)
    assert np.array_equal(dense, want) and np.array_equal(keys, want)
    assert ranking_metrics(gt_rowptr, gt_items, None, [10, 20], hit=keys) == \
        ranking_metrics(gt_rowptr, gt_items, np.where(pred == np.iinfo(np.int32).max, -1, pred).astype(np.int64), [10, 20])
    # an empty ground truth: no keys at all
    empty = tr._membership({"users": [0, 1], "gt_rowptr": np.zeros(3, np.int64), "gt_items": np.zeros(0, np.int64),
                            "gt_dense": None}, i[:2])
    assert not empty.any()


def test_bench_helpers_legs_summary_and_profile_clusters():
    """bench.legs_summary (the compact last key of the line) and tools/prof_summary._long_cluster (one kernel name launched
    on very different sizes: the long launches are what the roofline lines refer to)."""
    import importlib.util
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_f.json")))
    summ = bench.legs_summary({k: v for k, v in rec.items() if k != "legs_summary"})
    assert summ == rec["legs_summary"]
    assert summ["headline"] == [round(rec["ms_per_step"], 3), round(rec["roofline"]["frac"], 4)]
    assert summ["eval_f16.shard_8gpu"][1] == round(rec["eval_f16"]["shard_8gpu"]["frac_of_fp16_mfma_peak"], 4)
    spec = importlib.util.spec_from_file_location("prof_summary", os.path.join(ROOT, "tools", "prof_summary.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    assert ps._long_cluster([10.0, 11.0, 9.5]) == ([10.0, 11.0, 9.5], 0)
    assert ps._long_cluster([100.0, 98.0, 12.0, 12.5, 13.0]) == ([100.0, 98.0], 3)
    assert ps._long_cluster([0.0, 5.0]) == ([0.0, 5.0], 0)


@pytest.mark.parametrize("shape", ["movielens", "citeulike"])
def test_g14_builder_and_adjacency_bitwise_at_real_size(shape):
    """G14: at the BASELINE dataset shapes the product's builder assigns the reference's internal ids and its normalised
    bipartite adjacency (D^-1/2 A D^-1/2, fp32, CSR; util/databuilder.py:220-254) equals the reference's bit for bit --
    checksums of the reference's arrays, taken by tests/golden/make_golden.py g14."""
    import zlib
    from coldrec_amd.data.synth import make_dataset
    g = load_golden("g14_graph_real_size.npz")
    split = make_dataset(shape, "item", seed=int(g[shape + "_data_seed"]), with_content=False)
    info = split.info
    d = ColdStartDataBuilder(split.warm_train, split.warm_val, split.cold_val, split.overall_val, split.warm_test,
                             split.cold_test, split.overall_test, info["user_num"], info["item_num"], info["warm_user"],
                             info["warm_item"], info["cold_user"], info["cold_item"], None, None)
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes(), 0)
    assert crc(np.asarray(d.user_keys, np.int64)) == int(g[shape + "_user_keys_crc"])
    assert crc(np.asarray(d.item_keys, np.int64)) == int(g[shape + "_item_keys_crc"])
    rowptr, col, val = d.norm_adj_csr()
    assert len(rowptr) - 1 == int(g[shape + "_n"]) and len(col) == int(g[shape + "_nnz"])
    assert crc(np.asarray(rowptr, np.int64)) == int(g[shape + "_indptr_crc"])
    assert crc(np.asarray(col, np.int64)) == int(g[shape + "_indices_crc"])
    assert crc(np.asarray(val, np.float32)) == int(g[shape + "_data_crc"])
