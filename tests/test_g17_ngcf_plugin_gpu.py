"""G17: SURVEY.md 8(f)4 -- "other SpMM / BPR consumers" -- pinned to the REFERENCE numerically.

tests/golden/g17_ngcf.npz holds what the reference's own ``NGCF.run()`` (model/NGCF.py:15-104) did on the toy split: 24
dependent Adam steps over the two tables AND four dense layers, every batch's bpr / l2 terms, the propagated tables' norms per
epoch, best epoch, validation and test metrics, sampled rows of the final tables (tests/golden/make_golden.py g17).  The
reference's source cannot travel to the GPU box, so the plugin below is a builder-written stand-in with the same forward --
what a ColdRec user's model file looks like -- and it reaches the MI355X ONLY through the three hooks the drop-in boundary
offers an unmodified model/*.py:
  * ``TorchGraphInterface.convert_sparse_mat_to_tensor`` -> HipSparseAdj answering ``torch.sparse.mm`` (forward and autograd),
  * ``util.utils.bpr_loss`` / ``l2_reg_loss`` as autograd Functions over HIP kernels, ``next_batch_pairwise`` = the C++ sampler,
  * the inherited ``fast_evaluation`` / ``_evaluate`` recognising the stock ``batch_predict`` and ranking with the fused kernel.
Tolerance: north_star's 1e-5 on losses and norms (the dense layers run on rocBLAS in fp32)."""
import argparse
import json
import types
import zlib

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _toy_builder():
    from coldrec_amd.data.synth import make_dataset
    from coldrec_amd.util.databuilder import ColdStartDataBuilder
    split = make_dataset("toy", "item", seed=1)
    info = split.info
    return ColdStartDataBuilder(split.warm_train, split.warm_val, split.cold_val, split.overall_val, split.warm_test,
                                split.cold_test, split.overall_test, info["user_num"], info["item_num"], info["warm_user"],
                                info["warm_item"], info["cold_user"], info["cold_item"], None, split.content)


def _make_plugin():
    """Written the way a ColdRec model file is written, against coldrec_amd's mirror of the reference modules."""
    from coldrec_amd.model.BaseRecommender import BaseColdStartTrainer
    from coldrec_amd.util.databuilder import TorchGraphInterface
    from coldrec_amd.util.utils import bpr_loss, l2_reg_loss, next_batch_pairwise

    class Encoder(nn.Module):
        def __init__(self, data, emb_size, n_layers, device):
            super().__init__()
            self.data, self.layers = data, n_layers
            self.norm_adj = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj).to(device)
            init = nn.init.xavier_uniform_
            self.embedding_dict = nn.ParameterDict({
                "user_emb": nn.Parameter(init(torch.empty(data.user_num, emb_size))),
                "item_emb": nn.Parameter(init(torch.empty(data.item_num, emb_size)))})
            self.W_gc = nn.ModuleList(nn.Linear(emb_size, emb_size) for _ in range(n_layers))
            self.W_bi = nn.ModuleList(nn.Linear(emb_size, emb_size) for _ in range(n_layers))

        def forward(self):
            ego = torch.cat([self.embedding_dict["user_emb"], self.embedding_dict["item_emb"]], 0)
            outs = [ego]
            for layer in range(self.layers):
                side = torch.sparse.mm(self.norm_adj, ego)
                ego = F.leaky_relu(self.W_gc[layer](side) + self.W_bi[layer](ego * side))
                outs.append(ego)
            out = torch.mean(torch.stack(outs, dim=1), dim=1)
            return out[: self.data.user_num], out[self.data.user_num:]

    class Plugin(BaseColdStartTrainer):
        def __init__(self, config):
            super().__init__(config)
            self.model = Encoder(self.data, self.emb_size, self.args.layers, self.device)
            self.seen = dict(bpr=[], l2=[], crc=0, epoch_norm=[])

        def train(self):
            model = self.model.to(self.device)
            optimizer = torch.optim.Adam(model.parameters(), lr=self.lr)
            self.timer(start=True)
            epoch = -1
            for epoch in range(self.maxEpoch):
                model.train()
                for user_idx, pos_idx, neg_idx in next_batch_pairwise(self.data, self.batch_size):
                    c = 0
                    for arr in (user_idx, pos_idx, neg_idx):
                        c = zlib.crc32(np.asarray(arr, np.int32).tobytes(), c)
                    self.seen["crc"] = c ^ (self.seen["crc"] * 31 & 0xFFFFFFFF)
                    ua, ia = model()
                    ue, pe, ne = ua[user_idx], ia[pos_idx], ia[neg_idx]
                    lb, lr_ = bpr_loss(ue, pe, ne), l2_reg_loss(self.reg, ue, pe, ne)
                    self.seen["bpr"].append(float(lb.item()))
                    self.seen["l2"].append(float(lr_.item()))
                    optimizer.zero_grad()
                    (lb + lr_).backward()
                    optimizer.step()
                with torch.no_grad():
                    model.eval()
                    self.user_emb, self.item_emb = model()
                    self.seen["epoch_norm"].append([float(torch.linalg.norm(self.user_emb.double())),
                                                    float(torch.linalg.norm(self.item_emb.double()))])
                    if epoch % self.eval_every == 0:
                        self.fast_evaluation(epoch, valid_type="all")
                        if self.early_stop_flag and self.early_stop_patience <= 0:
                            break
            self.epochs_ran = (epoch + 1) if self.maxEpoch > 0 else 0
            self.timer(start=False)
            self.user_emb, self.item_emb = self.best_user_emb, self.best_item_emb

        def save(self):
            with torch.no_grad():
                self.best_user_emb, self.best_item_emb = self.model.forward()

        def predict(self, u):
            with torch.no_grad():
                u = self.data.get_user_id(u)
                score = torch.matmul(self.user_emb[u], self.item_emb.transpose(0, 1))
                return score.cpu().numpy()

        def batch_predict(self, users):
            with torch.no_grad():
                users = self.data.get_user_id_list(users)
                users = torch.tensor(users, device=self.device)
                score = torch.matmul(self.user_emb[users], self.item_emb.transpose(0, 1))
                return score

    return Plugin


def test_an_ngcf_style_plugin_through_the_three_hooks_matches_the_references_ngcf_g17(monkeypatch, capsys):
    from coldrec_amd import ops
    from coldrec_amd.graph import HipSparseAdj
    from coldrec_amd.model.BaseRecommender import _is_stock_batch_predict
    from coldrec_amd.util.utils import set_seed
    g = load_golden("g17_ngcf.npz")
    data = _toy_builder()
    Plugin = _make_plugin()
    a = dict(dataset="toy", model="NGCF", epochs=int(g["epochs"]), layers=int(g["layers"]), topN="10,20", bs=int(g["batch_size"]),
             emb_size=int(g["d"]), lr=float(g["lr"]), reg=float(g["reg"]), runs=1, seed=2024, use_gpu=True, save_emb=False,
             gpu_id=0, cold_object="item", backbone="MF", early_stop=10, eval_every=1)
    calls = dict(spmm=0, fused=0)
    real_spmm, real_rank = ops.spmm_csr, ops.score_topk

    def spmm_counted(*x, **kw):
        calls["spmm"] += 1
        return real_spmm(*x, **kw)

    def rank_counted(*x, **kw):
        calls["fused"] += 1
        return real_rank(*x, **kw)

    monkeypatch.setattr(ops, "spmm_csr", spmm_counted)
    monkeypatch.setattr(ops, "score_topk", rank_counted)
    set_seed(2024, True)
    tr = Plugin(types.SimpleNamespace(args=argparse.Namespace(**a), data=data, device=DEV))
    assert isinstance(tr.model.norm_adj, HipSparseAdj) and _is_stock_batch_predict(Plugin.batch_predict)
    # the reference's initial state (tables from the xavier stream, dense layers from nn.Linear's init): loaded, not re-drawn
    sd = {k[len("init__"):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("init__")}
    missing = tr.model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("norm_adj" in k for k in missing.missing_keys), missing
    tr.run()
    capsys.readouterr()
    n = len(g["bpr"])
    assert tr.seen["crc"] == int(g["triples_crc"]), "the sampler's triples differ from the reference's NumPy stream"
    assert len(tr.seen["bpr"]) == n
    np.testing.assert_allclose(tr.seen["bpr"], g["bpr"], rtol=1e-5, err_msg="bpr loss per batch")
    np.testing.assert_allclose(np.array(tr.seen["bpr"]) + np.array(tr.seen["l2"]), g["bpr"] + g["l2"], rtol=1e-5)
    np.testing.assert_allclose(tr.seen["l2"], g["l2"], rtol=1e-4)
    np.testing.assert_allclose(np.array(tr.seen["epoch_norm"]), g["epoch_norm"], rtol=1e-5, err_msg="propagated tables per epoch")
    fin = {k: v.detach().cpu().numpy() for k, v in tr.model.state_dict().items() if "norm_adj" not in k}
    names = json.loads(str(g["param_names"]))
    assert sorted(fin) == names
    np.testing.assert_allclose([np.linalg.norm(fin[k].astype(np.float64)) for k in names], g["final_param_norm"], rtol=1e-5)
    assert tr.epochs_ran == int(g["epochs_ran"]) and tr.bestPerformance[0] == int(g["best_epoch"])
    for k, v in json.loads(str(g["best_metrics"])).items():
        assert abs(tr.bestPerformance[1][k] - v) <= 2e-4, (k, tr.bestPerformance[1][k], v)
    U, V = tr.user_emb.detach().cpu().numpy(), tr.item_emb.detach().cpu().numpy()
    np.testing.assert_allclose([np.linalg.norm(U.astype(np.float64)), np.linalg.norm(V.astype(np.float64))], g["final_norm"], rtol=1e-5)
    for got, ref in ((U[g["rows_u"]], g["final_U"]), (V[g["rows_v"]], g["final_V"])):
        assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max()
    worst = 0.0
    for name, got in (("test_overall", tr.overall_test_results), ("test_cold", tr.cold_test_results), ("test_warm", tr.warm_test_results)):
        np.testing.assert_allclose(np.array(got), g[name], atol=2e-4, rtol=0, err_msg=name)
        worst = max(worst, float(np.abs(np.array(got) - g[name]).max()))
    L, epochs = int(g["layers"]), int(g["epochs"])
    # every product went through crh_spmm_csr_f32: per step L forward + L backward, L per epoch-end forward, L per save()
    assert n * 2 * L + epochs * L + L <= calls["spmm"] <= n * 2 * L + 2 * epochs * L, calls
    assert calls["fused"] >= epochs + 3, calls                          # every validation and the three tests ranked by the fused kernel
    err = np.abs(np.array(tr.seen["bpr"]) - g["bpr"]) / np.abs(g["bpr"])
    print("g17 ngcf plugin: %d batches, worst relative bpr error %.1e, norms %.1e, test metrics within %.1e; %d SpMM launches, "
          "%d fused rankings" % (n, err.max(), float(np.abs(np.array(tr.seen["epoch_norm"]) / g["epoch_norm"] - 1).max()), worst,
                                 calls["spmm"], calls["fused"]))
