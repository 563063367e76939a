"""CPU build container only (skipped where /root/reference does not exist, i.e. on the GPU box; nothing of the
reference travels): the reference's OWN model files -- MF, LightGCN, NGCF, SimGCL -- are imported with the four
modules swapped as INTEGRATION.md section A prescribes, constructed on the mirrored ColdStartDataBuilder, driven
through their forward pass and one sampled batch, and stopped at the first call that needs the GPU (our bpr_loss
refuses CPU tensors).  What this proves: the attribute / symbol surface those plugins read
is served by coldrec_amd, including the sparse-adjacency handle answering torch.sparse.mm and the regex that decides
which batch_predict implementations take the fused ranking kernel."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, types
from argparse import Namespace
REF, ROOT = "/root/reference", os.environ["CR_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, REF)
m = types.ModuleType("model"); m.__path__ = [os.path.join(REF, "model")]; sys.modules["model"] = m   # no model/__init__ (faiss)
import numpy as np, torch
import util.utils, util.databuilder, util.evaluator            # the REFERENCE's modules
import model.BaseRecommender                                    # the REFERENCE's base class module
# ---- INTEGRATION.md section A, verbatim
import coldrec_amd.util.utils, coldrec_amd.util.databuilder, coldrec_amd.util.evaluator
import coldrec_amd.model.BaseRecommender as base
sys.modules["util.utils"].bpr_loss = coldrec_amd.util.utils.bpr_loss
sys.modules["util.utils"].l2_reg_loss = coldrec_amd.util.utils.l2_reg_loss
sys.modules["util.utils"].next_batch_pairwise = coldrec_amd.util.utils.next_batch_pairwise
for name in ("next_batch_pairwise_LARA", "next_batch_pairwise_CLCRec", "next_batch_pairwise_CCFCRec", "next_batch_cgrc"):
    setattr(sys.modules["util.utils"], name, getattr(coldrec_amd.util.utils, name))
sys.modules["util.databuilder"].ColdStartDataBuilder = coldrec_amd.util.databuilder.ColdStartDataBuilder
sys.modules["util.databuilder"].TorchGraphInterface = coldrec_amd.util.databuilder.TorchGraphInterface
sys.modules["model.BaseRecommender"].BaseColdStartTrainer = base.BaseColdStartTrainer
# ---- the reference's model files, unmodified
from model.MF import MF
from model.LightGCN import LightGCN
from model.NGCF import NGCF
from model.SimGCL import SimGCL
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.graph import HipSparseAdj
from coldrec_amd.model.BaseRecommender import _is_stock_batch_predict

split = make_dataset("toy", "item", seed=3)
info = split.info
data = coldrec_amd.util.databuilder.ColdStartDataBuilder(
    split.as_lists("warm_train"), split.as_lists("warm_val"), split.as_lists("cold_val"), split.as_lists("overall_val"),
    split.as_lists("warm_test"), split.as_lists("cold_test"), split.as_lists("overall_test"), info["user_num"],
    info["item_num"], info["warm_user"], info["warm_item"], info["cold_user"], info["cold_item"], None, split.content)
args = Namespace(dataset="toy", model="MF", epochs=1, layers=2, topN="10,20", bs=256, emb_size=16, lr=1e-3, reg=1e-4,
                 runs=1, seed=1, use_gpu=False, save_emb=False, gpu_id=0, cold_object="item", backbone="MF",
                 early_stop=3, eval_every=1, cl_rate=0.5, tau=0.2, eps=0.1)
cfg = types.SimpleNamespace(args=args, data=data, device=torch.device("cpu"))
np.random.seed(1)
for cls in (MF, LightGCN, NGCF, SimGCL):
    args.model = cls.__name__
    tr = cls(cfg)
    assert isinstance(tr, base.BaseColdStartTrainer), cls             # subclass of OUR base through the swap
    for attr in ("config", "args", "data", "device", "model_name", "dataset_name", "emb_size", "maxEpoch", "batch_size",
                 "lr", "reg", "topN", "max_N", "early_stop_flag", "early_stop_patience", "eval_every", "epochs_ran",
                 "bestPerformance"):
        assert hasattr(tr, attr), (cls, attr)
    assert _is_stock_batch_predict(cls.batch_predict), cls            # these four rank through crh_score_topk_f32
    enc = tr.model
    adj = getattr(enc, "sparse_norm_adj", None)
    if adj is None:
        adj = getattr(enc, "norm_adj", None)
    if cls is not MF:
        assert isinstance(adj, HipSparseAdj), (cls, type(adj))        # the handle that answers torch.sparse.mm
    out = enc() if cls is not SimGCL else enc(perturbed=False)
    ue, ie = out[0], out[1]
    assert ue.shape == (data.user_num, 16) and ie.shape == (data.item_num, 16)
    u, p, n = next(util.utils.next_batch_pairwise(data, args.bs))     # our C++ sampler behind the reference's name
    assert isinstance(u, list) and len(u) == len(p) == len(n) and max(n) < data.item_num
    try:                                                              # the first call that needs the MI355X
        util.utils.bpr_loss(ue[u], ie[p], ie[n])
        raise SystemExit("bpr_loss accepted CPU tensors")
    except RuntimeError as e:
        assert "no CPU path" in str(e), e
    # the evaluation cache is built from the attributes _evaluate reads (no kernel runs with an empty user set)
    c = tr._get_eval_cache(data.overall_valid_set, "all")
    assert len(c["users"]) == len(data.overall_valid_set) and c["users_int"].dtype == torch.int32
print("REFERENCE_PLUGINS_OK")
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model")), reason="reference checkout not present (GPU box)")
def test_reference_model_files_construct_on_the_mirrored_modules(tmp_path):
    script = tmp_path / "ref_plugins_worker.py"
    script.write_text(_WORKER)
    out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, CR_ROOT=ROOT, OMP_NUM_THREADS="1"),
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0 and "REFERENCE_PLUGINS_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
