"""BPR-MF warm-embedding trainer on the MI355X (reference: model/MF.py).

Same loop, same random streams (torch CPU generator for the xavier tables, NumPy global MT19937
for the triples), same early-stopping and checkpoint semantics -- including the reference's
non-cloning ``save()`` (model/MF.py:48-50: the "best" tables alias the live parameters, SURVEY.md
F5) -- with every optimiser step executed by three HIP kernels on device-resident tables
(coldrec_amd/train.py).  An epoch's triples are sampled on the host in one call and uploaded once.
"""
import os

import torch
import torch.nn as nn

from ..sampler import EpochPrefetcher
from ..train import EpochRunner, MFEngine, dp_from_env
from .BaseRecommender import BaseColdStartTrainer


def _require_gpu(device):
    if torch.device(device).type != 'cuda':
        raise RuntimeError('coldrec_amd trainers run on the MI355X only (--use_gpu true); there is no CPU path')


class Matrix_Factorization(object):
    """Two xavier-uniform tables drawn exactly like model/MF.py:72-79 (user table first)."""

    def __init__(self, data, emb_size):
        self.data, self.latent_size = data, emb_size
        init = nn.init.xavier_uniform_
        self.user0 = init(torch.empty(data.user_num, emb_size))
        self.item0 = init(torch.empty(data.item_num, emb_size))


class MF(BaseColdStartTrainer):
    fused_eval = True

    def __init__(self, config):
        super(MF, self).__init__(config)
        self.model = Matrix_Factorization(self.data, self.emb_size)
        self.engine = None

    def _make_engine(self):
        return MFEngine(self.model.user0, self.model.item0, self.lr, self.reg, self.device,
                        optimizer=getattr(self.args, 'optimizer', 'adam'))

    def train(self):
        _require_gpu(self.device)
        eng = self.engine = self._make_engine()
        epoch = -1
        dp = dp_from_env()            # one rank per GPU: shard every batch, all-reduce sums + gradient
        if dp is not None:
            eng.enable_data_parallel(dp)
            # LightGCN, graphs beyond one GPU's comfort: row-shard the propagation too (--shard_graph / CRH_LGCN_ROW_SHARD=1;
            # 2L + 1 all-gathers per step instead of replicated SpMMs: train.LGCNEngine.enable_row_sharding)
            if hasattr(eng, 'enable_row_sharding') and (getattr(self.args, 'shard_graph', False) or
                                                        os.environ.get('CRH_LGCN_ROW_SHARD', '0') == '1'):
                eng.enable_row_sharding(dp)
        # catalogue-scale tables: replay dense Adam on the touched rows only (bit-identical, see crh_adam_rows_f32); 'auto' =
        # when a batch touches under ~5 % of the rows.  Under data parallelism the step is then split by row ownership with
        # ONE all-gather of gradient rows instead of the dense gradient all-reduce (train.MFEngine._lazy_step_dp)
        mode = getattr(self.args, 'lazy_adam', 'auto')
        rows = self.data.user_num + self.data.item_num
        if hasattr(eng, 'enable_lazy_adam') and type(eng).__name__ == 'MFEngine' and \
                (mode == 'on' or (mode == 'auto' and rows > 64 * self.batch_size)):
            eng.enable_lazy_adam()
        # collectives are kept out of graph capture: the data-parallel epoch is launched eagerly
        runner = EpochRunner(eng, len(self.data.train_u), self.batch_size, use_graph=dp is None)
        # epoch e+1 is sampled (same NumPy stream) while the GPU trains and ranks epoch e: by the C++ sampler on its
        # persistent worker thread, into pinned buffers uploaded asynchronously (EpochPrefetcher)
        triples = EpochPrefetcher(self.data.sampler, self.batch_size, device=self.device)
        # the reference starts its clock once model and optimiser are on the device (model/MF.py:13-16); what it samples
        # from was built when the data was loaded (util/databuilder.py).  Same here: engine, sampler tables, staging
        # buffers and the runner's device buffers exist before the clock starts; every epoch's sampling is inside it.
        self.timer(start=True)
        try:
            for epoch in range(self.maxEpoch):
                # one host call samples the epoch, one hipGraph replay trains it; losses come back in bulk
                if epoch == self.maxEpoch - 1:
                    triples.enabled = False            # nothing follows the last epoch
                losses = runner.run(*triples.get()).sum(dim=1).cpu().numpy()
                for n in range(0, len(losses), 50):
                    print('training:', epoch + 1, 'batch', n, 'batch_loss:', float(losses[n]))
                self.user_emb, self.item_emb = eng.forward()
                if epoch % self.eval_every == 0:
                    self.fast_evaluation(epoch, valid_type='all')
                    if self.early_stop_flag and self.early_stop_patience <= 0:
                        break
        finally:
            triples.close()
        self.epochs_ran = (epoch + 1) if self.maxEpoch > 0 else 0
        self.timer(start=False)
        self.user_emb, self.item_emb = self.best_user_emb, self.best_item_emb
        if self.args.save_emb:
            self._save_tables()

    def _save_tables(self, as_parameter=True):
        a = self.args
        os.makedirs('./emb', exist_ok=True)
        wrap = (lambda t: nn.Parameter(t.detach().clone())) if as_parameter else (lambda t: t.detach().clone())
        stem = f"./emb/{a.dataset}_cold_{a.cold_object}_{a.model}"
        torch.save(wrap(self.user_emb), stem + "_user_emb.pt")
        torch.save(wrap(self.item_emb), stem + "_item_emb.pt")

    def save(self):
        # model/MF.py:48-50 stores the live parameters, not a copy
        self.best_user_emb, self.best_item_emb = self.engine.forward()

    def predict(self, u):
        u = self.data.get_user_id(u)
        return (self.item_emb @ self.user_emb[u]).cpu().numpy()

    def batch_predict(self, users):
        users = torch.as_tensor(self.data.get_user_id_list(users), device=self.device)
        return self.user_emb[users] @ self.item_emb.T
