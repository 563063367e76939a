"""LightGCN warm-embedding trainer on the MI355X (reference: model/LightGCN.py).

The reference recomputes the full-graph L-layer propagation for every batch and lets autograd
replay it transposed (model/LightGCN.py:23,86-96); here that is L + L launches of the CSR SpMM
kernel with the layer mean fused in, around the same fused BPR and dense-Adam kernels as MF.
``save()`` keeps a real snapshot, as the reference's does (its forward() builds new tensors).
"""
import torch
import torch.nn as nn

from ..train import LGCNEngine
from .MF import MF


class LGCN_Encoder(object):
    """Xavier tables (user first, model/LightGCN.py:78-84) + the normalised adjacency as CSR."""

    def __init__(self, data, emb_size, n_layers, device):
        self.data, self.latent_size, self.layers, self.device = data, emb_size, n_layers, device
        init = nn.init.xavier_uniform_
        self.user0 = init(torch.empty(data.user_num, emb_size))
        self.item0 = init(torch.empty(data.item_num, emb_size))
        self.norm_adj = data.norm_adj


class LightGCN(MF):
    fused_eval = True

    def __init__(self, config):
        super(MF, self).__init__(config)
        self.n_layers = self.args.layers
        self.model = LGCN_Encoder(self.data, self.emb_size, self.n_layers, self.device)
        self.engine = None

    def _make_engine(self):
        rowptr, col, val = self.data.norm_adj_csr()
        return LGCNEngine(self.model.user0, self.model.item0, rowptr, col, val, self.n_layers, self.lr,
                          self.reg, self.device, optimizer=getattr(self.args, 'optimizer', 'adam'))

    def _save_tables(self, as_parameter=False):
        super()._save_tables(as_parameter=False)

    def save(self):
        u, i = self.engine.forward()
        self.best_user_emb, self.best_item_emb = u.clone(), i.clone()
