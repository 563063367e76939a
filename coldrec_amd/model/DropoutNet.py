"""DropoutNet cold-start generator on the MI355X (reference: model/DropoutNet.py; SURVEY.md 8(f)3,
BASELINE.json configs[4]).

What the reference does, kept here: the warm tables written by a backbone run
(``./emb/<dataset>_cold_<object>_<backbone>_{user,item}_emb.pt``, model/DropoutNet.py:95-100) become
TRAINABLE parameters; a two-tower "DeepCF" MLP (Linear -> BatchNorm1d(momentum .01, eps .001) -> tanh,
hidden [200, 100], then Linear to emb_size; truncated-normal(0.01) weights, zero biases,
model/DropoutNet.py:138-236) maps [embedding ; content] of the cold side and the bare embedding of the other
side to new embeddings; training regresses the towers' dot product onto the warm dot product (MSE over
positive and negative pairs, model/DropoutNet.py:20-30) while a random ``n_dropout`` fraction of the cold
side's input embeddings is zeroed (model/DropoutNet.py:107-124).  After every epoch ALL users and items are
pushed through the towers in eval mode (model/DropoutNet.py:126-135) and ranked.

The generator is GEMM-shaped dense work: it runs as stock PyTorch-ROCm modules (rocBLAS/hipBLASLt) on the
device.  The ranking -- the hot path of this package -- goes through the fused HIP kernel of the base class;
``--score_dtype fp16`` (addition) casts the generated tables to half and uses the fp16-MFMA build.
Random streams are consumed in the reference's order (module construction, truncated-normal draws,
``torch.randperm`` per forward on the CPU generator) so a seeded run follows the reference's run.
"""
import os

import torch
import torch.nn as nn

from ..util.utils import epoch_triples
from .BaseRecommender import BaseColdStartTrainer


def truncated_normal_(tensor, mean=0.0, std=1.0):
    """Four N(0,1) candidates per element, the first inside (-2, 2) wins (model/DropoutNet.py:138-144)."""
    cand = tensor.new_empty(tuple(tensor.shape) + (4,)).normal_()
    inside = (cand > -2) & (cand < 2)
    first = inside.max(-1, keepdim=True)[1]
    with torch.no_grad():
        tensor.copy_(cand.gather(-1, first).squeeze(-1)).mul_(std).add_(mean)
    return tensor


@torch.no_grad()
def init_weights(net):
    if isinstance(net, nn.Linear) and type(net) is nn.Linear:
        truncated_normal_(net.weight, std=0.01)
        if net.bias is not None:
            net.bias.zero_()


class TanHBlock(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.layer = nn.Linear(dim_in, dim_out)
        self.bn = nn.BatchNorm1d(dim_out, eps=0.001, momentum=0.01)

    def forward(self, x):
        return torch.tanh(self.bn(self.layer(x)))


class DeepCF(nn.Module):
    """Two towers; a side with content gets [embedding ; content] as input (model/DropoutNet.py:161-222)."""

    def __init__(self, latent_rank_in, user_content_rank, item_content_rank, model_select, rank_out):
        super().__init__()
        self.rank_in, self.rank_out = latent_rank_in, rank_out
        self.phi_u_dim, self.phi_v_dim = user_content_rank, item_content_rank
        widths_u = [latent_rank_in + max(user_content_rank, 0)] + list(model_select)
        widths_v = [latent_rank_in + max(item_content_rank, 0)] + list(model_select)
        # construction order = the reference's (all user blocks, all item blocks, the two heads): the
        # default nn.Linear initialisers draw from the global generator before init_weights overwrites them
        self.u_layers = nn.ModuleList(TanHBlock(a, b) for a, b in zip(widths_u[:-1], widths_u[1:]))
        self.v_layers = nn.ModuleList(TanHBlock(a, b) for a, b in zip(widths_v[:-1], widths_v[1:]))
        self.u_emb = nn.Linear(widths_u[-1], rank_out)
        self.v_emb = nn.Linear(widths_v[-1], rank_out)

    @staticmethod
    def _tower(x, content, blocks, head):
        if content is not None:
            x = torch.cat((x, content), 1)
        for blk in blocks:
            x = blk(x)
        return head(x)

    def encode(self, Uin, Vin, Ucontent, Vcontent):
        return (self._tower(Uin, Ucontent if self.phi_u_dim > 0 else None, self.u_layers, self.u_emb),
                self._tower(Vin, Vcontent if self.phi_v_dim > 0 else None, self.v_layers, self.v_emb))

    def forward(self, Uin, Vin, Ucontent, Vcontent):
        u, v = self.encode(Uin, Vin, Ucontent, Vcontent)
        return (u * v).sum(1)


def get_model(latent_rank_in, user_content_rank, item_content_rank, model_select, rank_out):
    net = DeepCF(latent_rank_in, user_content_rank, item_content_rank, model_select, rank_out)
    net.apply(init_weights)            # visiting order: u blocks, v blocks, u head, v head
    return net


class DropoutNet_Learner(nn.Module):
    def __init__(self, args, data, emb_size, device):
        super().__init__()
        self.args, self.data, self.emb_size, self.device = args, data, emb_size, device
        self.cold_item = args.cold_object == 'item'
        content = data.mapped_item_content if self.cold_item else data.mapped_user_content
        self.register_buffer('content', torch.as_tensor(content, dtype=torch.float32), persistent=False)
        stem = f'./emb/{args.dataset}_cold_{args.cold_object}_{args.backbone}'
        load = lambda side: nn.Parameter(torch.load(f'{stem}_{side}_emb.pt', map_location='cpu').detach().float().clone())
        self.embedding_dict = nn.ParameterDict({'user_emb': load('user'), 'item_emb': load('item')})
        hidden = [int(getattr(args, 'dropoutnet_hidden1', 200)), int(getattr(args, 'dropoutnet_hidden2', 100))]
        cdim = data.item_content_dim if self.cold_item else data.user_content_dim
        self.deepcf_encoder = get_model(emb_size, 0 if self.cold_item else cdim, cdim if self.cold_item else 0,
                                        hidden, emb_size)

    def _rows(self, idx):
        return torch.as_tensor(idx, dtype=torch.long, device=self.embedding_dict['user_emb'].device)

    def pair_score(self, uid, iid):
        u, i = self._rows(uid), self._rows(iid)
        return (self.embedding_dict['user_emb'][u] * self.embedding_dict['item_emb'][i]).sum(1)

    def deepcf_forward(self, uid, iid, is_drop=False):
        u, i = self._rows(uid), self._rows(iid)
        user_emb, item_emb = self.embedding_dict['user_emb'][u], self.embedding_dict['item_emb'][i]
        cold = item_emb if self.cold_item else user_emb
        if is_drop:
            # the CPU generator decides which rows lose their preference embedding (reference :111-112)
            n_zero = int(cold.shape[0] * self.args.n_dropout)
            keep = torch.ones(cold.shape[0], 1, dtype=cold.dtype)
            keep[torch.randperm(cold.shape[0])[:n_zero]] = 0
            cold = cold * keep.to(cold.device)
        if self.cold_item:
            return self.deepcf_encoder(user_emb, cold, None, self.content[i])
        return self.deepcf_encoder(cold, item_emb, self.content[u], None)

    def forward(self):
        ue, ie = self.embedding_dict['user_emb'], self.embedding_dict['item_emb']
        if self.cold_item:
            return self.deepcf_encoder.encode(ue, ie, None, self.content)
        return self.deepcf_encoder.encode(ue, ie, self.content, None)


class DropoutNet(BaseColdStartTrainer):
    fused_eval = True        # batch_predict below is the stock user_emb[users] @ item_emb.T

    def __init__(self, config):
        super().__init__(config)
        self.model = DropoutNet_Learner(self.args, self.data, self.emb_size, self.device)

    def train(self):
        if torch.device(self.device).type != 'cuda':
            raise RuntimeError('coldrec_amd trainers run on the MI355X only (--use_gpu true); there is no CPU path')
        model = self.model.to(self.device)
        optimizer = torch.optim.Adam(model.parameters(), lr=self.lr)
        self.timer(start=True)
        epoch = -1
        for epoch in range(self.maxEpoch):
            model.train()
            # the epoch's triples come from the same sampler stream as next_batch_pairwise, in one host call and
            # one upload (the reference converts three Python lists per batch)
            eu, ei, ej = (torch.from_numpy(x).to(self.device, torch.long) for x in epoch_triples(self.data, self.batch_size))
            for n, lo in enumerate(range(0, eu.shape[0], self.batch_size)):
                user_idx, pos_idx, neg_idx = (x[lo:lo + self.batch_size] for x in (eu, ei, ej))
                target = torch.cat((model.pair_score(user_idx, pos_idx), model.pair_score(user_idx, neg_idx)))
                pred = torch.cat((model.deepcf_forward(user_idx, pos_idx, is_drop=True),
                                  model.deepcf_forward(user_idx, neg_idx, is_drop=True)))
                batch_loss = nn.functional.mse_loss(pred, target)   # the target is NOT detached upstream either
                optimizer.zero_grad()
                batch_loss.backward()
                optimizer.step()
                if n % 50 == 0:
                    print('training:', epoch + 1, 'batch', n, 'batch_loss:', batch_loss.item())
            with torch.no_grad():
                model.eval()
                u, i = model()
                self.user_emb, self.item_emb = u.clone(), i.clone()
                if epoch % self.eval_every == 0:
                    self.fast_evaluation(epoch, valid_type='all')
                    if self.early_stop_flag and self.early_stop_patience <= 0:
                        break
        self.epochs_ran = (epoch + 1) if self.maxEpoch > 0 else 0
        self.timer(start=False)
        model.eval()
        self.user_emb, self.item_emb = self.best_user_emb, self.best_item_emb
        if self.args.save_emb:
            a = self.args
            os.makedirs('./emb', exist_ok=True)
            stem = f'./emb/{a.dataset}_cold_{a.cold_object}_{a.model}'
            torch.save(self.user_emb, stem + '_user_emb.pt')
            torch.save(self.item_emb, stem + '_item_emb.pt')

    def save(self):
        with torch.no_grad():
            u, i = self.model.forward()
            self.best_user_emb, self.best_item_emb = u.clone(), i.clone()

    def predict(self, u):
        with torch.no_grad():
            return (self.item_emb @ self.user_emb[self.data.get_user_id(u)]).cpu().numpy()

    def batch_predict(self, users):
        with torch.no_grad():
            users = torch.as_tensor(self.data.get_user_id_list(users), device=self.device)
            return self.user_emb[users] @ self.item_emb.T
