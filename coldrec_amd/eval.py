"""Full-catalogue evaluation over a row-sharded item table.

GPU g owns item rows [g*I/G, (g+1)*I/G); the user table, the rated CSR and the candidate bitmap
are replicated.  Each rank runs the fused scoring/top-k kernel over its shard (global ids via
``item_base``), then ONE all-gather of the packed per-shard top-k (k scores + k ids per user,
8*k bytes) and the canonical merge -- so the result is independent of G.  With G == 1 there is
no collective at all.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch

from . import ops


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank`` (same rule everywhere: bench, trainer, tests)."""
    return rank * n_items // world, (rank + 1) * n_items // world


class UserShardedTopK:
    """The zero-exchange alternative of SURVEY.md 8(e): the item table is REPLICATED (5.1 GB at S-EVAL) and the
    user block is cut across the ranks; every rank ranks its users against the whole catalogue and one
    all-gather puts the (k scores, k ids) rows back in user order.  No merge, no dependence on G by construction;
    used as the validation mode of the item-sharded path and when the catalogue fits every GPU."""

    def __init__(self, items: torch.Tensor, k: int, world: int = 1, rank: int = 0, group=None):
        self.items, self.k = items, int(k)
        self.world, self.rank, self.group = world, rank, group

    def topk(self, user_emb, users, rated_rowptr=None, rated_col=None, cand_bitmap=None):
        """``users`` must be given (int32 rows of user_emb for the whole block, identical on every rank)."""
        n = users.shape[0]
        lo, hi = self.rank * n // self.world, (self.rank + 1) * n // self.world
        rp = rc = None
        if rated_rowptr is not None:
            rp = (rated_rowptr[lo:hi + 1] - rated_rowptr[lo]).contiguous()
            rc = rated_col[int(rated_rowptr[lo]):int(rated_rowptr[hi])].contiguous()
        s, i = ops.score_topk(user_emb, users[lo:hi].contiguous(), self.items, self.k, rp, rc, cand_bitmap)
        if self.world == 1:
            return s, i
        import torch.distributed as dist
        # slices differ by at most one user: pad to the longest, gather, cut the padding away
        longest = (n + self.world - 1) // self.world
        packed = torch.zeros((longest, 2 * self.k), dtype=torch.int32, device=s.device)
        packed[: hi - lo] = torch.cat([s.view(torch.int32), i], dim=1)
        flat = torch.empty((self.world * longest, 2 * self.k), dtype=torch.int32, device=s.device)
        dist.all_gather_into_tensor(flat, packed, group=self.group)
        rows = [flat[r * longest: r * longest + ((r + 1) * n // self.world - r * n // self.world)]
                for r in range(self.world)]
        full = torch.cat(rows, 0)
        return full[:, : self.k].contiguous().view(torch.float32), full[:, self.k:].contiguous()


class ShardedTopK:
    def __init__(self, item_shard: torch.Tensor, item_base: int, n_items_global: int, k: int,
                 world: int = 1, rank: int = 0, group=None):
        self.items = item_shard
        self.item_base = int(item_base)
        self.n_items_global = int(n_items_global)
        self.k = int(k)
        self.world, self.rank, self.group = world, rank, group
        # test hook: run the all-gather + merge even with one rank (exercises the RCCL path on a 1-GPU box)
        self.force_collective = bool(int(os.environ.get("CRH_FORCE_COLLECTIVE", "0")))

    def topk(self, user_emb, users, rated_rowptr=None, rated_col=None, cand_bitmap=None, n_splits: int = 0,
             kernel_events=None):
        kw = {}
        if kernel_events is not None:
            kw["kernel_events"] = kernel_events
        if n_splits:
            kw["n_splits"] = n_splits
        s, i = ops.score_topk(user_emb, users, self.items, self.k, rated_rowptr, rated_col, cand_bitmap,
                              item_base=self.item_base, **kw)
        if self.world == 1 and not self.force_collective:
            return s, i
        import torch.distributed as dist
        packed = torch.cat([s.view(torch.int32), i], dim=1).contiguous()          # (Bu, 2k) int32
        flat = torch.empty((self.world * packed.shape[0], packed.shape[1]), dtype=torch.int32,
                           device=packed.device)                                   # rank-major concatenation
        dist.all_gather_into_tensor(flat, packed, group=self.group)
        gathered = flat.view(self.world, packed.shape[0], packed.shape[1])
        gs = gathered[:, :, :self.k].contiguous().view(torch.float32)
        gi = gathered[:, :, self.k:].contiguous()
        return ops.merge_topk(gs, gi, self.k)
