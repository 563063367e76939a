"""Device-resident training engines for the warm-embedding trainers (BPR-MF, LightGCN).

State layout in HBM: ONE (user_num + item_num, d) fp32 buffer per quantity -- parameters E,
gradient G, Adam moments M and V -- with the user rows first.  ``user_emb`` / ``item_emb`` are
views, so LightGCN's torch.cat (model/LightGCN.py:87) costs nothing and Adam is one segment.
A step is 3 kernel launches for MF (forward partials, row gradients, Adam) -- or ONE when the tables are
cache-sized (``MFEngine.enable_fused_step``, the default of ``EpochRunner``) -- and 3 + 2L + 1 for LightGCN;
nothing is copied to the host unless the caller asks for the loss.
"""
from __future__ import annotations

import os

from typing import Optional

import numpy as np
import torch

from . import ops


class DPContext:
    """Data-parallel training over the GPUs of one node: tables and Adam state are
    replicated, every rank sees the same global batch (one host sampler stream) and works on the slice
    [rank*B/G, (rank+1)*B/G).  Two exchanges per optimiser step, both RCCL all-reduce(sum) on the
    compute stream: the 4 batch sums the backward pass needs (16 bytes) and the dense gradient table.
    Every replica then applies the identical dense Adam update, so the replicas never diverge and the
    result equals the single-GPU step up to fp32 addition order."""

    def __init__(self, world: int, rank: int, group=None):
        assert 0 <= rank < world
        self.world, self.rank, self.group = int(world), int(rank), group

    def slice(self, n: int):
        return self.rank * n // self.world, (self.rank + 1) * n // self.world

    def all_reduce(self, t: torch.Tensor) -> None:
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


    def all_gather_rows(self, full: torch.Tensor, part: torch.Tensor) -> None:
        """full (world * rows, d) <- every rank's part (rows, d), in rank order (one RCCL all-gather on the compute
        stream; the list form is the fallback for backends without the flat collective)."""
        if self.world == 1:
            full.copy_(part)
            return
        import torch.distributed as dist
        assert full.is_contiguous() and part.is_contiguous() and full.shape[0] == self.world * part.shape[0]
        try:
            dist.all_gather_into_tensor(full, part, group=self.group)
        except (RuntimeError, NotImplementedError):
            dist.all_gather(list(full.view(self.world, part.shape[0], -1).unbind(0)), part, group=self.group)

    def check_replicated(self, t: torch.Tensor, what: str) -> None:
        """Every rank must hold the same `t` (the epoch's triples: replica consistency rests on all ranks drawing the
        same NumPy stream).  Position-weighted int64 checksum, all-reduced as max and min."""
        if self.world == 1:
            return
        import torch.distributed as dist
        v = t.reshape(-1).to(torch.int64)
        w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
        c = (v * w).sum().reshape(1)
        hi, lo = c.clone(), c.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        if int(hi.item()) != int(lo.item()):
            raise RuntimeError("data-parallel training: %s differs between ranks (rank %d has checksum %d, the ranks "
                               "span %d..%d): the ranks' samplers have diverged -- seed every rank identically"
                               % (what, self.rank, int(c.item()), int(lo.item()), int(hi.item())))


def dp_from_env() -> Optional[DPContext]:
    """The DPContext of this process when it was launched one-rank-per-GPU (torch.distributed initialised,
    world size > 1), else None."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return DPContext(dist.get_world_size(), dist.get_rank())
    return None


class _TableState:
    dp: Optional[DPContext] = None

    optimizer = 'adam'        # 'adam' = torch.optim.Adam as the reference (model/MF.py:14); 'sgd' = torch.optim.SGD(lr)

    def enable_data_parallel(self, dp: DPContext) -> None:
        self.dp = dp
        self.sums = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._dp_ws, self._dp_cap = None, 0
        self._xc = None                      # exchange buffers of the touched-rows step (_lazy_step_dp)
        self.exchange_bytes_per_step = 0

    def _ws(self, batch: int) -> torch.Tensor:
        """This engine's OWN BPR scratch (grown only outside graph capture): a captured epoch keeps its raw pointer."""
        if self._bpr_ws is None or self._bpr_cap < batch:
            if self.E.is_cuda:
                assert not torch.cuda.is_current_stream_capturing(), "size the BPR workspace before capturing the epoch"
            self._bpr_cap = max(int(batch), 1)
            self._bpr_ws = self.k.bpr_workspace(self._bpr_cap, self.device)
        return self._bpr_ws

    def _dp_loss_grad(self, tu, tp, user_idx, pos_idx, neg_idx, gu, gp, loss_out) -> None:
        """Slice forward -> all-reduce(4 sums) -> slice backward into the local gradient.  The backward uses the
        reverse index of the SLICE (one lane group per touched row, fixed summation order, stores): every rank's
        gradient is bit-reproducible, as the single-GPU step's is; what remains order-dependent is the all-reduce."""
        B = user_idx.shape[0]
        lo, hi = self.dp.slice(B)
        if self._dp_ws is None or self._dp_cap < hi - lo:
            self._dp_cap = max(hi - lo, 1)
            self._dp_ws = self.k.bpr_workspace(self._dp_cap, self.device)
        u, p, n = user_idx[lo:hi], pos_idx[lo:hi], neg_idx[lo:hi]
        if hi > lo:
            self.k.bpr_fwd(tu, tp, tp, u, p, n, self.sums, self._dp_ws)
        else:
            self.sums.zero_()
        self.dp.all_reduce(self.sums)
        if hi > lo:
            plan = None
            if hi - lo <= 8192 and hasattr(self.k, 'build_plans_device') and self.d <= 256:
                plan = self.k.build_plans_device(u, p, n, hi - lo)[0]
            self.k.bpr_bwd(tu, tp, tp, u, p, n, B, self.reg, self.sums, gu, gp, gp, loss_out, self._dp_ws, plan=plan)
        elif loss_out is not None:            # a rank without triples still reports the global loss
            loss_out[0] = self.sums[3] / B
            loss_out[1] = self.reg * self.sums[:3].sqrt().sum() / B

    def _alloc_state(self, optimizer: str) -> None:
        assert optimizer in ('adam', 'sgd'), "optimizer must be 'adam' (the reference's) or 'sgd'"
        self.optimizer = optimizer
        self.G = torch.zeros_like(self.E)
        # plain SGD keeps no optimiser state (M, V stay None)
        self.M, self.V = (torch.zeros_like(self.E), torch.zeros_like(self.E)) if optimizer == 'adam' else (None, None)
        self.step_count = 0
        self.loss = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._bpr_ws, self._bpr_cap = None, 0

    @classmethod
    def from_table(cls, E: torch.Tensor, user_num: int, lr: float, reg: float, optimizer: str = 'adam'):
        """Engine over an existing device-resident (user_num + item_num, d) fp32 table (rows of users first);
        no host copy is made -- for tables generated on the GPU (bench.py S-TRAIN-XL)."""
        self = cls.__new__(cls)
        self.k = ops
        assert E.is_cuda and E.dtype == torch.float32 and E.is_contiguous() and E.shape[1] % 4 == 0
        self.user_num, self.item_num, self.d, self.device = int(user_num), E.shape[0] - int(user_num), E.shape[1], E.device
        self.d_logical = self.d
        self.E = E
        self.lr, self.reg = float(lr), float(reg)
        self._alloc_state(optimizer)
        return self

    def __init__(self, user0, item0, lr: float, reg: float, device, optimizer: str = 'adam'):
        self.k = ops      # the HIP ops; the 2-rank gloo plumbing tests (CPU) substitute the module attribute train.ops
        u = torch.as_tensor(np.asarray(user0, np.float32) if not torch.is_tensor(user0) else user0.detach().float())
        v = torch.as_tensor(np.asarray(item0, np.float32) if not torch.is_tensor(item0) else item0.detach().float())
        assert u.shape[1] == v.shape[1], "user and item tables must have one embedding width"
        self.user_num, self.item_num, self.d_logical = u.shape[0], v.shape[0], u.shape[1]
        # the reference accepts any --emb_size; the kernels move 16 B per lane, so other widths get zero columns up to
        # the next multiple of 4: their gradient is 0 (every factor is 0), Adam/SGD leave them at 0, norms and scores
        # are unchanged -- exact, and invisible through user_emb / item_emb
        self.d = (self.d_logical + 3) // 4 * 4
        self.device = torch.device(device)
        E = torch.zeros((self.user_num + self.item_num, self.d), dtype=torch.float32)
        E[: self.user_num, : self.d_logical] = u
        E[self.user_num:, : self.d_logical] = v
        self.E = E.to(self.device).contiguous()
        self.lr, self.reg = float(lr), float(reg)
        self._alloc_state(optimizer)

    def _view(self, t):
        return t if self.d == self.d_logical else t[:, : self.d_logical]

    @property
    def user_emb(self):
        return self._view(self.E[: self.user_num])

    @property
    def item_emb(self):
        return self._view(self.E[self.user_num:])

    def last_loss(self) -> float:
        """bpr + l2 of the last step (device -> host sync; the reference prints it every 50 batches)."""
        return float(self.loss.sum().item())


class MFEngine(_TableState):
    """model/MF.py:12-29 with the tables resident on the GPU.

    ``enable_lazy_adam()`` switches the optimiser to the touched-rows replay of dense Adam
    (crh_adam_rows_f32): per step only the rows of the batch are read and written, every other row is
    brought up to date -- bit for bit what the dense pass would have produced -- when it is next touched or
    when the tables are read (``forward()`` / ``sync_tables()``).  For catalogue-scale tables."""

    lazy = False

    def enable_lazy_adam(self) -> None:
        """Single GPU, or data-parallel (``enable_data_parallel`` first or afterwards): see ``_lazy_step_dp``."""
        if self.optimizer != 'adam':       # SGD never moves an untouched row: sgd_rows already is the touched-rows form
            return
        self.lazy = True
        self.last_step = torch.zeros(self.E.shape[0], dtype=torch.int32, device=self.device)
        self._table = None
        self._table_steps = 0
        self._dirty = False

    # ---------------------------------------------------------------- one launch per step (cache-resident tables)
    fused = False
    FUSED_MAX_MAP_BYTES = 1 << 29        # the (batches, rows) per-row tables of an epoch (12 B per row and batch)

    def can_fuse(self, n_batches: int, batch_size: int) -> bool:
        return (hasattr(self.k, 'mf_step') and not self.lazy and self.dp is None and self.d <= 256 and self.E.is_cuda
                and batch_size <= 8192 and 1 <= n_batches <= 65535
                and n_batches * self.E.shape[0] * 12 <= self.FUSED_MAX_MAP_BYTES)

    def enable_fused_step(self) -> None:
        """Whole step (gather, loss, backward, dense Adam) in one launch, crh_mf_step_f32: the parameters
        ping-pong between ``E`` and a second buffer, no gradient table is used.  Driven by ``fused_epoch``."""
        assert not self.lazy and self.dp is None and self.d <= 256
        self.fused = True
        self.E2 = torch.empty_like(self.E)
        self._sgd_lr = self.lr if self.optimizer == 'sgd' else None
        self._nparts = ops.mf_step_parts(self.E.shape[0], self.d)
        self._parts = [torch.zeros(self._nparts * 4, dtype=torch.float32, device=self.device) for _ in range(2)]
        self._fws, self._fws_cap = None, 0
        self._fsums = torch.zeros(4, dtype=torch.float32, device=self.device)

    def fused_epoch(self, u, i, j, steps, plans, tables, losses, scalars) -> None:
        """All optimiser steps of one epoch (``steps`` = [(lo, hi)] into the triple arrays), one launch each; the Frobenius
        norms of the first batch come from a forward pass, those of batch s+1 from step s.  (The epoch as ONE persistent
        launch with a grid barrier between the steps was built and measured in round 3 -- 19.2 us per step against 18.0 --
        and removed in round 4: profiles/r03_mf_epoch.log)"""
        U, (lo, hi) = self.user_num, steps[0]
        if self._fws is None or self._fws_cap < hi - lo:
            self._fws_cap = hi - lo
            self._fws = ops.bpr_workspace(self._fws_cap, self.device)
        ops.bpr_fwd(self.E[:U], self.E[U:], self.E[U:], u[lo:hi], i[lo:hi], j[lo:hi], self._fsums, self._fws)
        part_in, n_in = self._fws.view(torch.float32), ops.bpr_fwd_parts(hi - lo, self.d)
        src, dst, prev = self.E, self.E2, 0
        rng, mult, ent = tables
        for s, (lo, hi) in enumerate(steps):
            part_out = self._parts[s & 1]
            ops.mf_step(src, dst, self.M, self.V, U, hi - lo, self.reg, plans[s], rng[s], ent[s],
                        mult[s + 1] if s + 1 < len(steps) else None, part_in, n_in, part_out,
                        losses[s - 1] if s else None, prev, losses[s], scalars[s], sgd_lr=self._sgd_lr)
            part_in, n_in, prev = part_out, self._nparts, hi - lo
            src, dst = dst, src
        ops.mf_step_finish(part_in, n_in, prev, losses[len(steps) - 1])
        if src is not self.E:
            self.E.copy_(src)
        self.loss.copy_(losses[len(steps) - 1])
        self.step_count += len(steps)

    def _scalar_table(self, upto: int) -> torch.Tensor:
        if self._table is None or upto > self._table_steps:
            n = max(1024, 2 * upto)
            sc = np.zeros((n + 1, 2), np.float32)
            sc[1:] = self.k.adam_step_scalars(1, n, self.lr)
            self._table, self._table_steps = torch.from_numpy(sc).to(self.device), n
        return self._table

    def sync_tables(self) -> None:
        """Bring every row up to the current step (no-op for the dense optimiser)."""
        if self.lazy and self._dirty:
            self.k.adam_rows(self.E, self.G, self.M, self.V, self.last_step, None, 0, self.user_num, self.step_count,
                             self._scalar_table(self.step_count), mode=2)
            self._dirty = False

    def _lazy_step(self, user_idx, pos_idx, neg_idx, plan, loss) -> None:
        U, B, t = self.user_num, user_idx.shape[0], self.step_count + 1
        if plan is None:
            plan = self.k.build_plans_device(user_idx, pos_idx, neg_idx, B)[0]
        tab = self._scalar_table(t)
        self.k.adam_rows(self.E, self.G, self.M, self.V, self.last_step, plan, B, U, t, tab, mode=0)
        self.k.bpr_fwd_bwd(self.E[:U], self.E[U:], self.E[U:], user_idx, pos_idx, neg_idx, self.reg,
                           self.G[:U], self.G[U:], self.G[U:], loss, plan=plan, workspace=self._ws(B))
        self.k.adam_rows(self.E, self.G, self.M, self.V, self.last_step, plan, B, U, t, tab, mode=1)
        self.step_count, self._dirty = t, True

    def _lazy_step_dp(self, user_idx, pos_idx, neg_idx, plan, loss) -> None:
        """The touched-rows step over G replicas (SURVEY.md 8(e) at S-TRAIN-XL: the dense split would all-reduce a 5.6 GB
        gradient table per step).  Split by ROW OWNERSHIP, not by batch slice: every rank holds the whole batch and builds
        the same plan, catches the batch's rows up and runs the (cheap) forward over the whole batch -- identical sums on
        every replica; the backward, the part that gathers 2-3 rows per entry, is cut: rank r sums the gradient rows of the
        plan's row slots w with w % G == r, each row in the plan's entry order, i.e. with the bits of the single-GPU launch.
        ONE all-gather of (row id, d floats) for at most ceil(3 B / G) rows per rank (100 MB per step at B = 65 536, d = 128,
        G = 8), every replica stores the rows into its gradient table and applies crh_adam_rows_f32 over the whole plan:
        replicas stay bit-identical and equal the single-GPU touched-rows run bit for bit."""
        U, B, t = self.user_num, user_idx.shape[0], self.step_count + 1
        G, r = self.dp.world, self.dp.rank
        if plan is None:
            plan = self.k.build_plans_device(user_idx, pos_idx, neg_idx, B)[0]
        tab = self._scalar_table(t)
        k = self.k
        k.adam_rows(self.E, self.G, self.M, self.V, self.last_step, plan, B, U, t, tab, mode=0)
        ws = self._ws(B)
        k.bpr_fwd(self.E[:U], self.E[U:], self.E[U:], user_idx, pos_idx, neg_idx, self.sums, ws)
        k.bpr_bwd_owned(self.E[:U], self.E[U:], user_idx, pos_idx, neg_idx, self.reg, self.sums, self.G[:U], self.G[U:], loss,
                        ws, plan, G, r)
        cap = k.rows_pack_cap(B, G)
        if self._xc is None or self._xc[0].shape[0] < G * cap:
            dev = self.device
            # ids and rows travel in ONE buffer of 32-bit words: slot = [row id, 3 pad words | d floats as bit patterns]
            self._xc = (torch.empty((G * cap, self.d + 4), dtype=torch.int32, device=dev),
                        torch.zeros((cap, self.d + 4), dtype=torch.int32, device=dev),
                        torch.empty(cap, dtype=torch.int32, device=dev), torch.empty((cap, self.d), dtype=torch.float32, device=dev))
        full, part, ids, rows = self._xc
        full, part, ids, rows = full[: G * cap], part[:cap], ids[:cap], rows[:cap]
        k.rows_pack(self.G, plan, B, U, G, r, ids, rows)
        part[:, 0] = ids
        part[:, 4:] = rows.view(torch.int32)
        self.dp.all_gather_rows(full, part)
        self.exchange_bytes_per_step = int(full.numel() * 4)
        k.rows_unpack(self.G, full[:, 0].contiguous(), full[:, 4:].contiguous().view(torch.float32))
        k.adam_rows(self.E, self.G, self.M, self.V, self.last_step, plan, B, U, t, tab, mode=1)
        self.step_count, self._dirty = t, True

    def step(self, user_idx: torch.Tensor, pos_idx: torch.Tensor, neg_idx: torch.Tensor,
             plan: Optional[torch.Tensor] = None, loss_out=None, step_scalars=None) -> None:
        """``plan``: the batch's reverse index (ops.build_plans_device) -> deterministic gradient rows
        without atomics; without it gradients are accumulated with fp32 atomics."""
        U = self.user_num
        loss = self.loss if loss_out is None else loss_out
        if self.lazy and self.dp is not None:
            return self._lazy_step_dp(user_idx, pos_idx, neg_idx, plan, loss)
        if self.lazy:
            return self._lazy_step(user_idx, pos_idx, neg_idx, plan, loss)
        if self.dp is not None:
            self._dp_loss_grad(self.E[:U], self.E[U:], user_idx, pos_idx, neg_idx, self.G[:U], self.G[U:], loss)
            self.dp.all_reduce(self.G)
            plan = None                       # the summed gradient has rows of every rank's slice
        else:
            self.k.bpr_fwd_bwd(self.E[:U], self.E[U:], self.E[U:], user_idx, pos_idx, neg_idx, self.reg,
                               self.G[:U], self.G[U:], self.G[U:], loss, plan=plan, workspace=self._ws(user_idx.shape[0]))
        self.step_count += 1
        if self.optimizer == 'sgd':
            if plan is not None:              # only the rows the batch touched move (and only their gradient is cleared)
                self.k.sgd_rows(self.E, self.G, plan, user_idx.shape[0], U, self.lr)
            else:
                self.k.sgd_dense(self.E, self.G, self.lr, zero_grad=True)
            return
        self.k.adam_dense(self.E, self.G, self.M, self.V, self.step_count, lr=self.lr, zero_grad=True,
                          step_scalars=step_scalars)

    def forward(self):
        self.sync_tables()
        return self.user_emb, self.item_emb


class LGCNEngine(_TableState):
    """model/LightGCN.py:14-29,86-96: full-graph L-layer propagation per batch and its backward."""

    def __init__(self, user0, item0, rowptr, col, val, n_layers: int, lr: float, reg: float, device,
                 optimizer: str = 'adam'):
        super().__init__(user0, item0, lr, reg, device, optimizer)
        assert n_layers >= 1
        self.L = int(n_layers)
        dev = self.device
        self.rowptr = torch.as_tensor(np.asarray(rowptr, np.int64)).to(dev)
        self.col = torch.as_tensor(np.asarray(col, np.int32)).to(dev)
        self.val = torch.as_tensor(np.asarray(val, np.float32)).to(dev)
        assert self.rowptr.shape[0] == self.E.shape[0] + 1
        self.sched = self.k.SpmmSchedule(np.asarray(rowptr), dev, col=np.asarray(col), val=np.asarray(val))
        # bind the engine's device arrays to the schedule's record stream NOW (one checksum pass): the first launch may
        # already be inside a stream capture, where the check cannot run and the graph would keep the descriptor path
        if hasattr(self.sched, "for_launch"):              # (the CPU stand-in backend of the gloo tests has no record stream)
            self.sched.for_launch(self.E.shape[0], self.E.shape[1], self.col, self.val)
        self.X = [torch.empty_like(self.E) for _ in range(2)]   # layer ping-pong
        self.OUT = torch.empty_like(self.E)                     # mean of the layer outputs
        self.dOUT = torch.zeros_like(self.E)
        self._dout_clean = True
        # Adam in the epilogue of the last backward SpMM (CRH_LGCN_FUSED=0: separate gradient table + Adam launch)
        self.fuse_adam = hasattr(self.k, 'spmm_csr_adam') and os.environ.get("CRH_LGCN_FUSED", "1") != "0"
        self.keep_grad = False       # also store dE0 into self.G (tests compare it with the reference's autograd)

    @classmethod
    def from_device(cls, E: torch.Tensor, user_num: int, rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor,
                    n_layers: int, lr: float, reg: float, optimizer: str = 'adam', seg: Optional[int] = None):
        """Engine over a device-resident (user_num + item_num, d) table and a device-resident CSR adjacency (int64 rowptr,
        int32 col, fp32 val): nothing is copied through the host except rowptr for the schedule -- for graphs built on the
        GPU (bench.py S-TRAIN-XL).  ``seg``: heavy-row threshold of the schedule (None = the automatic rule)."""
        self = cls.from_table(E, user_num, lr, reg, optimizer)
        assert n_layers >= 1 and rowptr.dtype == torch.int64 and col.dtype == torch.int32 and val.dtype == torch.float32
        assert rowptr.shape[0] == E.shape[0] + 1
        self.L = int(n_layers)
        self.rowptr, self.col, self.val = rowptr.contiguous(), col.contiguous(), val.contiguous()
        self.sched = self.k.SpmmSchedule(self.rowptr.cpu().numpy(), self.device, seg=seg, col=self.col, val=self.val)
        if hasattr(self.sched, "for_launch"):
            self.sched.for_launch(self.E.shape[0], self.E.shape[1], self.col, self.val)  # eager binding (see __init__)
        self.X = [torch.empty_like(self.E) for _ in range(2)]
        self.OUT = torch.empty_like(self.E)
        self.dOUT = torch.zeros_like(self.E)
        self._dout_clean = True
        self.fuse_adam = hasattr(self.k, 'spmm_csr_adam') and os.environ.get("CRH_LGCN_FUSED", "1") != "0"
        self.keep_grad = False
        return self

    def enable_row_sharding(self, dp: DPContext) -> None:
        """SURVEY.md 8(e), "scalable variant": the propagation itself is sharded.  Rank r owns the contiguous row block
        [r0, r1) of the adjacency and of every layer state: per layer it multiplies ITS rows (1/G of the SpMM work) and
        the layer outputs are exchanged by one all-gather each (N*d*4 bytes), forward and backward -- the adjacency
        D^-1/2 A D^-1/2 is symmetric, so the backward `A^T dY` is the same row-block product over the gathered dY and no
        transposed blocks or reduce-scatter are needed.  The batch is sharded as in plain data parallelism (all-reduce of
        the 4 sums and of dOUT); Adam / SGD runs on the owned rows only, in the epilogue of the last backward SpMM, and
        one all-gather republishes E.  m and v of rows a rank does not own are never read there and stay stale.
        Per row the arithmetic is that of the replicated engine (same edge order, same heavy-row split), so the tables
        are bit-identical to it.  2L + 1 all-gathers + 1 all-reduce per step: worth it when a row block's SpMM costs more
        than an all-gather of the table, i.e. for graphs far beyond the reference's datasets."""
        if self.dp is None:
            self.enable_data_parallel(dp)
        N, d = self.E.shape
        G = dp.world
        rows = (N + G - 1) // G
        r0 = min(N, dp.rank * rows)
        r1 = min(N, r0 + rows)
        dev = self.device
        e0, e1 = int(self.rowptr[r0]), int(self.rowptr[r1])
        self.rs_rowptr = (self.rowptr[r0:r1 + 1] - e0).contiguous()
        self.rs_col, self.rs_val = self.col[e0:e1].contiguous(), self.val[e0:e1].contiguous()
        # the heavy-row threshold of the FULL graph: a row block of a bipartite graph (all user rows / all item rows) can
        # have a mean degree on the other side of the automatic rule, and a row summed as "heavy" on one engine and
        # "light" on the other would differ in its fp32 association
        self.rs_sched = self.k.SpmmSchedule(self.rs_rowptr.cpu().numpy(), dev, seg=self.sched.seg, col=self.rs_col,
                                             val=self.rs_val) if r1 > r0 else None
        if self.rs_sched is not None and hasattr(self.rs_sched, "for_launch"):
            self.rs_sched.for_launch(r1 - r0, d, self.rs_col, self.rs_val)               # eager binding (see __init__)
        self.rs = (rows, r0, r1)
        pad = G * rows
        E_pad = torch.zeros((pad, d), dtype=torch.float32, device=dev)     # E becomes a view of the gather target
        E_pad[:N] = self.E
        self.rs_E, self.E = E_pad, E_pad[:N]
        self.rs_X = [torch.zeros((pad, d), dtype=torch.float32, device=dev) for _ in range(2)]
        self.rs_OUT = torch.zeros((pad, d), dtype=torch.float32, device=dev)
        self.rs_own = torch.zeros((rows, d), dtype=torch.float32, device=dev)   # send buffer (tail rows stay zero)
        self.rs_acc = torch.zeros((rows, d), dtype=torch.float32, device=dev)   # layer sum / Horner state of the own rows
        self.OUT = self.rs_OUT[:N]

    def _propagate_sharded(self) -> None:
        c = 1.0 / (self.L + 1)
        rows, r0, r1 = self.rs
        own = r1 - r0
        x = self.rs_E
        for k in range(self.L):
            last = k == self.L - 1
            if own:
                self.k.spmm_csr(self.rs_rowptr, self.rs_col, self.rs_val, x, y=None if last else self.rs_own[:own],
                                acc_in=self.E[r0:r1] if k == 0 else self.rs_acc[:own], s_in=1.0,
                                acc_out=self.rs_acc[:own], s_out=c if last else 1.0, sched=self.rs_sched)
            if not last:
                self.dp.all_gather_rows(self.rs_X[k & 1], self.rs_own)
                x = self.rs_X[k & 1]
        self.dp.all_gather_rows(self.rs_OUT, self.rs_acc)

    def _step_sharded(self, user_idx, pos_idx, neg_idx, loss, step_scalars) -> None:
        U, c = self.user_num, 1.0 / (self.L + 1)
        rows, r0, r1 = self.rs
        own = r1 - r0
        self._propagate_sharded()
        self.dOUT.zero_()
        self._dp_loss_grad(self.OUT[:U], self.OUT[U:], user_idx, pos_idx, neg_idx, self.dOUT[:U], self.dOUT[U:], loss)
        self.dp.all_reduce(self.dOUT)
        x = self.dOUT
        self.step_count += 1
        Eo, Go = self.E[r0:r1], self.G[r0:r1]
        Mo, Vo = (self.M[r0:r1], self.V[r0:r1]) if self.M is not None else (None, None)
        for j in range(self.L):
            last = j == self.L - 1
            s_in, s_out = (1.0 if j == 0 else c), (c if j == 0 else 1.0)
            if not own:
                pass
            elif last and self.fuse_adam and self.optimizer == 'sgd':
                self.k.spmm_csr_sgd(self.rs_rowptr, self.rs_col, self.rs_val, x, self.dOUT[r0:r1], s_in,
                                    Go if self.keep_grad else None, s_out, self.rs_sched, Eo, self.lr)
            elif last and self.fuse_adam:
                self.k.spmm_csr_adam(self.rs_rowptr, self.rs_col, self.rs_val, x, self.dOUT[r0:r1], s_in,
                                     Go if self.keep_grad else None, s_out, self.rs_sched, Eo, Mo, Vo, self.step_count,
                                     lr=self.lr, step_scalars=step_scalars)
            else:
                self.k.spmm_csr(self.rs_rowptr, self.rs_col, self.rs_val, x, y=None, acc_in=self.dOUT[r0:r1], s_in=s_in,
                                acc_out=Go if last else self.rs_own[:own], s_out=s_out, sched=self.rs_sched)
            if not last:
                self.dp.all_gather_rows(self.rs_X[j & 1], self.rs_own)
                x = self.rs_X[j & 1]
        if own and not self.fuse_adam:
            if self.optimizer == 'sgd':
                self.k.sgd_dense(Eo, Go, self.lr, zero_grad=False)
            else:
                self.k.adam_dense(Eo, Go, Mo, Vo, self.step_count, lr=self.lr, zero_grad=False, step_scalars=step_scalars)
        if own:
            self.rs_own[:own].copy_(Eo)
        self.dp.all_gather_rows(self.rs_E, self.rs_own)      # every rank sees the updated table (rows >= N stay zero)

    def _propagate(self, out: torch.Tensor) -> None:
        if getattr(self, "rs", None) is not None:
            self._propagate_sharded()
            if out is not self.OUT:
                out.copy_(self.OUT)
            return
        c = 1.0 / (self.L + 1)
        x = self.E
        for k in range(self.L):
            last = k == self.L - 1
            y = None if last else self.X[k & 1]
            self.k.spmm_csr(self.rowptr, self.col, self.val, x, y=y, acc_in=self.E if k == 0 else out, s_in=1.0,
                            acc_out=out, s_out=c if last else 1.0, sched=self.sched)
            x = y

    def forward(self):
        self._propagate(self.OUT)
        return self._view(self.OUT[: self.user_num]), self._view(self.OUT[self.user_num:])

    def step(self, user_idx, pos_idx, neg_idx, plan: Optional[torch.Tensor] = None, loss_out=None,
             step_scalars=None) -> None:
        U, c = self.user_num, 1.0 / (self.L + 1)
        loss = self.loss if loss_out is None else loss_out
        if getattr(self, "rs", None) is not None:
            return self._step_sharded(user_idx, pos_idx, neg_idx, loss, step_scalars)
        self._propagate(self.OUT)
        if not self._dout_clean:
            self.dOUT.zero_()
        if self.dp is not None:
            # propagation is replicated; only the loss/gradient of the batch is sharded, and the exchange
            # happens on dOUT so the backward SpMMs run on identical inputs everywhere
            self._dp_loss_grad(self.OUT[:U], self.OUT[U:], user_idx, pos_idx, neg_idx, self.dOUT[:U], self.dOUT[U:],
                               loss)
            self.dp.all_reduce(self.dOUT)
        else:
            self.k.bpr_fwd_bwd(self.OUT[:U], self.OUT[U:], self.OUT[U:], user_idx, pos_idx, neg_idx, self.reg,
                               self.dOUT[:U], self.dOUT[U:], self.dOUT[U:], loss, plan=plan,
                               workspace=self._ws(user_idx.shape[0]))
        self._backward(step_scalars)

    def _backward(self, step_scalars) -> None:
        """Backward propagation of dOUT and the optimiser step (replicated engine)."""
        c = 1.0 / (self.L + 1)
        # dE0 = c * sum_k A^k dOUT by Horner: H1 = (dOUT + A dOUT) c ; H_{j+1} = dOUT c + A H_j
        x = self.dOUT
        self.step_count += 1
        for j in range(self.L):
            last = j == self.L - 1
            if last and self.fuse_adam:
                # the optimiser runs in the last SpMM's epilogue (no gradient table, no Adam launch); for L >= 2 the
                # gathered operand is not dOUT, so the same epilogue clears dOUT for the next step's scatter
                self._dout_clean = self.L >= 2
                if self.optimizer == 'sgd':
                    self.k.spmm_csr_sgd(self.rowptr, self.col, self.val, x, self.dOUT, 1.0 if j == 0 else c,
                                        self.G if self.keep_grad else None, c if j == 0 else 1.0, self.sched, self.E,
                                        self.lr, zero_acc_in=self._dout_clean)
                    return
                self.k.spmm_csr_adam(self.rowptr, self.col, self.val, x, self.dOUT, 1.0 if j == 0 else c,
                                     self.G if self.keep_grad else None, c if j == 0 else 1.0, self.sched, self.E, self.M, self.V, self.step_count,
                                     lr=self.lr, step_scalars=step_scalars, zero_acc_in=self._dout_clean)
                return
            dst = self.G if last else self.X[j & 1]
            self.k.spmm_csr(self.rowptr, self.col, self.val, x, y=None, acc_in=self.dOUT,
                            s_in=1.0 if j == 0 else c, acc_out=dst, s_out=c if j == 0 else 1.0, sched=self.sched)
            x = dst
        self._dout_clean = False
        if self.optimizer == 'sgd':
            self.k.sgd_dense(self.E, self.G, self.lr, zero_grad=False)
            return
        self.k.adam_dense(self.E, self.G, self.M, self.V, self.step_count, lr=self.lr, zero_grad=False,
                          step_scalars=step_scalars)


class EpochRunner:
    """One epoch of optimiser steps as a replayable hipGraph.

    At MovieLens / CiteULike sizes a step is a handful of 5-30 us kernels, so eager launches (and the
    Python around them) cost more than the kernels.  The runner owns static device buffers for the
    epoch's triples, their reverse-index plans, the per-step Adam factors and the per-step losses; the
    first epoch runs eagerly (warm-up, allocations), the second is captured into a ``torch.cuda.CUDAGraph``
    (= hipGraph) and every later epoch is one ``replay()`` after the buffers were refreshed.
    """

    def __init__(self, engine, n_records: int, batch_size: int, use_graph: bool = True, fused: Optional[bool] = None):
        self.eng, self.n, self.B = engine, int(n_records), int(batch_size)
        dev = engine.device
        self.steps = [(lo, min(lo + self.B, self.n)) for lo in range(0, self.n, self.B)]
        self.u, self.i, self.j = (torch.empty(self.n, dtype=torch.int32, device=dev) for _ in range(3))
        self.plans = None
        self.scalars = torch.empty((len(self.steps), 2), dtype=torch.float32, device=dev)
        self.losses = torch.zeros((len(self.steps), 2), dtype=torch.float32, device=dev)
        self.use_graph, self.graph, self.epochs_done = use_graph and not getattr(engine, "lazy", False), None, 0
        self._sc_pinned = None
        # epochs run eagerly before the capture (CRH_EAGER_EPOCHS): the first epoch makes the lazy allocations and uploads
        # (LightGCN's record streams per lane-group width) that a stream capture does not allow
        self.eager_epochs = max(int(os.environ.get("CRH_EAGER_EPOCHS", "1")), 0)
        # BPR-MF with cache-resident tables: the whole step is one launch (CRH_MF_FUSED=0 keeps the three-kernel step)
        self.tables = None
        if fused is None:
            # BPR-MF: the one-launch step is the default (CRH_MF_FUSED=0 keeps the three-kernel step).  LightGCN keeps its
            # forward pass over the batch: the step without one (block norms from the last forward SpMM's epilogue, score
            # differences recomputed by the row-gradient kernel) was built and measured in round 3 -- 133.6 - 135.0 us per
            # step against 128.3 - 128.6 -- and removed in round 4 (profiles/r03_lgcn_fold.log)
            fused = os.environ.get("CRH_MF_FUSED", "1") != "0"
        if fused and isinstance(engine, MFEngine) and engine.can_fuse(len(self.steps), self.B):
            engine.enable_fused_step()

    def _all_steps(self):
        if getattr(self.eng, "fused", False):
            return self.eng.fused_epoch(self.u, self.i, self.j, self.steps, self.plans, self.tables, self.losses,
                                        self.scalars)
        for s, (lo, hi) in enumerate(self.steps):
            self.eng.step(self.u[lo:hi], self.i[lo:hi], self.j[lo:hi], self.plans[s], self.losses[s], self.scalars[s])

    def run(self, u, i, j) -> torch.Tensor:
        """Train one epoch on the given triples (host int32 arrays or device tensors); returns the
        (steps, 2) device tensor of per-step [bpr, l2] losses (valid after the stream is synchronised)."""
        eng = self.eng
        for dst, src in ((self.u, u), (self.i, i), (self.j, j)):
            dst.copy_(torch.as_tensor(src), non_blocking=True)
        if self.epochs_done == 0 and getattr(eng, "dp", None) is not None:   # once per run: same triples on every rank?
            eng.dp.check_replicated(torch.stack([self.u, self.i, self.j]), "the first epoch's (user, pos, neg) triples")
        if self.plans is None or not hasattr(ops, "plan_shape"):      # (the CPU stand-in of the plumbing tests has no `out`)
            self.plans = ops.build_plans_device(self.u, self.i, self.j, self.B)
        else:
            ops.build_plans_device(self.u, self.i, self.j, self.B, out=self.plans)      # in place: no 26 MB copy per epoch
        if getattr(eng, "fused", False):
            self.tables = ops.mf_step_tables(self.plans, self.u, self.i, self.j, self.B, eng.user_num, eng.item_num,
                                             out=self.tables)
        sc = ops.adam_step_scalars(eng.step_count + 1, len(self.steps), eng.lr)
        if self.scalars.is_cuda:
            # through a pinned slot: a copy from pageable memory is staged synchronously, i.e. the host would wait here
            # for the previous epoch's graph to drain instead of preparing the next one beside it
            k = self.epochs_done & 1
            if self._sc_pinned is None:
                self._sc_pinned = torch.empty((2,) + tuple(self.scalars.shape), dtype=torch.float32, pin_memory=True)
                self._sc_event = [None, None]
            if self._sc_event[k] is not None:
                self._sc_event[k].synchronize()          # the upload that last read this slot (two epochs ago)
            self._sc_pinned[k].copy_(torch.from_numpy(sc))
            self.scalars.copy_(self._sc_pinned[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._sc_event[k] = ev
        else:
            self.scalars.copy_(torch.from_numpy(sc))
        if self.graph is not None:
            self.graph.replay()
            eng.step_count += len(self.steps)
        elif self.use_graph and self.epochs_done >= self.eager_epochs:
            count = eng.step_count
            if hasattr(eng, "_ws"):
                eng._ws(self.B)                         # the engine's scratch keeps its address: sized before the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._all_steps()
            eng.step_count = count                      # capture launched nothing
            self.graph = g
            g.replay()
            eng.step_count += len(self.steps)
        else:
            self._all_steps()
        self.epochs_done += 1
        return self.losses
