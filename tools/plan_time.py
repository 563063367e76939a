"""Time crh_bpr_plan_build (one epoch of plans) on the GPU: python tools/plan_time.py"""
import os
import sys

import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coldrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
for name, n_u, n_i, n, B in (("movielens", 6040, 3706, 650161, 4096), ("citeulike", 5551, 16980, 129838, 4096),
                             ("B=2048", 5551, 16980, 129838, 2048), ("B=8192", 6040, 3706, 650161, 8192),
                             ("wide ids", 1 << 20, 1 << 22, 650161, 4096)):
    tu = torch.from_numpy(rng.integers(0, n_u, n).astype(np.int32)).to(dev)
    ti = torch.from_numpy(rng.integers(0, n_i, n).astype(np.int32)).to(dev)
    tj = torch.from_numpy(rng.integers(0, n_i, n).astype(np.int32)).to(dev)
    for _ in range(3):
        ops.build_plans_device(tu, ti, tj, B)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.build_plans_device(tu, ti, tj, B)
    b.record()
    torch.cuda.synchronize()
    print("%-10s %4d batches of %5d: %.1f us per epoch of plans" % (name, -(-n // B), B, a.elapsed_time(b) / 20 * 1e3))

# profile build (make -C coldrec_amd/csrc profile; CRH_LIB=coldrec_amd/lib/libcoldrec_hip_profile.so): phase stamps of workgroup 0
import ctypes
from coldrec_amd import _lib
L = _lib.lib()
if hasattr(L, "crh_profile_plan_clocks"):
    n_u, n_i, n, B = 5551, 16980, 129838, 4096
    tu = torch.from_numpy(rng.integers(0, n_u, n).astype(np.int32)).to(dev)
    ti = torch.from_numpy(rng.integers(0, n_i, n).astype(np.int32)).to(dev)
    tj = torch.from_numpy(rng.integers(0, n_i, n).astype(np.int32)).to(dev)
    for _ in range(3):
        ops.build_plans_device(tu, ti, tj, B)
        torch.cuda.synchronize()
        buf = np.zeros(16, np.uint64)
        assert L.crh_profile_plan_clocks(ctypes.c_void_p(buf.ctypes.data)) == 0
        t = buf.astype(np.int64)
        names = ["max of ids", "keys", "sort", "emit"]
        for side, nm in ((0, "user side"), (1, "item side")):
            q = t[8 * side:8 * side + 5]
            print(nm + ":", "  ".join("%s %.2f" % (x, (q[i + 1] - q[i]) * 0.01) for i, x in enumerate(names)),
                  " total %.2f us; started %.2f us after the user side" % ((q[4] - q[0]) * 0.01, (q[0] - t[0]) * 0.01))
