#!/usr/bin/env python3
"""LightGCN training at CiteULike shape, a few hipGraph epochs (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CR_ROOT"] = ROOT
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lgcn_sweep
exec(lgcn_sweep.CHILD)
