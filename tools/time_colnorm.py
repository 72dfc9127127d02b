#!/usr/bin/env python3
"""GPU box: the one-launch column normalisation of a Terrace batch's two embedding matrices (gnncca_normalize_columns2), median of 50 calls.
A/B: the default LDS-resident kernel against round 4's two-pass kernel --
    python tools/time_colnorm.py [rows]                                  (A)
    GNNCCA_DIAG=1 GNNCCA_COLNORM_NOLDS=1 python tools/time_colnorm.py    (B)"""
import sys,os,torch,time
sys.path.insert(0,os.environ.get("GRAFT_REPO_ROOT",os.getcwd()))
from gnn_cca_amd.graph_build import normalize_columns
rows=int(sys.argv[1]) if len(sys.argv)>1 else 1100
a=torch.randn(rows,2048,device='cuda'); b=torch.randn(rows,256,device='cuda')
for _ in range(20): normalize_columns(a,b)
torch.cuda.synchronize()
ts=[]
for _ in range(50):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); normalize_columns(a,b); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
ts.sort(); print("GNNCCA_COLNORM_NOLDS", os.environ.get("GNNCCA_COLNORM_NOLDS", "-"), "rows", rows, "median us", ts[len(ts) // 2], "min", ts[0])
