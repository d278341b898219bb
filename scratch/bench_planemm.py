import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discrete_dppm_graphs_amd import ops
n=int(sys.argv[1]) if len(sys.argv)>1 else 200
ld=(n+3)//4*4; batch=4096
A=torch.randn(batch,n,ld,device='cuda'); B=torch.randn(batch,n,ld,device='cuda')
for ta,tb in ((0,0),(1,0),(0,1)):
    for _ in range(3): C=ops.plane_matmul(A,B,n,trans_a=ta,trans_b=tb)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): C=ops.plane_matmul(A,B,n,trans_a=ta,trans_b=tb)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"n={n} ta={ta} tb={tb}: {ms:.3f} ms  {2*n**3*batch/ms/1e9:.1f} TF")
