import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, time
from vn_pointcloudcompletion_amd import ops
def t(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for (M,N,K,opb) in [(96,2048,2048,0),(96,2048,2048,1),(96,1024,1024,1),(96,256,2048,0),(96,2048,256,1),(96,1024,512,0)]:
    A=torch.randn(M,K).cuda(); B=(torch.randn(N,K) if opb==0 else torch.randn(K,N)).cuda()
    row=[]
    for ns in (None,1,2,4,8,16,32,64):
        try: row.append((ns, round(t(lambda: ops.gemm(A,B,op_b=opb,nsplit=ns)),1)))
        except Exception as e: row.append((ns,'err'))
    print(M,N,K,'NT' if opb==0 else 'NN', row)
for (M,N,K) in [(2048,2048,96),(1024,1024,96),(256,2048,96)]:
    A=torch.randn(K,M).cuda(); B=torch.randn(K,N).cuda()
    print('TN',M,N,K, round(t(lambda: ops.gemm(A,B,op_a=1,op_b=1)),1))
x=torch.randn(96,2048).cuda()
print('empty launch-ish (tiny add)', round(t(lambda: x.add_(1.0)),1))
