import torch, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from vn_pointcloudcompletion_amd import ops, _lib
for (B,N,C,H) in [(16,1024,384,8),(32,128,384,6)]:
    rows=torch.randn(B*N*3,C,device='cuda')
    for name,fn in [('to_heads',lambda: ops._to_heads(rows,B,N,H))]:
        for _ in range(3): h=fn()
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): h=fn()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/20
        print(B,N,C,H,name,'%.1f us'%(ms*1e3),'%.2f TB/s'%(2*rows.numel()*4/ms/1e9))
    heads=ops._to_heads(rows,B,N,H)
    for _ in range(3): r=ops._from_heads(heads,B,N,C,H)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): r=ops._from_heads(heads,B,N,C,H)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(B,N,C,H,'from_heads','%.1f us'%(ms*1e3),'%.2f TB/s'%(2*rows.numel()*4/ms/1e9))
# padded token-major / feature-major layouts (round 3)
for (B, N, C, H) in [(16, 1024, 384, 8)]:
    rows = torch.randn(B * N * 3, 3 * C, device='cuda')[:, C:2 * C]
    Dh, ldh, fr = ops._attn_dims(C, H)
    fm = ops._heads_fm(rows, B, N, H, fr)
    out = torch.empty(B * N * 3, C, device='cuda')
    for name, fn in [('rows->tm(pad)', lambda: ops._heads_tm(rows, B, N, H, ldh)), ('rows->fm', lambda: ops._heads_fm(rows, B, N, H, fr)),
                     ('fm->rows', lambda: ops._rows_from_fm(fm, B, N, C, H, out=out))]:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(B, N, C, H, name, '%.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
