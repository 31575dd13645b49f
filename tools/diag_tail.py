import torch, sys, os
sys.path.insert(0, os.getcwd())
from video_depth_anything_amd import ops, _lib
g = torch.Generator().manual_seed(74)
for (Cc,B,h,w_,H,W_) in [(128,1,296,296,518,518),(128,1,148,148,259,259),(32,1,296,296,518,518)]:
    x = (torch.randn(B,h,w_,Cc,generator=g)).half().cuda()
    w2 = ops.pack_conv3x3(torch.randn(32,Cc,3,3,generator=g)*(9*Cc)**-0.5).cuda(); b2=torch.randn(32,generator=g).cuda(); w3=(torch.randn(32,generator=g)*0.3).cuda()
    outs=[]
    for v in (1,0):
        _lib.lib.vda_depth_tail_set_variant(v)
        o=torch.full((B,H,W_),float('nan'),device='cuda'); ops.depth_tail(x,w2,b2,w3,0.4,o,B,h,w_,H,W_,Cc); outs.append(o)
    _lib.lib.vda_depth_tail_set_variant(0)
    d=(outs[0]-outs[1]).abs()
    bad=(outs[0]!=outs[1]).nonzero()
    print((Cc,B,h,H), 'mismatches', bad.shape[0], 'max', d.max().item())
    if bad.shape[0]:
        ys=bad[:,1].unique().tolist(); xs=bad[:,2].unique().tolist()
        print(' rows', ys[:40], ' n', len(ys)); print(' cols', xs[:40], ' n', len(xs))
