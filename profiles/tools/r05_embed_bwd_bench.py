import torch, sys
sys.path.insert(0, '.')
from protein_transformer_amd import kernels as K
dev=torch.device('cuda:0')
seq=torch.randint(0,20,(32,512),device=dev); dout=torch.randn(32*512,512,device=dev); demb=torch.zeros(22,512,device=dev)
for p in (0.1, 0.0):
    for _ in range(3): K.embed_bwd(seq,dout,512,p,123,demb)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): K.embed_bwd(seq,dout,512,p,123,demb)
    e1.record(); torch.cuda.synchronize(); print("embed_bwd p=%.1f: %.1f us" % (p, 1e3*e0.elapsed_time(e1)/50))
