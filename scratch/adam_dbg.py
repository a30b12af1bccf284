import torch, sys
sys.path.insert(0, '.')
from protein_transformer_amd import kernels as K_
dev = torch.device('cuda:0')
def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale
n = 1000003
w0, g = rnd((n,), 30), rnd((n,), 31, 0.01)
ref = w0.clone().requires_grad_()
opt = torch.optim.Adam([ref], betas=(0.9, 0.98), eps=1e-9, lr=1e-3, weight_decay=0.01)
w, m, v = w0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
sq = torch.zeros(1, device=dev)
for step in (1, 2, 3):
    gi = g * step
    ref.grad = gi.clone()
    tn = torch.nn.utils.clip_grad_norm_([ref], 1.0)
    opt.step()
    K_.grad_sqnorm(gi.to(dev), sq)
    K_.adam_step(w, gi.to(dev), m, v, sq, 1.0, 1e-3, 0.9, 0.98, 1e-9, 0.01, step)
    err = (w.cpu().double() - ref.detach().double()).abs()
    st = opt.state[ref]
    print(step, float(tn), sq.sqrt().item(), 'maxerr', err.max().item(), 'n>1e-6', int((err > 1e-6).sum()),
          'm err', (m.cpu() - st['exp_avg']).abs().max().item(), 'v err', (v.cpu() - st['exp_avg_sq']).abs().max().item())
    i = int(err.argmax())
    print('  worst idx', i, 'w0', w0[i].item(), 'g', gi[i].item(), 'ref', ref[i].item(), 'got', w[i].item(), 'm', m[i].item(), st['exp_avg'][i].item(), 'v', v[i].item(), st['exp_avg_sq'][i].item())
