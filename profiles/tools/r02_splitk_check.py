"""split-K on activation x weight products (K-contiguous A): split vs unsplit, every arithmetic."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for M, N, Kd in [(768, 512, 2048), (768, 512, 1536), (100, 512, 2048), (4500, 512, 2048)]:
    a = torch.randn(M, Kd, generator=g).to(dev)
    w_f = (torch.randn(N, Kd, generator=g) * 0.05).to(dev)       # forward: B [N, K]
    w_b = (torch.randn(Kd, N, generator=g) * 0.05).to(dev)       # dX: B stored [K_red, N_out]
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    for name, mode in (("f32", K.GEMM_F32), ("bf16x3", K.GEMM_BF16X3), ("f16x2", K.GEMM_F16X2)):
        for sk in (2, 3, 4):
            c0, c1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
            K.gemm(a, w_f, c0, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, residual=res, ldr=N, arith=mode)
            K.gemm(a, w_f, c1, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, residual=res, ldr=N, arith=mode, split_k=sk)
            ref = (a.double() @ w_f.double().t() + bias.double() + res.double())
            e_f = float((c0 - c1).abs().max() / c0.abs().max())
            e_ref0, e_ref1 = float((c0.double() - ref).abs().max() / ref.abs().max()), float((c1.double() - ref).abs().max() / ref.abs().max())
            K.gemm(a, w_b, c0, M=M, N=N, K=Kd, lda=Kd, ldb=N, ldc=N, b_kmajor=True, arith=mode)
            K.gemm(a, w_b, c1, M=M, N=N, K=Kd, lda=Kd, ldb=N, ldc=N, b_kmajor=True, arith=mode, split_k=sk)
            e_b = float((c0 - c1).abs().max() / c0.abs().max())
            flag = "" if max(e_f, e_b) < 1e-5 else "   <-- WRONG"
            print(f"{M}x{N}x{Kd} {name:7s} split {sk}: fwd {e_f:.2e} (vs fp64: unsplit {e_ref0:.2e} split {e_ref1:.2e})  dX {e_b:.2e}{flag}")
print("---- with caller-provided scales")
for M, N, Kd in [(4500, 512, 2048), (1236, 512, 2048)]:
    a = torch.randn(M, Kd, generator=g).to(dev)
    w_f = (torch.randn(N, Kd, generator=g) * 0.05).to(dev)
    w_b = (torch.randn(Kd, N, generator=g) * 0.05).to(dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    sa, rs, cs = (torch.zeros(n, dtype=torch.int32, device=dev) for n in (M, N, N))
    K.weight_scales([dict(w=a, row_scale=sa), dict(w=w_f, row_scale=rs), dict(w=w_b, col_scale=cs)])
    ub = torch.zeros(4, dtype=torch.int32, device=dev)
    K.weight_scales([dict(w=torch.full((1, 4), float(a.abs().max()) * 8, device=dev), col_scale=ub)])
    for sk in (1, 3, 4):
        c0, c1, c2 = (torch.empty(M, N, device=dev) for _ in range(3))
        K.gemm(a, w_f, c0, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, residual=res, ldr=N, arith=K.GEMM_F16X2)
        K.gemm(a, w_f, c1, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, residual=res, ldr=N, arith=K.GEMM_F16X2, split_k=sk, a_scale=sa, b_scale=rs)
        K.gemm(a, w_f, c2, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, residual=res, ldr=N, arith=K.GEMM_F16X2, split_k=sk, a_scale=ub, a_scale_stride=0, b_scale=rs)
        e1, e2 = float((c0 - c1).abs().max() / c0.abs().max()), float((c0 - c2).abs().max() / c0.abs().max())
        K.gemm(a, w_b, c0, M=M, N=N, K=Kd, lda=Kd, ldb=N, ldc=N, b_kmajor=True, arith=K.GEMM_F16X2)
        K.gemm(a, w_b, c1, M=M, N=N, K=Kd, lda=Kd, ldb=N, ldc=N, b_kmajor=True, arith=K.GEMM_F16X2, split_k=sk, a_scale=sa, b_scale=cs)
        e3 = float((c0 - c1).abs().max() / c0.abs().max())
        print(f"{M}x{N}x{Kd} split {sk}: fwd exact scales {e1:.2e}  fwd uniform bound {e2:.2e}  dX exact scales {e3:.2e}")
