import sys, math, torch
sys.path.insert(0, '.')
import dimx
from dimx import engine
dev = torch.device('cuda:0')
torch.manual_seed(0)
for bf16 in (False, True):
    for (M, N, K) in [(64, 64, 32 if not bf16 else 64), (64, 64, 128), (64, 64, 256), (128, 128, 384), (1000, 1152, 384)]:
        a = torch.randn(M, K); w = torch.randn(N, K) / math.sqrt(K)
        if bf16:
            a = a.bfloat16().float(); w = w.bfloat16().float()
        ref = a.double() @ w.double().t()
        o = engine.op_gemm(a.to(dev), w.to(dev), bf16=bf16).cpu().double()
        o2 = engine.op_gemm(a.to(dev), w.to(dev), bf16=bf16, force_simple=True).cpu().double()
        e = (o - ref).abs()
        print("bf16" if bf16 else "f32", M, N, K, "glds err %.3g simple err %.3g" % (e.max(), (o2 - ref).abs().max()))
        if e.max() > 1e-2:
            bad = (e > 1e-2)
            print("   bad frac %.3f; bad rows %s ; bad cols %s" % (bad.float().mean(), bad.any(1).nonzero().view(-1)[:12].tolist(), bad.any(0).nonzero().view(-1)[:12].tolist()))
            # which k contributions are wrong? test with one-hot k
            for k in range(0, K, max(1, K // 8)):
                a1 = torch.zeros(M, K); a1[:, k] = 1.0
                o1 = engine.op_gemm(a1.to(dev), w.to(dev), bf16=bf16).cpu().double()
                r1 = a1.double() @ w.double().t()
                print("      k=%d err %.3g" % (k, (o1 - r1).abs().max()))
