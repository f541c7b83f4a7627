"""debug: staged GEMM epilogue vs torch, determinism, several configs (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
import dimx  # noqa
from dimx import engine as E

torch.manual_seed(0)
dev = "cuda:0"
for (M, N, K) in ((4800, 384, 384), (2400, 1536, 384), (4784, 384, 1536), (4800, 2304, 384)):
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    for bf in (False, True):
        for cfg in (0, 1, 4, 14, 18, 19):
            for act in (0, 2):
                outs = []
                for rep in range(2):
                    o = E.op_gemm(a, w, bias, act, residual=res if not bf else None, bf16=bf, out_bf16=bf, cfg=cfg)
                    outs.append(o.float())
                ref = (a.bfloat16().float() if bf else a) @ (w.bfloat16().float() if bf else w).t() + bias
                if act == 2:
                    ref = torch.nn.functional.gelu(ref, approximate="tanh")
                if not bf:
                    ref = ref + res
                err = (outs[0] - ref).abs().max().item()
                det = torch.equal(outs[0], outs[1])
                flag = "" if (det and err < (0.05 if bf else 2e-3)) else "   <<<<<< BAD"
                print(M, N, K, "bf16" if bf else "f32", "cfg", cfg, "act", act, "err %.3e" % err, "det", det, flag)
