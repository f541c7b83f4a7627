"""debug probe for attention_tr.hip: structured inputs that expose index mappings."""
import sys
import torch
sys.path.insert(0, ".")
import dimx  # noqa
from dimx import engine

dev = torch.device("cuda:0")
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)


def run(q, k, v, scale, causal=False, lens=None):
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    return engine.op_attention(q.to(dev), k.to(dev), v.to(dev), scale, causal, lens_t, None, bf16=True, row_v=True).float().cpu()


for (H, D, L) in ((1, 64, 64), (1, 48, 64), (2, 64, 40), (8, 48, 40)):
    B = 1
    print("==== H %d D %d L %d" % (H, D, L))
    # 1. uniform softmax, v[key][d] = d  -> out[q][d] = d
    q = torch.zeros(B, L, H, D); k = torch.zeros(B, L, H, D)
    v = torch.arange(D).float()[None, None, None, :].expand(B, L, H, D).contiguous()
    o = run(q, k, v, 0.125)
    print("v=d   : out[0,0,0,:] =", o[0, 0, 0, :].tolist())
    print("        max |out - d| over all rows:", (o - v).abs().max().item())
    # 2. uniform softmax, v[key][d] = key -> out = mean(key) = (L-1)/2
    v = torch.arange(L).float()[None, :, None, None].expand(B, L, H, D).contiguous()
    o = run(q, k, v, 0.125)
    print("v=key : expect %.2f, out[0,0,0,:8] =" % ((L - 1) / 2), o[0, 0, 0, :8].tolist(), " max dev", (o - (L - 1) / 2).abs().max().item())
    # 3. one-hot attention: q_i . k_j large iff i == j  (q = k = 8 * one-hot over d for i < D) -> out[i] = v[i]
    n = min(L, D)
    q = torch.zeros(B, L, H, D); k = torch.zeros(B, L, H, D)
    for i in range(n):
        q[:, i, :, i] = 16.0
        k[:, i, :, i] = 16.0
    v = (torch.arange(L).float()[None, :, None, None] + 0.01 * torch.arange(D).float()[None, None, None, :]).expand(B, L, H, D).contiguous()
    o = run(q, k, v, 1.0)
    print("onehot: out[0,i,0,0] for i<16 =", [round(x, 2) for x in o[0, :16, 0, 0].tolist()])
    print("        out[0,5,0,:8] =", [round(x, 2) for x in o[0, 5, 0, :8].tolist()], " max |out - v| rows < n:", (o[:, :n] - v[:, :n]).abs().max().item())
    # 4. random vs f64
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, L, H, D, generator=g) for _ in range(3))
    sc = 384 ** -0.5 if D == 48 else 0.125
    o = run(q, k, v, sc)
    qd, kd, vd = (t.to(torch.bfloat16).double() for t in (q, k, v))
    ref = torch.einsum("bhij,bjhd->bihd", torch.softmax(torch.einsum("bihd,bjhd->bhij", qd, kd) * sc, -1), vd).float()
    e = (o - ref).abs()
    print("random: max err %.4f; per-head max:" % e.max().item(), [round(x, 3) for x in e.amax(dim=(0, 1, 3)).tolist()], "per-d-block max:", [round(e[..., i:i + 16].max().item(), 3) for i in range(0, D, 16)])
