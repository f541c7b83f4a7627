"""GPU: each HIP kernel through the C-ABI against a plain PyTorch fp32/fp64 reference of the same op."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


ACTS = {0: lambda x: x, 1: lambda x: F.leaky_relu(x, 0.2),
        2: lambda x: x * 0.5 * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3))),
        3: lambda x: F.gelu(x)}


@pytest.mark.parametrize("M,N,K,act,use_bias,use_res", [
    (300, 384, 56, 1, True, False), (1000, 1152, 384, 0, False, True), (77, 56, 384, 0, False, False),
    (256, 4608, 1152, 3, True, False), (130, 128, 1536, 2, True, True), (40000, 384, 384, 0, True, True),
    (1, 512, 1152, 0, False, False), (33000, 1536, 384, 2, True, False)])
@pytest.mark.parametrize("bf16", [False, True])
def test_gemm(dev, M, N, K, act, use_bias, use_res, bf16):
    from dimx import engine
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g) if use_bias else None
    res = torch.randn(M, N, generator=g) if use_res else None
    aa, ww = (_bf(a), _bf(w)) if bf16 else (a, w)
    ref = aa.double() @ ww.double().t()
    if bias is not None:
        ref = ref + bias.double()
    ref = ACTS[act](ref)
    if res is not None:
        ref = ref + res.double()
    out = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev) if use_bias else None, act,
                         res.to(dev) if use_res else None, bf16=bf16)
    e = _err(out, ref)
    tol = 2e-3 if bf16 else 2e-5 * max(1.0, math.sqrt(K) / 8)
    assert e < tol, "gemm err %g" % e


@pytest.mark.parametrize("M,N,K,S", [(256, 1152, 4608, 4), (256, 1152, 768, 3), (17, 1152, 4608, 8), (64, 384, 1536, 2),
                                     (256, 2304, 1152, 1)])
@pytest.mark.parametrize("bf16", [False, True])
def test_gemm_splitk_slabs_and_simple_kernel(dev, M, N, K, S, bf16):
    """decode-step projections: split-K partial sums land in S f32 slabs that the consumer adds in order
    (deterministic); plus the register-staged kernel as a cross-check of the LDS-DMA pipeline."""
    from dimx import engine
    g = torch.Generator().manual_seed(K + M)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    aa, ww = (_bf(a), _bf(w)) if bf16 else (a, w)
    ref = aa.double() @ ww.double().t() + bias.double()
    tol = 2e-3 if bf16 else 1e-4
    sl = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), 0, None, bf16=bf16, slabs=S)
    assert sl.shape == (S, M, N) and torch.isfinite(sl).all()
    assert _err(sl.sum(0), ref) < tol
    sl2 = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), 0, None, bf16=bf16, slabs=S)
    assert torch.equal(sl, sl2), "split-K slabs must be bit-reproducible"
    o2 = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), 0, None, bf16=bf16, force_simple=True)
    o3 = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), 0, None, bf16=bf16)
    assert _err(o2, ref) < tol and _err(o3, ref) < tol
    assert _err(o2, o3) < (1e-5 if not bf16 else 1e-4)   # same products, only the summation order differs


@pytest.mark.parametrize("M,N,K,S,act", [(256, 4608, 1152, 0, 3), (256, 3456, 1152, 0, 0), (256, 1152, 4608, 4, 0),
                                         (200, 1100, 1152, 0, 0), (1, 512, 1152, 0, 0), (64, 384, 64, 0, 1),
                                         (256, 1152, 192, 3, 0)])
@pytest.mark.parametrize("cfg", [34, 35])
@pytest.mark.parametrize("bf16", [False, True])
def test_gemm_loader_consumer_kernel(dev, M, N, K, S, act, cfg, bf16):
    """the decode GEMM with loader and consumer waves (cfg 34 / 35): same tiles, same k order, same epilogue as the
    4-wave kernel (cfg 3) -> bit-identical results, ragged M / N edges, 1..18 k-tiles, split-K slabs included."""
    from dimx import engine
    g = torch.Generator().manual_seed(M + N + K + S)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    aa, ww = (_bf(a), _bf(w)) if bf16 else (a, w)
    ref = ACTS[act](aa.double() @ ww.double().t() + bias.double())
    kw = dict(bf16=bf16, slabs=S) if S else dict(bf16=bf16, out_bf16=bf16 and act == 3)
    new = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), act, None, cfg=cfg, **kw)
    old = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), act, None, cfg=3, **kw)
    assert torch.equal(new, old)
    if N % 8 == 0:   # the same weights in 8-row x 128-byte blocks (the decode step's layout): same bits
        til = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), act, None, cfg=cfg, w_tiled=True, **kw)
        assert torch.equal(til, old)
    out = new.sum(0) if S else new.float()
    assert _err(out, ref) < (0.05 if kw.get("out_bf16") else (2e-3 if bf16 else 1e-4))


@pytest.mark.parametrize("M,N,K,S,act", [(256, 4608, 1152, 0, 3), (256, 2304, 1152, 2, 0), (256, 1152, 4608, 4, 0),
                                         (200, 1152, 768, 2, 0), (1, 1152, 1152, 0, 0), (33, 4608, 1152, 0, 3),
                                         (256, 144, 64, 0, 1), (256, 1152, 4608, 8, 0)])
def test_gemm_one_block_per_cu_kernel(dev, M, N, K, S, act):
    """round 5: the decode GEMM on 64 x 72 tiles (cfg 72; what M <= 256, N % 72 == 0 takes by default): same k order and the
    same split partition as the 64 x 64 kernels -> bit-identical results; ragged M, a single row, a single k-tile, slabs;
    columns 72..95 of the third MFMA column block are never stored (the neighbouring tile's columns stay intact)."""
    from dimx import engine
    g = torch.Generator().manual_seed(M + N + K + S)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    ref = ACTS[act](_bf(a).double() @ _bf(w).double().t() + bias.double())
    kw = dict(bf16=True, slabs=S) if S else dict(bf16=True, out_bf16=act == 3)
    new = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), act, None, cfg=72, **kw)
    old = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), act, None, cfg=3, **kw)
    auto = engine.op_gemm(a.to(dev), w.to(dev), bias.to(dev), act, None, **kw)   # the default route
    assert torch.equal(new, old)
    if not S:
        assert torch.equal(auto, old)
    out = new.sum(0) if S else new.float()
    assert _err(out, ref) < (0.05 if kw.get("out_bf16") else 2e-3)


@pytest.mark.parametrize("bf16", [False, True])
def test_gemm_bf16_out(dev, bf16):
    from dimx import engine
    if not bf16:
        pytest.skip("f32 inputs always produce f32")
    g = torch.Generator().manual_seed(3)
    a, w = torch.randn(513, 384, generator=g), torch.randn(768, 384, generator=g) / 20
    ref = _bf(a).double() @ _bf(w).double().t()
    out = engine.op_gemm(a.to(dev), w.to(dev), bf16=True, out_bf16=True)
    assert out.dtype == torch.bfloat16
    assert _err(out.float(), ref) < 0.05


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,T,lens", [(3, 40, [40, 33, 5]), (2, 300, None), (4, 7, [1, 2, 7, 3])])
def test_conv5_gemm(dev, B, T, lens, bf16):
    from dimx import engine
    C = 384
    g = torch.Generator().manual_seed(B * T)
    x = torch.randn(B, T, C, generator=g)
    w = torch.randn(C, C, 5, generator=g) / math.sqrt(5 * C)
    bias = torch.randn(C, generator=g)
    xx, ww = (_bf(x), _bf(w)) if bf16 else (x, w)
    ref = torch.zeros(B, T, C, dtype=torch.float64)
    for b in range(B):
        n = lens[b] if lens else T
        xb = xx[b, :n].double().t()[None]
        yb = F.conv1d(F.pad(xb, (2, 2), mode="replicate"), ww.double(), bias.double())
        ref[b, :n] = F.leaky_relu(yb[0].t(), 0.2)
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    out = engine.op_gemm(x.view(B * T, C).to(dev), w.to(dev), bias.to(dev), 1, bf16=bf16, conv_T=T,
                         conv_lens=lens_t).view(B, T, C)
    for b in range(B):
        n = lens[b] if lens else T
        e = _err(out[b, :n], ref[b, :n])
        assert e < (3e-3 if bf16 else 5e-5), "conv err %g (clip %d)" % (e, b)


@pytest.mark.parametrize("C", [384, 1152])
@pytest.mark.parametrize("use_beta", [True, False])
def test_layernorm(dev, C, use_beta):
    from dimx import engine
    g = torch.Generator().manual_seed(C)
    x = torch.randn(1001, C, generator=g) * 3 + 1
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref = F.layer_norm(x.double(), (C,), gamma.double(), beta.double() if use_beta else None, 1e-5)
    out = engine.op_layernorm(x.to(dev), gamma.to(dev), beta.to(dev) if use_beta else None)
    assert _err(out, ref) < 1e-5
    outb = engine.op_layernorm(x.to(dev), gamma.to(dev), beta.to(dev) if use_beta else None, out_bf16=True)
    assert _err(outb.float(), ref) < 0.05


@pytest.mark.parametrize("B,T,lens", [(3, 40, [40, 33, 5]), (2, 299, None)])
def test_instnorm(dev, B, T, lens):
    from dimx import engine
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, 384, generator=g) * 2 + 0.5
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    out = engine.op_instnorm(x.to(dev), lens_t)
    for b in range(B):
        n = lens[b] if lens else T
        ref = F.instance_norm(x[b, :n].double().t()[None], eps=1e-5)[0].t()
        assert _err(out[b, :n], ref) < 2e-5


def _attn_ref(q, k, v, scale, causal, lens, kmask):
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    dots = torch.einsum("bihd,bjhd->bhij", q.double(), k.double()) * scale
    keep = torch.ones(B, 1, Lq, Lk, dtype=torch.bool)
    if causal:
        keep = keep & ~torch.triu(torch.ones(Lq, Lk, dtype=torch.bool), diagonal=1)
    if lens is not None:
        keep = keep & (torch.arange(Lk)[None, :] < torch.tensor(lens)[:, None])[:, None, None, :]
    if kmask is not None:
        keep = keep & kmask.bool()[:, None, None, :]
    dots = dots.masked_fill(~keep, -torch.finfo(torch.float64).max)
    return torch.einsum("bhij,bjhd->bihd", dots.softmax(-1), v.double())


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,H,Lq,Lk,D,causal,lens,rand_mask", [
    (2, 8, 40, 40, 48, False, [40, 13], False), (1, 8, 300, 300, 48, False, None, False),
    (2, 12, 299, 299, 64, True, None, True), (3, 12, 70, 70, 64, True, None, False),
    (2, 12, 129, 130, 64, False, None, True), (1, 12, 5, 5, 64, True, None, False),
    (1, 8, 1500, 1500, 48, False, None, False)])
def test_attention(dev, B, H, Lq, Lk, D, causal, lens, rand_mask, bf16):
    from dimx import engine
    g = torch.Generator().manual_seed(Lq * 31 + D)
    q, k, v = (torch.randn(B, L_, H, D, generator=g) for L_ in (Lq, Lk, Lk))
    scale = 384 ** -0.5 if D == 48 else 0.125
    kmask = None
    if rand_mask:
        kmask = (torch.rand(B, Lk, generator=g) > 0.3).to(torch.uint8)
        kmask[:, 0] = 1
    qq, kk, vv = ((_bf(q), _bf(k), _bf(v)) if bf16 else (q, k, v))
    ref = _attn_ref(qq, kk, vv, scale, causal, lens, kmask)
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    out = engine.op_attention(q.to(dev), k.to(dev), v.to(dev), scale, causal, lens_t,
                              kmask.to(dev) if kmask is not None else None, bf16=bf16)
    assert torch.isfinite(out.float()).all()
    for b in range(B):
        n = lens[b] if lens else Lq
        e = _err(out[b, :n].float(), ref[b, :n])
        assert e < (2e-2 if bf16 else 2e-5), "attention err %g" % e


def _attention_f64(q, k, v, scale, causal, lens, kmask):
    """softmax(q k^T scale + masks) v in float64 on bf16-rounded operands (x-transformers' -max fill == -inf for rows with a visible key)."""
    qd, kd, vd = (t.to(torch.bfloat16).double() for t in (q, k, v))
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    s = torch.einsum("bihd,bjhd->bhij", qd, kd) * scale
    keep = torch.ones(B, 1, Lq, Lk, dtype=torch.bool)
    if causal:
        keep = keep & (torch.arange(Lk)[None, :] <= torch.arange(Lq)[:, None])[None, None]
    if lens is not None:
        keep = keep & (torch.arange(Lk)[None, :] < torch.tensor(lens)[:, None])[:, None, None, :]
    if kmask is not None:
        keep = keep & kmask.bool()[:, None, None, :]
    s = s.masked_fill(~keep, float("-inf"))
    return torch.einsum("bhij,bjhd->bihd", torch.softmax(s, -1), vd)


@pytest.mark.parametrize("B,H,Lq,Lk,D,causal,lens,rand_mask", [
    (2, 8, 40, 40, 48, False, [40, 13], False), (1, 8, 300, 300, 48, False, None, False),
    (2, 12, 299, 299, 64, True, None, True), (2, 12, 129, 130, 64, False, None, True), (1, 12, 5, 5, 64, True, None, False),
    (3, 12, 300, 300, 64, True, [300, 171, 64], False), (2, 12, 299, 300, 64, False, [300, 201], False),
    (1, 8, 299, 299, 48, False, None, False), (1, 12, 700, 700, 64, True, None, True), (9, 3, 33, 65, 64, False, None, True),
    (2, 8, 64, 64, 48, True, None, False), (1, 12, 1500, 1500, 64, True, None, False)])
def test_row_major_v_attention_matches_f64_and_the_transposed_v_kernel(dev, B, H, Lq, Lk, D, causal, lens, rand_mask):
    """round 4: row-major q / k / v with 48- / 64-wide heads run on attention_tr.hip (5-wave blocks on one XCD per (clip, head),
    V through the transposing LDS read).  Checked against float64 on the same bf16-rounded operands and against the
    transposed-V kernel of attention.hip (whose VROW form was bit-identical to it in round 3): both within bf16 output
    rounding + the bf16 rounding of P."""
    from dimx import engine
    g = torch.Generator().manual_seed(Lq * 17 + D)
    q, k, v = (torch.randn(B, L_, H, D, generator=g) for L_ in (Lq, Lk, Lk))
    scale = 384 ** -0.5 if D == 48 else 0.125
    kmask = None
    if rand_mask:
        kmask = (torch.rand(B, Lk, generator=g) > 0.3).to(torch.uint8)
        kmask[:, 0] = 1
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    args = (q.to(dev), k.to(dev), v.to(dev), scale, causal, lens_t, kmask.to(dev) if kmask is not None else None)
    a = engine.op_attention(*args, bf16=True).float().cpu()
    b = engine.op_attention(*args, bf16=True, row_v=True).float().cpu()
    ref = _attention_f64(q, k, v, scale, causal, lens, kmask).float()
    assert torch.isfinite(b).all()
    for i in range(B):
        n = min(lens[i], Lq) if lens else Lq           # rows of padded queries are never read by valid rows
        assert (b[i, :n] - ref[i, :n]).abs().max().item() < 2.5e-2
        assert (b[i, :n] - a[i, :n]).abs().max().item() < 2.5e-2
        assert (b[i, :n] - ref[i, :n]).abs().mean().item() < 2e-3


def test_vq_argmin_and_sampler(dev, golden_dir):
    from dimx import engine
    from oracle import ref_cpu
    g = torch.Generator().manual_seed(5)
    # sampler vs the oracle and vs the torch.multinomial fixture
    fx = np.load(os.path.join(golden_dir, "sampler_multinomial.npz"))
    tok = engine.op_sample(torch.from_numpy(fx["logits"]).to(dev), 52, 1.0, torch.from_numpy(fx["noise"]).to(dev))
    assert np.array_equal(tok.cpu().numpy(), fx["ids"].astype(np.int32))
    logits = torch.randn(300, 512, generator=g) * 2
    noise = torch.empty(300, 512).exponential_(1, generator=g)
    ref = ref_cpu.sample_tokens(logits, noise)
    tok = engine.op_sample(logits.to(dev), 52, 1.0, noise.to(dev))
    assert (tok.cpu().long() == ref).float().mean().item() > 0.995
    greedy = engine.op_sample(logits.to(dev), 52, 0.0)
    assert torch.equal(greedy.cpu().long(), logits.argmax(-1))
    dr = engine.op_sample(logits.to(dev), 52, 1.0, None, seed=1234, step=3)
    top = ref_cpu.top_k_filter(logits, 52)
    assert torch.isfinite(top.gather(1, dr.cpu().long()[:, None])).all(), "device-RNG sample outside the top-k set"


@pytest.mark.parametrize("B", [256, 200, 40, 8])
def test_chain_kernels_match_torch(B):
    """csrc/chain.hip, the three launch shapes of the decode step at the SLMFT geometry: {out-projection, residual +
    LayerNorm, q-projection}, {out-projection, residual + LayerNorm}, {slabs + LayerNorm, logits}; partial and
    missing 32-clip groups included."""
    from dimx import engine
    torch.manual_seed(B)
    dev = torch.device("cuda:0")
    C, K1, N2 = 1152, 768, 768

    def ref(x, gamma, a1=None, w1=None, slabs=None, w2=None):
        x = x.double()
        if w1 is not None:
            x = x + a1.bfloat16().double() @ w1.bfloat16().double().t()
        if slabs is not None:
            for s in slabs:
                x = x + s.double()
        y = torch.nn.functional.layer_norm(x, (x.shape[1],), gamma.double(), None, 1e-5)
        yb = y.float().bfloat16()
        out2 = (yb.double() @ w2.bfloat16().double().t()) if w2 is not None else None
        return x, y, out2

    a1 = torch.randn(B, K1, device=dev)
    w1 = torch.randn(C, K1, device=dev) / K1 ** 0.5
    w2 = torch.randn(N2, C, device=dev) / C ** 0.5
    wl = torch.randn(512, C, device=dev) / C ** 0.5
    gamma = torch.rand(C, device=dev) * 0.4 + 0.8
    slabs = torch.randn(4, B, C, device=dev) * 0.3
    x0 = torch.randn(B, C, device=dev)
    for kw in (dict(a1=a1, w1=w1, w2=w2), dict(a1=a1, w1=w1), dict(slabs=slabs, w2=wl)):
        x = x0.clone()
        y, out2 = engine.op_chain(x, gamma, **kw)
        rx, ry, ro = ref(x0, gamma, **kw)
        assert (x.double() - rx).abs().max() < 2e-4, "residual stream"
        assert (y.float().double() - ry).abs().max() < 3e-2, "normalised row (bf16)"
        if ro is not None:
            # the kernel's y may differ from the reference's by one bf16 ulp in a few places: compare through its own y
            ro_own = y.double() @ kw["w2"].bfloat16().double().t()
            assert (out2.double() - ro_own).abs().max() < 2e-3, "second projection"
            assert (out2.double() - ro).abs().max() < 5e-2
    # determinism
    x1, x2 = x0.clone(), x0.clone()
    r1 = engine.op_chain(x1, gamma, a1=a1, w1=w1, w2=w2)
    r2 = engine.op_chain(x2, gamma, a1=a1, w1=w1, w2=w2)
    assert torch.equal(x1, x2) and torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])


@pytest.mark.parametrize("B", [256, 200, 33, 8])
def test_chain_deferred_layernorm_matches_torch(B):
    """The deferred-LayerNorm form of the decode step (chain.hip defer = 1 + the decode GEMM's ln epilogue): the chain writes
    x and bf16(x) un-normalised with partial row sums; q-projection (inside the chain) and ff1 (next launch) run on
    gamma-scaled weights and must equal LayerNorm(x) * gamma . W^T; a mean far from zero is included."""
    from dimx import engine
    torch.manual_seed(B + 1)
    dev = torch.device("cuda:0")
    C, K1, N2, NF = 1152, 768, 768, 4608
    a1 = torch.randn(B, K1, device=dev)
    w1 = torch.randn(C, K1, device=dev) / K1 ** 0.5
    w2 = torch.randn(N2, C, device=dev) / C ** 0.5
    wf = torch.randn(NF, C, device=dev) / C ** 0.5
    bf = torch.randn(NF, device=dev)
    gamma = torch.rand(C, device=dev) * 0.4 + 0.8
    x0 = torch.randn(B, C, device=dev) * 1.5 + 0.7            # row means ~0.7 at a standard deviation of ~1.8
    w2s = (w2 * gamma).bfloat16()
    wfs = (wf * gamma).bfloat16()
    cs2, csf = w2s.float().sum(1).contiguous(), wfs.float().sum(1).contiguous()

    rx = x0.double() + a1.bfloat16().double() @ w1.bfloat16().double().t()
    ln = torch.nn.functional.layer_norm(rx, (C,), gamma.double(), None, 1e-5)
    ro2 = ln @ w2.double().t()
    rof = torch.nn.functional.gelu(ln @ wf.double().t() + bf.double())

    # {out-projection, residual, q-projection with the LayerNorm folded in}
    x = x0.clone()
    y, stats, out2 = engine.op_chain_ln(x, a1, w1, w2s, cs2)
    assert (x.double() - rx).abs().max() < 2e-4, "residual stream"
    assert torch.equal(y, x.bfloat16()), "y is the rounded residual stream"
    # stats[g, cu, r] = {sum, centred sum of squares} of a 36-column slice: recombined they are the row's mean / variance
    st = stats[:(B + 31) // 32].double()                   # [groups, 32 CUs, 32 rows, 2]
    n = C // 32
    mean_c = st[..., 0] / n
    mean = st[..., 0].sum(1) / C                           # [groups, 32 rows]
    var = (st[..., 1] + n * (mean_c - mean[:, None]) ** 2).sum(1) / C
    assert (mean.reshape(-1)[:B] - rx.mean(1)).abs().max() < 2e-5
    assert ((var.reshape(-1)[:B] - rx.var(1, unbiased=False)) / rx.var(1, unbiased=False)).abs().max() < 1e-5
    e2 = (out2.double() - ro2).abs().max().item()
    assert e2 < 4e-2, "q projection of the normalised row: %g" % e2   # bf16 operands, K = 1152
    # {out-projection, residual} + ff1 with the ln epilogue
    x = x0.clone()
    y, stats, none = engine.op_chain_ln(x, a1, w1)
    assert none is None and (x.double() - rx).abs().max() < 2e-4
    f = engine.op_gemm_ln(y, wfs, stats, csf, bias=bf, act=3, out_bf16=True)
    ef = (f.float().double() - rof).abs().max().item()
    assert ef < 6e-2, "ff1 of the normalised row: %g" % ef
    # against the same arithmetic done in float64 on the kernel's own bf16 operands: only summation order + bf16 output rounding
    mean, var = rx.mean(1, keepdim=True), rx.var(1, unbiased=False, keepdim=True)
    own = (y.double() @ w2s.double().t() - mean * cs2.double()) / torch.sqrt(var + 1e-5)
    assert (out2.double() - own).abs().max() < 2e-3
    # determinism
    xa, xb = x0.clone(), x0.clone()
    ra = engine.op_chain_ln(xa, a1, w1, w2s, cs2)
    rb = engine.op_chain_ln(xb, a1, w1, w2s, cs2)
    assert torch.equal(xa, xb) and torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2])


def test_deferred_layernorm_statistics_survive_a_large_row_mean_and_raise_the_guard():
    """sum x^2 - mean^2 in f32 cancels catastrophically for |mean| >> std.  The partial statistics are now
    {sum, centred squares} combined by the parallel-variance formula: rows with a mean of 300 at a standard deviation of ~1.8
    keep their variance to 1e-4 relative; and because the deferred form multiplies bf16(x) (not bf16(x - mean)), such rows
    raise flag bit 2 -- dimx_generate answers it by regenerating the batch with the row-phase LayerNorm."""
    from dimx import engine
    torch.manual_seed(11)
    dev = torch.device("cuda:0")
    B, C, K1, NF = 64, 1152, 768, 512
    a1 = torch.randn(B, K1, device=dev)
    w1 = torch.randn(C, K1, device=dev) / K1 ** 0.5
    x0 = torch.randn(B, C, device=dev) * 1.5
    x0[:7] += 300.0                                         # seven rows far from zero
    rx = x0.double() + a1.bfloat16().double() @ w1.bfloat16().double().t()
    x = x0.clone()
    y, stats, none, flags = engine.op_chain_ln(x, a1, w1, return_flags=True)
    st = stats[:2].double()
    n = C // 32
    mean = st[..., 0].sum(1) / C
    var = (st[..., 1] + n * (st[..., 0] / n - mean[:, None]) ** 2).sum(1) / C
    rv = rx.var(1, unbiased=False)
    assert ((var.reshape(-1)[:B] - rv) / rv).abs().max() < 1e-4      # the old form lost every digit here (300^2 vs 3)
    # the consumer's {mean, rstd}: ff1 through the ln epilogue on the small-mean rows is as exact as before, and the
    # consumer raises the precision guard for the offset rows
    wf = torch.randn(NF, C, device=dev) / C ** 0.5
    wfs = wf.bfloat16()
    f = engine.op_gemm_ln(y, wfs, stats, wfs.float().sum(1).contiguous())
    ln = torch.nn.functional.layer_norm(rx, (C,), None, None, 1e-5)
    ref = ln @ wf.double().t()
    assert (f.double()[7:] - ref[7:]).abs().max() < 6e-2
    # the chain's own consumer (second projection inside the launch) raises flag bit 2 for the offset rows, and only then
    w2s = torch.randn(768, C, device=dev).div(C ** 0.5).bfloat16()
    cs2 = w2s.float().sum(1).contiguous()
    x = x0.clone()
    assert engine.op_chain_ln(x, a1, w1, w2s, cs2, return_flags=True)[3] == 4
    x = x0[7:].clone()
    assert engine.op_chain_ln(x, a1[7:], w1, w2s, cs2, return_flags=True)[3] == 0


@pytest.mark.parametrize("M,N,K", [(4096, 512, 384), (8192, 384, 1536), (4096 + 256 * 3, 1152, 448), (6000, 768, 64)])
def test_gemm256_prefill_kernel(M, N, K):
    """csrc/gemm256.hip (phase-pipelined 256 x 256 kernel, taken for M >= 4096): plain / bias + GELU / f32 residual
    epilogues against float64 on the bf16-rounded operands, and against the one-barrier kernel (cfg 14) bit for bit
    where both produce f32 (same products, different summation order -> tolerance, not equality)."""
    from dimx import engine
    torch.manual_seed(M + N + K)
    dev = torch.device("cuda:0")
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    ab, wb = a.bfloat16().double(), w.bfloat16().double()
    ref = ab @ wb.t()
    out = engine.op_gemm(a, w, bf16=True, out_bf16=True)
    assert (out.double() - ref).abs().max() < 0.06
    out = engine.op_gemm(a, w, bias=bias, act=2, bf16=True, out_bf16=True)
    r2 = torch.nn.functional.gelu(ref + bias.double(), approximate="tanh")
    assert (out.double() - r2).abs().max() < 0.06
    out = engine.op_gemm(a, w, bias=bias, residual=res, bf16=True)
    old = engine.op_gemm(a, w, bias=bias, residual=res, bf16=True, cfg=14)
    assert (out.double() - (ref + bias.double() + res.double())).abs().max() < 2e-3
    assert (out - old).abs().max() < 1e-3
    # repeated launches are bit-identical (the pipeline has no timing-dependent summation order / races)
    for _ in range(3):
        assert torch.equal(out, engine.op_gemm(a, w, bias=bias, residual=res, bf16=True))


@pytest.mark.parametrize("nlayers", [1, 4])
def test_gemm256_head_major_kv_layout(nlayers):
    """the cross-attention K/V projection's destination: [B,H,Tp,64] K and V caches of 1 or all 4 decoder layers in one
    launch (dimx_op_gemm_headmajor)."""
    from dimx import lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, T, H, K = 24, 300, 12, 1152
    nseg = 2 * nlayers
    Tp, N, M = 304, nseg * H * 64, B * T
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    out = torch.zeros(nseg, B, H, Tp, 64, device=dev, dtype=torch.bfloat16)
    L.check(lib.dimx_op_gemm_headmajor(L.BF16, L.ptr(a), K, L.ptr(w), K, L.ptr(out), M, N, K, T, Tp, nlayers,
                                       L.stream_ptr(dev)), "gemm_headmajor")
    ref = (a.double() @ w.double().t()).view(B, T, nseg, H, 64).permute(2, 0, 3, 1, 4)     # [nseg,B,H,T,64]
    assert (out[:, :, :, :T].double() - ref).abs().max() < 0.06
    assert float(out[:, :, :, T:].abs().max()) == 0.0                                      # padding rows untouched
