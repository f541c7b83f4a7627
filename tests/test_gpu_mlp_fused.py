"""GPU: the fused feed-forward sublayer of the prefill (csrc/mlp_fused.hip through dimx_op_mlp_fused) against a float64 evaluation of
x + W2 . gelu(W1 . LayerNorm(x) + b1) + b2 -- the MLP sublayers of the VQ-VAE transformers (reference
code/models/lib/base_models.py:56-68: pre-LN with bias, tanh-GELU) and of the x-transformers encoders (pre-LN without bias, erf-GELU).
The kernel multiplies bf16 operands with f32 accumulation; the reference rounds at the same three points (LayerNorm output, weights,
GELU output), so what is left is summation order and the fast exp / rcp forms: <= 2e-3 of the sublayer's output scale.  Against the
unrounded float64 sublayer the error is bf16's (reported, bounded at 2e-2)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

C, F = 384, 1536


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _gelu(h, act):
    if act == 2:
        return 0.5 * h * (1.0 + torch.tanh(0.7978845608028654 * (h + 0.044715 * h ** 3)))
    return 0.5 * h * (1.0 + torch.erf(h * 0.7071067811865476))


def _reference(x, w1, b1, w2, b2, g, be, act, rounded):
    x = x.double()
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) / torch.sqrt(var + 1e-5) * g.double() + (be.double() if be is not None else 0.0)
    r = _bf if rounded else (lambda t: t.double())
    h = r(y.float()) @ r(w1).t() + b1.double()
    return x + r(_gelu(h, act).float()) @ r(w2).t() + b2.double()


@pytest.mark.parametrize("M,act,beta", [(128, 2, True), (300, 3, False), (8200, 2, True), (4099, 3, False), (1, 3, True)])
def test_fused_mlp_matches_float64(M, act, beta):
    from dimx import lib as L
    from dimx import prng
    lib = L.load()
    dev = torch.device("cuda:0")
    seed = 100 + M
    x = torch.from_numpy(prng.normal(seed, "mlp.x", (M, C))) * 1.5 + 0.3
    x[:, 5] += 4.0                                                    # a row mean far from zero: the pivoted variance must hold
    w1 = torch.from_numpy(prng.uniform(seed, "mlp.w1", (F, C), -C ** -0.5, C ** -0.5))
    b1 = torch.from_numpy(prng.uniform(seed, "mlp.b1", (F,), -C ** -0.5, C ** -0.5))
    w2 = torch.from_numpy(prng.uniform(seed, "mlp.w2", (C, F), -F ** -0.5, F ** -0.5))
    b2 = torch.from_numpy(prng.uniform(seed, "mlp.b2", (C,), -F ** -0.5, F ** -0.5))
    g = torch.from_numpy(prng.uniform(seed, "mlp.g", (C,), 0.8, 1.2))
    be = torch.from_numpy(prng.uniform(seed, "mlp.be", (C,), -0.1, 0.1)) if beta else None
    xd = x.to(dev).contiguous()
    keep = [t.to(dev).contiguous() if t is not None else None for t in (b2, g, be)]
    host = [t.contiguous() for t in (w1, b1, w2)]
    hp = lambda t: ctypes.c_void_p(t.data_ptr())
    L.check(lib.dimx_op_mlp_fused(L.ptr(xd), hp(host[0]), hp(host[1]), hp(host[2]), L.ptr(keep[0]), L.ptr(keep[1]), L.ptr(keep[2]),
                                  M, C, F, act, L.stream_ptr(dev)), "dimx_op_mlp_fused")
    got = xd.cpu().double()
    want = _reference(x, w1, b1, w2, b2, g, be, act, rounded=True)
    exact = _reference(x, w1, b1, w2, b2, g, be, act, rounded=False)
    scale = (exact - x.double()).abs().max().item()
    err = (got - want).abs().max().item()
    err_exact = (got - exact).abs().max().item()
    print("fused MLP M=%d act=%d: max |err| vs the bf16-rounded float64 form %.2e, vs the exact sublayer %.2e (output scale %.2f)" % (
        M, act, err, err_exact, scale))
    assert err < 2e-3 * max(scale, 1.0), (err, scale)
    assert err_exact < 2e-2 * max(scale, 1.0), (err_exact, scale)


def test_fused_mlp_full_size_is_row_independent_and_deterministic():
    """M = 76 800 (the headline row count, 600 blocks = 2.34 rounds of the chip): a row's result depends on nothing but the row --
    permuting the rows permutes the output bit for bit, a row copied to another block's tile gives the same bits -- and two launches
    give identical results (no atomics, fixed summation order); a sample of rows against the float64 form."""
    from dimx import lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    M = 76800
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, C, generator=g)
    x[12345] = x[7]                                   # the same row in two different blocks / waves / lanes
    w1 = torch.randn(F, C, generator=g) * C ** -0.5
    b1 = torch.randn(F, generator=g) * 0.05
    w2 = torch.randn(C, F, generator=g) * F ** -0.5
    b2 = (torch.randn(C, generator=g) * 0.02).to(dev)
    gam = (torch.rand(C, generator=g) + 0.5).to(dev)
    nbytes = int(lib.dimx_mlp_fused_packed_bytes(C, F))
    host = torch.empty(nbytes, dtype=torch.uint8)
    hp = lambda t: ctypes.c_void_p(t.data_ptr())
    L.check(lib.dimx_mlp_fused_pack(hp(w1), hp(b1), hp(w2), C, F, hp(host), nbytes), "mlp_fused_pack")
    packed = host.to(dev)

    def run(inp):
        y = inp.to(dev).contiguous()
        L.check(lib.dimx_op_mlp_fused_packed(L.ptr(y), L.ptr(packed), L.ptr(b2), L.ptr(gam), None, M, C, F, 3, L.stream_ptr(dev)), "mlp_fused")
        torch.cuda.synchronize()
        return y.cpu()
    y0, y1 = run(x), run(x)
    assert torch.equal(y0, y1)
    assert torch.equal(y0[12345], y0[7])
    perm = torch.randperm(M, generator=g)
    assert torch.equal(run(x[perm]), y0[perm])
    rows = torch.tensor([0, 7, 127, 128, 4099, 65535, 65536, 76799])
    want = _reference(x[rows], w1, b1, w2, b2.cpu(), gam.cpu(), None, 3, rounded=True)
    assert (y0[rows].double() - want).abs().max().item() < 4e-3
