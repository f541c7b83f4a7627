"""GPU: the f32 parity mode's split-bf16 decode GEMM (csrc/gemm_x3.hip) through the C-ABI.

An f32 number is the exact sum of three bf16 numbers; the kernel multiplies the planes on the bf16 matrix cores and accumulates in
f32, i.e. it computes an f32 GEMM (reference arithmetic: the fp32 Linear layers of the decoder, code/seq2seq_pretrain.py:413-418).
Checked here: the split is EXACT on both operands (one-hot probes come back bit for bit), the result is as close to float64 as the
exact-f32 MFMA kernel it replaces, the slab form sums to the same, rows do not depend on the batch they are computed in."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

ACTS = {0: lambda x: x, 1: lambda x: F.leaky_relu(x, 0.2),
        2: lambda x: x * 0.5 * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3))), 3: lambda x: F.gelu(x)}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def test_both_operand_splits_are_exact(dev):
    """one-hot A rows return W's columns, one-hot W rows return A's columns -- bit for bit, for values spanning 60 binades"""
    from dimx import engine as E
    K, N, M = 1152, 1152, 256
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-30, 30, (N, K), generator=g).float())).to(dev)
    a = torch.zeros(M, K, device=dev)
    a[torch.arange(M), torch.arange(M) * 4 + 1] = 1.0
    out = E.op_gemm_x3(a, w)
    assert torch.equal(out, w[:, torch.arange(M) * 4 + 1].t().contiguous())
    a = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-30, 30, (M, K), generator=g).float())).to(dev)
    w = torch.zeros(N, K, device=dev)
    w[torch.arange(N), torch.arange(N) % K] = 1.0
    out = E.op_gemm_x3(a, w)
    assert torch.equal(out, a[:, torch.arange(N) % K])


@pytest.mark.parametrize("M,N,K,act,use_bias,use_res", [
    (256, 2304, 1152, 0, False, False),      # fused q/k/v                (64-column tiles)
    (256, 4608, 1152, 3, True, False),       # ff1 + bias + erf-GELU      (no slabs: one pass over K)
    (256, 1152, 4608, 0, True, True),        # ff2 with bias + residual
    (200, 768, 1152, 0, False, False),       # cross-q, ragged M          (96-column tiles)
    (4, 512, 1152, 0, False, False),         # logits at B = 4
    (128, 1152, 768, 2, True, True),         # out-projection shape, tanh-GELU
    (37, 1536, 512, 1, True, False),         # legacy decoder (dim 512) fused q/k/v
    (256, 36 * 5, 64, 0, False, False)])     # 36-column tiles, two k-tiles
def test_x3_gemm_is_an_f32_gemm(dev, M, N, K, act, use_bias, use_res):
    from dimx import engine as E
    a, w = _rand((M, K), dev, 2 + M), _rand((N, K), dev, 3 + N, K ** -0.5)
    bias = _rand((N,), dev, 4) if use_bias else None
    res = _rand((M, N), dev, 5) if use_res else None
    out = E.op_gemm_x3(a, w, bias, act, res)
    f32 = E.op_gemm(a, w, bias, act, res)                      # the exact-f32 MFMA kernel it replaces
    ref = a.double() @ w.double().t()
    if bias is not None:
        ref = ref + bias.double()
    ref = ACTS[act](ref)
    if res is not None:
        ref = ref + res.double()
    scale = (a.double().abs() @ w.double().abs().t()).max().item()      # what the rounding errors scale with
    e_x3, e_f32 = (out.double() - ref).abs().max().item() / scale, (f32.double() - ref).abs().max().item() / scale
    assert torch.isfinite(out).all()
    assert e_x3 < 3e-7, (e_x3, e_f32)                          # f32 accumulation of K terms: ~1e-7 of sum |a||w|
    assert e_x3 < 4 * e_f32 + 2e-8, (e_x3, e_f32)              # and no worse than the f32 MFMA kernel


@pytest.mark.parametrize("M,N,K", [(256, 2304, 1152), (256, 1152, 4608), (256, 1152, 768), (130, 512, 1152), (256, 768, 1152),
                                   (600, 1152, 768)])   # more rows than the design point: the decode step of a batch > 256 takes the same kernel
def test_x3_slabs_sum_to_the_product_and_rows_do_not_depend_on_the_batch(dev, M, N, K):
    from dimx import engine as E
    a, w = _rand((M, K), dev, 7 + N), _rand((N, K), dev, 8 + K, K ** -0.5)
    slabs = E.op_gemm_x3(a, w, slabs=True)
    assert 1 <= slabs.shape[0] <= 8 and torch.isfinite(slabs).all()
    total = slabs[0].clone()
    for s in range(1, slabs.shape[0]):
        total += slabs[s]                                        # slab order, what the consumers do
    ref = a.double() @ w.double().t()
    scale = (a.double().abs() @ w.double().abs().t()).max().item()
    assert (total.double() - ref).abs().max().item() / scale < 3e-7
    # a shard of the batch (a rank's rows) reproduces the rows of the whole batch bit for bit: tile and split count
    # depend on (N, K) only
    lo, hi = M // 4, M // 4 + max(1, M // 2)
    part = E.op_gemm_x3(a[lo:hi].contiguous(), w, slabs=True)
    assert part.shape[0] == slabs.shape[0] and torch.equal(part, slabs[:, lo:hi])
    one = E.op_gemm_x3(a[lo:lo + 1].contiguous(), w, slabs=True)
    assert torch.equal(one, slabs[:, lo:lo + 1])
