"""CPU: the host-side weight packing of the fused feed-forward kernel (dimx_mlp_fused_pack, csrc/mlp_fused.hip) against the layout its
header documents: per chunk of 32 hidden units 25 W1 fragments (lane l, element j = W1[32 c + l % 32][16 s + 8 (l / 32) + j]; the 25th
carries b1 as a bf16 hi + lo pair in k-slots 0 / 1 of the lower half-wave) and 24 W2 fragments whose k-slots follow the accumulator-row
order of the first product (hidden index (2 s2 + j / 4) 8 + 4 (l / 32) + j % 4).  No GPU involved: a pure host function of the library."""
import ctypes

import numpy as np
import torch


def _bf16_bits(x):
    return (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).view(torch.int16).numpy().astype(np.int64)) & 0xFFFF


def _bf16_val(bits):
    return (torch.from_numpy(bits.astype(np.int16)).view(torch.bfloat16).float().numpy())


def test_pack_layout_matches_the_documented_fragment_order():
    from dimx import lib as L
    lib = L.load()
    C, F = 384, 96
    rng = np.random.default_rng(3)
    w1 = rng.standard_normal((F, C)).astype(np.float32)
    b1 = rng.standard_normal(F).astype(np.float32)
    w2 = rng.standard_normal((C, F)).astype(np.float32)
    nbytes = int(lib.dimx_mlp_fused_packed_bytes(C, F))
    assert nbytes == (F // 32) * 49 * 1024
    assert int(lib.dimx_mlp_fused_packed_bytes(512, F)) == 0 and int(lib.dimx_mlp_fused_packed_bytes(C, 100)) == 0
    out = np.zeros(nbytes // 2, dtype=np.uint16)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    L.check(lib.dimx_mlp_fused_pack(p(w1), p(b1), p(w2), C, F, p(out), nbytes), "mlp_fused_pack")
    img = out.reshape(F // 32, 49, 64, 8).astype(np.int64)
    lanes = np.arange(64)
    row, half = lanes % 32, lanes // 32
    for c in range(F // 32):
        for s in range(24):          # W1 k-steps
            cols = 16 * s + 8 * half[:, None] + np.arange(8)[None, :]
            want = _bf16_bits(w1[(32 * c + row)[:, None], cols])
            assert np.array_equal(img[c, s], want), (c, s)
        hi = _bf16_bits(b1[32 * c + row])
        lo = _bf16_bits(b1[32 * c + row] - _bf16_val(hi))
        want = np.zeros((64, 8), dtype=np.int64)
        want[:32, 0], want[:32, 1] = hi[:32], lo[:32]
        assert np.array_equal(img[c, 24], want), c
        # hi + lo reproduces b1 to ~2^-17 relative
        assert np.allclose(_bf16_val(hi[:32]) + _bf16_val(lo[:32]), b1[32 * c:32 * c + 32], rtol=2e-5, atol=1e-7)
        for ob in range(12):         # W2: out block ob, k-step s2
            for s2 in range(2):
                j = np.arange(8)
                hid = 32 * c + (2 * s2 + j[None, :] // 4) * 8 + 4 * half[:, None] + j[None, :] % 4
                want = _bf16_bits(w2[(32 * ob + row)[:, None], hid])
                assert np.array_equal(img[c, 25 + 2 * ob + s2], want), (c, ob, s2)
    # every W2 hidden column of the chunk appears exactly once per k-step pair and half-wave pair
    seen = sorted(((2 * s2 + j // 4) * 8 + 4 * h + j % 4) for s2 in range(2) for h in range(2) for j in range(8))
    assert seen == list(range(32))


def test_pack_rejects_a_short_buffer():
    from dimx import lib as L
    lib = L.load()
    w = np.zeros((32, 384), dtype=np.float32)
    w2 = np.zeros((384, 32), dtype=np.float32)
    out = np.zeros(16, dtype=np.uint16)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert lib.dimx_mlp_fused_pack(p(w), None, p(w2), 384, 32, p(out), 32) != 0
