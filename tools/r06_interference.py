"""Round 6, VERDICT item 1: what do two half-chip engines cost each other?

Engine-sized pieces of the decode step at 128 clips -- the HBM stream (decode cross attention, 128 clips x 300 keys) and the
latency-bound projections (the three chip-wide decode GEMMs at M = 128: 128 one-block-per-CU blocks of gemm_ws72_kernel) -- timed
with HIP events on their own stream, alone and while the OTHER piece loops on a second stream:
  * plain streams (the dispatcher places blocks where it likes);
  * CU-masked streams, engine A = CUs 0-15 of every XCD, engine B = CUs 16-31 of every XCD (the only half-chip partition this
    platform honours: a mask that leaves an XCD without CUs is dropped, profiles/r06_xcdmask_probe.txt).
"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import engine as E
from dimx import lib as L

dev = torch.device("cuda:0")
lib = L.load()
hip = ctypes.CDLL("libamdhip64.so")
H, T, Tp, LAYERS = 12, 300, 304, 4
SPIN = 200_000_000   # cycles: ~100 ms, longer than the host needs to enqueue a measurement


def masked_stream(half):
    """half 0: CUs 0-15 of every XCD (mask bits i with (i // 8) < 16), half 1: the others; None: an ordinary stream"""
    if half is None:
        return torch.cuda.Stream()
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if ((i // 8) < 16) == (half == 0):
            words[i >> 5] |= 1 << (i & 31)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def make_attn(B):
    kc = [torch.randn(B, H, Tp, 64, device=dev).bfloat16() for _ in range(LAYERS)]
    vc = [torch.randn(B, H, Tp, 64, device=dev).bfloat16() for _ in range(LAYERS)]
    q = torch.randn(B, H * 64, device=dev)
    km = torch.ones(B, T, dtype=torch.uint8, device=dev)
    return lambda i: E.op_decode_attn(q, kc[i % LAYERS], vc[i % LAYERS], T, 0.125, km)


def make_gemms(M):
    shapes = [(2304, 1152, 2), (4608, 1152, 1), (1152, 4608, 4)]      # qkv, ff1, ff2: (N, K, split-K slabs) = 128 blocks each at M = 128
    a = {K: torch.randn(M, K, device=dev).bfloat16() for K in (1152, 4608)}
    ws = [[(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for (N, K, _) in shapes] for _ in range(LAYERS)]
    outs = [torch.empty(sp, M, N, device=dev) for (N, K, sp) in shapes]

    def run(i):
        for j, (N, K, sp) in enumerate(shapes):
            flags = 1 | (1 << 2) | (sp << 16)
            L.check(lib.dimx_op_gemm(L.BF16, L.F32, L.ptr(a[K]), K, L.ptr(ws[i % LAYERS][j]), K, L.ptr(outs[j]), N, M, N, K, None, 0,
                                     None, 0, 0, None, flags, L.stream_ptr(dev)), "gemm")
    return run


def timed(fn, stream, n, warm=20):
    with torch.cuda.stream(stream):
        for i in range(warm):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(n):
            fn(i)
        e1.record(stream)
    return e0, e1


def beside(fg, fg_stream, bg, bg_stream, n_fg, n_bg):
    """time n_fg launches of `fg` while `bg` keeps its stream busy (n_bg launches, issued first and long enough to outlast fg)"""
    torch.cuda.synchronize()
    # both streams are parked behind a spin kernel while the host enqueues (a Python launch costs about what a launch runs): the
    # two loops then run from full queues, side by side from their first launch
    with torch.cuda.stream(bg_stream):
        torch.cuda._sleep(SPIN)
    with torch.cuda.stream(fg_stream):
        torch.cuda._sleep(SPIN)
    with torch.cuda.stream(bg_stream):
        for i in range(n_bg):
            bg(i)
    e0, e1 = timed(fg, fg_stream, n_fg, warm=10)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n_fg


def alone(fn, stream, n):
    torch.cuda.synchronize()
    e0, e1 = timed(fn, stream, n)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    attn128, gemm128 = make_attn(128), make_gemms(128)
    attn256, gemm256 = make_attn(256), make_gemms(256)
    bytes128 = 128 * H * T * 64 * 2 * 2
    print("decode cross attention, 300 keys, bf16 (us per launch; GB/s of K/V) and the three chip-wide decode GEMMs (us per {qkv, ff1, ff2} triple)")
    s = torch.cuda.Stream()
    a256, g256 = alone(attn256, s, 300), alone(gemm256, s, 300)
    print("one engine of 256 clips:   attention %6.2f us (%.0f GB/s)   GEMM triple %6.2f us" % (a256, 2 * bytes128 / a256 / 1e3, g256))
    for label, ha, hb in (("plain streams", None, None), ("CU-masked halves (16 CUs of every XCD each)", 0, 1)):
        sa, sb = masked_stream(ha), masked_stream(hb)
        a_al, g_al = alone(attn128, sa, 300), alone(gemm128, sb, 300)
        # background loops long enough to outlast the timed foreground
        a_bg = beside(attn128, sa, gemm128, sb, 200, 1200)
        g_bg = beside(gemm128, sb, attn128, sa, 200, 1500)
        a_aa = beside(attn128, sa, attn128, sb, 200, 500)
        g_gg = beside(gemm128, sb, gemm128, sa, 200, 500)
        print("two engines of 128 clips, %s:" % label)
        print("  attention   alone %6.2f us (%.0f GB/s) | beside the other engine's GEMMs %6.2f (%.0f GB/s) | beside its attention %6.2f (%.0f GB/s each)"
              % (a_al, bytes128 / a_al / 1e3, a_bg, bytes128 / a_bg / 1e3, a_aa, bytes128 / a_aa / 1e3))
        print("  GEMM triple alone %6.2f us             | beside the other engine's attention %6.2f          | beside its GEMMs %6.2f"
              % (g_al, g_bg, g_gg))
        # what the pair costs per layer: one engine = attention x 2 (self at mean fill ~ half) + triple ...: report the simple sum model
        per_engine_alone = 1.5 * a_al + g_al
        per_engine_beside = 1.5 * a_bg + g_bg
        print("  model, per layer and engine (1.5 attention launches + one GEMM triple): alone %.1f us, beside the other engine %.1f us; "
              "one 256-clip engine %.1f us for both halves" % (per_engine_alone, per_engine_beside, 1.5 * a256 + g256))


if __name__ == "__main__":
    main()
