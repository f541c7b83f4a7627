"""Round 5 experiment: does the prefill (listener VQ encode + encoders + context + cross K/V projection) of a 256-clip batch finish
sooner as two clip groups on two streams (218 + 38 clips: the big group's 128-row kernels then run whole rounds of 256 blocks, the
small group fills the tails) than as one batch?  Two engines (own workspaces), same weights."""
import sys
import time

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import engine, lib, weights
from bench import synth_batch

dev = torch.device("cuda:0")
sd = weights.synth_state_dict(weights.slmft_spec(), 20260928)
B, T = 256, 300
v_s, v_l, v_a, mask = synth_batch(B, T, dev, salt=0)
m8 = mask.to(torch.uint8).contiguous()
lens = mask.sum(1).to(torch.int32)


def prefill(e, sl):
    e.vq_encode(1, v_l[sl].contiguous(), lens[sl].contiguous(), pe_mode=0, pad_value=-100)
    e.encode_ctx(v_s[sl].contiguous(), v_a[sl].contiguous(), m8[sl].contiguous(), True)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


e0 = engine.Engine(dev, lib.MODE_PERF_BF16)
e0.load_state_dict(sd)
print("one batch of 256:            %.2f ms" % timed(lambda: prefill(e0, slice(0, B))))
# all three prefill-sized stages of a forward (VQ encode, encoders + context + K/V projection, VQ decode of 299 codes) in G equal groups on G streams
idx = torch.randint(0, 512, (B, T - 1), device=dev, dtype=torch.int32)


def stages(e, sl):
    prefill(e, sl)
    e.vq_decode(1, idx[sl].contiguous())


print("three stages, one batch:     %.2f ms" % timed(lambda: stages(e0, slice(0, B))))
for G in (2, 3, 4):
    es = []
    for _ in range(G):
        e = engine.Engine(dev, lib.MODE_PERF_BF16)
        e.load_state_dict(sd)
        es.append(e)
    ss = [torch.cuda.Stream() for _ in range(G)]
    bounds = [B * i // G for i in range(G + 1)]

    def groups():
        ev = torch.cuda.Event()
        ev.record()
        for i in range(G):
            ss[i].wait_event(ev)
            with torch.cuda.stream(ss[i]):
                stages(es[i], slice(bounds[i], bounds[i + 1]))
        for i in range(G):
            torch.cuda.current_stream().wait_stream(ss[i])
    print("three stages, %d groups on %d streams: %.2f ms" % (G, G, timed(groups)))
    for e in es:
        e.close()
for nbig in (218, 192, 128):
    e1, e2 = engine.Engine(dev, lib.MODE_PERF_BF16), engine.Engine(dev, lib.MODE_PERF_BF16)
    e1.load_state_dict(sd)
    e2.load_state_dict(sd)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        ev = torch.cuda.Event()
        ev.record()
        s1.wait_event(ev)
        s2.wait_event(ev)
        with torch.cuda.stream(s1):
            prefill(e1, slice(0, nbig))
        with torch.cuda.stream(s2):
            prefill(e2, slice(nbig, B))
        torch.cuda.current_stream().wait_stream(s1)
        torch.cuda.current_stream().wait_stream(s2)

    def serial():
        prefill(e1, slice(0, nbig))
        prefill(e2, slice(nbig, B))
    print("%3d + %3d on two streams:     %.2f ms   (the same two calls on one stream: %.2f ms)" % (nbig, B - nbig, timed(both), timed(serial)))
    e1.close()
    e2.close()
