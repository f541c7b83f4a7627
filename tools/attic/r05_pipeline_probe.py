"""Round 5 probe: can the next batch's prefill hide under the current batch's decode loop?  Two engines (own workspaces): engine A generates
(299 steps, latency-bound) on one stream while engine B runs VQ encode + encode_ctx of another batch on a second stream (optionally a
low-priority one); against the same two pieces of work one after the other."""
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
ea, eb = engine.Engine(dev, lib.MODE_PERF_BF16), engine.Engine(dev, lib.MODE_PERF_BF16)
ea.load_state_dict(sd)
eb.load_state_dict(sd)
z = ea.vq_encode(1, v_l, lens, pe_mode=0, pad_value=-100)
start = z[:, 0].contiguous()
ea.encode_ctx(v_s, v_a, m8, True)


def prefill(e):
    e.vq_encode(1, v_l, lens, pe_mode=0, pad_value=-100)
    e.encode_ctx(v_s, v_a, m8, True)


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


gen = lambda: ea.generate(start, m8, T, 1.0, 52, None, 7)
print("generate alone           %.2f ms" % timed(gen))
print("prefill alone            %.2f ms" % timed(lambda: prefill(eb)))
print("generate, then prefill   %.2f ms" % timed(lambda: (gen(), prefill(eb))))
for prio, name in ((0, "default priority"), (None, "lowest priority")):
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    s2 = torch.cuda.Stream(priority=0 if prio == 0 else 0)
    if prio is None:
        try:
            s2 = torch.cuda.Stream(priority=lo)
        except Exception as e:  # noqa
            print("no low-priority stream:", e)

    def both():
        ev = torch.cuda.Event()
        ev.record()
        s2.wait_event(ev)
        with torch.cuda.stream(s2):
            prefill(eb)
        gen()
        torch.cuda.current_stream().wait_stream(s2)
    print("generate || prefill (%s) %.2f ms" % (name, timed(both)))
