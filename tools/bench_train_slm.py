"""Wall time of the SLM pre-training step and the legacy generator's step on the HIP kernels (dimx.train_hip.SlmHipTrainer /
LegacyHipTrainer: forward + backward + clip + AdamW) next to the PyTorch-autograd restatements they replace (dimx.train).
    python tools/bench_train_slm.py [B=16] [T=300] [steps=5] [which=all|slm|legacy] [autograd=1]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L
from dimx import prng
from dimx import train as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Tn = int(sys.argv[2]) if len(sys.argv) > 2 else 300
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
which = sys.argv[4] if len(sys.argv) > 4 else "all"
with_autograd = (sys.argv[5] if len(sys.argv) > 5 else "1") != "0"
dev = torch.device("cuda:0")
mask = torch.ones(B, Tn, dtype=torch.bool, device=dev)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if which in ("all", "slm"):
    from dimx.seq2seq_pretrain import SLM
    from dimx.train_hip import SlmHipTrainer
    v_s = torch.from_numpy(prng.normal(1, "bt.vs", (B, Tn, 56))).to(dev)
    v_l = torch.from_numpy(prng.normal(1, "bt.vl", (B, Tn, 56))).to(dev)
    v_a = torch.from_numpy(prng.normal(1, "bt.va", (B, Tn, 768))).to(dev)
    for mode, name in ((L.MODE_PERF_BF16, "bf16"), (L.MODE_PARITY_F32, "f32")):
        m = SLM(numeric_mode=mode).to(dev)
        T.set_slm_trainable(m)
        m.train()
        with torch.no_grad():
            z_s, z_l = m.forward_vq(v_s, v_l, mask)
        ms_, ml_ = m.random_masking_unstructured(v_s, mask, 0.15), m.random_masking_unstructured(v_l, mask, 0.15)
        tr = SlmHipTrainer(m, lr=1e-5, clip=1.0)
        t = timed(lambda: tr.train_step(v_s, v_l, v_a, mask, mask_speaker=ms_, mask_listener=ml_, z_s=z_s, z_l=z_l))
        print("HIP SLM pre-training step  %-4s B=%d T=%d: %8.1f ms  (%.1f clips/s)" % (name, B, Tn, t, B / t * 1e3), flush=True)
        del tr
        if mode == L.MODE_PARITY_F32 and with_autograd:
            opt = torch.optim.AdamW([p for _, p in T.slm_trainable_parameters(m)], lr=1e-5)

            def step():
                opt.zero_grad()
                with torch.enable_grad():
                    total, _, _ = m(v_s, v_l, v_a, mask, mask_speaker=ms_, mask_listener=ml_, z_s=z_s, z_l=z_l)
                    total.backward()
                torch.nn.utils.clip_grad_norm_([p for _, p in T.slm_trainable_parameters(m)], 1.0)
                opt.step()
            t = timed(step)
            print("autograd SLM step (dimx.train, rocBLAS / hipBLASLt) f32 B=%d T=%d: %8.1f ms  (%.1f clips/s)" % (B, Tn, t, B / t * 1e3), flush=True)
        del m
        torch.cuda.empty_cache()

if which in ("all", "legacy"):
    from dimx.seq2seq import ListenerGenerator
    from dimx.train_hip import LegacyHipTrainer
    v_s = torch.from_numpy(prng.normal(1, "bt.lvs", (B, Tn, 824))).to(dev)
    v_l = torch.from_numpy(prng.normal(1, "bt.lvl", (B, Tn, 56))).to(dev)
    lid = (torch.arange(B) % 100).to(dev)
    for mode, name in ((L.MODE_PERF_BF16, "bf16"), (L.MODE_PARITY_F32, "f32")):
        m = ListenerGenerator(numeric_mode=mode).to(dev)
        T.set_legacy_trainable(m)
        m.train()
        tr = LegacyHipTrainer(m, lr=1e-5, clip=1.0)
        t = timed(lambda: tr.train_step(v_s, v_l, mask, listener_ids=lid))
        print("HIP legacy generator step  %-4s B=%d T=%d: %8.1f ms  (%.1f clips/s)   [incl. the frozen speaker VQ-VAE encoder + listener VQ encode on the engine]"
              % (name, B, Tn, t, B / t * 1e3), flush=True)
        del tr
        if mode == L.MODE_PARITY_F32 and with_autograd:
            opt = torch.optim.AdamW([p for _, p in T.legacy_trainable_parameters(m)], lr=1e-5)

            def step():
                opt.zero_grad()
                with torch.enable_grad():
                    loss, _ = m(v_s, v_l, mask, speaker_ids=None, listener_ids=lid)
                    loss.backward()
                torch.nn.utils.clip_grad_norm_([p for _, p in T.legacy_trainable_parameters(m)], 1.0)
                opt.step()
            t = timed(step)
            print("autograd legacy step (dimx.train) f32 B=%d T=%d: %8.1f ms  (%.1f clips/s)" % (B, Tn, t, B / t * 1e3), flush=True)
        del m
        torch.cuda.empty_cache()
