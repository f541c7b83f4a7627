"""Round 6 ("measure a 3-product variant too and report its token-flip rate, do not ship it as parity").
One numeric variant of the f32 parity mode's decode GEMMs per process (the switches are read once):
    exact    DIMX_NO_X3=1                      v_mfma_f32_32x32x2_f32, split-K planned by the f32 rule
    exact2   DIMX_NO_X3=1 DIMX_F32_NO_SPLIT=1  the same products, another summation order (no split-K): the yardstick -- two f32 GEMMs
                                               that differ only in the order of their additions
    x3       (default)                         three bf16 planes per operand, six products (csrc/gemm_x3.hip)
    x3_3     DIMX_X3_ABL=4                     the three leading products only (16 significand bits per operand)
usage: python tools/r06_x3_terms.py gen <name> <out.npz>   |   python tools/r06_x3_terms.py cmp <a.npz> <b.npz> ...
C3's 256 synthetic clips, T = 300, sampler seed fixed: free-running generation, tokens + decoded motion."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")


def gen(name, out):
    import dimx  # noqa
    import bench
    from dimx import lib as L
    from dimx.seq2seq_pretrain import SLMFT
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    m = SLMFT(synthetic_seed=bench.SEED, numeric_mode=L.MODE_PARITY_F32).eval()
    v_s, v_l, v_a, mask = bench.synth_batch(256, 300, dev, salt=0)
    for i in range(2):
        m(v_s, v_l, v_a, mask, mode="val", seed=5 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        _, _, pred, tok = m(v_s, v_l, v_a, mask, mode="val", seed=bench.SEED + 777, return_tokens=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    np.savez(out, name=name, tokens=tok.reshape(256, -1).cpu().numpy(), pred=pred.float().cpu().numpy(), clips_per_s=256 / dt)
    print(name, "%.1f clips/s" % (256 / dt))


def cmp(files):
    d = [np.load(f) for f in files]
    ref = d[0]
    rows = []
    for x in d:
        same = x["tokens"] == ref["tokens"]
        whole = same.all(1)
        first = np.where(whole, same.shape[1], (~same).argmax(1))
        rows.append({"variant": str(x["name"]), "clips_per_s": float(x["clips_per_s"]), "vs": str(ref["name"]),
                     "clips_with_identical_sequences": int(whole.sum()), "clips": int(same.shape[0]),
                     "free_running_token_agreement": float(same.mean()), "mean_steps_before_first_flip": float(first.mean()),
                     "max_abs_motion_difference": float(np.abs(x["pred"] - ref["pred"]).max())})
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "gen":
        gen(sys.argv[2], sys.argv[3])
    else:
        cmp(sys.argv[2:])
