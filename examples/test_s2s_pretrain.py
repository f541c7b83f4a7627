"""The reference's evaluation driver (code/test_s2s_pretrain.py) on the dimx drop-ins: the four imports are the
only lines that change.  Without the ViCo files / checkpoint it runs on synthetic clips and weights.

    python examples/test_s2s_pretrain.py [--clips 32] [--batch 8] [--beam 10] [--bf16] [--ckpt best_vico_causal.pt]
"""
import argparse
import os
import pickle
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dimx  # noqa: E402,F401
from dimx import lib as L  # noqa: E402
from dimx.dataset.data_loader import get_vico_dataloaders  # noqa: E402   (was: from dataset.data_loader import ...)
from dimx.mymetrics import print_metrics, print_metrics_full  # noqa: E402 (was: from mymetrics import ...)
from dimx.seq2seq_pretrain import SLMFT  # noqa: E402                     (was: from seq2seq_pretrain import SLMFT)
from dimx.x_engine_pt import evaluate_test_epoch  # noqa: E402            (was: from x_engine_pt import ...)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--beam", type=int, default=10)
    ap.add_argument("--max-len", type=int, default=300)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--ckpt", default="best_vico_causal.pt")
    ap.add_argument("--out", default="l2l_listener_predictions.pkl")
    args = ap.parse_args()

    crank = 0
    device = torch.device("cuda:{}".format(crank))
    model = SLMFT(numeric_mode=L.MODE_PERF_BF16 if args.bf16 else L.MODE_PARITY_F32).to(device)
    if os.path.isfile(args.ckpt):
        model.load_state_dict(torch.load(args.ckpt, map_location="cpu"))
    else:
        print("no checkpoint at %s: synthetic weights" % args.ckpt)

    have_vico = os.path.isdir("../data/vico_processed_30fps")
    if not have_vico:
        print("no ViCo data under ../data: SYNTHETIC clips -- the metrics below are not ViCo results")
    dataset = get_vico_dataloaders(batch_size=args.batch,
                                   synthetic=None if have_vico
                                   else {"n_clips": args.clips, "max_len": args.max_len, "min_len": 24})
    val_loader = dataset["valid"]

    t0 = time.time()
    y_true, y_pred, x, data_ids = evaluate_test_epoch(model, val_loader, device, beam_size=args.beam)
    torch.cuda.synchronize()
    print("generated %d clips x best-of-%d in %.2f s" % (len(y_true), args.beam, time.time() - t0))
    print_metrics(y_true, y_pred, x)
    print_metrics_full(y_true, y_pred, x)

    d = {"y_true": y_true, "y_pred": y_pred, "data_ids": data_ids, "synthetic": not have_vico}
    with open(args.out, "wb") as f:
        pickle.dump(d, f, protocol=pickle.HIGHEST_PROTOCOL)


if __name__ == "__main__":
    main()
