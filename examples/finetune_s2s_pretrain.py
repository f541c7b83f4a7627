"""The reference's fine-tuning driver (code/finetune_s2s_pretrain.py:105-143) on the dimx drop-ins: AdamW lr 1e-5,
clip 1.0, frozen VQ-VAEs, evaluate_finetune_epoch + print_metrics after every epoch, best checkpoint by FD sum.
Single process or `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/finetune_s2s_pretrain.py`
(one process per GPU, one flat gradient all-reduce per step over RCCL; every rank reads its own shard of the clips).
The training step -- forward, backward, clip, AdamW -- runs on the hand-written HIP kernels (dimx.train_hip.HipTrainer);
`--backward autograd` selects the PyTorch-autograd restatement that serves as its checker.

    python examples/finetune_s2s_pretrain.py [--epochs 2] [--clips 64] [--batch 4] [--max-len 120] [--backward hip|autograd]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dimx  # noqa: E402,F401
from dimx import dist as ddist  # noqa: E402
from dimx.dataset.data_loader import get_vico_dataloaders  # noqa: E402
from dimx.mymetrics import print_metrics  # noqa: E402
from dimx.seq2seq_pretrain import SLMFT  # noqa: E402
from dimx.x_engine_pt import evaluate_finetune_epoch, train_epoch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--max-len", type=int, default=120)
    ap.add_argument("--out", default="best_vico_causal.pt")
    ap.add_argument("--backward", default="hip", choices=["hip", "autograd"],
                    help="hip: forward + backward + clip + AdamW on csrc/train*.hip; autograd: the PyTorch restatement (checker)")
    args = ap.parse_args()
    rank, world, local = ddist.init_from_env()
    device = torch.device("cuda:{}".format(local))
    torch.cuda.set_device(device)
    model = SLMFT().to(device)
    # the reference's own lines (code/finetune_s2s_pretrain.py:118-119): train_epoch maps this AdamW onto the HIP training step
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-5)
    have_vico = os.path.isdir("../data/vico_processed_30fps")
    if not have_vico and rank == 0:
        print("no ViCo data under ../data: SYNTHETIC clips -- the numbers below are not ViCo results")
    dataset = get_vico_dataloaders(batch_size=args.batch,
                                   synthetic=None if have_vico else {"n_clips": args.clips, "max_len": args.max_len,
                                                                     "min_len": 24, "seed": 20260928})   # same clips on every rank: the sampler shards them
    best = float("inf")
    for epoch in range(args.epochs):
        loss = train_epoch(model, dataset["train"], optimizer, device, scheduler=None, clip=1.0, print_freq=100,
                           epoch=epoch, log=print if rank == 0 else (lambda *_: None), backward=args.backward)
        y_true, y_pred, x, _ = evaluate_finetune_epoch(model, dataset["valid"], device)
        if rank == 0:
            a, b = print_metrics(y_true, y_pred, x)
            print("epoch %d: mean loss %.4f, FD pose %.4f + exp %.4f" % (epoch, loss, a, b))
            if a + b < best:
                best = a + b
                torch.save(model.state_dict(), args.out)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
