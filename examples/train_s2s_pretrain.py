"""The reference's pre-training driver (code/train_s2s_pretrain.py:41-64) on the dimx drop-ins: SLM, AdamW lr 1e-5, clip
1.0, train_epoch + evaluate_epoch per epoch, best checkpoint by validation loss.  The frozen VQ encoders and every
evaluation forward run on the HIP engine; the training step -- forward, backward, clip and AdamW -- runs on the hand-written HIP
kernels too: train_epoch hands the torch AdamW to dimx.train_hip.SlmHipTrainer (csrc/train.hip: slm_run), `--backward autograd`
selects the PyTorch-autograd restatement (dimx.train.slm_loss), its checker.  Single process, or one process per GPU with
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_s2s_pretrain.py` (gradients
averaged over RCCL: ONE all-reduce of the flat gradient arena per step; every rank reads its own shard of the clips).

    python examples/train_s2s_pretrain.py [--epochs 2] [--clips 64] [--batch 4] [--max-len 120] [--backward auto|hip|autograd]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dimx  # noqa: E402,F401
from dimx import dist as ddist  # noqa: E402
from dimx.dataset.data_loader import get_vico_dataloaders  # noqa: E402
from dimx.seq2seq_pretrain import SLM  # noqa: E402
from dimx.x_engine_pt import evaluate_epoch, train_epoch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--max-len", type=int, default=120)
    ap.add_argument("--out", default="best_model_pretrain_15.pt")
    ap.add_argument("--backward", default="auto", choices=["auto", "hip", "autograd"])
    args = ap.parse_args()
    rank, world, local = ddist.init_from_env()
    device = torch.device("cuda:{}".format(local))
    torch.cuda.set_device(device)
    model = SLM().to(device)
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-5)       # code/train_s2s_pretrain.py:45
    have_vico = os.path.isdir("../data/vico_processed_30fps")
    if not have_vico and rank == 0:
        print("no dyad data under ../data: SYNTHETIC clips -- the losses below say nothing about the real task")
    dataset = get_vico_dataloaders(batch_size=args.batch,
                                   synthetic=None if have_vico else {"n_clips": args.clips, "max_len": args.max_len,
                                                                     "min_len": 24, "seed": 20260928})   # same clips on every rank: the sampler shards them
    log = print if rank == 0 else (lambda *_: None)
    best = float("inf")
    for epoch in range(args.epochs):
        model.train()
        train_epoch(model, dataset["train"], optimizer, device, scheduler=None, clip=1.0, print_freq=2000, epoch=epoch, log=log, backward=args.backward)
        val_loss = evaluate_epoch(model, dataset["valid"], device, log=log)
        log("Epoch %d val loss: %.4f" % (epoch, val_loss))
        if rank == 0 and val_loss < best:
            best = val_loss
            torch.save(model.state_dict(), args.out)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
