"""Input side of the evaluation path: the ViCo clip dataset, ``pad_collate`` and ``get_vico_dataloaders`` with the
sample / batch format of reference ``code/dataset/data_loader.py:108-152,429-439,461-478``, plus a synthetic
dataset of the same format (the real pickles are not distributable).

Sample (ViCoDataset.__getitem__, reference :137-152):
    (combined_feats [L, 56+768] f32, video_feats_listener [L, 56] f32, path, speaker_id, listener_id, sentiment)
  where ``combined_feats[:, :56]`` is ``torch.ones_like(video_speaker)`` -- the ViCo protocol feeds a constant
  speaker-motion stream (reference :147) -- and ``combined_feats[:, 56:]`` the audio features.
Batch (pad_collate, reference :429-439):
    (xx_pad [B, Lmax, 824], yy_pad [B, Lmax, 56], x_lens list[int], (speaker_ids, listener_ids), names)
  zero padded; this is exactly what ``dimx.x_engine_pt.evaluate_*`` consumes.

Host -> device staging: the loaders use pinned host memory (``pin_memory=True``) so that the engine's
``.to(device, non_blocking=True)`` copies overlap the previous batch's kernels.
"""
import os
import pickle

import numpy as np
import torch
from torch.utils import data

from .. import prng


class ViCoDataset(data.Dataset):
    """reference :108-152.  ``data_path``: directory of ``<id>.pkl`` files holding a dict with the arrays
    ``video_speaker`` [L,56], ``audio`` [L,768], ``video_listener`` [L,56]; ``meta_data_path``: the RLD csv whose
    columns 0,1,4,5,6 are sentiment, clip id, listener id, speaker id, split."""

    SENTIMENT = {"neutral": 0, "positive": 1, "negative": 2}

    def __init__(self, data_path, meta_data_path, mode="train"):
        import pandas as pd
        meta = pd.read_csv(meta_data_path).values
        ids = [row[1] for row in meta if row[6] == mode]
        self.data = []
        for cid in ids:
            f = os.path.join(data_path, cid + ".pkl")
            if not os.path.exists(f):
                continue
            with open(f, "rb") as fh:
                cur = pickle.load(fh)
            n = len(cur["video_speaker"])
            if n == len(cur["audio"]) == len(cur["video_listener"]) and 5 <= n <= 1024:
                self.data.append(f)
        print(f"Loaded {len(self.data)} data points for {mode}")
        self.id2speaker_id = {row[1]: row[5] for row in meta}
        self.id2listener_id = {row[1]: row[4] for row in meta}
        self.id2sentiment = {row[1]: self.SENTIMENT[row[0]] for row in meta}

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        with open(self.data[index], "rb") as fh:
            d = pickle.load(fh)
        uid = os.path.basename(self.data[index]).split(".")[0]
        v_s = torch.ones_like(torch.FloatTensor(np.asarray(d["video_speaker"])))     # reference :147
        v_l = torch.FloatTensor(np.asarray(d["video_listener"]))
        aud = torch.FloatTensor(np.asarray(d["audio"]))
        return (torch.cat((v_s, aud), dim=1), v_l, self.data[index], self.id2speaker_id[uid],
                self.id2listener_id[uid], self.id2sentiment[uid])


class SyntheticDyadDataset(data.Dataset):
    """Clips of the ViCoDataset sample format drawn from the repository's counter-based PRNG:
    listener motion and audio ~ N(0,1), lengths ~ U[min_len, max_len]; ``vico_like=True`` keeps the reference's
    constant speaker stream (ones), False draws speaker motion ~ N(0,1)."""

    def __init__(self, n_clips=64, min_len=5, max_len=300, seed=20260928, vico_like=True, fixed_len=None):
        self.n, self.seed, self.vico_like = n_clips, seed, vico_like
        if fixed_len:
            self.lens = np.full(n_clips, int(fixed_len), dtype=np.int64)
        else:
            self.lens = prng.integers(seed, "synthetic.lens", (n_clips,), min_len, max_len + 1)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        L = int(self.lens[i])
        v_l = torch.from_numpy(prng.normal(self.seed, "synthetic.vl.%d" % i, (L, 56)))
        aud = torch.from_numpy(prng.normal(self.seed, "synthetic.a.%d" % i, (L, 768)))
        v_s = torch.ones(L, 56) if self.vico_like else torch.from_numpy(prng.normal(self.seed, "synthetic.vs.%d" % i, (L, 56)))
        return torch.cat((v_s, aud), dim=1), v_l, "synthetic_%05d" % i, i % 100, (i * 7) % 100, i % 3


def pad_collate(batch):
    """reference :429-439."""
    xx, yy, zz, speaker_ids, listener_ids, sentiment = zip(*batch)
    x_lens = [len(x) for x in xx]
    xx_pad = torch.nn.utils.rnn.pad_sequence(xx, batch_first=True, padding_value=0)
    yy_pad = torch.nn.utils.rnn.pad_sequence(yy, batch_first=True, padding_value=0)
    return xx_pad, yy_pad, x_lens, (torch.LongTensor(speaker_ids), torch.LongTensor(listener_ids)), list(zz)


def _loader(ds, batch_size, shuffle, rank=0, world=1, seed=0):
    """world > 1: every rank iterates its own 1/world of the (shuffled) set through a DistributedSampler -- same number
    of batches on every rank (the sampler pads by wrapping around), so a collective per batch never waits for a rank that
    has run out (without it every rank read the whole set and the effective batch was world x too large)."""
    sampler = None
    if world > 1:
        sampler = data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed,
                                                      drop_last=False)
        shuffle = False
    return data.DataLoader(dataset=ds, batch_size=batch_size, shuffle=shuffle, sampler=sampler, num_workers=0,
                           collate_fn=pad_collate, pin_memory=torch.cuda.is_available())


def get_vico_dataloaders(batch_size, data_path="../data/vico_processed_30fps", meta_data_path="../data/RLD_data.csv",
                         synthetic=None, rank=None, world=None, seed=0, shard_eval=False):
    """reference :461-478 -> {'train', 'valid', 'all'} loaders.  ``synthetic`` (a dict of SyntheticDyadDataset
    kwargs) asks for synthetic clips of the same format; without it the ViCo files must exist -- like the reference,
    which fails on a missing data directory -- so metrics on random clips can never pass for ViCo results.
    ``rank`` / ``world`` (default: the initialised process group) shard the TRAINING loaders ('train', 'all') across the ranks
    of a multi-GPU job (``train_epoch`` calls ``loader.sampler.set_epoch(epoch)`` for a fresh shuffle per epoch).  The
    evaluation loader 'valid' is NOT sharded by default: the evaluation loops (``evaluate_test_epoch`` /
    ``evaluate_finetune_epoch``) expect every rank to see the same full batch -- the first splits its rows over the ranks itself
    and all-gathers the winners, the second scores the whole set on every rank -- so a rank-sharded 'valid' loader would pair
    gathered predictions with another rank's targets.  ``shard_eval=True`` shards 'valid' too, for callers
    that gather targets / ids themselves."""
    if world is None:
        from .. import dist as ddist
        rank, world = ddist.rank(), ddist.world_size()
    if synthetic is None:
        if not (os.path.isdir(data_path) and os.path.isfile(meta_data_path)):
            raise FileNotFoundError("ViCo data not found (%s, %s); pass synthetic={...} to get synthetic clips"
                                    % (data_path, meta_data_path))
        train, val = ViCoDataset(data_path, meta_data_path, "train"), ViCoDataset(data_path, meta_data_path, "test")
    else:
        kw = dict(synthetic or {})
        train = SyntheticDyadDataset(**kw)
        val = SyntheticDyadDataset(**{**kw, "seed": kw.get("seed", 20260928) + 1})
    ev_rank, ev_world = (rank, world) if shard_eval else (0, 1)
    return {"train": _loader(train, batch_size, True, rank, world, seed), "valid": _loader(val, batch_size, False, ev_rank, ev_world, seed),
            "all": _loader(data.ConcatDataset([train, val]), batch_size, True, rank, world, seed)}
