"""GPU: the N-rank path of bench.py with the REAL model on a 1-GPU box -- `python bench.py --gpus 2 --rehearse-shared-gpu` launches
its own two ranks (one process each, both on cuda:0, collectives on gloo), shards the clips, all-gathers the generated code indices
and prints ONE line whose shard check must say that the gathered tokens equal a single rank's on the same seeded global batch
(clips shard, one all-gather per batch; reference parallelism: nn.DataParallel, code/finetune_s2s_pretrain.py:47,105).
Not a measurement (the line says so); it is the part of the multi-GPU path a 1-GPU box can execute end to end."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_reproduce_the_single_rank_tokens():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--batch", "8", "--frames", "60",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-parity-mode", "--no-train-step"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["global_batch"] == 16
    assert "REHEARSAL" in d["metric"] and d["data"].startswith("synthetic")
    sc = d["shard_check"]
    assert sc["identical"] and sc["tokens_sha256_sharded"] == sc["tokens_sha256_one_rank"] and sc["max_abs_pred_diff"] == 0.0
