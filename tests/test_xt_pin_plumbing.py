"""The offline pin path of the x-transformers half of the oracle (tools/verify_against_xtransformers.py --write-golden
-> tests/golden/xt_{slmft,legacy}.npz -> test_oracle_xt_golden.py / test_gpu_xt_golden.py) executed once end to end
against tests/fake_xtransformers.py, a stand-in module made from the oracle itself.  This proves NOTHING about the
library (parity of that half stays "unpinned"); it proves that the writer runs, that its hooks fire
(the key-mask draw is captured, torch.multinomial is replaced by argmax(p / q)), and that the consumers read the keys the
writer writes -- so the first person with the real wheel gets fixtures, not a traceback."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SLMFT_KEYS = {"B", "T", "lens", "x_s", "kv_mask", "tf_logits", "tf_loss", "gen_greedy", "gen_sampled", "xt_version"}
LEGACY_KEYS = {"B", "T", "lens", "enc_out", "tf_logits", "tf_loss", "gen_sampled"}


def test_writer_and_consumers_run_end_to_end_on_a_stand_in(tmp_path, full_sd, monkeypatch, capsys):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_xtransformers as fake
    spec = importlib.util.spec_from_file_location("verify_xt", os.path.join(ROOT, "tools", "verify_against_xtransformers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    real_gold = mod.GOLD
    before = sorted(os.listdir(real_gold))
    monkeypatch.setattr(mod, "GOLD", str(tmp_path))
    torch.set_grad_enabled(False)
    mod.slmft_case(fake, True)
    mod.legacy_case(fake, True)
    out = capsys.readouterr().out
    # the stand-in IS the oracle: every stage agrees exactly, which shows the hooks fed both sides the same randomness
    assert "greedy equal: True, injected-noise equal: True" in out
    assert "legacy generate            injected-noise equal: True" in out
    g1, g2 = np.load(tmp_path / "xt_slmft.npz"), np.load(tmp_path / "xt_legacy.npz")
    assert set(g1.files) == SLMFT_KEYS and set(g2.files) == LEGACY_KEYS
    assert g1["kv_mask"].shape == (3, 39) and (~g1["kv_mask"]).sum(1).tolist() == [6, 6, 6]   # int(40 * 0.15) hidden keys
    assert g1["kv_mask"][:, 0].all()                                                           # never position 0
    # consumers: the CPU tests that skip while the real fixtures are absent, pointed at the temporary directory
    import test_oracle_xt_golden as consumer
    consumer.test_oracle_slmft_stages_match_the_library(str(tmp_path), full_sd)
    consumer.test_oracle_legacy_stages_match_the_library(str(tmp_path))
    # nothing from the stand-in may ever land next to the real fixtures
    assert sorted(os.listdir(real_gold)) == before
    assert not any(f.startswith("xt_") for f in before), "xt_*.npz in tests/golden must come from the real wheel"
    sys.path.remove(os.path.join(ROOT, "tests"))
