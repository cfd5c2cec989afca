#!/usr/bin/env python3
"""Build-container check (TEST INFRASTRUCTURE, needs /root/reference): a checkpoint written by THIS package's
save_checkpoint() is read by the REFERENCE's load_checkpoint(), and the reference model then holds the same values.

Two processes, because both packages are called `glow_tts_train`:
    python oracle/check_reference_reads_ours.py            # writes /tmp/ours.pth with this package, then re-runs itself
    python oracle/check_reference_reads_ours.py --ref FILE # (internal) loads FILE with the reference
"""
import json
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TINY = dict(num_symbols=20, hidden_channels=16, filter_channels=32, filter_channels_dp=16, n_blocks_dec=2, n_layers_enc=1,
            n_block_layers=2, hidden_channels_enc=16, hidden_channels_dec=16, p_dropout=0.0, p_dropout_dec=0.0,
            window_size=4, prenet=False)


def ours(path):
    sys.path.insert(0, os.path.join(ROOT, "glow-tts-train_amd"))
    from glow_tts_train import checkpoint as C
    from glow_tts_train.config import AudioConfig, ModelConfig, TrainingConfig

    cfg = TrainingConfig(model=ModelConfig(**TINY), audio=AudioConfig(mel_channels=8), warmup_steps=10)
    ck = C.load_checkpoint(os.path.join(ROOT, "tests", "golden", "host_ref_checkpoint.pth"), cfg, use_cuda=False)
    C.save_checkpoint(C.Checkpoint(model=ck.model, optimizer=ck.optimizer, learning_rate=ck.optimizer.cur_lr,
                                   global_step=11, version=1), path)
    with open(path + ".config.json", "w") as f:
        cfg.save(f)


def reference(path):
    sys.path.insert(0, HERE)
    import make_golden

    make_golden.import_reference()
    from glow_tts_train import checkpoint as ref_ckpt
    from glow_tts_train import config as ref_cfg

    d = json.load(open(path + ".config.json"))        # our config JSON, fed to the reference's dataclasses
    cfg = ref_cfg.TrainingConfig(**{k: v for k, v in d.items() if k not in ("audio", "model", "betas")},
                                 betas=tuple(d["betas"]), audio=ref_cfg.AudioConfig(**d["audio"]),
                                 model=ref_cfg.ModelConfig(**d["model"]))
    ck = ref_ckpt.load_checkpoint(path, cfg, use_cuda=False)          # torch.load's default weights_only=True: no numpy
    want = torch.load(os.path.join(ROOT, "tests", "golden", "host_ref_checkpoint.pth"), weights_only=False)
    assert ck.global_step == 11
    for k, v in ck.model.state_dict().items():
        assert torch.equal(v, want["model"][k]), k
    st = ck.optimizer._optim.state_dict()["state"]
    for i, s in want["optimizer"]["state"].items():
        assert torch.equal(st[i]["exp_avg"], s["exp_avg"]) and torch.equal(st[i]["exp_avg_sq"], s["exp_avg_sq"])
        assert float(st[i]["step"]) == float(s["step"])
    assert np.isclose(ck.optimizer._optim.param_groups[0]["lr"], want["optimizer"]["param_groups"][0]["lr"])
    print("reference read our checkpoint and config: OK")


if __name__ == "__main__":
    if sys.argv[1:2] == ["--ref"]:
        reference(sys.argv[2])
    else:
        out = "/tmp/ours.pth"
        ours(out)
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--ref", out]))
