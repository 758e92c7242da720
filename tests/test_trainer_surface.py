"""CPU checks of the trainer surface: reference-style YAML parsing, CfgNode, hook names, checkpoint shape."""
import os

import torch

from plankassembly_amd.config import CfgNode, load_cli_config
from plankassembly_amd.trainer import SidefaceTrainer, SyntheticDrawings, Trainer, VisibleTrainer

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_hparams(**over):
    _, _, hp = load_cli_config(os.path.join(REPO, "configs", "train_complete.yaml"))
    hp["MODEL"].update(NUM_MODEL=64, NUM_HEAD=4, NUM_FEEDFORWARD=128, NUM_ENCODER_LAYERS=1, NUM_DECODER_LAYERS=1,
                       COMPUTE_DTYPE="f32")
    hp["DATA"].update(MAX_INPUT_LENGTH=65, MAX_OUTPUT_LENGTH=36)
    hp["BATCH_SIZE"] = 4
    hp.update(over)
    return hp


def test_configs_parse_like_the_reference_cli():
    for name, L in (("train_complete", 1200), ("train_visible", 1000), ("train_sideface", 300)):
        seed, tkw, hp = load_cli_config(os.path.join(REPO, "configs", name + ".yaml"))
        assert seed == 2022 and tkw["strategy"] == "ddp"
        cfg = CfgNode(hp)
        assert cfg.DATA.MAX_INPUT_LENGTH == L and cfg.TOKEN.PAD == 513 and cfg.MODEL.NORMALIZE_BEFORE is True
        assert isinstance(cfg.LR, float) and cfg.LR == 1e-4


def test_trainer_has_the_lightning_hook_surface():
    t = Trainer(small_hparams())
    for hook in ("train_dataloader", "val_dataloader", "test_dataloader", "training_step", "validation_step",
                 "validation_epoch_end", "test_step", "test_epoch_end", "configure_optimizers"):
        assert callable(getattr(t, hook))
    assert t.cfg.MODEL.NUM_MODEL == 64
    ck = t.checkpoint(0)
    assert all(k.startswith("model.") for k in ck["state_dict"])
    assert "hparams" in ck["hyper_parameters"]
    # round trip through a Lightning-shaped checkpoint
    path = os.path.join(REPO, ".pytest_cache_ckpt.pt")
    try:
        torch.save(ck, path)
        t2 = Trainer(small_hparams())
        t2.load_checkpoint(path)
        for a, b in zip(t.model.state_dict().values(), t2.model.state_dict().values()):
            assert torch.equal(a, b)
    finally:
        if os.path.exists(path):
            os.remove(path)


def test_synthetic_loaders_follow_the_batch_contract():
    t = SidefaceTrainer(small_hparams())
    batch = next(iter(t.train_dataloader()))
    assert "input_type" not in batch and batch["input_value"].shape == (4, 64) and batch["output_value"].shape == (4, 36)
    assert batch["input_mask"].dtype == torch.bool and len(batch["name"]) == 4
    v = VisibleTrainer(small_hparams())
    b2 = next(iter(v.val_dataloader()))
    assert "input_type" in b2
    assert torch.equal(b2["input_mask"], b2["input_value"] == 513)
    lab, val = b2["output_label"], b2["output_value"]
    ptr = lab >= 514
    assert torch.equal(lab[~ptr], val[~ptr])
