"""The reference's command line, end to end on the GPU (reference trainer_complete.py:132-133 `LightningCLI(Trainer)`,
README.md:111-123: `trainer_complete.py fit --config ...`, `fit --ckpt_path`, `test --ckpt_path`, then `evaluate.py`).

A small model (d_model 64, 2+2 layers) trains on nine synthetic info files (reference schema) through the real loop of
plankassembly_amd.trainer.run: LineDataset -> DataLoader -> DevicePrefetcher -> HIP train step -> FusedAdam, validation
every epoch (HIP greedy decode -> box filter -> Hungarian matcher -> running means), `last.ckpt` + best-F1 checkpoint,
resume, test with pred_jsons, offline re-scoring.
"""
import glob
import json
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_infos(root, n=9, seed=11):
    """`n` info files in the reference's schema (dataset/prepare_info.py:59-70): 3-25 lines in the three
    views, 2-8 planks (plank 0 = the overall bounding box) with pointer attachments that the reference's pointer mask
    allows.  (The golden infos hold only three drawings with lines; the f1 cases have none, and a drawing without lines is
    an error in the reference's LineDataset too.)"""
    from plankassembly_amd.data import pointer_mask_row
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    names = []
    for i in range(n):
        grid = np.round(rng.uniform(-1, 1, size=10), 3)
        nl, npk = int(rng.integers(3, 26)), int(rng.integers(2, 9))
        a, b = rng.choice(grid, size=(nl, 2)), rng.choice(grid, size=(nl, 2))
        lines = np.concatenate([np.minimum(a, b), np.maximum(a, b)], axis=1)
        lo, hi = rng.choice(grid, size=(npk, 3)), rng.choice(grid, size=(npk, 3))
        coords = np.concatenate([np.minimum(lo, hi), np.maximum(lo, hi) + 0.05], axis=1).round(3)
        coords[0] = np.concatenate([coords[:, :3].min(0), coords[:, 3:].max(0)])
        flat, attach = coords.reshape(-1), np.full(npk * 6, -1)
        for t in range(6, npk * 6):
            cand = np.nonzero(pointer_mask_row(t, t))[0]
            cand = cand[np.isclose(flat[cand], flat[t])]
            if len(cand):
                attach[t] = int(cand[0])
        name = f"drawing{i:02d}"
        with open(os.path.join(root, name + ".json"), "w") as f:
            json.dump({"name": name, "lines": lines.tolist(), "views": rng.integers(0, 3, nl).tolist(),
                       "types": rng.integers(0, 2, nl).tolist(), "svgs": [], "coords": coords.tolist(),
                       "attach": attach.reshape(npk, 6).tolist()}, f)
        names.append(name + ".json")
    return names


def _write_config(tmp_path, max_epochs, n=9):
    names = _write_infos(str(tmp_path / "data" / "infos"), n=n)
    split = tmp_path / "all.txt"
    split.write_text("\n".join(names))
    with open(os.path.join(REPO, "configs", "train_complete.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["trainer"].update(max_epochs=max_epochs, check_val_every_n_epoch=1, devices=1)
    hp = cfg["model"]["hparams"]
    hp.update(ROOT=str(tmp_path / "data" / "infos"), DATASETS_TRAIN=str(split), DATASETS_VALID=str(split), DATASETS_TEST=str(split),
              BATCH_SIZE=4, NUM_WORKERS=0, LR=2e-3)
    hp["DATA"].update(MAX_INPUT_LENGTH=129, MAX_OUTPUT_LENGTH=60, AUG_RATIO=0.0)
    hp["MODEL"].update(NUM_MODEL=64, NUM_HEAD=4, NUM_FEEDFORWARD=128, NUM_ENCODER_LAYERS=2, NUM_DECODER_LAYERS=2, DROPOUT=0.0,
                       COMPUTE_DTYPE="f32")
    path = tmp_path / "train_small.yaml"
    path.write_text(yaml.safe_dump(cfg))
    return str(path), len(names)


def test_cli_fit_resume_test_evaluate(tmp_path, monkeypatch):
    from plankassembly_amd import lightning_state as LS
    from plankassembly_amd.trainer import Trainer, cli
    import evaluate as EV
    monkeypatch.chdir(tmp_path)                                   # lightning_logs/version_N lands here
    config, n_files = _write_config(tmp_path, max_epochs=3)
    steps_per_epoch = n_files // 4                                # drop_last, as the reference's train loader
    assert steps_per_epoch == 2

    # ---- fit
    mod = cli(Trainer, ["fit", "--config", config])
    losses = [v for _, name, v in mod.logger.history if name == "train/loss"]
    assert len(losses) == 3 and losses[-1] < losses[0], losses
    assert mod.global_step == 3 * steps_per_epoch
    ckdir = os.path.join(mod.logger.log_dir, "checkpoints")
    last = os.path.join(ckdir, "last.ckpt")
    best = glob.glob(os.path.join(ckdir, "checkpoint_*-precision=*-recall=*-f1=*.ckpt"))
    assert os.path.exists(last) and len(best) == 1, os.listdir(ckdir)          # save_last + save_top_k: 1
    ck = torch.load(last, map_location="cpu", weights_only=True)
    assert ck["epoch"] == 2 and ck["global_step"] == 6 and ck["pytorch-lightning_version"].startswith("1.7")
    assert all(k.startswith("model.") for k in ck["state_dict"]) and ck["hyper_parameters"]["hparams"]["BATCH_SIZE"] == 4
    assert list(ck["callbacks"]) == [LS.checkpoint_callback_key()]
    assert ck["callbacks"][LS.checkpoint_callback_key()]["best_model_path"] == best[0]
    st = ck["optimizer_states"][0]
    assert float(st["state"][0]["step"]) == 6.0 and len(st["state"]) == len(st["param_groups"][0]["params"])
    vals = [v for _, name, v in mod.logger.history if name == "val/fmeasure"]
    assert len(vals) == 3 and all(0.0 <= v <= 1.0 for v in vals)
    p_end = {k: v.detach().cpu().clone() for k, v in mod.model.state_dict().items()}
    m_end = mod.optimizer._m.detach().cpu().clone()

    # ---- fit --ckpt_path: continues at the stored epoch / global step with the stored Adam moments
    mod2 = cli(Trainer, ["fit", "--config", config, "--ckpt_path", last, "--trainer.max_epochs", "5"])
    assert mod2.resume_epoch == 3 and mod2.global_step == 5 * steps_per_epoch and mod2.optimizer._step == 10
    l2 = [v for _, name, v in mod2.logger.history if name == "train/loss"]
    assert len(l2) == 2 and l2[-1] < losses[0]                                # epochs 3 and 4 only
    # the resumed run started from exactly the saved state: re-load the file into a fresh trainer and compare with the
    # end of the first run
    probe = Trainer(ck["hyper_parameters"]["hparams"])
    probe.model.cuda()
    popt = probe.configure_optimizers()["optimizer"]
    probe.load_checkpoint(last, optimizer=popt)
    assert all(torch.equal(v.cpu(), p_end[k]) for k, v in probe.model.state_dict().items())
    assert torch.equal(popt._m.cpu(), m_end) and popt._step == 6
    last2 = os.path.join(mod2.logger.log_dir, "checkpoints", "last.ckpt")
    assert torch.load(last2, map_location="cpu", weights_only=True)["epoch"] == 4
    # the resumed run inherits best_model_path from the checkpoint, but save_top_k only prunes its OWN directory -
    # the first run's best file and the file passed as --ckpt_path are still there
    assert os.path.exists(best[0]) and os.path.exists(last), os.listdir(ckdir)
    # a run that already reached max_steps takes no further optimizer step when resumed
    mod2b = cli(Trainer, ["fit", "--config", config, "--ckpt_path", last, "--trainer.max_epochs", "5", "--trainer.max_steps", "6"])
    assert mod2b.global_step == 6 and mod2b.optimizer._step == 6

    # ---- test --ckpt_path: pred_jsons in the reference's format, test/* logged
    mod3 = cli(Trainer, ["test", "--config", config, "--ckpt_path", last2])
    files = sorted(os.listdir(os.path.join(mod3.logger.log_dir, "pred_jsons")))
    assert len(files) == n_files
    per_file = []
    for fn in files:
        with open(os.path.join(mod3.logger.log_dir, "pred_jsons", fn)) as f:
            d = json.load(f)
        assert sorted(d) == sorted(["prediction", "attach", "groundtruth", "precision", "recall", "fmeasure"])
        assert all(len(row) == 6 for row in d["prediction"] + d["groundtruth"] + d["attach"])
        per_file.append(d["fmeasure"])
    assert abs(np.mean(per_file) - mod3._logged["test/fmeasure"]) < 1e-6

    # ---- evaluate.py on those files: the same scores up to dequantisation (quantised boxes vs the continuous ground truth)
    p, r, f, metrics = EV.evaluate(str(tmp_path / "data"), mod3.logger.log_dir, 0.5, 9, verbose=False)
    assert os.path.exists(os.path.join(mod3.logger.log_dir, "metrics.json")) and len(metrics) <= n_files
    assert abs(f - mod3._logged["test/fmeasure"]) <= 0.2, (f, mod3._logged)
    print(f"    CLI: train/loss {losses} -> {l2}; val/fmeasure {vals}; test/fmeasure {mod3._logged['test/fmeasure']:.3f}; "
          f"evaluate.py f1 {f:.3f}")


def test_cli_fit_with_bf16x3_products(tmp_path, monkeypatch):
    """MODEL.COMPUTE_DTYPE: x3 through the command line: the f32 path with split (hi / lo bf16) matrix products trains like the f32
    path - same first-epoch loss to 1e-3, decreasing afterwards - and its GEMMs really ran split."""
    import ctypes as C
    from plankassembly_amd import _lib as L
    from plankassembly_amd.trainer import Trainer, cli
    monkeypatch.chdir(tmp_path)
    config, _ = _write_config(tmp_path, max_epochs=2)
    ref = cli(Trainer, ["fit", "--config", config])
    l_ref = [v for _, name, v in ref.logger.history if name == "train/loss"]
    with open(config) as f:
        cfg = yaml.safe_load(f)
    cfg["model"]["hparams"]["MODEL"]["COMPUTE_DTYPE"] = "x3"
    path = tmp_path / "train_small_x3.yaml"
    path.write_text(yaml.safe_dump(cfg))
    out = (C.c_int64 * 2)()
    L.check(L.lib().pa_gemm_split_stats(out, 1), "pa_gemm_split_stats")
    mod = cli(Trainer, ["fit", "--config", str(path)])
    L.check(L.lib().pa_gemm_split_stats(out, 0), "pa_gemm_split_stats")
    assert mod.model.compute_mode == "x3" and int(out[0]) > 100, (mod.model.compute_mode, int(out[0]), int(out[1]))
    losses = [v for _, name, v in mod.logger.history if name == "train/loss"]
    assert len(losses) == 2 and losses[1] < losses[0]
    assert abs(losses[0] - l_ref[0]) < 1e-3 * max(1.0, abs(l_ref[0])), (losses, l_ref)


def test_cli_two_ranks_sharing_the_gpu(tmp_path, monkeypatch):
    """The reference trains with `strategy: ddp` on 4 devices (configs/train_complete.yaml:18-21).  Two ranks of this trainer's
    command line - launched as torchrun would, sharing the one GPU of the test box through gloo - run `test` and `fit`:
    * `test`: DistributedSampler shards the 16 drawings, each rank decodes and scores its 8, the metric sums are exchanged:
      precision / recall / F1 equal the single-process run on the same checkpoint, all 16 pred_jsons land in ONE version dir;
    * `fit`: both ranks end every epoch with bit-identical parameters (broadcast at start, exchanged gradients, same Adam step),
      rank 0 alone writes the checkpoints, the validation metrics agree across ranks."""
    import socket
    import subprocess
    import sys
    from plankassembly_amd.trainer import Trainer, cli
    monkeypatch.chdir(tmp_path)
    config, n_files = _write_config(tmp_path, max_epochs=2, n=16)
    mod = cli(Trainer, ["fit", "--config", config])
    last = os.path.join(mod.logger.log_dir, "checkpoints", "last.ckpt")
    single = cli(Trainer, ["test", "--config", config, "--ckpt_path", last])

    def launch(tag, argv):
        out = tmp_path / tag
        out.mkdir()
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, PLANK_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(REPO, "tests", "cli_ddp_worker.py"), str(out)] + argv
        r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-4000:]
        recs = []
        for rank in range(2):
            with open(out / f"rank{rank}.json") as f:
                recs.append(json.load(f))
        return recs

    # ---- test on two ranks == test on one
    r0, r1 = launch("ddp_test", ["test", "--config", config, "--ckpt_path", last])
    assert r0["log_dir"] == r1["log_dir"]
    files = sorted(os.listdir(os.path.join(str(tmp_path), r0["log_dir"], "pred_jsons")))
    assert len(files) == n_files == 16
    for key in ("test/precision", "test/recall", "test/fmeasure"):
        assert abs(r0["logged"][key] - single._logged[key]) < 1e-6, (key, r0["logged"], single._logged)
        assert r0["logged"][key] == r1["logged"][key]

    # ---- fit on two ranks
    f0, f1 = launch("ddp_fit", ["fit", "--config", config])
    assert f0["param_sha"] == f1["param_sha"]                                 # DDP: identical replicas after every step
    assert f0["global_step"] == f1["global_step"] == 2 * (n_files // 2 // 4)   # 8 drawings per rank, batch 4, drop_last
    l0 = [v for _, name, v in f0["history"] if name == "train/loss"]
    assert len(l0) == 2 and all(np.isfinite(l0)) and l0[-1] < l0[0]
    v0 = [v for _, name, v in f0["history"] if name == "val/fmeasure"]
    v1 = [v for _, name, v in f1["history"] if name == "val/fmeasure"]
    assert v0 == v1 and len(v0) == 2
    ckdir = os.path.join(str(tmp_path), f0["log_dir"], "checkpoints")
    assert f0["log_dir"] == f1["log_dir"] and os.path.exists(os.path.join(ckdir, "last.ckpt"))
    ck = torch.load(os.path.join(ckdir, "last.ckpt"), map_location="cpu", weights_only=True)
    assert ck["global_step"] == f0["global_step"] and ck["epoch"] == 1
