"""Trainer surface of the reference (trainer_complete.py / trainer_visible.py / trainer_sideface.py)
for the MI355X hot path, without Lightning / detectron2 / torchmetrics / jsonargparse (none of
them is installed in this image): same class, same hook names (Lightning 1.7: ``training_step``,
``validation_step``, ``validation_epoch_end``, ``test_step``, ``test_epoch_end``,
``configure_optimizers``, ``*_dataloader``), same YAML config files, same CLI shape
(``fit`` / ``test``, ``--config``, ``--ckpt_path``, ``--trainer.<key> value``).

What differs is underneath: ``self.model`` is the HIP-backed PlankModel, the optimizer is the fused
Adam over the flat parameter buffer, and the ``ddp`` strategy is one process per GPU with the
segmented RCCL gradient exchange of ``plankassembly_amd.distributed`` (launch with
``python -m torch.distributed.run --nproc-per-node N trainer_complete.py fit --config ...``).
Checkpoints are Lightning-shaped (``state_dict`` with the ``model.`` prefix + ``hyper_parameters``),
so the reference's published checkpoints load and ours load in the reference.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from .config import CfgNode, load_cli_config
from .data import SynthSpec, synth_sample
from .datasets import LineDataset, SidefaceDataset, parse_splits_list  # noqa: F401  (parse_splits_list re-exported)
from .metric import PlankScorer
from .models import build_model


class SyntheticDrawings(torch.utils.data.Dataset):
    """Seeded synthetic samples with the reference dataloader's tensor layout (the real
    PlankAssembly dataset needs network access)."""

    def __init__(self, n, spec: SynthSpec, seed=2022):
        self.n, self.spec, self.seed = n, spec, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed + i)
        s = synth_sample(rng, self.spec)
        out = {"name": f"synth_{self.seed}_{i:06d}"}
        out.update({k: torch.from_numpy(np.asarray(v)) for k, v in s.items()})
        return out


class _Logger:
    def __init__(self, root="lightning_logs"):
        v = 0
        while os.path.exists(os.path.join(root, f"version_{v}")):
            v += 1
        self.log_dir = os.path.join(root, f"version_{v}")
        self.history = []

    def log(self, name, value, step):
        self.history.append((step, name, float(value)))


class Trainer(torch.nn.Module):
    """reference trainer_complete.py:19-129."""

    with_type = True
    default_lines = (8, 299)
    dataset_cls = LineDataset
    train_augmentation = True

    def __init__(self, hparams):
        super().__init__()
        self.hparams_dict = dict(hparams)
        cfg = CfgNode(hparams)
        self.cfg = cfg
        self.model = build_model(cfg)
        self.scorer = PlankScorer(cfg.THRESHOLD)
        self.matcher, self.criterion = self.scorer.matcher, self.scorer.criterion      # the reference's attribute names
        self.logger = None
        self.global_step = 0
        self._logged = {}

    # ------------------------------------------------------------------ logging (self.log of Lightning)
    def log(self, name, value, **kw):
        v = float(value)
        self._logged[name] = v
        if self.logger is not None:
            self.logger.log(name, v, self.global_step)

    # ------------------------------------------------------------------ data
    def _spec(self):
        d = self.cfg.DATA
        hi = min(self.default_lines[1], (d.MAX_INPUT_LENGTH - 2) // d.NUM_INPUT_DOF)
        return SynthSpec(d.MAX_INPUT_LENGTH, d.MAX_OUTPUT_LENGTH, (min(self.default_lines[0], hi), hi),
                         (2, (d.MAX_OUTPUT_LENGTH - 1) // d.NUM_OUTPUT_DOF), self.with_type)

    def _dataset(self, split_key, n_default, seed, augmentation):
        """The reference's dataset over ``ROOT`` + split file when both exist on disk (trainer_complete.py:35-61),
        else the seeded synthetic generator (there is no network to fetch the real dataset here)."""
        split = self.cfg.get(split_key)
        root = str(self.cfg.get("ROOT", ""))
        have_split = bool(split) and all(os.path.exists(s) for s in str(split).split())
        if have_split and os.path.isdir(root):
            return self.dataset_cls(root, parse_splits_list(split), self.cfg.TOKEN, self.cfg.DATA, augmentation)
        return SyntheticDrawings(int(self.cfg.get("SYNTHETIC_SAMPLES", n_default)), self._spec(), seed)

    def _loader(self, split_key, n_default, shuffle, drop_last, seed, augmentation=False):
        ds = self._dataset(split_key, n_default, seed, augmentation)
        sampler = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=shuffle, drop_last=drop_last)
            shuffle = False
        workers = int(self.cfg.get("NUM_WORKERS", 0)) if not isinstance(ds, SyntheticDrawings) else 0
        return torch.utils.data.DataLoader(ds, batch_size=self.cfg.BATCH_SIZE, shuffle=shuffle, drop_last=drop_last,
                                           num_workers=workers, sampler=sampler)

    def train_dataloader(self):
        return self._loader("DATASETS_TRAIN", 256, True, True, 2022, self.train_augmentation)

    def val_dataloader(self):
        return self._loader("DATASETS_VALID", 64, False, False, 9_000_000)

    def test_dataloader(self):
        return self._loader("DATASETS_TEST", 64, False, False, 9_500_000)

    # ------------------------------------------------------------------ steps
    def training_step(self, batch, batch_idx):
        outputs = self.model(batch)
        loss = torch.mean(outputs["loss"])
        self._train_stats = (loss.detach(), torch.mean(outputs["accuracy"]).detach())
        return loss

    def _valid_pred(self, pred):
        """reference trainer_complete.py:78-79 / 100-101: drop boxes with a zero extent (row 0 = bbox kept)."""
        if len(pred) <= 1:
            return pred
        ok = torch.all(torch.abs(pred[1:, 3:] - pred[1:, :3]) != 0, dim=1)
        return torch.concat((pred[:1], pred[1:][ok]))

    def validation_step(self, batch, batch_idx):
        outputs = self.model(batch)
        for pred, gt in zip(outputs["predicts"], outputs["groundtruths"]):
            self.scorer.add(self._valid_pred(pred), gt)

    def _log_means(self, stage):
        for key, value in zip(("precision", "recall", "fmeasure"), self.scorer.means()):
            self.log(f"{stage}/{key}", value)

    def validation_epoch_end(self, outputs=None):
        self._log_means("val")

    def test_step(self, batch, batch_idx):
        outputs = self.model(batch)
        out_dir = os.path.join(self.logger.log_dir, "pred_jsons")
        os.makedirs(out_dir, exist_ok=True)
        for name, pred, gt, atta in zip(batch["name"], outputs["predicts"], outputs["groundtruths"], outputs["attach"]):
            vp = self._valid_pred(pred)
            scores = self.scorer.add(vp, gt)
            atta = atta[: vp.numel()].cpu().numpy()
            atta = atta[: len(atta) // 6 * 6].reshape(-1, 6).tolist()
            self._write_pred_json(out_dir, name, {"prediction": vp.cpu().numpy().reshape(-1, 6).tolist(), "attach": atta,
                                                  "groundtruth": gt.cpu().numpy().reshape(-1, 6).tolist(), **scores})

    @staticmethod
    def _write_pred_json(out_dir, name, record):
        """One `pred_jsons/<name>.json` in the reference's on-disk format (trainer_complete.py:104-118)."""
        with open(os.path.join(out_dir, f"{name}.json"), "w") as f:
            json.dump(record, f, indent=4, separators=(", ", ": "))

    def test_epoch_end(self, outputs=None):
        self._log_means("test")

    def configure_optimizers(self):
        from .optim import FusedAdam
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        return {"optimizer": FusedAdam(self.model, lr=self.cfg.LR, grad_scale=1.0 / world)}

    # ------------------------------------------------------------------ checkpoints (Lightning-shaped)
    def checkpoint(self, epoch, optimizer=None, best_score=None, steps_this_epoch=0, best_path="", last_path="",
                   epoch_finished=True):
        """A pytorch_lightning-1.7-shaped checkpoint dict, the layout ModelCheckpoint writes for the reference
        (configs/train_complete.yaml:6-14): `state_dict` with the `model.` prefix, `optimizer_states` in
        torch.optim.Adam's layout, `hyper_parameters` (save_hyperparameters(hparams), trainer_complete.py:24), epoch /
        global_step, and the loop / callback records of `lightning_state`.  Guaranteed (tested) interop: weights +
        hyper-parameters + Adam state, both directions, and `fit --ckpt_path` resume inside THIS trainer.  The loop and
        callback records follow Lightning 1.7's schema from its source but no Lightning process has read them (it is
        not installable here), so resuming one of these files inside Lightning itself is unverified - INTEGRATION.md."""
        from . import lightning_state as LS
        ck = {"epoch": epoch, "global_step": self.global_step, "pytorch-lightning_version": "1.7.7",
              "state_dict": {"model." + k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
              "loops": LS.loops_state(epoch + 1, self.global_step, steps_this_epoch, epoch_finished=epoch_finished),
              "callbacks": {LS.checkpoint_callback_key(): LS.checkpoint_callback_state(best_score, best_path, last_path,
                                                                                       os.path.dirname(last_path))},
              "optimizer_states": [], "lr_schedulers": [],
              "hparams_name": "hparams", "hyper_parameters": {"hparams": self.hparams_dict}}
        if optimizer is not None:
            ck["optimizer_states"] = [optimizer.torch_state_dict() if hasattr(optimizer, "torch_state_dict")
                                      else optimizer.state_dict()]
        return ck

    @staticmethod
    def _read_checkpoint_file(path):
        """torch.load restricted to tensors / containers / plain scalars (`weights_only=True`).  A file that needs more
        than that (arbitrary pickled objects - older Lightning files can carry argparse Namespaces or callback objects) is
        refused unless PLANK_TRUST_CHECKPOINT=1: unpickling runs code from the file, and `--ckpt_path` is user input."""
        import pickle
        try:
            return torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as exc:
            if os.environ.get("PLANK_TRUST_CHECKPOINT") != "1":
                raise RuntimeError(f"{path}: not loadable with weights_only=True ({exc}); set PLANK_TRUST_CHECKPOINT=1 to "
                                   f"unpickle it anyway - only for files you trust") from exc
            print(f"[plankassembly_amd] WARNING: unpickling {path} without restrictions (PLANK_TRUST_CHECKPOINT=1)")
            return torch.load(path, map_location="cpu", weights_only=False)

    def load_checkpoint(self, path, optimizer=None):
        """Weights always; with ``optimizer`` also the Adam moments / step, epoch and global_step (what Lightning's
        ``fit --ckpt_path`` resumes).  Returns the checkpoint dict."""
        from . import lightning_state as LS
        ck = self._read_checkpoint_file(path)
        sd = ck.get("state_dict", ck)
        sd = {(k[6:] if k.startswith("model.") else k): v for k, v in sd.items()}
        self.model.load_state_dict(sd)
        if optimizer is not None and ck.get("optimizer_states"):
            optimizer.load_state_dict(ck["optimizer_states"][0])
        if optimizer is not None:
            self.global_step = int(ck.get("global_step", 0))
            self.resume_epoch = LS.epochs_done_of(ck)
            for key, state in (ck.get("callbacks") or {}).items():
                if str(key).startswith("ModelCheckpoint") and isinstance(state, dict) and state.get("best_model_score") is not None:
                    self.resume_best = float(state["best_model_score"])
                    # ModelCheckpoint restores best_model_path too: the file a better epoch must replace (save_top_k: 1)
                    self.resume_best_path = str(state.get("best_model_path") or "")
        return ck


class VisibleTrainer(Trainer):
    """reference trainer_visible.py: no augmentation in the train loader."""
    default_lines = (8, 249)
    train_augmentation = False


class SidefaceTrainer(Trainer):
    """reference trainer_sideface.py: side-face tokens, no ``input_type``; a sample with no detected
    side faces (input = [END, PAD...]) scores 0 and writes an empty prediction (:46-52)."""
    with_type = False
    default_lines = (0, 74)
    dataset_cls = SidefaceDataset

    def test_step(self, batch, batch_idx):
        outputs = self.model(batch)
        out_dir = os.path.join(self.logger.log_dir, "pred_jsons")
        os.makedirs(out_dir, exist_ok=True)
        for name, mask, pred, gt in zip(batch["name"], batch["input_mask"], outputs["predicts"], outputs["groundtruths"]):
            gtl = gt.cpu().numpy().reshape(-1, 6).tolist()
            if bool(torch.all(mask[1:])):
                predl, scores = [], {"precision": 0.0, "recall": 0.0, "fmeasure": 0.0}
            else:
                vp = self._valid_pred(pred)
                scores = self.scorer.add(vp, gt)
                predl = vp.cpu().numpy().reshape(-1, 6).tolist()
            self._write_pred_json(out_dir, name, {"prediction": predl, "groundtruth": gtl, **scores})


# ====================================================================================== loop + CLI
def _to_device(batch, dev, model=None):
    if model is not None and hasattr(model, "prepare_batch"):
        return model.prepare_batch(batch)
    return {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


def run(trainer_cls, subcommand, config, ckpt_path=None, overrides=None):
    """The part of ``pl.Trainer.fit/test`` the reference relies on."""
    seed, tkw, hparams = load_cli_config(config)
    for k, v in (overrides or {}).items():
        tkw[k] = v
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PLANK_DIST_BACKEND=gloo is for TESTS (tests/test_cli_gpu.py): RCCL refuses two ranks on one device; with gloo the ranks of a
    # one-GPU box share cuda:0 and the whole DDP path of this loop (DistributedSampler shards, GradSync, metric sums) still runs.
    backend = os.environ.get("PLANK_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if seed is not None:
        torch.manual_seed(int(seed)); np.random.seed(int(seed))
    module = trainer_cls(hparams)
    module.logger = _Logger()
    if world > 1:                                  # one lightning_logs/version_N for the whole job: rank 0's choice
        box = [module.logger.log_dir]
        dist.broadcast_object_list(box, src=0)
        module.logger.log_dir = box[0]
    dev = torch.device("cuda", local)
    module.model.to(dev)
    if ckpt_path and subcommand == "test":
        module.load_checkpoint(ckpt_path)
    if subcommand == "test":
        module.model.eval()
        with torch.no_grad():
            for i, batch in enumerate(module.test_dataloader()):
                module.test_step(_to_device(batch, dev, module.model), i)
        module.test_epoch_end()
        if rank == 0:
            print({k: round(v, 4) for k, v in module._logged.items()})
        return module
    opt = module.configure_optimizers()["optimizer"]
    module.optimizer = opt
    start_epoch, best, best_file = 0, -1.0, ""
    if ckpt_path:                                  # fit --ckpt_path: weights + Adam state + counters, like Lightning
        module.load_checkpoint(ckpt_path, optimizer=opt)
        start_epoch = getattr(module, "resume_epoch", 0)
        best = getattr(module, "resume_best", -1.0)
        best_file = getattr(module, "resume_best_path", "")
        if best_file and not os.path.exists(best_file):
            best_file = ""
    sync = None
    if world > 1:
        from .distributed import GradSync
        sync = GradSync(module.model)
        sync.broadcast_parameters(0)
    max_epochs = int(tkw.get("max_epochs", 1))
    every = int(tkw.get("check_val_every_n_epoch", 1))
    max_steps = int(tkw.get("max_steps", -1))
    loader = module.train_dataloader()
    for epoch in range(start_epoch, max_epochs):
        if 0 < max_steps <= module.global_step:            # a resumed run whose epoch was cut by max_steps: nothing left to do
            break
        module.model.train()
        if hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(epoch)
        t0, n, steps_here = time.perf_counter(), 0, 0
        from .data import DevicePrefetcher
        n_batches = len(loader) if hasattr(loader, "__len__") else -1
        for i, batch in enumerate(DevicePrefetcher(module.model, loader)):     # batch i + 1 is prepared while step i runs
            opt.zero_grad()
            loss = module.training_step(batch, i)
            loss.backward()
            opt.step()
            module.global_step += 1
            steps_here += 1
            n += batch["input_value"].shape[0]
            if 0 < max_steps <= module.global_step:
                break
        l, a = module._train_stats
        module.log("train/loss", l); module.log("train/accuracy", a)
        torch.cuda.synchronize()
        if rank == 0:
            print(f"epoch {epoch}: train/loss {float(l):.4f} train/accuracy {float(a):.4f} "
                  f"{n * world / (time.perf_counter() - t0):.1f} samples/s")
        finished = not (0 <= steps_here < n_batches)       # False: max_steps cut the epoch short - a resume runs it again
        if (epoch + 1) % every == 0:
            ckdir = os.path.join(module.logger.log_dir, "checkpoints")
            last = os.path.join(ckdir, "last.ckpt")
            if rank == 0:                                              # save_last: before the metric exchange, so a failure
                os.makedirs(ckdir, exist_ok=True)                      # in validation cannot lose the epoch's weights
                torch.save(module.checkpoint(epoch, opt, best if best >= 0 else None, steps_here, best_file, last,
                                             epoch_finished=finished), last)
            module.model.eval()
            with torch.no_grad():
                for i, batch in enumerate(module.val_dataloader()):
                    module.validation_step(_to_device(batch, dev, module.model), i)
            module.validation_epoch_end()
            f1 = module._logged.get("val/fmeasure", 0.0)
            if rank == 0:
                improved = f1 > best                                   # ModelCheckpoint(monitor=val/fmeasure, mode=max)
                if improved:
                    best = f1
                    stale, best_file = best_file, os.path.join(ckdir, (
                        f"checkpoint_{epoch:03d}-precision={module._logged['val/precision']:.3f}-"
                        f"recall={module._logged['val/recall']:.3f}-f1={f1:.3f}.ckpt"))
                ck = module.checkpoint(epoch, opt, best, steps_here, best_file, last, epoch_finished=finished)
                torch.save(ck, last)
                if improved:
                    torch.save(ck, best_file)
                    # save_top_k: 1 - but only among THIS run's checkpoints.  A best file inherited from the checkpoint being
                    # resumed lives in an earlier run's version_N/checkpoints (often it IS --ckpt_path): Lightning 1.7 restores
                    # best_model_path only for a matching dirpath and never deletes another run's file; neither do we.
                    if (stale and stale != best_file and os.path.exists(stale)
                            and os.path.abspath(os.path.dirname(stale)) == os.path.abspath(ckdir)
                            and not (ckpt_path and os.path.abspath(stale) == os.path.abspath(ckpt_path))):
                        os.remove(stale)
                print({k: round(v, 4) for k, v in module._logged.items() if k.startswith("val/")})
        if 0 < max_steps <= module.global_step:
            break
    if sync is not None:
        dist.barrier()
    return module


def cli(trainer_cls, argv=None):
    """``python trainer_complete.py fit --config configs/train_complete.yaml [--ckpt_path x] [--trainer.devices 1]``"""
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("fit", "test", "validate"):
        raise SystemExit("usage: <trainer>.py {fit,test} --config CONFIG [--ckpt_path CKPT] [--trainer.KEY VALUE ...]")
    sub, config, ckpt, over = argv[0], None, None, {}
    i = 1
    while i < len(argv):
        key, val = argv[i], argv[i + 1] if i + 1 < len(argv) else None
        if "=" in key:
            key, val = key.split("=", 1)
            i += 1
        else:
            i += 2
        if key == "--config":
            config = val
        elif key == "--ckpt_path":
            ckpt = val
        elif key.startswith("--trainer."):
            try:
                val = json.loads(val)
            except (ValueError, TypeError):
                pass
            over[key[len("--trainer."):]] = val
        else:
            raise SystemExit(f"unknown argument {key}")
    if config is None:
        raise SystemExit("--config is required")
    return run(trainer_cls, "test" if sub != "fit" else "fit", config, ckpt, over)
