"""The non-tensor half of a pytorch_lightning 1.7 checkpoint: loop progress trackers and callback state.

Lightning itself is not installable in this image, so this module restates the layout its 1.7.x `Trainer.save_checkpoint`
writes (pytorch_lightning/loops/fit_loop.py, loops/epoch/training_epoch_loop.py, trainer/progress.py,
callbacks/model_checkpoint.py in the 1.7 source tree) for the reference's trainer set-up (configs/train_complete.yaml:3-22:
automatic optimisation, one optimizer, no LR scheduler, ModelCheckpoint(monitor=val/fmeasure, mode=max, save_top_k=1,
save_last=True)).  What can be checked here is checked: tests/test_callers.py validates the schema (every progress record has
its total/current pair with the field set of its tracker class; the ModelCheckpoint entry sits under Lightning's own
`state_key`) and that the counters are mutually consistent.  What cannot be checked offline is an actual `fit --ckpt_path`
of such a file inside Lightning - INTEGRATION.md says so; weights-only use (`test --ckpt_path`, `load_from_checkpoint`)
needs nothing from this module.
"""
from __future__ import annotations

READY_COMPLETED = ("ready", "completed")                                 # progress.ReadyCompletedTracker
STARTED = ("ready", "completed", "started")                              # progress.StartedTracker
PROCESSED = ("ready", "completed", "started", "processed")               # progress.ProcessedTracker


def _progress(fields, total, current):
    return {"total": {f: int(total) for f in fields}, "current": {f: int(current) for f in fields}}


def _batch_progress(total, current, last):
    rec = _progress(PROCESSED, total, current)
    rec["is_last_batch"] = bool(last)
    return rec


def checkpoint_callback_key(monitor="val/fmeasure", mode="max"):
    """`ModelCheckpoint.state_key` (callbacks/model_checkpoint.py: `_generate_state_key` over these six arguments)."""
    args = {"monitor": monitor, "mode": mode, "every_n_train_steps": 0, "every_n_epochs": 1, "train_time_interval": None,
            "save_on_train_epoch_end": None}
    return "ModelCheckpoint" + repr(args)


def checkpoint_callback_state(best_score, best_path="", last_path="", dirpath="", monitor="val/fmeasure"):
    import torch
    score = None if best_score is None else torch.tensor(float(best_score))
    return {"monitor": monitor, "best_model_score": score, "best_model_path": best_path, "current_score": score,
            "dirpath": dirpath, "best_k_models": {best_path: score} if (best_path and score is not None) else {},
            "kth_best_model_path": best_path, "kth_value": score, "last_model_path": last_path}


def _eval_loop(batches_seen=0, runs=0):
    return {"state_dict": {}, "dataloader_progress": _progress(READY_COMPLETED, runs, 0),
            "epoch_loop.state_dict": {}, "epoch_loop.batch_progress": _batch_progress(batches_seen, 0, False)}


def loops_state(epochs_done, global_step, steps_this_epoch, val_batches_seen=0, val_runs=0, epoch_finished=True):
    """`checkpoint["loops"]` at the end of training epoch number `epochs_done` (1-based count of finished epochs), after
    `global_step` optimizer steps in total of which `steps_this_epoch` fell into the last epoch."""
    val = _eval_loop(val_batches_seen, val_runs)
    fit = {
        "state_dict": {},
        "epoch_loop.state_dict": {"_batches_that_stepped": int(global_step)},
        "epoch_loop.batch_progress": _batch_progress(global_step, steps_this_epoch, epoch_finished),
        "epoch_loop.scheduler_progress": _progress(READY_COMPLETED, 0, 0),
        "epoch_loop.batch_loop.state_dict": {},
        "epoch_loop.batch_loop.optimizer_loop.state_dict": {},
        "epoch_loop.batch_loop.optimizer_loop.optim_progress": {
            "optimizer": {"step": _progress(READY_COMPLETED, global_step, steps_this_epoch),
                          "zero_grad": _progress(STARTED, global_step, steps_this_epoch)},
            "optimizer_position": 1},
        "epoch_loop.batch_loop.manual_loop.state_dict": {},
        "epoch_loop.batch_loop.manual_loop.optim_step_progress": _progress(READY_COMPLETED, 0, 0),
        "epoch_loop.val_loop.state_dict": val["state_dict"],
        "epoch_loop.val_loop.dataloader_progress": val["dataloader_progress"],
        "epoch_loop.val_loop.epoch_loop.state_dict": val["epoch_loop.state_dict"],
        "epoch_loop.val_loop.epoch_loop.batch_progress": val["epoch_loop.batch_progress"],
        "epoch_progress": _progress(PROCESSED, epochs_done, epochs_done),
    }
    # ModelCheckpoint(monitor=<a validation metric>) with check_val_every_n_epoch != 1 saves from on_validation_end
    # (model_checkpoint.py `_should_save_on_train_epoch_end`), i.e. inside the epoch: FitLoop has incremented `ready` and
    # `started` for it but neither `processed` nor `completed` (fit_loop.py `on_advance_end`).  The top-level `epoch` is
    # `current.completed`.  On restart Lightning sees `is_last_batch`, finishes the epoch's bookkeeping and goes on with the
    # next one - so "epochs done" of such a file is `epoch` + 1 = `started`.
    for scope in ("total", "current"):
        fit["epoch_progress"][scope]["processed"] = int(epochs_done) - 1
        fit["epoch_progress"][scope]["completed"] = int(epochs_done) - 1
    return {"fit_loop": fit, "validate_loop": _eval_loop(), "test_loop": _eval_loop(), "predict_loop": {
        "state_dict": {}, "dataloader_progress": _progress(READY_COMPLETED, 0, 0), "epoch_loop.state_dict": {},
        "epoch_loop.batch_progress": _progress(PROCESSED, 0, 0)}}


def epochs_done_of(ck) -> int:
    """Training epochs a `fit --ckpt_path` of this checkpoint must NOT run again: the top-level `epoch` (0-based index of
    the epoch the file was written in) + 1, for Lightning's files and ours alike; a dict without it is read through the
    fit loop's `started` counter."""
    if "epoch" in ck:
        try:            # a file written when max_steps cut an epoch short (is_last_batch False): that epoch runs again
            cut = ck["loops"]["fit_loop"]["epoch_loop.batch_progress"]["is_last_batch"] is False
        except (KeyError, TypeError):
            cut = False
        return int(ck["epoch"]) + (0 if cut else 1)
    try:
        return int(ck["loops"]["fit_loop"]["epoch_progress"]["current"]["started"])
    except (KeyError, TypeError):
        return 0
