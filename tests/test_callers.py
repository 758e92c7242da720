"""The callers either side of the hot path against golden vectors of the reference's own code
(tests/golden/make_golden_callers.py): tokenisers + info-JSON reader (f2), validation / test step box pipeline and
pred-json writer (f1, f3), evaluate.py (f3), Lightning-shaped checkpoints (f4).  CPU parts here; the parts that need
the HIP model carry the gpu marker."""
import json
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, load_fixture
from plankassembly_amd import datasets as DS
from plankassembly_amd.config import CfgNode

TOKEN = types.SimpleNamespace(END=512, PAD=513)


def data_cfg(max_in, max_out):
    return CfgNode(dict(VOCAB_SIZE=514, NUM_INPUT_DOF=4, MAX_INPUT_LENGTH=max_in, MAX_OUTPUT_LENGTH=max_out, NUM_BITS=9,
                        AUG_RATIO=0.1, NOISE_RATIO=0.15, NOISE_LENGTH=0.02))


# ------------------------------------------------------------------------------------------------ f2: tokenisers
def test_g10_quantize_dequantize_vectors():
    z = np.load(os.path.join(GOLDEN, "data_tokens.npz"))
    for bits in (9, 8):
        assert np.array_equal(DS.quantize_values(z["q::in"], bits), z[f"q::quant{bits}"])
        assert np.array_equal(DS.dequantize_values(np.arange(2 ** bits), bits), z[f"q::dequant{bits}"])
    assert DS.quantize_values(np.array([-1.0, 1.0]), 9).tolist() == [0, 511]


@pytest.mark.parametrize("i", [0, 1, 2])
def test_line_dataset_reads_info_json_like_the_reference(i):
    """info file -> batch rows == the reference's prepare_input_sequence / prepare_output_sequence on the same data."""
    z = np.load(os.path.join(GOLDEN, "data_tokens.npz"))
    ds = DS.LineDataset(os.path.join(GOLDEN, "infos"), [f"item{j}.json" for j in range(3)], TOKEN, data_cfg(120, 64))
    item = ds[i]
    assert item["name"] == f"item{i}"
    for k in ("input_value", "input_pos", "input_coord", "input_view", "input_type", "input_mask", "output_value",
              "output_label", "output_mask"):
        assert np.array_equal(np.asarray(item[k]), z[f"item{i}::{k}"]), k
        assert np.asarray(item[k]).dtype == z[f"item{i}::{k}"].dtype, k
    assert len(item["input_value"]) == 119 and len(item["input_pos"]) == 119       # MAX_INPUT_LENGTH - 1 (a0)
    assert list(item)[:7] == ["name", "input_value", "input_pos", "input_coord", "input_view", "input_type", "input_mask"]


@pytest.mark.parametrize("i", [0, 1, 2])
def test_sideface_dataset_tokens(i):
    z = np.load(os.path.join(GOLDEN, "data_tokens.npz"))
    ds = DS.SidefaceDataset(os.path.join(GOLDEN, "infos"), [f"item{j}.json" for j in range(3)], TOKEN, data_cfg(60, 64))
    item = ds[i]
    assert "input_type" not in item
    for k in ("input_value", "input_pos", "input_coord", "input_view", "input_mask"):
        assert np.array_equal(np.asarray(item[k]), z[f"side{i}::{k}"]), k
    empty = ds.prepare_input_sequence([], [])
    for k, v in empty.items():
        assert np.array_equal(np.asarray(v), z[f"side_empty::{k}"]), k
    assert empty["input_value"][0] == 512 and bool(empty["input_mask"][1:].all())


def test_dataloader_collates_real_infos(tmp_path):
    """Trainer.train_dataloader() uses the info files when ROOT and the split file exist (trainer_complete.py:35-43)."""
    from plankassembly_amd.config import load_cli_config
    from plankassembly_amd.trainer import Trainer
    _, _, hp = load_cli_config(os.path.join(REPO, "configs", "train_complete.yaml"))
    hp["MODEL"].update(NUM_MODEL=64, NUM_HEAD=4, NUM_FEEDFORWARD=128, NUM_ENCODER_LAYERS=1, NUM_DECODER_LAYERS=1)
    hp["DATA"].update(MAX_INPUT_LENGTH=120, MAX_OUTPUT_LENGTH=64, AUG_RATIO=0.0)
    split = tmp_path / "train.txt"
    split.write_text("item0.json\nitem1.json\nitem2.json\n")
    hp.update(ROOT=os.path.join(GOLDEN, "infos"), DATASETS_TRAIN=str(split), BATCH_SIZE=3, NUM_WORKERS=0)
    t = Trainer(hp)
    loader = t.train_dataloader()
    assert isinstance(loader.dataset, DS.LineDataset)
    batch = next(iter(loader))
    assert batch["input_value"].shape == (3, 119) and batch["input_value"].dtype == torch.int64
    assert batch["input_mask"].dtype == torch.bool and sorted(batch["name"]) == ["item0", "item1", "item2"]


def test_add_noise_shortens_or_deletes_segments():
    np.random.seed(3)
    segs = [np.array([[0.0, 0.0], [1.0, 0.0]]), np.array([[0.0, 0.0], [0.0, 0.5]]), np.array([[0.2, 0.2], [0.2, 0.201]])] * 8
    out, views, types = DS.add_noise(segs, list(range(24)), [0] * 24, 0.5, 0.02)
    assert 0 < len(out) <= 24 and len(out) == len(views) == len(types)
    for v, s in zip(views, out):
        ref = segs[v]
        assert np.linalg.norm(s[1] - s[0]) <= np.linalg.norm(ref[1] - ref[0]) + 1e-12


# ------------------------------------------------------------------------------------------------ f1 / f3: box pipeline
class _FakeModel(torch.nn.Module):
    """Returns stored decode outputs (CPU stand-in for the HIP model in the host-logic tests)."""

    def __init__(self, samples, attach, parse):
        super().__init__()
        self.samples, self.attach, self.parse = samples, attach, parse

    def forward(self, batch):
        return {"samples": self.samples, "attach": self.attach,
                "predicts": [self.parse(s) for s in self.samples],
                "groundtruths": [self.parse(s) for s in batch["output_value"]]}


def _f1_trainer(cls=None):
    from plankassembly_amd.config import load_cli_config
    from plankassembly_amd.trainer import Trainer
    _, _, hp = load_cli_config(os.path.join(REPO, "configs", "train_complete.yaml"))
    hp["MODEL"].update(NUM_MODEL=64, NUM_HEAD=4, NUM_FEEDFORWARD=128, NUM_ENCODER_LAYERS=2, NUM_DECODER_LAYERS=2,
                       DROPOUT=0.0, COMPUTE_DTYPE="f32")
    hp["DATA"].update(MAX_INPUT_LENGTH=65, MAX_OUTPUT_LENGTH=36)
    return (cls or Trainer)(hp)


def _check_pred_jsons(out_dir, g, n):
    for i in range(n):
        with open(os.path.join(out_dir, "pred_jsons", f"f1case{i}.json")) as f:
            d = json.load(f)
        assert d["prediction"] == g[f"valid_pred{i}"].tolist()
        assert d["attach"] == g[f"attach_rows{i}"].tolist()
        assert np.allclose([d["precision"], d["recall"], d["fmeasure"]], g[f"prf{i}"], atol=1e-7)
        assert sorted(d) == sorted(["prediction", "attach", "groundtruth", "precision", "recall", "fmeasure"])


def test_validation_and_test_steps_reproduce_reference_f1_host_logic(tmp_path):
    """validation_step / test_step / epoch_end on the reference's decode outputs == the reference's P/R/F1, and the
    pred_jsons + evaluate.py round trip == the reference's dequantised re-scoring."""
    import evaluate as EV
    sd, batch, g = load_fixture("fixture_f1.npz")
    t = _f1_trainer()
    t.model = _FakeModel(torch.from_numpy(g["samples"]), torch.from_numpy(g["attach"]), t.model.parse_sequence)
    t.validation_step(batch, 0)
    t.validation_epoch_end()
    got = [t._logged[k] for k in ("val/precision", "val/recall", "val/fmeasure")]
    assert np.allclose(got, g["epoch_prf"], atol=1e-7), (got, g["epoch_prf"])
    t.logger = types.SimpleNamespace(log_dir=str(tmp_path), log=lambda *a: None)
    n = int(g["n"])
    tb = dict(batch, name=[f"f1case{i}" for i in range(n)])
    t.test_step(tb, 0)
    t.test_epoch_end()
    assert np.allclose([t._logged[k] for k in ("test/precision", "test/recall", "test/fmeasure")], g["epoch_prf"], atol=1e-7)
    _check_pred_jsons(str(tmp_path), g, n)
    p, r, f, metrics = EV.evaluate(GOLDEN, str(tmp_path), 0.5, 9, verbose=False)
    assert np.allclose([p, r, f], g["eval_epoch_prf"], atol=1e-7)
    for i in range(n):
        m = metrics[f"f1case{i}"]
        assert np.allclose([m["precision"], m["recall"], m["fmeasure"]], g[f"eval_prf{i}"], atol=1e-7)
    assert os.path.exists(os.path.join(str(tmp_path), "metrics.json"))


@pytest.mark.gpu
def test_f1_end_to_end_on_the_gpu(tmp_path):
    """f1: HIP greedy decode -> box filter -> matcher -> Criterion == the reference's P/R/F1 for the same weights
    (trainer_complete.py:73-89), through the Trainer's own validation / test hooks."""
    sd, batch, g = load_fixture("fixture_f1.npz")
    t = _f1_trainer()
    t.model.load_state_dict(sd)
    t.model.cuda().eval()
    gb = t.model.prepare_batch(batch)
    with torch.no_grad():
        out = t.model(gb)
        assert np.array_equal(out["samples"].cpu().numpy(), g["samples"])
        assert np.array_equal(out["attach"].cpu().numpy(), g["attach"])
        t.validation_step(gb, 0)
    t.validation_epoch_end()
    got = [t._logged[k] for k in ("val/precision", "val/recall", "val/fmeasure")]
    assert np.allclose(got, g["epoch_prf"], atol=1e-7), (got, g["epoch_prf"])
    assert 0.5 < got[2] < 1.0
    t.logger = types.SimpleNamespace(log_dir=str(tmp_path), log=lambda *a: None)
    n = int(g["n"])
    with torch.no_grad():
        t.test_step(dict(gb, name=[f"f1case{i}" for i in range(n)]), 0)
    t.test_epoch_end()
    _check_pred_jsons(str(tmp_path), g, n)


# ------------------------------------------------------------------------------------------------ f4: checkpoints
def _ckpt_trainer():
    ck = torch.load(os.path.join(GOLDEN, "lightning_small.ckpt"), map_location="cpu", weights_only=True)
    from plankassembly_amd.trainer import Trainer
    hp = ck["hyper_parameters"]["hparams"]
    hp["MODEL"]["COMPUTE_DTYPE"] = "f32"
    return Trainer(hp), ck


def test_lightning_checkpoint_loads_weights_optimizer_and_counters():
    from plankassembly_amd.optim import FusedAdam
    t, ck = _ckpt_trainer()
    assert ck["pytorch-lightning_version"].startswith("1.7") and "optimizer_states" in ck and "callbacks" in ck
    opt = FusedAdam(t.model, lr=123.0)
    t.load_checkpoint(os.path.join(GOLDEN, "lightning_small.ckpt"), optimizer=opt)
    for k, v in t.model.state_dict().items():
        assert torch.equal(v, ck["state_dict"]["model." + k]), k
    assert t.global_step == 2 and t.resume_epoch == 2 and abs(t.resume_best - 0.5) < 1e-12
    assert opt._step == 2 and opt.param_groups[0]["lr"] == 1e-3
    st = ck["optimizer_states"][0]["state"]
    names = [k for k, p in t.model.named_parameters() if p.requires_grad]
    for i in (0, 7, len(names) - 1):
        off, n = t.model._offsets[names[i]], st[i]["exp_avg"].numel()
        assert torch.equal(opt._m[off:off + n].view_as(st[i]["exp_avg"]), st[i]["exp_avg"])
        assert torch.equal(opt._v[off:off + n].view_as(st[i]["exp_avg_sq"]), st[i]["exp_avg_sq"])
    # and back: the checkpoint this trainer writes has the same shape and content
    out = t.checkpoint(1, opt, 0.5)
    assert set(ck) <= set(out)
    assert all(torch.equal(out["state_dict"][k], ck["state_dict"][k]) for k in ck["state_dict"])
    o2 = out["optimizer_states"][0]
    assert o2["param_groups"][0]["params"] == ck["optimizer_states"][0]["param_groups"][0]["params"]
    for i in st:
        assert torch.equal(o2["state"][i]["exp_avg"], st[i]["exp_avg"]) and float(o2["state"][i]["step"]) == 2.0
    # torch.optim.Adam accepts it (what the reference's Lightning resume does)
    ref_opt = torch.optim.Adam(t.model.parameters(), lr=1.0)
    ref_opt.load_state_dict(o2)
    assert ref_opt.param_groups[0]["lr"] == 1e-3


def test_written_checkpoint_has_lightning_17_loop_and_callback_schema():
    """the dict `Trainer.checkpoint` writes must carry Lightning 1.7's full loop state (every progress record a
    total/current pair with its tracker's field set) and the ModelCheckpoint state under Lightning's own `state_key` -
    the key the golden file (typed from the 1.7 source, tests/golden/make_golden_callers.py) uses."""
    from plankassembly_amd import lightning_state as LS
    t, ck = _ckpt_trainer()
    t.global_step = 37
    out = t.checkpoint(4, None, 0.25, steps_this_epoch=9, best_path="/x/checkpoints/best.ckpt", last_path="/x/checkpoints/last.ckpt")
    (golden_key,) = [k for k in ck["callbacks"] if k.startswith("ModelCheckpoint")]
    assert list(out["callbacks"]) == [golden_key] == [LS.checkpoint_callback_key()]
    assert set(out["callbacks"][golden_key]) == set(ck["callbacks"][golden_key])
    cb = out["callbacks"][golden_key]
    assert float(cb["best_model_score"]) == 0.25 and cb["best_model_path"].endswith("best.ckpt") and cb["dirpath"] == "/x/checkpoints"
    assert set(out["loops"]) == {"fit_loop", "validate_loop", "test_loop", "predict_loop"}
    fit = out["loops"]["fit_loop"]

    def fields(rec, names):
        assert set(rec) >= {"total", "current"}, rec
        assert set(rec["total"]) == set(names) == set(rec["current"]), rec

    fields(fit["epoch_progress"], ("ready", "completed", "started", "processed"))
    fields(fit["epoch_loop.batch_progress"], ("ready", "completed", "started", "processed"))
    assert fit["epoch_loop.batch_progress"]["is_last_batch"] is True
    fields(fit["epoch_loop.scheduler_progress"], ("ready", "completed"))
    op = fit["epoch_loop.batch_loop.optimizer_loop.optim_progress"]
    fields(op["optimizer"]["step"], ("ready", "completed"))
    fields(op["optimizer"]["zero_grad"], ("ready", "completed", "started"))
    fields(fit["epoch_loop.val_loop.dataloader_progress"], ("ready", "completed"))
    fields(fit["epoch_loop.val_loop.epoch_loop.batch_progress"], ("ready", "completed", "started", "processed"))
    for k in ("state_dict", "epoch_loop.state_dict", "epoch_loop.batch_loop.state_dict", "epoch_loop.batch_loop.optimizer_loop.state_dict",
              "epoch_loop.batch_loop.manual_loop.state_dict", "epoch_loop.val_loop.state_dict"):
        assert k in fit, k
    # counters agree with each other and with the top level: written inside epoch index 4 -> 5 started, 4 completed
    assert out["epoch"] == 4 == fit["epoch_progress"]["current"]["completed"] and fit["epoch_progress"]["current"]["started"] == 5
    assert fit["epoch_loop.state_dict"]["_batches_that_stepped"] == 37 == out["global_step"]
    assert op["optimizer"]["step"]["total"]["completed"] == 37 and op["optimizer"]["step"]["current"]["completed"] == 9
    assert LS.epochs_done_of(out) == 5 and LS.epochs_done_of(ck) == 2
    import io
    buf = io.BytesIO()
    torch.save(out, buf)                                      # and it is a weights_only-loadable file
    buf.seek(0)
    assert torch.load(buf, weights_only=True)["global_step"] == 37


def test_load_checkpoint_refuses_arbitrary_pickles(tmp_path, monkeypatch):
    """a --ckpt_path file that needs full unpickling is refused unless PLANK_TRUST_CHECKPOINT=1."""
    import argparse
    t, ck = _ckpt_trainer()
    bad = dict(ck, hyper_parameters=argparse.Namespace(x=1))
    path = str(tmp_path / "bad.ckpt")
    torch.save(bad, path)
    monkeypatch.delenv("PLANK_TRUST_CHECKPOINT", raising=False)
    with pytest.raises(RuntimeError, match="PLANK_TRUST_CHECKPOINT"):
        t.load_checkpoint(path)
    monkeypatch.setenv("PLANK_TRUST_CHECKPOINT", "1")
    assert "state_dict" in t.load_checkpoint(path)


@pytest.mark.gpu
def test_resume_from_lightning_checkpoint_continues_the_reference_run():
    """fit --ckpt_path semantics: after loading weights + Adam moments + step, the next FusedAdam step lands on the
    parameters the reference's torch.optim.Adam reached on its 3rd step."""
    from plankassembly_amd.optim import FusedAdam
    _, batch, _ = load_fixture("fixture_small.npz")
    exp = np.load(os.path.join(GOLDEN, "lightning_small_expect.npz"))
    t, ck = _ckpt_trainer()
    t.model.cuda().train()
    opt = FusedAdam(t.model, lr=1.0)
    t.load_checkpoint(os.path.join(GOLDEN, "lightning_small.ckpt"), optimizer=opt)
    opt.zero_grad()
    out = t.model(t.model.prepare_batch(batch))
    assert abs(out["loss"].item() - float(exp["loss3"])) < 1e-4
    out["loss"].backward()
    opt.step()
    worst = 0.0
    for k, v in t.model.state_dict().items():
        worst = max(worst, float((v.cpu() - torch.from_numpy(exp["p3::" + k])).abs().max()))
    assert worst < 5e-5, worst            # lr 1e-3: an Adam step moves entries by <= 1e-3; agreement to 5 % of a step
