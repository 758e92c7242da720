"""CPU-only checks of the drop-in surface of plankassembly_amd.models (no GPU, no compute calls)."""
import ctypes
import os
import re
import types

import pytest
import torch

from plankassembly_amd import _lib as L
from plankassembly_amd.config import CfgNode
from plankassembly_amd.models import PlankModel, build_model, param_order

TOKEN = types.SimpleNamespace(END=512, PAD=513)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_model(**kw):
    return PlankModel(64, 4, 128, 0.0, "relu", True, 2, 2, 3, 2, 4, 6, 65, 36, 514, TOKEN, **kw)


def test_state_dict_matches_reference_fixture(small_fixture):
    sd, _, _ = small_fixture
    m = small_model()
    mine = m.state_dict()
    assert list(mine.keys()) == list(sd.keys())            # same names, same ORDER as the reference
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])
    # parameters stay views of the one flat buffer after loading
    base = m.flat_params.untyped_storage().data_ptr()
    assert all(p.untyped_storage().data_ptr() == base for p in m.parameters())
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in sd.values())


def test_default_config_parameter_count():
    cfg = CfgNode({"MODEL": dict(NUM_MODEL=512, NUM_HEAD=8, NUM_FEEDFORWARD=1024, DROPOUT=0.2, ACTIVATION="relu",
                                 NORMALIZE_BEFORE=True, NUM_ENCODER_LAYERS=6, NUM_DECODER_LAYERS=6),
                   "DATA": dict(NUM_VIEW=3, NUM_TYPE=2, NUM_INPUT_DOF=4, NUM_OUTPUT_DOF=6, MAX_INPUT_LENGTH=1200,
                                MAX_OUTPUT_LENGTH=128, VOCAB_SIZE=514),
                   "TOKEN": dict(END=512, PAD=513)})
    m = build_model(cfg)
    assert sum(p.numel() for p in m.parameters()) == 32_507_907
    assert m.eps_layer == 1.0 and m.has_enc_norm                       # the positional-argument slip
    assert m.state_dict()["input_embeddings.input_pos.weight"].shape == (300, 512)
    assert m.state_dict()["query_pos_embedding.weight"].shape == (22, 512)


def test_initial_ffn_biases_are_one_draw_per_stack_like_torchs_deep_copied_layers():
    """torch's TransformerEncoder / TransformerDecoder deep-copy one layer and the reference re-draws only dim > 1 parameters
    (/root/reference/plankassembly/models.py:60-69,78-83): linear1.bias / linear2.bias start identical in every
    layer of a stack, non-zero, and different between the two stacks; the matrices differ per layer."""
    torch.manual_seed(5)
    m = PlankModel(64, 4, 128, 0.0, "relu", True, 3, 2, 3, 2, 4, 6, 65, 36, 514, TOKEN)
    sd = m.state_dict()
    for name in ("linear1.bias", "linear2.bias"):
        e0 = sd["encoder.layers.0." + name]
        assert float(e0.abs().max()) > 0
        assert torch.equal(sd["encoder.layers.1." + name], e0) and torch.equal(sd["encoder.layers.2." + name], e0)
        assert torch.equal(sd["decoder.layers.1." + name], sd["decoder.layers.0." + name])
        assert not torch.equal(sd["decoder.layers.0." + name], e0)
    assert not torch.equal(sd["encoder.layers.0.linear1.weight"], sd["encoder.layers.1.linear1.weight"])


def test_normalize_before_false_has_no_encoder_norm():
    m = PlankModel(64, 4, 128, 0.0, "relu", False, 1, 1, 3, 2, 4, 6, 65, 36, 514, TOKEN)
    assert "encoder.norm.weight" not in m.state_dict() and "decoder.norm.weight" in m.state_dict()
    assert m.eps_layer == 0.0


def test_segment_slices_partition_the_flat_buffer():
    m = small_model()
    sl = m.segment_slices()
    assert len(sl) == 2 + 2 + 4
    cover = sorted(sl)
    assert cover[0][0] == 0 and cover[-1][1] == m.flat_params.numel()
    for (a, b), (c, d) in zip(cover, cover[1:]):
        assert b == c
    assert param_order(2, 2) == list(m._shapes)


def test_cpu_module_fails_loudly():
    m = small_model()
    m.train()
    with pytest.raises(L.PlankHipError):
        m({"input_value": torch.zeros(1, 64, dtype=torch.long)})


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads and exports every function include/plank_hip.h declares."""
    hdr = open(os.path.join(REPO, "include", "plank_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", hdr))
    names -= {"pa_gemm_args", "pa_attn_args"}
    assert len(names) >= 25
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in plank_hip.h but not exported"
    assert L.lib().pa_version() >= 1


def test_group_rows_by_id_segments():
    """Host side of the segment-sum embedding gradients: every entry lands in exactly the segment of its table row."""
    import torch
    from plankassembly_amd.models import group_rows_by_id
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 11, (257,), generator=g)
    rows_idx = torch.arange(1000, 1257)
    order, seg = group_rows_by_id(ids, 13, rows_idx)
    assert seg[0] == 0 and seg[-1] == 257 and order.dtype == torch.int32 and seg.dtype == torch.int32
    seen = torch.zeros(257, dtype=torch.bool)
    for r in range(13):
        sl = order[seg[r]:seg[r + 1]].long() - 1000
        assert bool((ids[sl] == r).all())
        seen[sl] = True
    assert bool(seen.all())
    # emulate the kernel: table gradient = sum of the gradient rows of each segment
    dout = torch.randn(257, 8, generator=g)
    ref = torch.zeros(13, 8).index_add_(0, ids, dout)
    got = torch.stack([dout[order[seg[r]:seg[r + 1]].long() - 1000].sum(0) for r in range(13)])
    assert torch.allclose(got, ref, atol=1e-5)
