import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def usable_cores():
    """Host cores this process may really use: affinity mask clipped by the cgroup CPU quota (the GPU boxes show every core of
    the host but grant 16: torch's default of one thread per visible core oversubscribes the quota, and the float64 oracle runs
    of the parity tests then crawl), capped at 32."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    torch.set_num_threads(usable_cores())


def _gpu_ready():
    """A ROCm device is visible AND the in-tree HIP library is built (the product has no fallback)."""
    lib = os.path.join(REPO, "plankassembly_amd", "libplank_hip.so")
    return torch.cuda.is_available() and os.path.exists(lib)


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests are skipped (not errored) on a machine without a ROCm device.  On a GPU box nothing is
    skipped - a missing libplank_hip.so there must fail loudly; PLANK_REQUIRE_GPU=1 forces that behaviour anywhere."""
    if os.environ.get("PLANK_REQUIRE_GPU") == "1" or torch.cuda.is_available():
        return                      # with a device present a missing library is an error, never a skip
    skip = pytest.mark.skip(reason="needs an MI355X (no ROCm device visible)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_fixture(name):
    """npz -> (state_dict, batch, rest) with torch tensors."""
    z = np.load(os.path.join(GOLDEN, name))
    sd, batch, rest = {}, {}, {}
    for k in z.files:
        v = z[k]
        if k.startswith("sd::"):
            sd[k[4:]] = torch.from_numpy(v.copy())
        elif k.startswith("batch::"):
            batch[k[7:]] = torch.from_numpy(v.copy())
        else:
            rest[k] = v
    # batch-dict order matters for the embedding sum (reference models.py:107): value,pos,coord,view,type
    order = ["input_value", "input_pos", "input_coord", "input_view", "input_type", "input_mask",
             "output_value", "output_label", "output_mask"]
    batch = {k: batch[k] for k in order if k in batch}
    return sd, batch, rest


@pytest.fixture(scope="session")
def small_fixture():
    return load_fixture("fixture_small.npz")


@pytest.fixture(scope="session")
def ragged_fixture():
    return load_fixture("fixture_ragged.npz")


@pytest.fixture(scope="session")
def tiny_fixture():
    return load_fixture("fixture_tiny.npz")


def has_gpu():
    return _gpu_ready()
