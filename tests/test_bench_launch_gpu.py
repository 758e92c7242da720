"""`bench.py` under the DRIVER's own launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`).

N = 1 goes through RCCL (world size 1).  N = 2 needs two devices for RCCL, so on the one-GPU test box the two ranks share
`cuda:0` and exchange through gloo (`PLANK_BENCH_BACKEND=gloo`, a switch that exists for this test only): what runs is every
line of the N > 1 path of bench.py - per-rank model + GradSync hooks + FusedAdam with grad_scale 1/N, the barrier +
synchronize bracket, the max-over-ranks reduction of the clock and the whole-job `value` - not a scaling number."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(n, extra_env, steps=4, warmup=2, flags=("--no-cpu", "--no-decode", "--no-kernels", "--long-steps", "6")):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup),
           *flags]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                # rank 0 prints ONE JSON line, the other ranks none
    return json.loads(lines[0]), r.stderr


def _check_line(line, n, steps, warmup):
    assert line["n_gpus"] == n and line["steps"] == steps and line["warmup"] == warmup
    assert line["unit"] == "samples/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 16 * n and line["config"]["parallelism"] == f"dp{n}"
    # value = samples of ALL ranks / the slowest rank's time
    assert abs(line["value"] - 16 * n / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert line["final_loss"] == line["final_loss"] and 0.0 < line["final_loss"] < 20.0


def test_driver_launch_line_one_rank_rccl():
    line, _ = _launch(1, {}, flags=("--no-cpu", "--no-decode", "--long-steps", "6"))
    _check_line(line, 1, 4, 2)
    assert line["roofline"]["bound"] == "mfma" and 0.0 < line["roofline"]["frac"] < 1.0
    # the driver checks that RCCL saw N ranks from these fields; the parity-meeting (f32) training figure rides in the same line
    assert line["rccl_ranks"]["world_size"] == 1 and line["rccl_ranks"]["backend"] == "nccl" and line["rccl_ranks"]["process_group"]
    assert line["rccl_ranks"]["ms_per_step_by_rank"]["train"]["ranks"] == 1
    assert line["train"]["parity_meeting"]["dtype"] in ("f32", "x3") and 0 < line["train"]["f32"]["value"] < line["train"]["bf16"]["value"]
    assert 0 < line["train"]["x3"]["value"] < line["train"]["bf16"]["value"]
    assert line["train"]["parity_meeting"]["value"] == max(line["train"]["f32"]["value"], line["train"]["x3"]["value"])
    assert line["steady_state"]["steps"] == 6 and "f32" in line["metric"]


def test_driver_launch_line_two_ranks_sharing_the_gpu():
    line, err = _launch(2, {"PLANK_BENCH_BACKEND": "gloo"}, flags=("--long-steps", "6"))      # the driver's flags: decode + census legs included
    _check_line(line, 2, 4, 2)
    by_rank = line["rccl_ranks"]["ms_per_step_by_rank"]
    assert line["rccl_ranks"]["world_size"] == 2 and by_rank["train"]["ranks"] == 2 and by_rank["train_f32"]["ranks"] == 2
    assert by_rank["train"]["max"] <= line["ms_per_step"] * 1.0001 and by_rank["train"]["min"] <= by_rank["train"]["max"]
    assert line.get("cpu_baseline") is None                      # the CPU leg is rank 0 at N = 1 only
    assert line["decode"]["bf16"]["value"] > 0 and line["decode"]["bf16"]["token_exact"] is None
    assert "rank 0/2" in err                                 # (rank 1 logs nothing: only rank 0 reports)
    # both ranks draw the same batches and start from broadcast parameters: after the exchange the mean gradient equals each
    # rank's own, so the two-rank loss trajectory is the one-rank trajectory (same backend on both sides: bench.py runs 30
    # set-up steps in front of the warm-up under RCCL only, which would move the one-rank run 30 optimizer steps ahead)
    one, _ = _launch(1, {"PLANK_BENCH_BACKEND": "gloo"})
    assert abs(line["final_loss"] - one["final_loss"]) <= 2e-3 * abs(one["final_loss"])


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two devices: RCCL refuses two ranks on one")
def test_driver_launch_line_two_ranks_over_rccl():
    """the driver's N = 2 line over the product transport (backend nccl = RCCL, one device per rank)."""
    line, _ = _launch(2, {}, flags=("--no-cpu", "--no-decode", "--no-kernels", "--long-steps", "6"))
    _check_line(line, 2, 4, 2)
    assert line["rccl_ranks"]["world_size"] == 2 and line["rccl_ranks"]["backend"] == "nccl" and line["rccl_ranks"]["process_group"]
    assert line["rccl_ranks"]["ms_per_step_by_rank"]["train"]["ranks"] == 2
