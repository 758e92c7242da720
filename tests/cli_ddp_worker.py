"""One rank of tests/test_cli_gpu.py::test_cli_two_ranks_sharing_the_gpu (launched through torch.distributed.run): runs the
command line of plankassembly_amd.trainer and leaves what the test compares in <out>/rank<r>.json."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out, argv = sys.argv[1], sys.argv[2:]
    from plankassembly_amd.trainer import Trainer, cli
    import torch
    mod = cli(Trainer, argv)
    torch.cuda.synchronize()
    flat = mod.model.flat_params.detach().float().cpu().contiguous()
    rec = {"rank": int(os.environ.get("RANK", "0")), "log_dir": mod.logger.log_dir, "logged": {k: float(v) for k, v in mod._logged.items()},
           "global_step": int(mod.global_step), "param_sha": hashlib.sha256(flat.numpy().tobytes()).hexdigest(),
           "history": [(int(s), n, float(v)) for s, n, v in mod.logger.history]}
    with open(os.path.join(out, f"rank{rec['rank']}.json"), "w") as f:
        json.dump(rec, f)


if __name__ == "__main__":
    main()
