"""`python trainer_sideface.py fit --config configs/train_sideface.yaml` (reference trainer_sideface.py:87-88)."""
from plankassembly_amd.trainer import SidefaceTrainer, cli

if __name__ == "__main__":
    cli(SidefaceTrainer)
