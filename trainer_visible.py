"""`python trainer_visible.py fit --config configs/train_visible.yaml` (reference trainer_visible.py:26-27)."""
from plankassembly_amd.trainer import VisibleTrainer, cli

if __name__ == "__main__":
    cli(VisibleTrainer)
