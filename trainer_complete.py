"""`python trainer_complete.py fit --config configs/train_complete.yaml` -- same command line as the
reference's trainer_complete.py:132-133 (LightningCLI(Trainer)), running the MI355X hot path."""
from plankassembly_amd.trainer import Trainer, cli

if __name__ == "__main__":
    cli(Trainer)
