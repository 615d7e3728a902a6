"""Observability shell (the reference writes TensorBoard scalars, src/callbacks/loggers/base_logger.py:40-48);
tensorboard is not in this image, so the scalars go to a JSON-lines file with the same per-epoch content."""
import json
from pathlib import Path


class AcdcVSRLogger:
    def __init__(self, log_dir, net=None, dummy_input=None, **_):
        self.path = Path(log_dir)
        self.path.mkdir(parents=True, exist_ok=True)
        self.f = open(self.path / 'scalars.jsonl', 'a')

    def write(self, epoch, train_log, train_batch, train_outputs, valid_log, valid_batch, valid_outputs):
        self.f.write(json.dumps({'epoch': epoch, 'train': train_log, 'valid': valid_log}) + '\n')
        self.f.flush()

    def close(self):
        self.f.close()


BaseLogger = AcdcVSRLogger
