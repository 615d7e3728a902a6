"""Testing shell around the hot path (reference src/runner/predictors/base_predictor.py:6-136): constructor contract,
device placement, log bookkeeping and ``load`` (reads the 'net' entry of a trainer checkpoint).  Host logic only."""
import torch


class BasePredictor:
    def __init__(self, device, test_dataloader, net, loss_fns, loss_weights, metric_fns):
        self.device = device
        self.test_dataloader = test_dataloader
        self.net = net.to(device)
        self.loss_fns = [fn.to(device) for fn in loss_fns]
        self.loss_weights = torch.tensor(loss_weights, dtype=torch.float, device=device)
        self.metric_fns = [fn.to(device) for fn in metric_fns]

    def predict(self):
        raise NotImplementedError

    def _allocate_data(self, batch):
        if isinstance(batch, dict):
            return {k: self._allocate_data(v) for k, v in batch.items()}
        if isinstance(batch, list):
            return [self._allocate_data(v) for v in batch]
        if isinstance(batch, tuple):
            return tuple(self._allocate_data(v) for v in batch)
        if isinstance(batch, torch.Tensor):
            return batch.to(self.device)
        return batch

    def _init_log(self):
        log = {'Loss': 0}
        for fn in list(self.loss_fns) + list(self.metric_fns):
            log[type(fn).__name__] = 0
        return log

    def load(self, path):
        checkpoint = torch.load(path, map_location=self.device, weights_only=False)
        self.net.load_state_dict(checkpoint['net'])
