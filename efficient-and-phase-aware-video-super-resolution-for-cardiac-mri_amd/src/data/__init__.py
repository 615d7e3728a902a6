from . import datasets, dataloader   # noqa: F401
