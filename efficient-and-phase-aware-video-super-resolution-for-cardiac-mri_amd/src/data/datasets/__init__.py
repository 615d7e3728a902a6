from .acdc_vsr_refinenet_dataset import AcdcVSRRefineNetDataset, SyntheticCineDataset   # noqa: F401
