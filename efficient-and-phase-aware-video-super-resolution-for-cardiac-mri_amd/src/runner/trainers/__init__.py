from .base_trainer import BaseTrainer                                # noqa: F401
from .acdc_vsr_refinenet_trainer import AcdcVSRRefineNetTrainer      # noqa: F401
