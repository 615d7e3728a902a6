from .base_predictor import BasePredictor                              # noqa: F401
from .acdc_vsr_refinenet_predictor import AcdcVSRRefineNetPredictor    # noqa: F401
