"""Drop-in counterpart of the reference's ``src`` package for the RefineNet hot path only
(``src.model.nets.RefineNet``, ``src.model.losses``, ``src.model.metrics``, the RefineNet trainer and predictor,
``src.main``).  The reference's other nets, datasets, loggers and predictors are out of scope."""
from . import model, runner, data, callbacks, utils   # noqa: F401
