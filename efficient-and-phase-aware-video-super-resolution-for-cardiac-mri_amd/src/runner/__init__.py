from . import trainers, predictors   # noqa: F401
