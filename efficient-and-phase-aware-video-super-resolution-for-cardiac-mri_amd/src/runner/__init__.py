from . import trainers   # noqa: F401
