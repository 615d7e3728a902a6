from . import loggers, monitor   # noqa: F401
