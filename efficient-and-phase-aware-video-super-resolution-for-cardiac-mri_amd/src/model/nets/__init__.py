from .refine_net import RefineNet   # noqa: F401
