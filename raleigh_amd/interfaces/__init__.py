from .partial_hevp import partial_hevp  # noqa: F401
