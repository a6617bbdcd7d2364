from . import gloria_loss  # noqa: F401
