from . import image  # noqa: F401
