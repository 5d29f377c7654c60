from .application import Application  # noqa: F401
