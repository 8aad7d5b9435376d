from .make_renderer import make_renderer  # noqa: F401
