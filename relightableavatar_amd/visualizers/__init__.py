from .base_visualizer import Output, Visualizer  # noqa: F401
