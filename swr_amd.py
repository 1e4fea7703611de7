"""Alias so that `import swr_amd` gives the hyphen-named package `software-renderer_amd`."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("software-renderer_amd")
