"""Test-only stand-in for the subset of PyG the reference hot path touches (see README.md)."""
from . import nn, utils, data, loader, typing  # noqa: F401
