"""Minimal gym.spaces stand-in (the agents only read .shape/.low/.high). Test infrastructure."""
from . import spaces  # noqa: F401
