"""isaacgym.torch_utils.to_torch — the only symbol the agents use (amp_agent.py:426,583)."""
import numpy as np  # noqa: F401  (the reference relies on `from isaacgym.torch_utils import *` exporting np)
import torch


def to_torch(x, dtype=torch.float, device='cuda:0', requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)
