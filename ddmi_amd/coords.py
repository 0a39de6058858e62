"""Coordinate helpers - mirror of utils/general_utils.py:16-35, 204-206 (host-side plumbing, torch like the reference)."""
import random

import numpy as np
import torch


def random_seed(seed):
    """utils/general_utils.py:16-23."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def convert_to_coord_format_2d(b, h, w, device="cpu", integer_values=False, hstart=-1, hend=1, wstart=-1, wend=1):
    """utils/general_utils.py:27-35 (including its square-grid-only repeat pattern)."""
    if integer_values:
        x_channel = torch.arange(w, dtype=torch.float, device=device).view(1, 1, 1, -1).repeat(b, 1, w, 1)
        y_channel = torch.arange(h, dtype=torch.float, device=device).view(1, 1, -1, 1).repeat(b, 1, 1, h)
    else:
        x_channel = torch.linspace(wstart, wend, w, device=device).view(1, 1, 1, -1).repeat(b, 1, w, 1)
        y_channel = torch.linspace(hstart, hend, h, device=device).view(1, 1, -1, 1).repeat(b, 1, 1, h)
    return torch.cat((x_channel, y_channel), dim=1)


def generation_coords(res, device="cpu"):
    """The grid generate()/eval() build (tools/ldm/image.py:227-233): +-(R-1)/R."""
    e = (res - 1) / res
    return convert_to_coord_format_2d(1, res, res, device=device, hstart=-e, hend=e, wstart=-e, wend=e)


def get_scale_injection(current_res, anchor_res=256):
    """utils/general_utils.py:204-206."""
    return anchor_res / current_res
