import torch.nn.functional as _F


def pad(img, padding, fill=0, padding_mode="constant"):
    """symbol needed by /root/reference/models/custom_functions.py:8 (dead code path)."""
    return _F.pad(img, padding, mode=padding_mode, value=fill)
