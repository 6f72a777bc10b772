"""Tensor contracts of the reference's network/operator.py:4-23, kept so callers and tests read the same."""
import torch


def normalize_coords(coords: torch.Tensor, h, w):
    """pixel (x,y) -> [-1,1] with the half-pixel centre convention: ((c+0.5)/size - 0.5)*2 (operator.py:4-17)."""
    size = torch.tensor([w, h], dtype=coords.dtype, device=coords.device)
    return ((coords + 0.5) / size - 0.5) * 2


def pose_apply_th(poses, pts):
    """poses [b,3,4], pts [b,n,3] -> R pts + t (operator.py:19-20)."""
    return pts @ poses[:, :, :3].transpose(1, 2) + poses[:, :, 3:].transpose(1, 2)


def generate_coords(h, w, device):
    """[h,w,2] integer (x,y) coordinates (operator.py:22-24)."""
    ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack([xs, ys], -1)
