"""Small helpers with the names of VoGE/Utils.py that the demos import."""
import torch


def rotation_theta(theta, device_=None):
    """In-plane rotation matrices [[cos,-sin,0],[sin,cos,0],[0,0,1]] for angle(s) theta
    (VoGE/Utils.py:336-359) -> [n,3,3]."""
    if isinstance(theta, float):
        theta = torch.full((1,), theta, device=device_ or 'cpu')
    theta = theta.reshape(-1).to(device_ or theta.device)
    c, s = torch.cos(theta), torch.sin(theta)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack([c, -s, z, s, c, z, z, z, o], dim=1).view(-1, 3, 3)


def eye_like(tensor: torch.Tensor):
    """Identity matrices broadcast to tensor's shape [..., n, n] (VoGE/Utils.py:9-10)."""
    n = tensor.shape[-1]
    return torch.eye(n, device=tensor.device, dtype=tensor.dtype).expand(tensor.shape[:-2] + (n, n))
