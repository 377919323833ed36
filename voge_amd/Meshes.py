"""Gaussian parameter containers with the interface of VoGE/Meshes.py:5-57: calling one
returns (verts, sigmas, radians)."""
import torch
import torch.nn as nn


class GaussianMeshesNaive:
    """Plain references to tensors (Meshes.py:5-27)."""

    def __init__(self, verts, sigmas, radians=None):
        self.verts, self.sigmas, self.radians = verts, sigmas, radians

    def to(self, device):
        self.verts = self.verts.to(device)
        self.sigmas = self.sigmas.to(device)
        if self.radians is not None:
            self.radians = self.radians.to(device)
        return self

    def __call__(self):
        return self.verts, self.sigmas, self.radians

    def __getitem__(self, item):
        rad = None if self.radians is None else self.radians[item]
        return GaussianMeshesNaive(self.verts[item], self.sigmas[item], rad)


class GaussianMeshes(nn.Module):
    """Parameters with per-argument requires_grad switches (Meshes.py:30-54)."""

    def __init__(self, verts, sigmas, radians=None, gradianted_args=None):
        super().__init__()
        flags = [True, True, True] if gradianted_args is None else list(gradianted_args)
        self.verts = nn.Parameter(verts, requires_grad=flags[0])
        self.sigmas = nn.Parameter(sigmas, requires_grad=flags[1])
        if radians is None:
            self.radians = None
            flags[2] = False
        else:
            self.radians = nn.Parameter(radians, requires_grad=flags[2])
        self.gradianted_args = flags

    def grad_parameters(self):
        params = (self.verts, self.sigmas, self.radians)
        return tuple(p for p, f in zip(params, self.gradianted_args) if f)

    def forward(self):
        return self.verts, self.sigmas, self.radians


DeformedGaussianMeshes = GaussianMeshes
