"""Inverse rendering helpers with the interface of VoGE/Sampler.py: scatter image features back
to the Gaussians through the fragments (sample_features :5-29, scatter_max_weight :32-42).

sample_voge is the transpose of merge_final, so it runs on the merge kernels (voge_amd.ops
_ScatterAttr); scatter_max has its own small kernel."""
import torch

from . import ops


def _n_vert(frag, n_vert):
    if n_vert is None:
        n_vert = frag.num_vertices if hasattr(frag, 'num_vertices') else int(frag.vert_index.max()) + 1
    return int(n_vert)


def sample_features(frag, image, n_vert=None):
    """image [B,H,W,C] -> (vert_feature [n,C] = sum over pixels/slots of weight * image,
    vert_sum_weight [n]); equals the dense formulation quoted in Sampler.py:7-11."""
    n_vert = _n_vert(frag, n_vert)
    assert image.device == frag.vert_index.device
    assert frag.vert_weight.shape[:3] == image.shape[:3]
    aug = torch.cat([image, torch.ones_like(image[..., :1])], dim=-1).contiguous()
    out = ops.scatter_attr(aug, frag.vert_weight, frag.vert_index, frag.valid_num, n_vert)
    C = image.shape[-1]
    return out[:, :C], out[:, C]


def scatter_max_weight(frag, n_vert=None):
    """Per-Gaussian maximum of the compositing weight over all pixels (non-differentiable)."""
    n_vert = _n_vert(frag, n_vert)
    # slots beyond valid_num are empty: mask them so that a rewritten index (-1 -> 0) cannot count
    K = frag.vert_weight.shape[-1]
    mask = torch.arange(K, device=frag.vert_weight.device) < frag.valid_num.unsqueeze(-1)
    idx = torch.where(mask, frag.vert_index, torch.full_like(frag.vert_index, -1))
    return ops.scatter_max(frag.vert_weight, idx.contiguous(), n_vert)
