"""Host-side mirror of VoGE/Aggregation.py (aggregation :82-107, merge_final :111-141,
expend_sigma :144-175, inverse_cumsum :7-8).  The K x K occlusion integral and the attribute
merge run as fused HIP kernels (voge_amd.ops); nothing here builds a [pixels,K,K] tensor."""
import torch

from . import ops


def inverse_cumsum(x, dim):
    return x + torch.sum(x, dim=dim, keepdim=True) - torch.cumsum(x, dim=dim)


def aggregation(sel_idx: torch.Tensor, sel_act: torch.Tensor, sel_len: torch.Tensor, sel_dsd: torch.Tensor,
                occupation_weight: float = 1.):
    """[..., K] hit lists -> (weight [..., K] f32, sel_idx (same tensor), valid_num [...] int64,
    sel_len (same tensor)), as Aggregation.py:82-107."""
    weight, valid_num = ops.composite(sel_idx, sel_act, sel_len, sel_dsd, occupation_weight)
    return weight, sel_idx, valid_num, sel_len


def merge_final(vert_attr: torch.Tensor, weight: torch.Tensor, vert_assign: torch.Tensor, valid_num: torch.Tensor):
    """out[..., c] = sum over the first valid_num slots of vert_attr[idx] * weight; vert_assign is
    updated in place (-1 -> 0) like Aggregation.py:131."""
    assert vert_attr.dim() == 2
    return ops.merge(vert_attr, weight, vert_assign, valid_num)


def expend_sigma(sigma, rotation_matrix=None):
    """(N,) -> s*R, (N,3) -> diag(s)-scaled R rows, (N,3,3) passthrough; R defaults to identity
    (Aggregation.py:144-175)."""
    if sigma.dim() == 3:
        if tuple(sigma.shape[1:]) == (3, 3):
            return sigma
        raise Exception('Got unexpected sigma, which has shape: ' + str(sigma.shape))
    if rotation_matrix is None:
        rotation_matrix = torch.eye(3, device=sigma.device).unsqueeze(0)
    rotation_matrix = rotation_matrix[..., :3, :3]
    if rotation_matrix.dim() == 2:
        rotation_matrix = rotation_matrix.unsqueeze(0)
    if sigma.dim() == 1:
        return sigma[:, None, None] * rotation_matrix
    if sigma.dim() == 2:
        return sigma[:, :, None] * rotation_matrix
    raise Exception('Got unexpected sigma, which has shape: ' + str(sigma.shape))
