"""`torch_geometric.utils.softmax` as published (scatter-max, exp, scatter-sum, +1e-16)."""
import torch


def softmax(src, index, ptr=None, num_nodes=None, dim=0):
    assert ptr is None and dim == 0
    n = int(index.max()) + 1 if num_nodes is None else int(num_nodes)
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    src_max = torch.full(shape, float('-inf'), dtype=src.dtype, device=src.device)
    src_max = src_max.scatter_reduce(0, idx, src.detach(), 'amax', include_self=True)
    out = (src - src_max.gather(0, idx)).exp()
    out_sum = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add(0, idx, out)
    return out / (out_sum.gather(0, idx) + 1e-16)
