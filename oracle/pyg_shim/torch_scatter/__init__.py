"""TEST INFRASTRUCTURE ONLY -- `torch_scatter.scatter(reduce='add')` as `index_add_`
(call site `Ob_propagation.py:227`); `gather_csr`/`segment_csr` are imported but never called."""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    assert reduce in ('add', 'sum') and out is None
    dim = dim % src.dim()
    n = int(index.max()) + 1 if dim_size is None else int(dim_size)
    shape = list(src.shape)
    shape[dim] = n
    return torch.zeros(shape, dtype=src.dtype, device=src.device).index_add_(dim, index, src)


def gather_csr(*a, **k):
    raise NotImplementedError


def segment_csr(*a, **k):
    raise NotImplementedError
