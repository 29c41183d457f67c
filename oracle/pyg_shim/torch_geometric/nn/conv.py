"""Minimal `MessagePassing` with the collect/message/aggregate contract the reference relies on."""
import inspect

import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2, **kwargs):
        super().__init__()
        assert flow == 'source_to_target'
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim
        self._msg_params = list(inspect.signature(self.message).parameters)

    def propagate(self, edge_index, size=None, **kwargs):
        assert isinstance(edge_index, torch.Tensor) and edge_index.dim() == 2
        src_idx, dst_idx = edge_index[0], edge_index[1]   # j = source, i = target
        collected = {}
        dim_size = None
        for name in self._msg_params:
            if name.endswith('_i') or name.endswith('_j'):
                data = kwargs.get(name[:-2])
                if data is None:
                    collected[name] = None
                    continue
                pair = data if isinstance(data, (tuple, list)) else (data, data)
                if dim_size is None:
                    dim_size = pair[1].size(self.node_dim)          # number of target nodes
                if name.endswith('_i'):
                    collected[name] = pair[1].index_select(self.node_dim, dst_idx)
                else:
                    collected[name] = pair[0].index_select(self.node_dim, src_idx)
            elif name == 'index':
                collected[name] = dst_idx
            elif name == 'ptr':
                collected[name] = None
            elif name in ('size_i', 'dim_size'):
                collected[name] = None  # filled below
            else:
                collected[name] = kwargs.get(name)
        if dim_size is None:
            dim_size = int(edge_index.max()) + 1
        if 'size_i' in collected:
            collected['size_i'] = dim_size
        out = self.message(**collected)
        return self.aggregate(out, dst_idx, ptr=None, dim_size=dim_size)

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        from torch_scatter import scatter
        return scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce=self.aggr)
