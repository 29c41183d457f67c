"""`TransformerConv` -- drop-in for the reference's modified PyG operator
(code/transformer_conv.py:13-212): out_i = lin_skip(x_i) + sum_j softmax_i(w_ji) lin_value(x_j).

When `edge_weights` is given they REPLACE the q.k attention scores (code/transformer_conv.py:
198-200), which is the only way the reference ever calls it (legacy `Raindrop` v1,
code/models_rd.py:158-160).  The returned alpha is POST-softmax (:201-202).
"""
import torch
from torch import nn
from torch.nn import Linear

from . import _lib, ops


class TransformerConv(nn.Module):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, beta=False, dropout=0.,
                 edge_dim=None, bias=True, root_weight=True, **kwargs):
        super().__init__()
        self.aggr = kwargs.get("aggr", "add")
        self.node_dim = 0
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.heads = heads
        self.beta = beta and root_weight
        self.root_weight = root_weight
        self.concat = concat
        self.dropout = dropout
        self.edge_dim = edge_dim
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.lin_key = Linear(in_channels[0], heads * out_channels)
        self.lin_query = Linear(in_channels[1], heads * out_channels)
        self.lin_value = Linear(in_channels[0], heads * out_channels)
        if edge_dim is not None:
            self.lin_edge = Linear(edge_dim, heads * out_channels, bias=False)
        else:
            self.lin_edge = self.register_parameter('lin_edge', None)
        skip_out = heads * out_channels if concat else out_channels
        self.lin_skip = Linear(in_channels[1], skip_out, bias=bias)
        if self.beta:
            self.lin_beta = Linear(3 * skip_out, 1, bias=False)
        else:
            self.lin_beta = self.register_parameter('lin_beta', None)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_key.reset_parameters()
        self.lin_query.reset_parameters()
        self.lin_value.reset_parameters()
        if self.edge_dim:
            self.lin_edge.reset_parameters()
        self.lin_skip.reset_parameters()
        if self.beta:
            self.lin_beta.reset_parameters()

    def forward(self, x, edge_index, edge_weights=None, edge_attr=None, return_attention_weights=None):
        if isinstance(x, (tuple, list)):
            x = x[1]
        if edge_weights is None or self.heads != 1 or self.lin_beta is not None or edge_attr is not None \
                or self.dropout != 0.:
            raise _lib.RaindropHipError(
                "RD_EUNSUPPORTED: TransformerConv is built for the reference's only use: heads=1, "
                "edge_weights given (they replace q.k scores), no edge_attr/beta/dropout")
        # x [N, C] (the reference's call) or [B, N, C]: B feature matrices on the same graph in one batched product
        batched = x.dim() == 3
        n = x.shape[-2]
        gamma_e, _ = ops.edge_softmax_list(edge_index, edge_weights, n, norm_row=1)
        # dense coefficient matrix gamma[j, i]; duplicate edges accumulate in edge order (== scatter-add of messages), on device
        gamma = ops.edge_gamma_dense(edge_index, gamma_e, n)
        x2 = x.reshape(-1, x.shape[-1])
        v = ops.linear(x2, self.lin_value.weight, self.lin_value.bias, act=0)
        skip = ops.linear(x2, self.lin_skip.weight, self.lin_skip.bias, act=0) if self.root_weight else None
        if batched:
            B = x.shape[0]
            out = ops.aggregate_batched(gamma, v.view(B, n, -1), None if skip is None else skip.view(B, n, -1))
        else:
            out = ops.aggregate(gamma, v, skip)
        if isinstance(return_attention_weights, bool):
            return out, (edge_index, gamma_e.unsqueeze(-1))
        return out

    def __repr__(self):
        return '{}({}, {}, heads={})'.format(self.__class__.__name__, self.in_channels,
                                             self.out_channels, self.heads)
