"""`TransformerConv` -- drop-in for the reference's modified PyG operator
(code/transformer_conv.py:13-212): out_i = lin_skip(x_i) + sum_j softmax_i(w_ji) lin_value(x_j).

When `edge_weights` is given they REPLACE the q.k attention scores (code/transformer_conv.py:
198-200), which is the only way the reference ever calls it (legacy `Raindrop` v1,
code/models_rd.py:158-160).  The returned alpha is POST-softmax (:201-202).  Without `edge_weights` the operator is PyG's
TransformerConv as this fork has it -- heads, q.k scores, `lin_edge(edge_attr)` on the key (not on the value, :205), coefficient
dropout, concat / mean, beta gate: `_forward_general` (round 6).
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
        if edge_weights is None:
            return self._forward_general(x, edge_index, edge_attr, return_attention_weights)
        if self.heads != 1 or self.lin_beta is not None or edge_attr is not None or (self.dropout != 0. and self.training):
            # given edge weights REPLACE the scores as one [E,1] column (code/transformer_conv.py:198-200): the reference's own
            # `out *= alpha.view(-1, heads, 1)` only works for heads = 1 then, which is also the only way it is ever called
            raise _lib.RaindropHipError(
                "RD_EUNSUPPORTED: TransformerConv with edge_weights is built for the reference's use of it: heads=1, "
                "no edge_attr / beta gate / coefficient dropout (without edge_weights every form is built)")
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

    def _forward_general(self, x, edge_index, edge_attr, return_attention_weights):
        """code/transformer_conv.py:139-207 with `edge_weights=None` (round 6): q.k scores per edge and head (+ lin_edge(edge_attr)
        on the key), softmax per target, coefficient dropout in training mode, source-valued aggregate, concat or mean over the
        heads, root weight with or without the beta gate.  The four projections are rd_linear_fwd, the graph part
        rd_edge_attention_fwd / _bwd; the head mean and the gate's blend are elementwise torch (a few KB)."""
        if x.dim() != 2:
            raise _lib.RaindropHipError("RD_EUNSUPPORTED: the general TransformerConv form takes one graph, x [N, C]")
        H, C = self.heads, self.out_channels
        q = ops.linear(x, self.lin_query.weight, self.lin_query.bias)
        k = ops.linear(x, self.lin_key.weight, self.lin_key.bias)
        v = ops.linear(x, self.lin_value.weight, self.lin_value.bias)
        ef = None
        if self.lin_edge is not None:
            if edge_attr is None:
                raise ValueError("edge_attr is required (the operator was built with edge_dim)")       # the reference asserts, :192
            ef = ops.linear(edge_attr.float(), self.lin_edge.weight, None)
        p_drop, seed = (float(self.dropout) if self.training else 0.0), 0
        if p_drop > 0.0:
            self._drop_calls = getattr(self, "_drop_calls", 0) + 1
            seed = (torch.initial_seed() * 1000003 + 7919 * self._drop_calls + ops.rank_seed_offset()) & 0x7FFFFFFFFFFFFFFF
        out, alpha = ops.edge_attention(q, k, v, ef, edge_index, H, C, p_drop, seed)
        if not self.concat:
            out = out.view(-1, H, C).mean(dim=1)
        if self.root_weight:
            x_r = ops.linear(x, self.lin_skip.weight, self.lin_skip.bias)
            if self.lin_beta is not None:
                beta = torch.sigmoid(ops.linear(torch.cat([out, x_r, out - x_r], dim=-1), self.lin_beta.weight, None))
                out = beta * x_r + (1 - beta) * out
            else:
                out = out + x_r
        if isinstance(return_attention_weights, bool):
            return out, (edge_index, alpha)
        return out

    def __repr__(self):
        return '{}({}, {}, heads={})'.format(self.__class__.__name__, self.in_channels,
                                             self.out_channels, self.heads)
