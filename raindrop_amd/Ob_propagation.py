"""`Observation_progation` -- drop-in for the reference operator class (code/Ob_propagation.py:17-233).

Same constructor, parameters (state_dict keys/shapes, including the ones the shipped forward
never touches), initialisation and `forward` signature; the arithmetic runs on the HIP kernels.
Unlike the reference the module keeps no per-call state on `self` (it is re-entrant).
"""
import math

import torch
from torch import nn
from torch.nn import Linear, Parameter, init

from . import _lib, ops


def glorot(t):
    """torch_geometric.nn.inits.glorot (used at code/Ob_propagation.py:85,90-91)."""
    if t is not None:
        a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
        t.data.uniform_(-a, a)


class Observation_progation(nn.Module):
    def __init__(self, in_channels, out_channels, n_nodes, ob_dim, heads=1, concat=True, beta=False,
                 dropout=0., edge_dim=None, bias=True, root_weight=True, **kwargs):
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
        # registration order and names follow code/Ob_propagation.py:40-69
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
        self.weight = Parameter(torch.Tensor(in_channels[1], heads * out_channels))
        self.bias = Parameter(torch.Tensor(heads * out_channels))
        self.n_nodes = n_nodes
        self.nodewise_weights = Parameter(torch.Tensor(self.n_nodes, heads * out_channels))
        self.increase_dim = Linear(in_channels[1], heads * out_channels * 8)
        self.map_weights = Parameter(torch.Tensor(self.n_nodes, heads * 16))
        self.ob_dim = ob_dim
        self.reset_parameters()

    def reset_parameters(self):
        """code/Ob_propagation.py:75-92."""
        self.lin_key.reset_parameters()
        self.lin_query.reset_parameters()
        self.lin_value.reset_parameters()
        if self.edge_dim:
            self.lin_edge.reset_parameters()
        self.lin_skip.reset_parameters()
        if self.beta:
            self.lin_beta.reset_parameters()
        glorot(self.weight)
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)
        glorot(self.nodewise_weights)
        glorot(self.map_weights)
        self.increase_dim.reset_parameters()

    def forward(self, x, p_t, edge_index, edge_weights=None, use_beta=False, edge_attr=None,
                return_attention_weights=None):
        """x [N,K]; edge_index int64 [2,E] (row 0 source j, row 1 target i); edge_weights [E].
        Default branch (code/Ob_propagation.py:186-200,209-228): message = relu(lin_value(x_i)) *
        softmax_i(edge_weights); scatter-add onto the target -> relu(lin_value(x)) * ssum[i].
        Returns out [N, heads*out_channels] or (out, (edge_index, alpha)) with alpha the
        PRE-softmax weights [E,1] (:193)."""
        if isinstance(x, (tuple, list)):
            x = x[1]
        # code/Ob_propagation.py:196: F.dropout of the edge coefficients AFTER the softmax (the shipped model builds the operator with
        # dropout = 0., models_rd.py:243-247).  Default branch: round 6 (the mask is a function of torch's seed and a per-module call
        # counter, like the model's own dropout); the use_beta branch with coefficient dropout is not built.
        p_edge = float(self.dropout) if self.training else 0.0
        if p_edge > 0.0 and use_beta:
            raise _lib.RaindropHipError("RD_EUNSUPPORTED: Observation_progation(dropout > 0) with use_beta=True in training mode is not built")
        if use_beta:
            return self._forward_beta(x, p_t, edge_index, edge_weights, return_attention_weights)
        if edge_weights is None:
            raise ValueError("edge_weights is required (the reference raises UnboundLocalError at "
                             "code/Ob_propagation.py:193 without it)")
        if self.heads != 1:
            raise _lib.RaindropHipError("RD_EUNSUPPORTED: heads != 1")
        n = x.shape[0]
        seed = 0
        if p_edge > 0.0:
            self._drop_calls = getattr(self, "_drop_calls", 0) + 1
            seed = (torch.initial_seed() * 1000003 + 7919 * self._drop_calls + ops.rank_seed_offset()) & 0x7FFFFFFFFFFFFFFF
        _, ssum = ops.edge_softmax_list(edge_index, edge_weights, n, norm_row=1, p_drop=p_edge, seed=seed)
        v = ops.linear(x, self.lin_value.weight, self.lin_value.bias, act=1)
        out = v * ssum[:, None]
        out = out.view(-1, self.heads * self.out_channels) if self.concat else out
        if isinstance(return_attention_weights, bool):
            return out, (edge_index, edge_weights.unsqueeze(-1))
        return out

    def _forward_beta(self, x, p_t, edge_index, edge_weights, return_attention_weights):
        """use_beta branch (code/Ob_propagation.py:161-185,190-191,195,200,207-208,227): per-time-step edge scores
        beta * w, pruning to the top half of the edges, per-channel softmax over the edges of a SOURCE node, aggregation
        of the TARGETS' values.  lin_value and increase_dim run once per node (rd_linear_fwd); the graph part is
        rd_graph_beta_fwd.  Returns out [N,K] or (out, (edge_index' [2,E/2], alpha [E/2])) like the reference."""
        if edge_weights is None:
            raise ValueError("edge_weights is required on the use_beta branch")
        if self.heads != 1:
            raise _lib.RaindropHipError("RD_EUNSUPPORTED: heads != 1")
        V = ops.linear(x, self.lin_value.weight, self.lin_value.bias, act=1)
        H = ops.linear(x, self.increase_dim.weight, self.increase_dim.bias, exact=True)     # feeds the top-K pruning: exact fp32 in every mode
        out, ei2, alpha = ops.graph_beta(V.unsqueeze(0), H.unsqueeze(0), self.map_weights, p_t.unsqueeze(0).float(), edge_index,
                                         edge_weights.reshape(1, -1).float(), self.ob_dim)
        out = out[0]
        out = out.view(-1, self.heads * self.out_channels) if self.concat else out
        if isinstance(return_attention_weights, bool):
            return out, (ei2[0], alpha[0])
        return out

    def __repr__(self):
        return '{}({}, {}, heads={})'.format(self.__class__.__name__, self.in_channels,
                                             self.out_channels, self.heads)
