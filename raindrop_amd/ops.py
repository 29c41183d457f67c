"""torch.autograd seams over the C-ABI (libraindrop_hip.so).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the autograd graph.
Every numerical step of the hot path is a HIP kernel reached through `raindrop_amd._lib.call`.
Tensors handed to the library must be fp32 (int64 / bool where stated), contiguous, and live on
a ROCm device; anything else raises -- there is deliberately no eager fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(*tensors, dtype=torch.float32):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.RaindropHipError(
                "Raindrop HIP ops need device tensors (got %s); no CPU fallback exists" % t.device)
        if t.dtype != dtype:
            raise _lib.RaindropHipError("expected %s tensor, got %s" % (dtype, t.dtype))
        if not t.is_contiguous():
            raise _lib.RaindropHipError("tensor must be contiguous")


def rank_seed_offset():
    """Added to every dropout seed: data-parallel ranks must draw DIFFERENT masks (the masks are functions of
    (seed, site, local element index), so equal seeds would give sample b of every rank the same mask, which is not
    what one process with the global batch does)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank() * 0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF
    return 0


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# integer work
# ------------------------------------------------------------------------------------------------

def graph_build(global_structure):
    """code/models_rd.py:307-311 on device.  Returns (adj [F,F], edge_index int64 [2,E],
    edge_weights [E]); the single host read is the edge count (the reference syncs here too,
    inside torch.nonzero)."""
    gs = global_structure.contiguous()
    _check(gs)
    F = gs.shape[0]
    adj = torch.empty_like(gs)
    ei = torch.empty((2, F * F), dtype=torch.int64, device=gs.device)
    ew = torch.empty((F * F,), dtype=torch.float32, device=gs.device)
    n = torch.zeros((1,), dtype=torch.int32, device=gs.device)
    _lib.call("rd_graph_build", F, _ptr(gs), _ptr(adj), _ptr(ei), _ptr(ew), _ptr(n), _stream())
    E = int(n.item())
    return adj, ei[:, :E], ew[:E]


def edge_softmax_dense(adj):
    """Per-target softmax of the dense adjacency + coefficient sums: (gamma [F,F], ssum [F])."""
    adj = adj.contiguous()
    _check(adj)
    F = adj.shape[0]
    gamma = torch.empty_like(adj)
    ssum = torch.empty((F,), dtype=torch.float32, device=adj.device)
    _lib.call("rd_edge_softmax", F, _ptr(adj), _ptr(gamma), _ptr(ssum), _stream())
    return gamma, ssum


def timescales(max_len, d_pe=16):
    """float64 `max_len ** linspace(0,1,d_pe/2)` cast to fp32 (code/models_rd.py:31,34)."""
    return torch.from_numpy((max_len ** np.linspace(0, 1, d_pe // 2)).astype(np.float32))


# ------------------------------------------------------------------------------------------------
# sensor stage: observation embedding + 2 x Observation_progation + PE concat + padding mask
# ------------------------------------------------------------------------------------------------

def sensor_stage_fwd_raw(src, times, lengths, ts, ssum, R_u, W1, b1, W2, b2, shp, p_drop, seed):
    """rd_sensor_stage_fwd: PE + padding mask + message passing into one [T,B,D] buffer.
    Returns (z, mask, saved)."""
    _check(src, times, ts, ssum, R_u, W1, b1, W2, b2)
    _check(lengths, dtype=torch.int64)
    T, B, F, d = shp.T, shp.B, shp.F, shp.d_ob
    D = F * d + shp.d_pe
    dev = src.device
    z = torch.empty((T, B, D), dtype=torch.float32, device=dev)
    mask = torch.empty((B, T), dtype=torch.bool, device=dev)
    sp = ctypes.byref(shp)
    saved = _workspace(_lib.load().rd_msgpass_saved_bytes(sp), dev)
    _lib.call("rd_sensor_stage_fwd", sp, _ptr(src), _ptr(times), _ptr(lengths), _ptr(ts), _ptr(R_u), _ptr(W1),
              _ptr(b1), _ptr(W2), _ptr(b2), _ptr(ssum), float(p_drop), int(seed), _ptr(z), _ptr(mask), _ptr(saved),
              saved.numel(), _stream())
    return z, mask, saved


def sensor_stage_bwd_raw(src, R_u, W1, W2, ssum, saved, z, dz, shp, p_drop):
    """rd_msgpass_bwd.  Returns (dR_u, dW1, db1, dW2, db2)."""
    K = shp.T * shp.d_ob
    D = shp.F * shp.d_ob + shp.d_pe
    dev = dz.device
    dW1 = torch.empty((K, K), dtype=torch.float32, device=dev)
    dW2 = torch.empty((K, K), dtype=torch.float32, device=dev)
    db1 = torch.empty((K,), dtype=torch.float32, device=dev)
    db2 = torch.empty((K,), dtype=torch.float32, device=dev)
    dRu = torch.empty_like(R_u)
    sp = ctypes.byref(shp)
    ws = _workspace(_lib.load().rd_msgpass_workspace_bytes(sp), dev)
    _lib.call("rd_msgpass_bwd", sp, _ptr(src), _ptr(R_u), _ptr(W1), _ptr(W2), _ptr(ssum), float(p_drop),
              _ptr(saved), saved.numel(), _ptr(z), _ptr(dz), D, _ptr(dW1), _ptr(db1), _ptr(dW2), _ptr(db2),
              _ptr(dRu), _ptr(ws), ws.numel(), _stream())
    return dRu, dW1, db1, dW2, db2


class _SensorStage(torch.autograd.Function):
    """z[T,B,D] = cat(message_passing(src), PE(times));  mask[B,T] = t >= lengths.

    Forward : rd_pe_mask + rd_msgpass_fwd write disjoint column ranges of one buffer.
    Backward: rd_msgpass_bwd (dW1, db1, dW2, db2, dR_u); PE / mask carry no gradient."""

    @staticmethod
    def forward(ctx, src, times, lengths, ts, ssum, R_u, W1, b1, W2, b2, shp, p_drop, seed):
        z, mask, saved = sensor_stage_fwd_raw(src, times, lengths, ts, ssum, R_u, W1, b1, W2, b2, shp, p_drop, seed)
        ctx.shp = shp
        ctx.p_drop = float(p_drop)
        ctx.save_for_backward(src, R_u, W1, W2, ssum, saved, z)
        ctx.mark_non_differentiable(mask)
        return z, mask

    @staticmethod
    def backward(ctx, dz, _dmask):
        src, R_u, W1, W2, ssum, saved, z = ctx.saved_tensors
        dRu, dW1, db1, dW2, db2 = sensor_stage_bwd_raw(src, R_u, W1, W2, ssum, saved, z, dz.contiguous(), ctx.shp,
                                                       ctx.p_drop)
        return None, None, None, None, None, dRu, dW1, db1, dW2, db2, None, None, None


def sensor_stage(src, times, lengths, ts, ssum, R_u, W1, b1, W2, b2, shp, p_drop=0.0, seed=0):
    return _SensorStage.apply(src.contiguous(), times.contiguous(), lengths.contiguous(), ts, ssum,
                              R_u.contiguous(), W1, b1, W2, b2, shp, p_drop, seed)


# ------------------------------------------------------------------------------------------------
# dense layers
# ------------------------------------------------------------------------------------------------

class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) through rd_linear_fwd; backward through rd_linear_bwd_{input,weight}."""

    @staticmethod
    def forward(ctx, x, W, b, act, exact=False):
        _check(x, W, b)
        N, K = W.shape
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        # exact: the product on the fp32 matrix instruction (bitwise an fmaf chain) whatever the process's precision mode
        # (rd_linear_fwd_fp32: a per-call, per-thread override inside the library; round 5 toggled the process-wide mode around the
        # call) -- for values that feed INDEX work (the use_beta branch's edge scores -> top-K pruning): a few MFLOP, and index work is
        # bit-exact by contract
        _lib.call("rd_linear_fwd_fp32" if exact else "rd_linear_fwd", M, N, K, _ptr(x2), K, _ptr(W), _ptr(b), _ptr(y), N, int(act), _stream())
        ctx.act = int(act)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x2, W, y if act else None)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        N, K = W.shape
        M = x2.shape[0]
        dy2 = dy.reshape(M, N)
        if ctx.act:
            dy2 = dy2 * (y > 0)          # ReLU gate (elementwise glue)
        dy2 = dy2.contiguous()
        dev = dy.device
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dev)
            _lib.call("rd_linear_bwd_input", M, N, K, _ptr(dy2), N, _ptr(W), _ptr(dx), K, _stream())
            dx = dx.view(ctx.xshape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.empty((N, K), dtype=torch.float32, device=dev)
            db = torch.empty((N,), dtype=torch.float32, device=dev) if ctx.has_bias else None
            nbytes = _lib.load().rd_linear_bwd_weight_workspace_bytes(M, N, K)
            ws = _workspace(nbytes, dev)
            _lib.call("rd_linear_bwd_weight", M, N, K, _ptr(dy2), N, _ptr(x2), K, _ptr(dW), _ptr(db),
                      _ptr(ws), ws.numel(), _stream())
        return dx, dW, db, None, None


def linear(x, W, b=None, act=0, exact=False):
    return _Linear.apply(x.contiguous(), W.contiguous(), None if b is None else b.contiguous(), act, bool(exact))


# ------------------------------------------------------------------------------------------------
# stand-alone graph operators (PyG operator API: one graph per call)
# ------------------------------------------------------------------------------------------------

def edge_softmax_list(edge_index, edge_weights, n_nodes, norm_row=1, p_drop=0.0, seed=0):
    """PyG softmax over an explicit edge list; returns (gamma_e [E], ssum [n_nodes]).  p_drop > 0: F.dropout of the coefficients
    after the softmax (code/Ob_propagation.py:196), ssum is then the sum of the dropped-and-rescaled coefficients."""
    ei = edge_index.contiguous()
    w = edge_weights.contiguous()
    _check(ei, dtype=torch.int64)
    _check(w)
    E = ei.shape[1]
    gamma = torch.empty((E,), dtype=torch.float32, device=w.device)
    ssum = torch.empty((n_nodes,), dtype=torch.float32, device=w.device)
    if p_drop > 0.0:
        _lib.call("rd_edge_softmax_list_dropout", int(n_nodes), int(E), _ptr(ei), ei.stride(0), int(norm_row), _ptr(w), float(p_drop),
                  int(seed) & 0x7FFFFFFFFFFFFFFF, _ptr(gamma), _ptr(ssum), _stream())
    else:
        _lib.call("rd_edge_softmax_list", int(n_nodes), int(E), _ptr(ei), ei.stride(0), int(norm_row),
                  _ptr(w), _ptr(gamma), _ptr(ssum), _stream())
    return gamma, ssum


class _Aggregate(torch.autograd.Function):
    """out = gamma^T V (+ skip): source-valued aggregate with fixed (non-learned) coefficients."""

    @staticmethod
    def forward(ctx, gamma, V, skip):
        _check(gamma, V, skip)
        N, C = V.shape
        out = torch.empty_like(V)
        _lib.call("rd_aggregate_fwd", N, C, _ptr(gamma), _ptr(V), _ptr(skip), _ptr(out), _stream())
        ctx.save_for_backward(gamma)
        ctx.has_skip = skip is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (gamma,) = ctx.saved_tensors
        dout = dout.contiguous()
        N, C = dout.shape
        dV = torch.empty_like(dout)
        _lib.call("rd_aggregate_bwd", N, C, _ptr(gamma), _ptr(dout), _ptr(dV), _stream())
        return None, dV, (dout if ctx.has_skip else None)


def aggregate(gamma, V, skip=None):
    return _Aggregate.apply(gamma.contiguous(), V.contiguous(), None if skip is None else skip.contiguous())


class _EdgeAttention(torch.autograd.Function):
    """The general message / aggregate of TransformerConv (include/raindrop_hip.h rd_edge_attention_fwd / _bwd): q.k scores per
    edge and head, softmax per target, coefficient dropout, source-valued aggregate."""

    @staticmethod
    def forward(ctx, q, k, v, ef, edge_index, H, C, p_drop, seed):
        _check(q, k, v, ef)
        _check(edge_index, dtype=torch.int64)
        N, E = q.shape[0], edge_index.shape[1]
        alpha = torch.empty((E, H), dtype=torch.float32, device=q.device)
        alpha_d = torch.empty((E, H), dtype=torch.float32, device=q.device)
        out = torch.empty((N, H * C), dtype=torch.float32, device=q.device)
        _lib.call("rd_edge_attention_fwd", N, E, H, C, _ptr(q), _ptr(k), _ptr(v), _ptr(ef), _ptr(edge_index), edge_index.stride(0),
                  float(p_drop), int(seed) & 0x7FFFFFFFFFFFFFFF, _ptr(alpha), _ptr(alpha_d), _ptr(out), _stream())
        ctx.save_for_backward(q, k, v, ef, edge_index, alpha, alpha_d)
        ctx.dims = (N, E, H, C, float(p_drop), int(seed) & 0x7FFFFFFFFFFFFFFF)
        ctx.mark_non_differentiable(alpha)
        return out, alpha

    @staticmethod
    def backward(ctx, dout, _dalpha):
        q, k, v, ef, ei, alpha, alpha_d = ctx.saved_tensors
        N, E, H, C, p_drop, seed = ctx.dims
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        # rows of nodes without in- / out-edges are written too (zeros): every workgroup covers its node
        dea = torch.zeros_like(ef) if ef is not None else None
        ws = torch.empty((max(E * H, 1),), dtype=torch.float32, device=q.device)
        _lib.call("rd_edge_attention_bwd", N, E, H, C, _ptr(q), _ptr(k), _ptr(v), _ptr(ef), _ptr(ei), ei.stride(0), p_drop, seed,
                  _ptr(alpha), _ptr(alpha_d), _ptr(dout), _ptr(ws), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dea), _stream())
        return dq, dk, dv, dea, None, None, None, None, None


def edge_attention(q, k, v, edge_feat, edge_index, heads, channels, p_drop=0.0, seed=0):
    """(out [N, heads*channels], alpha [E, heads] post-softmax) -- code/transformer_conv.py:186-207 without given edge weights."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_attention: edge_index [2,E] expected, got %s" % (tuple(edge_index.shape),))
    N, HC = q.shape[0], int(heads) * int(channels)
    for name, t in (("q", q), ("k", k), ("v", v)):
        if tuple(t.shape) != (N, HC):
            raise ValueError("edge_attention: %s must be [N, heads*channels] = %s, got %s" % (name, (N, HC), tuple(t.shape)))
    if edge_feat is not None and tuple(edge_feat.shape) != (edge_index.shape[1], HC):
        raise ValueError("edge_attention: edge features must be [E, heads*channels] = %s, got %s"
                         % ((edge_index.shape[1], HC), tuple(edge_feat.shape)))
    _validate_edges(edge_index, N, "edge_attention")
    ef = None if edge_feat is None else edge_feat.contiguous()
    return _EdgeAttention.apply(q.contiguous(), k.contiguous(), v.contiguous(), ef, edge_index.contiguous(), int(heads), int(channels),
                                float(p_drop), int(seed))


def edge_softmax_list_batched(edge_index, edge_weights, n_nodes, norm_row=1):
    """B edge lists in one launch: edge_index int64 [B,2,E] (or [2,E] shared), edge_weights [B,E] -> (gamma_e [B,E], ssum [B,N])."""
    ei = edge_index.contiguous()
    w = edge_weights.contiguous()
    _check(ei, dtype=torch.int64)
    _check(w)
    B, E = w.shape
    shared = ei.dim() == 2
    if not shared and (ei.shape[0] != B or ei.shape[1] != 2 or ei.shape[2] != E):
        raise ValueError("edge_softmax_list_batched: edge_index must be [B,2,E] or [2,E], got %s" % (tuple(ei.shape),))
    gamma = torch.empty((B, E), dtype=torch.float32, device=w.device)
    ssum = torch.empty((B, n_nodes), dtype=torch.float32, device=w.device)
    _lib.call("rd_edge_softmax_list_batched", B, int(n_nodes), E, _ptr(ei), 0 if shared else 2 * E, E, int(norm_row), _ptr(w), E,
              _ptr(gamma), _ptr(ssum), _stream())
    return gamma, ssum


def edge_gamma_dense(edge_index, gamma_e, n_nodes):
    """Dense coefficient matrix gamma[j, i] of an edge list (duplicates added in edge order, on device, deterministic)."""
    ei = edge_index.contiguous()
    g = gamma_e.contiguous()
    _check(ei, dtype=torch.int64)
    _check(g)
    out = torch.empty((n_nodes, n_nodes), dtype=torch.float32, device=g.device)
    _lib.call("rd_edge_gamma_dense", int(n_nodes), ei.shape[1], _ptr(ei), ei.stride(0), _ptr(g), _ptr(out), _stream())
    return out


class _AggregateBatched(torch.autograd.Function):
    """out[b] = gamma^T V[b] (+ skip[b]) for V [B,N,C]: one batched product (rd_aggregate_batched_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, gamma, V, skip):
        _check(gamma, V, skip)
        B, N, C = V.shape
        out = torch.empty_like(V)
        _lib.call("rd_aggregate_batched_fwd", B, N, C, _ptr(gamma), _ptr(V), _ptr(skip), _ptr(out), _stream())
        ctx.save_for_backward(gamma)
        ctx.has_skip = skip is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (gamma,) = ctx.saved_tensors
        dout = dout.contiguous()
        B, N, C = dout.shape
        dV = torch.empty_like(dout)
        _lib.call("rd_aggregate_batched_bwd", B, N, C, _ptr(gamma), _ptr(dout), _ptr(dV), _stream())
        return None, dV, (dout if ctx.has_skip else None)


def aggregate_batched(gamma, V, skip=None):
    return _AggregateBatched.apply(gamma.contiguous(), V.contiguous(), None if skip is None else skip.contiguous())


class _ScaleDropout(torch.autograd.Function):
    """x * scale followed by nn.Dropout on the device RNG (rd_scale_dropout): forward and backward regenerate the same mask."""

    @staticmethod
    def forward(ctx, x, scale, p, seed, site):
        _check(x)
        out = torch.empty_like(x)
        _lib.call("rd_scale_dropout", x.numel(), _ptr(x), float(scale), float(p), int(seed), int(site), _ptr(out), _stream())
        ctx.args = (float(scale), float(p), int(seed), int(site))
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dx = torch.empty_like(dout)
        scale, p, seed, site = ctx.args
        _lib.call("rd_scale_dropout", dout.numel(), _ptr(dout), scale, p, seed, site, _ptr(dx), _stream())
        return dx, None, None, None, None


def scale_dropout(x, scale=1.0, p=0.0, seed=0, site=2):
    if p <= 0.0 and scale == 1.0:
        return x
    return _ScaleDropout.apply(x.contiguous(), scale, p, seed, site)


class _ObsEmbed(torch.autograd.Function):
    """X[b,f,t*d+c] = dropout(relu(src[t,b,f] * R_u[f*d+c])): rd_obs_embed_fwd / _bwd (gradient w.r.t. R_u only)."""

    @staticmethod
    def forward(ctx, src, R_u, shp, p_drop, seed):
        _check(src, R_u)
        B, T, F, d = shp.B, shp.T, shp.F, shp.d_ob
        X = torch.empty((B, F, T * d), dtype=torch.float32, device=src.device)
        _lib.call("rd_obs_embed_fwd", ctypes.byref(shp), _ptr(src), _ptr(R_u), float(p_drop), int(seed), _ptr(X), _stream())
        ctx.save_for_backward(src, X)
        ctx.shp, ctx.p_drop, ctx.ru_shape = shp, float(p_drop), tuple(R_u.shape)
        return X

    @staticmethod
    def backward(ctx, dX):
        src, X = ctx.saved_tensors
        dX = dX.contiguous()
        dRu = torch.empty(ctx.ru_shape, dtype=torch.float32, device=dX.device)
        ws = _workspace(_lib.load().rd_obs_embed_bwd_workspace_bytes(ctypes.byref(ctx.shp)), dX.device)
        _lib.call("rd_obs_embed_bwd", ctypes.byref(ctx.shp), _ptr(src), _ptr(X), _ptr(dX), ctx.p_drop, _ptr(dRu), _ptr(ws), ws.numel(),
                  _stream())
        return None, dRu, None, None, None


def obs_embed(src, R_u, shp, p_drop=0.0, seed=0):
    return _ObsEmbed.apply(src.contiguous(), R_u.contiguous(), shp, p_drop, seed)


class _RowsToTokens(torch.autograd.Function):
    """z[:, :, :F*d] of a [T,B,ldz] buffer <- Y [B,F,T*d] * rowscale [B,F] (in place into `z`, which already holds the PE columns)."""

    @staticmethod
    def forward(ctx, Y, rowscale, z, shp):
        _check(Y, rowscale, z)
        _lib.call("rd_rows_to_tokens_fwd", ctypes.byref(shp), _ptr(Y), _ptr(rowscale), _ptr(z), z.shape[2], _stream())
        ctx.save_for_backward(rowscale)
        ctx.shp, ctx.yshape = shp, tuple(Y.shape)
        ctx.mark_dirty(z)
        return z

    @staticmethod
    def backward(ctx, dz):
        (rowscale,) = ctx.saved_tensors
        dz = dz.contiguous()
        dY = torch.empty(ctx.yshape, dtype=torch.float32, device=dz.device)
        _lib.call("rd_rows_to_tokens_bwd", ctypes.byref(ctx.shp), _ptr(dz), dz.shape[2], _ptr(rowscale), _ptr(dY), _stream())
        return dY, None, None, None


def rows_to_tokens(Y, rowscale, z, shp):
    return _RowsToTokens.apply(Y.contiguous(), None if rowscale is None else rowscale.contiguous(), z, shp)


class _GraphBeta(torch.autograd.Function):
    """The use_beta graph operator (rd_graph_beta_fwd / _bwd), batched: V [B,N,K], H [B,N,T*32], map_w [N,16],
    p_t [B or 1, T, 16], edge_index int64 [2,E], edge_weights [B or 1, E] -> out [B,N,K], edge_index' [B,2,Kk], alpha [B,Kk]."""

    @staticmethod
    def forward(ctx, V, H, map_w, p_t, edge_index, edge_weights, d_ob):
        _check(V, H, map_w, p_t, edge_weights)
        _check(edge_index, dtype=torch.int64)
        B, N, K = V.shape
        T = K // d_ob
        E = edge_index.shape[1]
        lib = _lib.load()
        Kk = int(lib.rd_graph_beta_kept(E))
        dev = V.device
        out = torch.empty_like(V)
        ei_out = torch.empty((B, 2, Kk), dtype=torch.int64, device=dev)
        alpha = torch.empty((B, Kk), dtype=torch.float32, device=dev)
        beta = torch.empty((B, N, T), dtype=torch.float32, device=dev)
        kept = torch.empty((B, max(Kk, 1)), dtype=torch.int32, device=dev)
        pts = 0 if p_t.shape[0] == 1 else T * 16
        ws = 0 if edge_weights.shape[0] == 1 else E
        scratch = _workspace(lib.rd_graph_beta_workspace_bytes(B, N, K, T, E), dev)     # 0 bytes where the graph fits LDS
        _lib.call("rd_graph_beta_fwd", B, N, K, T, d_ob, E, _ptr(V), _ptr(H), _ptr(map_w), _ptr(p_t), pts, _ptr(edge_index),
                  edge_index.stride(0), _ptr(edge_weights), ws, _ptr(out), _ptr(ei_out), _ptr(alpha), _ptr(beta), _ptr(kept),
                  _ptr(scratch), scratch.numel(), _stream())
        ctx.save_for_backward(V, H, map_w, p_t, edge_index, edge_weights, beta, kept)
        ctx.dims = (B, N, K, T, d_ob, E, pts, ws)
        ctx.mark_non_differentiable(ei_out, alpha)
        return out, ei_out, alpha

    @staticmethod
    def backward(ctx, dout, _dei, _dalpha):
        V, H, map_w, p_t, edge_index, edge_weights, beta, kept = ctx.saved_tensors
        B, N, K, T, d_ob, E, pts, ws = ctx.dims
        dout = dout.contiguous()
        dV, dH = torch.empty_like(V), torch.empty_like(H)
        dmap_part = torch.empty((B, N, 16), dtype=torch.float32, device=V.device)
        want_dw = ctx.needs_input_grad[5]
        dw = torch.empty((B, E), dtype=torch.float32, device=V.device) if want_dw else None
        scratch = _workspace(_lib.load().rd_graph_beta_workspace_bytes(B, N, K, T, E), V.device)
        _lib.call("rd_graph_beta_bwd", B, N, K, T, d_ob, E, _ptr(V), _ptr(H), _ptr(map_w), _ptr(p_t), pts, _ptr(edge_index),
                  edge_index.stride(0), _ptr(edge_weights), ws, _ptr(beta), _ptr(kept), _ptr(dout), _ptr(dV), _ptr(dH),
                  _ptr(dmap_part), _ptr(dw), _ptr(scratch), scratch.numel(), _stream())
        dmap = dmap_part[0] if B == 1 else _colsum_rows(dmap_part.view(B, N * 16)).view(N, 16)
        if want_dw and edge_weights.shape[0] == 1 and B > 1:
            dw = _colsum_rows(dw).view(1, E)
        return dV, dH, dmap, None, None, dw, None


def _colsum_rows(x):
    """Deterministic sum over the rows of a small [B, n] matrix on the device: rd_linear_bwd_weight with a ones column
    (dW[1, n] = ones[B,1]^T x[B,n]), i.e. the library's fixed-order split reduction -- no torch math in the product."""
    B, n = x.shape
    ones = torch.ones((B, 1), dtype=torch.float32, device=x.device)
    dW = torch.empty((1, n), dtype=torch.float32, device=x.device)
    ws = _workspace(_lib.load().rd_linear_bwd_weight_workspace_bytes(B, 1, n), x.device)
    _lib.call("rd_linear_bwd_weight", B, 1, n, _ptr(ones), 1, _ptr(x.contiguous()), n, _ptr(dW), _ptr(None), _ptr(ws), ws.numel(),
              _stream())
    return dW


_EDGES_CHECKED = set()


def _validate_edges(edge_index, N, who):
    """Edge endpoints in [0, N): one device read per edge LIST (keyed by storage, version and size -- the models pass their cached
    graph every call), because the read syncs and is not permitted while a hipGraph is being captured (AutogradStep).  The
    reference's index_select raises IndexError at the same point."""
    E = edge_index.shape[1]
    if E == 0:
        return
    key = (edge_index.data_ptr(), edge_index._version, E, N, str(edge_index.device))
    if key in _EDGES_CHECKED:
        return
    if torch.cuda.is_current_stream_capturing():
        raise _lib.RaindropHipError("%s: this edge list has not been validated yet and a stream capture is in progress; "
                                    "run one eager step first" % who)
    lo, hi = int(edge_index.min()), int(edge_index.max())
    if lo < 0 or hi >= N:
        raise IndexError("%s: edge endpoint out of range [0, %d): min %d, max %d" % (who, N, lo, hi))
    if len(_EDGES_CHECKED) > 64:
        _EDGES_CHECKED.clear()
    _EDGES_CHECKED.add(key)


def graph_beta(V, H, map_w, p_t, edge_index, edge_weights, d_ob=4):
    """use_beta branch of Observation_progation.message, batched (include/raindrop_hip.h: rd_graph_beta_fwd).  V [B,N,K],
    H [B,N,T*32], map_w [N,16], p_t [B or 1, T, 16], edge_index int64 [2,E], edge_weights [B or 1, E].  Shapes and edge
    endpoints are validated here (one device read for the endpoint range: the reference's index_select syncs and raises
    IndexError at the same point)."""
    if V.dim() != 3 or H.dim() != 3 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("graph_beta: V [B,N,K], H [B,N,T*32], edge_index [2,E] expected")
    B, N, K = V.shape
    if K % d_ob:
        raise ValueError("graph_beta: K (%d) is not a multiple of d_ob (%d)" % (K, d_ob))
    T, E = K // d_ob, edge_index.shape[1]
    if tuple(H.shape) != (B, N, T * 32):
        raise ValueError("graph_beta: H must be [B,N,T*32] = %s, got %s" % ((B, N, T * 32), tuple(H.shape)))
    if tuple(map_w.shape) != (N, 16):
        raise ValueError("graph_beta: map_weights must be [N,16] = %s, got %s" % ((N, 16), tuple(map_w.shape)))
    if p_t.dim() != 3 or p_t.shape[0] not in (1, B) or tuple(p_t.shape[1:]) != (T, 16):
        raise ValueError("graph_beta: p_t must be [1 or B, T, 16], got %s" % (tuple(p_t.shape),))
    if edge_weights.dim() != 2 or edge_weights.shape[0] not in (1, B) or edge_weights.shape[1] != E:
        raise ValueError("graph_beta: edge_weights must be [1 or B, E], got %s" % (tuple(edge_weights.shape),))
    _validate_edges(edge_index, N, "graph_beta")
    return _GraphBeta.apply(V.contiguous(), H.contiguous(), map_w.contiguous(), p_t.contiguous(), edge_index.contiguous(),
                            edge_weights.contiguous(), int(d_ob))


def structure_distance(alpha_all):
    """code/models_rd.py:345-346: mean(cdist(alpha_all.T, alpha_all.T, p=2)) for alpha_all [E,B] (no gradient: the
    reference's training loss does not use it, code/Raindrop.py:319-322)."""
    a = alpha_all.detach().contiguous()
    _check(a)
    E, B = a.shape
    ws = torch.empty((B,), dtype=torch.float32, device=a.device)
    out = torch.empty((), dtype=torch.float32, device=a.device)
    _lib.call("rd_structure_distance", E, B, _ptr(a), _ptr(ws), _ptr(out), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# temporal stage: nn.TransformerEncoderLayer and the masked mean, on the HIP kernels
# ------------------------------------------------------------------------------------------------

ENC_PARAM_NAMES = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                   "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                   "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


def _enc_ptrs(tensors):
    return _lib.RdEncoderPtrs(*[t.data_ptr() for t in tensors])


class _EncoderLayer(torch.autograd.Function):
    """One post-norm TransformerEncoderLayer: rd_encoder_layer_fwd / rd_encoder_layer_bwd.
    Inputs: x [T,B,D], mask [B,T] bool, the 12 parameters in ENC_PARAM_NAMES order."""

    @staticmethod
    def forward(ctx, x, mask, shp, layer, p_drop, seed, *params):
        _check(x, *params)
        _check(mask, dtype=torch.bool)
        lib = _lib.load()
        sp = ctypes.byref(shp)
        dev = x.device
        saved = _workspace(lib.rd_encoder_layer_saved_bytes(sp), dev)
        ws = _workspace(lib.rd_encoder_layer_workspace_bytes(sp), dev)
        y = torch.empty_like(x)
        w = _enc_ptrs(params)
        _lib.call("rd_encoder_layer_fwd", sp, int(layer), _ptr(x), _ptr(mask), ctypes.byref(w), float(p_drop),
                  int(seed), _ptr(y), _ptr(saved), saved.numel(), _ptr(ws), ws.numel(), _stream())
        ctx.shp, ctx.layer, ctx.p_drop, ctx.seed = shp, int(layer), float(p_drop), int(seed)
        ctx.save_for_backward(x, mask, saved, *params)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask, saved, *params = ctx.saved_tensors
        dy = dy.contiguous()
        lib = _lib.load()
        sp = ctypes.byref(ctx.shp)
        dev = dy.device
        ws = _workspace(lib.rd_encoder_layer_workspace_bytes(sp), dev)
        dx = torch.empty_like(x)
        grads = [torch.empty_like(p) for p in params]
        w, g = _enc_ptrs(params), _enc_ptrs(grads)
        _lib.call("rd_encoder_layer_bwd", sp, ctx.layer, _ptr(x), _ptr(mask), ctypes.byref(w), ctx.p_drop, ctx.seed,
                  _ptr(saved), saved.numel(), _ptr(dy), _ptr(dx), ctypes.byref(g), _ptr(ws), ws.numel(), _stream())
        return (dx, None, None, None, None, None, *grads)


def encoder_layer(x, mask, shp, layer, p_drop, seed, params):
    return _EncoderLayer.apply(x.contiguous(), mask, shp, layer, p_drop, seed, *[p.contiguous() for p in params])


class _MaskedMean(torch.autograd.Function):
    """code/models_rd.py:366-367,379 -- writes the mean into the left D columns of a [B, D+extra]
    buffer so the static embedding can be concatenated without a copy."""

    @staticmethod
    def forward(ctx, r, mask, lengths, shp, extra):
        _check(r)
        T, B, D = r.shape
        out = torch.empty((B, D + extra), dtype=torch.float32, device=r.device)
        _lib.call("rd_masked_mean_fwd", ctypes.byref(shp), D, _ptr(r), _ptr(mask), _ptr(lengths), _ptr(out),
                  D + extra, _stream())
        ctx.shp, ctx.dims = shp, (T, B, D, extra)
        ctx.save_for_backward(mask, lengths)
        return out

    @staticmethod
    def backward(ctx, dout):
        mask, lengths = ctx.saved_tensors
        T, B, D, extra = ctx.dims
        dout = dout.contiguous()
        dr = torch.empty((T, B, D), dtype=torch.float32, device=dout.device)
        _lib.call("rd_masked_mean_bwd", ctypes.byref(ctx.shp), D, _ptr(dout), D + extra, _ptr(mask), _ptr(lengths),
                  _ptr(dr), _stream())
        return dr, None, None, None, None


def masked_mean(r, mask, lengths, shp, extra=0):
    return _MaskedMean.apply(r.contiguous(), mask, lengths, shp, extra)
