"""TEST INFRASTRUCTURE ONLY -- stand-in for the un-vendored `torch_geometric` dependency.

The reference (`/root/reference/code/{models_rd,Ob_propagation,transformer_conv}.py`) imports
PyTorch-Geometric, which is neither pinned in `requirements.txt:1-9` nor installed here.  This
package restates, in plain CPU torch, exactly the pieces the reference calls:

* `MessagePassing(aggr='add', node_dim=0).propagate`   (`Ob_propagation.py:114`, `transformer_conv.py:158`)
* `utils.softmax(src, index, ptr, num_nodes)`          (`Ob_propagation.py:195`, `transformer_conv.py:201`)
* `nn.inits.{glorot,uniform,zeros,ones,reset}`         (`Ob_propagation.py:85,90-91`, `models_rd.py:276`)

Semantics follow the published PyG 1.7-2.0 sources (flow='source_to_target', `x_j = x[edge_index[0]]`,
`x_i = x[edge_index[1]]`, softmax adds 1e-16 to the denominator).  It exists so that the reference
files can be executed UNMODIFIED as oracle O1; nothing under `raindrop_amd/` may import it.
"""
