"""Type aliases the reference imports (`Ob_propagation.py:6`, `transformer_conv.py:3`)."""
from typing import Optional, Tuple, Union
from torch import Tensor

Adj = Union[Tensor, "SparseTensor"]
OptTensor = Optional[Tensor]
PairTensor = Tuple[Tensor, Tensor]
OptPairTensor = Tuple[Tensor, Optional[Tensor]]
Size = Optional[Tuple[int, int]]
