"""TEST INFRASTRUCTURE ONLY -- the reference only uses `SparseTensor` in isinstance checks
(`Ob_propagation.py:11,127`, `transformer_conv.py:7,181`)."""


class SparseTensor:  # never instantiated on the path
    pass
