"""MI355X mirrors of the reference's gnns package (gnns/__init__.py:1-7)."""
from .ggnn import sparse_ggnn_layer, ggnn_layer_variables
from .rgcn import sparse_rgcn_layer, rgcn_layer_variables
