"""MI355X mirrors of the reference's gnns package (gnns/__init__.py:1-7)."""
from .ggnn import sparse_ggnn_layer, ggnn_layer_variables
from .gnn_edge_mlp import sparse_gnn_edge_mlp_layer, gnn_edge_mlp_layer_variables
from .gnn_film import sparse_gnn_film_layer, gnn_film_layer_variables
from .rgat import sparse_rgat_layer, rgat_layer_variables
from .rgcn import sparse_rgcn_layer, rgcn_layer_variables
from .rgin import sparse_rgin_layer, rgin_layer_variables
from .rgdcn import sparse_rgdcn_layer, rgdcn_layer_variables
