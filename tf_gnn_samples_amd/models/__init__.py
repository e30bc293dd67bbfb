from .sparse_graph_model import Sparse_Graph_Model
from .rgcn_model import RGCN_Model
from .ggnn_model import GGNN_Model
