from .sparse_graph_task import DataFold, DeviceBatch, MinibatchData, Sparse_Graph_Task
from .ppi_task import PPI_Task
