from .sparse_graph_task import DataFold, DeviceBatch, MinibatchData, Sparse_Graph_Task
from .ppi_task import PPI_Task
from .qm9_task import QM9_Task

TASK_CLASSES = {"ppi": PPI_Task, "qm9": QM9_Task}   # utils/model_utils.py:12-29 (VarMisuse / citation tasks: out of scope)


def name_to_task_class(name: str):
    """-> (class, extra task parameters), utils/model_utils.py:12-29 (the citation / VarMisuse names are unknown here)."""
    key = name.lower()
    if key not in TASK_CLASSES:
        raise ValueError("Unknown task type '%s'" % key)
    return TASK_CLASSES[key], {}
