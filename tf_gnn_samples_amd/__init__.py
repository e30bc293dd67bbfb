"""tf_gnn_samples_amd — MI355X-native sparse relational message passing, a drop-in for the
hot path of microsoft/tf-gnn-samples (gnns/*.py driven from models/sparse_graph_model.py).

Compute runs ONLY through librelgnn.so (hand-written HIP for gfx950, C ABI in include/relgnn.h);
there is no CPU fallback.  Importing the package does not need a GPU or the built library —
calling a kernel does.
"""
__version__ = "0.1.0"
