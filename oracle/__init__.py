"""CPU oracle for the sparse relational message-passing path of microsoft/tf-gnn-samples.

TEST INFRASTRUCTURE ONLY.  Nothing under tf_gnn_samples_amd/ imports this package; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker
/ the reported CPU baseline — never as the thing shipped.

PARITY UNPINNED: the reference cannot run here (tensorflow 1.13, dpu_utils and docopt are not
installed and cannot be: Python 3.10, no network) and it ships no tests, golden vectors or
fixtures for this path.  This oracle is therefore a NumPy restatement of the reference's
arithmetic, op for op and in the reference's op order, with every function citing the
reference file:line it follows and the TensorFlow-internal semantics it assumes spelled out
(marked [TF-internal]).  The pins that do exist are checked in tests/: the published
parameter count 699 257 (README.md:29), the RGIN G1/G2 docstring example (gnns/rgin.py:29-35),
hand-computed tiny graphs, fp64-vs-fp32 agreement and size-independent properties.
"""
