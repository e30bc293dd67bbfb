"""CPU oracle for the sparse relational message-passing path of microsoft/tf-gnn-samples.

TEST INFRASTRUCTURE ONLY.  Nothing under tf_gnn_samples_amd/ imports this package; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker
/ the reported CPU baseline — never as the thing shipped.

PARITY: PINNED AS A TRANSCRIPTION, UNPINNED AT THE LEVEL OF TENSORFLOW'S OPS.
TensorFlow 1.13, dpu_utils and docopt are not installed and cannot be (Python 3.10, no network), and the reference ships no
tests, golden vectors or fixtures for this path.  This oracle is a NumPy restatement of the reference's arithmetic, op for op
and in the reference's op order, every function citing the reference file:line it follows and the TensorFlow-internal
semantics it assumes (marked [TF-internal]).  What pins it:

  * Since round 4 the REFERENCE'S OWN SOURCES are executed in the build container — gnns/*.py (all seven layer functions),
    utils/utils.py, models/sparse_graph_model.py's __make_model with every model adapter and the PPI / QM9 output heads,
    tasks/ppi_task.py and tasks/qm9_task.py (loaders and minibatch iterators) — unmodified, over a NumPy shim of the ~60
    TensorFlow / dpu_utils symbols they touch (tests/golden/tf_numpy_shim.py, make_reference_run.py).  Their outputs are
    committed under tests/golden/reference_run_*.npz; oracle/gnns.py, oracle/model.py and oracle/bookkeeping.py must reproduce
    them (tests/test_reference_run_cpu.py: to the bit for the bookkeeping, within one ulp for the layers), and so must the
    package's host code (bit for bit) and the HIP path (1e-5, tests/test_gpu_reference_run.py).  That removes the
    transcription risk: which ops run on which operands in which order, constants, variable names, concat / segment orders,
    batch packing.  The run also reproduces a known answer of the reference: "Model has 699257 parameters." (README.md:29).
    A float64 torch backend of the same shim (tf_torch_shim.py, over oracle/torch_ref.py's primitives) runs the sources under
    autograd: gradients of every layer case and the reference's own train step (clip_by_norm per variable, optimizer
    selection, learning-rate normalisation) pin oracle/torch_ref.py and oracle/optim.py the same way.
  * The semantics of the individual TensorFlow ops are the shim's, i.e. oracle/tf_ops.py's: NOT pinned against TensorFlow.
    They are cross-checked against PyTorch's independent CPU kernels (tests/test_oracle_crosscheck_cpu.py), hand-derived
    vectors that fail under look-alike semantics, the RGIN G1/G2 docstring example (gnns/rgin.py:29-35), fp64-vs-fp32
    agreement and size-independent properties.  scripts/dump_tf_golden.py writes the fixtures that would close this part
    wherever TF 1.13-1.15 exists; tests/test_tf_golden.py consumes them.
"""
