"""CPU oracle for the iterative-inference hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy restatement (float64 by default) of the arithmetic that the
reference (adri-romsor/iterative_inference_segm, Theano/Lasagne, Python 2) performs on the
path `pred_fcn_fn -> de_fn x num_iter -> val_fn`.  Every function cites the reference
file:line it follows.

PARITY UNPINNED vs. the real Theano reference: the reference cannot be executed (Python 2,
Theano/Lasagne/dataset_loaders absent, no trained weights) and it
ships no tests or numeric fixtures for this path.  The only golden vector the reference holds
is the `build_experiment_name` string recorded in plots.ipynb:84, which `oracle.naming` is
pinned against in tests/test_oracle_naming.py.  Lasagne/Theano layer semantics (SURVEY.md
section 2.1 pins P1-P15) are restated from the pinned upstream versions and covered by
hand-computed known-answer tests in tests/test_oracle_kat.py.

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of `bench.py` -- as the checker / the timed CPU baseline, never as the product path.  The
product package `iterative_inference_segm_amd` must not import it (tests/test_boundary.py
greps for that).
"""

from . import nn, fcn8, dae, contextmod, refine, metrics, naming  # noqa: F401
