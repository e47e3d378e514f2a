"""CPU oracle for the RPN proposal path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``tf_rpn_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / reported CPU baseline, never as the product path.

PARITY UNPINNED: the reference (FurkanOM/tf-rpn) ships no tests, golden
vectors or fixtures, and TensorFlow 2.0.0 / keras-applications 1.0.8 (which own
Conv2D, CombinedNonMaxSuppression, exp, sqrt) cannot be imported in the build
container.  The functions here are line-by-line restatements of the reference's
Python (each cites file:line under /root/reference) plus a restatement of the
documented TF kernel semantics.  Golden fixtures under
``tests/golden`` are therefore "restatement-generated", never "TF-generated".
"""
