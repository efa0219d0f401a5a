"""CPU oracle for the RetinaNet hot path (TEST INFRASTRUCTURE ONLY).

This package restates, on the CPU (numpy + torch-CPU fp32), the algorithm of the
reference's hot path so that the HIP kernels can be parity-checked.  It is NOT part of
the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / timed baseline.  The
product path (``retinanet-tensorflow_amd/``) never imports it and fails loudly when the
HIP library is missing.

Pinning status:
  * levels / anchor sizes: PINNED against the reference's own ``levels.py`` imported in
    the build container (fixture ``tests/golden/levels_reference.npz`` written by
    ``tests/golden/make_golden.py``) and against ``levels_test.py:5-14``.
  * iou, grid/decode transforms, scale_regression, Huber regression loss, class
    assignment: PINNED against the known-answer vectors of ``utils_test.py:7-118``,
    ``retinanet_old_test.py:15-37``, ``losses_test.py:17-27``, ``dataset_test.py:8-45``
    (transcribed as arrays in ``tests/golden/reference_kats.py``).
  * conv / GroupNorm / ELU / NN-resize / BCE+dice / focal / boxes_decode / NMS /
    optimizers: **parity unpinned** -- the reference holds no test or golden value for
    them and TensorFlow 1.x (un-vendored, un-pinned third-party dependency) is not
    installable here.  Their TensorFlow semantics are restated from TF 1.x's published
    behaviour and each rule is a named, tested function tagged [TF-sem].
"""
