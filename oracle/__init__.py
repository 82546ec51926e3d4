"""CPU oracle for the PoP-Net inference hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(``/root/reference``; citations in each function) used to CHECK the HIP path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package (``pop-net_amd/`` == ``popnet_amd``)
never does: it fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * parse / grouping / read-out / YOLO decode / metric glue: PINNED against the
    reference's own Python, imported in the build container with import shims
    (``tests/golden/make_golden.py``), outputs committed under ``tests/golden``.
  * ``process_paf`` (COCO-18 C++): PINNED against the reference ``pafprocess.cpp``
    compiled as-is into ``oracle/_ref`` (``oracle/Makefile``).
  * network forward: PINNED (tolerance, fp32) against the reference ``nn.Module``s.
  * the OpenCV ``cv2.resize`` arithmetic (bilinear input resize, bicubic patch /
    PAF up-sampling): **parity unpinned** -- OpenCV 4.2.0 (pinned in the
    reference's environment.yaml:325) is a third-party dependency absent from
    both ``/root/reference`` and this image.  ``oracle/cv2_resize.py`` restates its
    published algorithm; it is cross-checked against ``torch.nn.functional
    .interpolate`` (same kernel, different evaluation order) only.
"""
