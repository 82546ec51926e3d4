"""CPU oracle for the PoP-Net inference hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(``/root/reference``; citations in each function) used to CHECK the HIP path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package (``popnet_amd/`` == ``popnet_amd``)
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
    .interpolate`` (same kernel, different evaluation order) and pinned by hand-derived
    known answers (``tests/test_cv2_kat.py``: an exact-rational evaluator that follows the
    scalar code path of resize.cpp operation by operation and rounds every float32 operation
    itself, plus frozen hex answers and analytic properties of the A = -0.75 kernel).
    What can still move the LAST ULP against a real OpenCV build, and is therefore not claimed
    (none of it can be checked here; these are the known degrees of freedom of resize.cpp builds):
      - SIMD code paths compiled with FMA (the AVX2 / AVX-512 dispatch of the ``VResize*Vec_32f`` /
        ``HResize*`` helpers): a fused multiply-add skips the rounding of a product that the
        scalar path performs;
      - a SIMD vertical pass is free to associate the four (cubic) row products differently from
        the scalar left-to-right sum;
      - vendor back ends a wheel may be built with (IPP on x86) whose float results are not
        specified bit for bit;
      - INTER_LINEAR at exactly 2x decimation runs INTER_AREA (ResizeAreaFast): refused by the
        oracle and by ``pn_preprocess`` instead of guessed.
"""
