"""CPU oracle for the vtx hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The shipped path
(``videotransformer-pytorch_amd/``) never imports this package and raises if
its HIP library is missing.

Contents
--------
vt_oracle.py   fp32 torch-CPU restatement of the reference transformer path
               (reference transformer.py / video_transformer.py, cited per
               function), functional style over a plain ``state_dict``.
hog_oracle.py  NumPy restatement of skimage-0.18.3 ``hog`` as the reference
               calls it (reference dataset.py:39-45), bit-exact vs real skimage.
hog_ref.c      plain-C restatement of the same arithmetic (built by
               ``__graft_entry__.build()`` into oracle/_build/libhogref.so).
ref_loader.py  loads the *unmodified* reference modules from /root/reference
               (dev container only; used to pin the restatement and to generate
               tests/golden/).  Not available on the GPU box.

Parity pin: the restatement is checked against (a) the running reference in
this container (tests/test_oracle_pin.py, skipped when /root/reference is
absent) and (b) the committed golden vectors in tests/golden/ generated from
the reference by tests/golden/make_golden.py.
"""
