"""CPU oracle for the ubdvss hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference's algorithm
(asmekal/ubdvss, files cited per function) used as the *checker* for the HIP
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product package ``ubdvss_amd`` never
imports, links or executes anything from here and fails loudly when its HIP
library is missing.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and
its third-party engines (Keras 2.2.x / TensorFlow 1.x / OpenCV 3.4) are neither
vendored nor installable in the build container, so these restatements are
checked against each other (numpy fp64 vs torch-CPU, C vs scipy) and against
analytic known-answer cases -- not against outputs of the reference itself.
"""
