"""CPU oracle for the sketch-transformer-tf2 train step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sketchformer_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / baseline.

Parity status: **parity unpinned by the reference's own tests** (the reference
has none, and TensorFlow cannot be imported in the build container).  The
oracle is pinned instead by (i) goldens captured from the TF-free reference
functions (``tests/golden``), (ii) an independent torch-autograd witness
(``tests/witness_torch.py``) and (iii) the analytic known-answer tests of
SURVEY.md section 8(c).
"""
from .sketchformer_oracle import *  # noqa: F401,F403
