"""CPU oracle for the domain-adaptive hand-pose hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this
directory; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may, and there only as the checker.

It is a pure-torch (CPU, fp32) restatement of the reference's live classes
(reference = CVlab315/Domain-Adaptative-Hand-Pose-Estimation; every function
cites the reference file:line it follows).  The restatement issues the same
ATen ops as the reference, so it is bit-identical to it on CPU for
neck / heads / losses / pseudo-labels / decode; this is pinned by the golden
vectors under ``tests/golden/`` that were captured by importing the reference
itself (``tests/golden/make_golden.py``).

The ResNet backbone arithmetic lives in third-party ``torchvision`` (absent
from the reference tree and from this image, version unpinned by the
reference: ``torchvision>=0.5.0``); ``backbone.py`` restates the published
torchvision ResNet-v1.5 structure.  For that part parity is UNPINNED by any
reference fixture (the reference holds no tests); its arithmetic oracle is
torch's own CPU conv2d / batch_norm / max_pool2d.
"""
