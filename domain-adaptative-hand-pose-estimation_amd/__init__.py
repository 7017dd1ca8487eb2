"""MI355X-native hot path of the domain-adaptive hand-pose pipeline.

The directory name (the reference repo's name + ``_amd``) is not a valid Python identifier, so the
package is used as a *source root*: put this directory on ``sys.path`` and import ``uda.model``,
``utils`` (the mirrors of the reference's modules) and ``mi355`` (the HIP boundary) from it:

    import sys; sys.path.insert(0, '<repo>/domain-adaptative-hand-pose-estimation_amd')
    import uda.model as models
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
