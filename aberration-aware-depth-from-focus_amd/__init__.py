"""MI355X-native focal-stack rendering path (package root).

The directory name is not a Python identifier; put this directory on sys.path and import
`deeplens` (drop-in API of the reference) and `aadff` (ABI binding, stack renderer)."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
