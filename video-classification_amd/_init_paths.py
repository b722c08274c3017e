"""Put lib/ (reference-style: core, models, utils, dataset) and the vc3d engine on sys.path.
Mirrors the role of /root/reference/_init_paths.py:14-17."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
for _p in (_here, os.path.join(_here, "lib")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
