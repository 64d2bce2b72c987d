#!/usr/bin/env python3
"""Run a script of this repository against ANOTHER build of libvnpcc.so (A/B of two builds on one box, one after the other):
    python scripts/experiments/with_lib.py path/to/libvnpcc_variant.so scripts/bench_configs.py 3 --steps 10
The library path is swapped before anything loads it; nothing in the product reads an environment variable for this."""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from vn_pointcloudcompletion_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
_lib.build = lambda *a, **k: _lib.LIB_PATH          # the variant is what it is: no rebuild from the tree's sources
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
