#!/usr/bin/env python
"""Run a script with Winograd tuning keys preset: python tools/run_tuned.py KEY=VALUE[,KEY=VALUE] script.py [args...]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcomp_cvpr_amd import _lib
for kv in sys.argv[1].split(','):
    k, v = kv.split('=')
    _lib.lib.ic_wino3x3_c128_set_tuning(int(k), int(v))
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
