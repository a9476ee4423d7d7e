#!/usr/bin/env python3
"""Dense stencil micro-benchmark only: python tools_micro.py [n]"""
import json, sys
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench  # noqa: E402 (repo root on sys.path above)
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
print(json.dumps(bench.stencil_microbench(fs, n, 0)))
