#!/usr/bin/env python
"""A/B of the cached limb images of the 128-column panel products (dense._SEL_CACHE) on the C3 / C5 steps of bench_other.py:
usage: ab_sel_cache.py {0|1} [C3|C5 ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tf_gnn_samples_amd.dense as D
D._SEL_CACHE = sys.argv[1] == "1"
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench_other
bench_other.main()
