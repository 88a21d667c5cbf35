#!/usr/bin/env python3
"""bench.py against another build of the library (same-box A-B): SFRON_LIB_NAME=libsfron_base.so python tools/bench_ab.py --steps 10 ..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib  # noqa: E402
ab_lib.select()
import bench  # noqa: E402
bench.main()
