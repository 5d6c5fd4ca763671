#!/usr/bin/env python3
"""Run a tool against another build of the product library (A/B of compile-time variants):
with_lib.py <path to libsbv variant .so> <script.py> [args ...].  The variant replaces consensus_amd.LIB_PATH before
anything loads it."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import consensus_amd  # noqa: E402

consensus_amd.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
