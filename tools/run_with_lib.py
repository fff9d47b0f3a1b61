#!/usr/bin/env python3
"""usage: run_with_lib.py <libvvm_hip variant .so> <script.py> [args...]   runs a script of this repository against another build of
the library (kernel experiments)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from renderer_amd import abi

abi._HIP_LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
