#!/usr/bin/env python3
"""Run a tool script against another build of the library (same box, same inputs):
  python3 tools/ab_run.py tools/ab_libs/libsr_hip_base.so tools/quick_encode_budget.py 16384
The product package never looks anywhere but scaling_retriever_amd/libsr_hip.so; this wrapper points the loader elsewhere for ONE process."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scaling_retriever_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")
