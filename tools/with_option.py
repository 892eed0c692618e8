"""Run another tool with library options set first: python tools/with_option.py key=value[,key=value] tools/<tool>.py [args]"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fidelityfusion_amd import _lib  # noqa: E402

for kv in sys.argv[1].split(","):
    k, v = kv.split("=")
    _lib.set_option(k, float(v), 0)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
