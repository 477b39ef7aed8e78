"""Build libpiehip.so in-tree with hipcc (cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpiehip.so")


def build(force=False):
    args = ["make", "-j%d" % min(8, os.cpu_count() or 1), "-C", os.path.join(HERE, "csrc")]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc did not produce %s" % LIB_PATH)
    return LIB_PATH
