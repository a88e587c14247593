"""Build csrc/libtcl_hip.so for gfx950:  python -m master_thesis_amd.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


def build(force=False, jobs=8):
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j%d" % jobs])
    return os.path.join(CSRC, "libtcl_hip.so")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
