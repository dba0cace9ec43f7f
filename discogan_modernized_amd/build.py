"""Build csrc/*.hip into libdiscogan_hip.so for gfx950:  python -m discogan_modernized_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=True, jobs=6):
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), f"-j{jobs}"]
    r = subprocess.run(cmd, capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc build failed:\n" + (r.stdout or "") + (r.stderr or ""))
    lib = os.path.join(HERE, "libdiscogan_hip.so")
    if not os.path.exists(lib):
        raise RuntimeError("build finished but libdiscogan_hip.so is missing")
    return lib


if __name__ == "__main__":
    print(build())
    sys.exit(0)
