"""CPU oracle for the Go-RIO hot path (APD-GICP scan matching + UGPM GP pre-integration).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (go-rio_amd/) never does.  See oracle/apd_oracle.c and oracle/ugpm_oracle.cpp for the restated
reference lines, and DESIGN.md for how the oracle is pinned.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.join(_HERE, "build")


def build(force: bool = False) -> None:
    """Compile the C/C++ restatements with the recipe in oracle/Makefile (gcc/g++ only, no GPU needed)."""
    subprocess.check_call(["make", "-C", _HERE, "all"] + (["-B"] if force else []), stdout=sys.stderr)  # stdout belongs to the caller (bench.py prints one JSON line)
