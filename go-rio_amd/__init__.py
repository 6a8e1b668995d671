"""go-rio_amd -- MI355X-native back end of Go-RIO's hot path (APD-GICP scan matching + UGPM GP pre-integration).

The product is the C-ABI shared library go-rio_amd/lib/libgorio_amd.so (hand-written HIP for gfx950, sources in
go-rio_amd/csrc, ABI in include/gorio_apd.h and include/gorio_ugpm.h) plus the C++ host classes in go-rio_amd/host that
mirror the reference's own class surfaces (fast_gicp::FastAPDGICP, ugpm::VelPreintegration).  This Python module is a thin
ctypes binding of that ABI for tests and bench.py -- it adds no numerics of its own and has NO CPU fallback: if the
library is missing, or no HIP device is usable, calls raise.

The directory name contains a hyphen, so import it with
    import importlib; gorio = importlib.import_module("go-rio_amd")
"""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GORIO_AMD_LIB") or os.path.join(_HERE, "lib", "libgorio_amd.so")  # GORIO_AMD_LIB: a variant build (development A/B runs)
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")
CSRC_DIR = os.path.join(_HERE, "csrc")


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into go-rio_amd/lib/libgorio_amd.so (hipcc cross-compiles without a GPU)."""
    import subprocess
    import sys

    subprocess.check_call(["make", "-C", CSRC_DIR] + (["-B"] if force else []), stdout=sys.stderr)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


from . import synth  # noqa: E402  (pure-numpy synthetic inputs, no GPU)
from .apd import ApdGicp, ApdParams, DeviceInputs, GorioError, align_batch, load_library  # noqa: E402
from . import prep  # noqa: E402
from .ugpm import PreintOption, PreintPrior, UgpmBatch, VelPreintegration, ugpm_combine_preints, ugpm_preint_batch, ugpm_stage_times  # noqa: E402
