"""CPU-side checks of the drop-in boundary: the library builds, loads and exports every symbol the headers declare.
No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gorio_[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.parametrize("header", [h for h in sorted(os.listdir(os.path.join(ROOT, "include"))) if h.endswith(".h")])
def test_library_exports_every_declared_symbol(gorio, header):
    lib = gorio.load_library()
    names = _declared(header)
    assert names, header
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/{header} but not exported by libgorio_amd.so"


def test_binding_covers_header(gorio):
    from importlib import import_module

    apd = import_module("go-rio_amd.apd")
    assert sorted(apd.APD_SYMBOLS) == _declared("gorio_apd.h")
    ugpm = import_module("go-rio_amd.ugpm")
    assert sorted(ugpm.UGPM_SYMBOLS) == _declared("gorio_ugpm.h")
    prep = import_module("go-rio_amd.prep")
    assert sorted(prep.PREP_SYMBOLS) == _declared("gorio_prep.h")


def test_default_params_match_reference_defaults(gorio):
    """Constructor defaults of FastAPDGICP / LsqRegistration (fast_apdgicp_impl.hpp:14-28, lsq_registration_impl.hpp:10-24,
    fast_apdgicp.hpp:116-118)."""
    lib = gorio.load_library()
    p = gorio.ApdParams()
    lib.gorio_apd_default_params(C.byref(p))
    assert p.k_correspondences == 20 and p.regularization == 3  # PLANE
    assert (p.dist_var, p.azimuth_var, p.elevation_var) == (0.86, 0.5, 1.0)
    assert p.corr_dist_threshold == pytest.approx(3.4028234663852886e38)
    assert p.max_iterations == 64 and p.rotation_epsilon == 2e-3 and p.transformation_epsilon == 5e-4
    assert p.optimizer == 1 and p.lm_max_iterations == 10 and p.lm_init_lambda_factor == 1e-9


def test_no_cpu_fallback_without_device(gorio):
    """On a box without a GPU the product must refuse, not compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = gorio.load_library()
    h = C.c_void_p()
    assert lib.gorio_apd_create(C.byref(h), 0) == -2  # GORIO_ERR_NO_DEVICE
    with pytest.raises(gorio.GorioError):
        gorio.ApdGicp()


def test_product_never_touches_the_oracle():
    """The product tree must not import, include, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "go-rio_amd")
    bad = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.+oracle|#include\s*[<\"][^>\"]*oracle|oracle/|oracle\\|_oracle\.so|-l\w*oracle)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".c", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                m = bad.search(txt)
                assert m is None, f"{os.path.join(dirpath, f)} references the oracle: {m.group(0)!r}"
