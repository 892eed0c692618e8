"""CPU: the C-ABI library loads and exports every symbol include/ffgp.h declares (no compute without a GPU),
and the package fails loudly -- never falls back -- when no GPU is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "fidelityfusion_amd", "libffgp.so")
HDR = os.path.join(ROOT, "include", "ffgp.h")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ffgp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(SO):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(SO)


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("ffgp_create", "ffgp_destroy", "ffgp_set_stream", "ffgp_assemble", "ffgp_potrf", "ffgp_potrf_rows",
                 "ffgp_trsm_lower", "ffgp_potrs", "ffgp_potri", "ffgp_nll_reduce", "ffgp_nlml_fused", "ffgp_predict",
                 "ffgp_gemm", "ffgp_last_timings", "ffgp_syrk_stats", "ffgp_mfma_f64_peak"):
        assert must in names


def test_so_exports_every_declared_symbol(built):
    missing = [n for n in declared_functions() if not hasattr(built, n)]
    assert not missing, missing


def test_so_exports_nothing_but_the_declared_symbols(built):
    """the dynamic symbol table of libffgp.so IS the header (-fvisibility=hidden, the header's visibility pragma and the version
    script generated from the header): no internal C++ function, kernel handle, device stub or std:: instantiation leaks"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", SO], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert exported == declared_functions(), sorted(set(exported) ^ set(declared_functions()))


def test_python_binding_covers_the_header(built):
    from fidelityfusion_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_functions()
    assert _lib.lib.ffgp_version().decode().startswith("ffgp")


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fidelityfusion_amd import _lib, kernel
    with pytest.raises(_lib.FFGPError):
        _lib.handle(0)
    k = kernel.ARDKernel(2)
    with pytest.raises(_lib.FFGPError):
        k(torch.zeros(3, 2), torch.zeros(3, 2))
    # ffgp_create itself refuses without a device
    out = ctypes.c_void_p()
    assert _lib.lib.ffgp_create(0, ctypes.byref(out)) == -4


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fidelityfusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), os.path.join(dirpath, f)


def test_host_asan_build_walks_the_no_device_paths():
    """`make asan` (SURVEY section 5's sanitizer row, CPU side only -- GPU ASAN is not available on the pool): the whole
    library built with host AddressSanitizer + LeakSanitizer, device code uninstrumented, and a host driver walking the
    create / argument-check / destroy paths"""
    import subprocess
    csrc = os.path.join(ROOT, "fidelityfusion_amd", "csrc")
    p = subprocess.run(["make", "-C", csrc, "-j8", "asan"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "asan_host_check:" in p.stdout and "clean" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr
