"""
C-ABI checks that need no GPU: libmdx.so loads, exports every symbol that
include/mdx.h declares, reports errors through mdx_last_error(), and the product
package fails loudly (no CPU fallback) when no HIP device is visible.
"""
import ctypes
import pathlib
import re

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "mdx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from mdhelper_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_library_exports_every_declared_symbol():
    from mdhelper_amd import _lib
    lib = _lib.lib()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.mdx_version() == 100


def test_argument_errors_need_no_device():
    from mdhelper_amd import _lib
    lib = _lib.lib()
    h = ctypes.c_void_p()
    edges = np.array([0.0, 1.0, 0.5])
    rc = lib.mdx_rdf_create(ctypes.byref(h), 0, 2, edges.ctypes.data_as(ctypes.c_void_p), 0, 0, 0)
    assert rc == -1 and b"edges" in lib.mdx_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_product_fails_loudly_without_gpu():
    from mdhelper_amd import _core, _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback|not available"):
        _core.RdfEngine(np.linspace(0, 1, 5))
    with pytest.raises(RuntimeError):
        _lib.require_device(0)
    import mdhelper_amd
    from mdhelper_amd.analysis import Onsager, RadialDistributionFunction, StructureFactor
    u = mdhelper_amd.ArrayUniverse(np.random.rand(4, 50, 3).astype("f4") * 10, [10, 10, 10, 90, 90, 90])
    with pytest.raises(RuntimeError):
        RadialDistributionFunction(u.atoms).run()
    with pytest.raises(RuntimeError):
        StructureFactor(u.atoms, n_points=2).run()
    with pytest.raises(RuntimeError):
        Onsager(u.atoms, reduced=True, temperature=1).run()
    from mdhelper_amd.analysis import EndToEndVector, IntermediateScatteringFunction
    with pytest.raises(RuntimeError):
        EndToEndVector(u.atoms, n_chains=5, n_monomers=10, verbose=False).run()
    with pytest.raises(RuntimeError):
        IntermediateScatteringFunction(u.atoms, n_points=2, n_lags=2, verbose=False).run()
    from mdhelper_amd.algorithm import correlation
    with pytest.raises(RuntimeError):
        correlation.correlation_fft(np.ones(8))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under mdhelper_amd/ may import or load it."""
    for path in (ROOT / "mdhelper_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".hpp", ".h"} or path.name == "Makefile":
            text = path.read_text()
            assert "import oracle" not in text and "from oracle" not in text, path
            assert "rdf_oracle" not in text and "oracle/" not in text, path


def test_one_rocm_runtime_and_a_foreign_one_is_refused():
    """One process, one runtime (reference src/mdhelper/analysis/base.py:137-172 runs in one process by
    construction): libmdx.so's HIP / rocFFT / RCCL calls are bound to the installation it was built for and every
    library is mapped once; a process that loaded another copy first (torch's wheel bundles one) is refused with an
    ImportError that says so; nothing under mdhelper_amd/ or bench.py / __graft_entry__.py imports torch."""
    import subprocess
    import sys
    from mdhelper_amd import _lib
    r = _lib.runtime()
    assert r["problems"] == [], r["problems"]
    root = r["rocm_root"]
    for key in ("libamdhip64", "librocfft", "librccl"):
        assert r[key].startswith(root + "/"), (key, r[key])
        assert r["mapped"][key.split(".")[0]] == [r[key]]
    assert r["rccl_version"] > 20000 and r["hip_runtime_version"] > 0
    s = _lib.runtime_summary()
    assert s["one_runtime"] is True and isinstance(s["libhsa-runtime64"], str)
    child = subprocess.run(
        [sys.executable, "-c",
         "import torch, sys; sys.path.insert(0, %r)\n"
         "from mdhelper_amd import _lib\n"
         "try:\n    _lib.lib()\nexcept ImportError as e:\n    print('REFUSED', e)\n" % str(ROOT)],
        capture_output=True, text=True, timeout=600)
    assert "REFUSED" in child.stdout and "torch/lib/libamdhip64" in child.stdout, child.stdout + child.stderr
    for path in [*(ROOT / "mdhelper_amd").rglob("*.py"), ROOT / "bench.py", ROOT / "__graft_entry__.py"]:
        text = path.read_text()
        assert not re.search(r"^\s*(import torch|from torch)", text, flags=re.M), path


def test_the_runtime_is_told_not_to_page_lock_caller_memory():
    """csrc/mdx_runtime.hip, mdx_process_init: GPU_PINNED_MIN_XFER_SIZE is out of reach from the moment the library is
    loaded (a value the user has set stays) — the HIP runtime then stages pageable copies instead of locking caller
    pages and keeping the registrations (NOTES.md round 5)."""
    import os
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); from mdhelper_amd import _lib; _lib.lib(); "
            "print(os.environ.get('GPU_PINNED_MIN_XFER_SIZE'))" % str(ROOT))
    env = {k: v for k, v in os.environ.items() if k != "GPU_PINNED_MIN_XFER_SIZE"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "1048576"
    out = subprocess.run([sys.executable, "-c", code], env={**env, "GPU_PINNED_MIN_XFER_SIZE": "7"},
                         capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "7"
