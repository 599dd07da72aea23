import os
import pathlib
import sys

import pytest


ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Before libmdx.so is loaded: a SIGABRT inside a runtime library writes the native call stack — and the last
    # lines the runtime wrote to the CAPTURED stderr, which die with the process otherwise — to the stderr pytest
    # itself was started with (csrc/mdx_runtime.hip, MDX_ABORT_TRACE).
    fd = 2
    try:
        cap = config.pluginmanager.getplugin("capturemanager")._global_capturing
        saved = getattr(cap.err, "targetfd_save", None)
        if isinstance(saved, int) and saved > 2:
            fd = saved
            os.set_inheritable(fd, False)
    except Exception:
        pass
    os.environ.setdefault("MDX_ABORT_TRACE", str(fd))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _runtime_problems():
    """One process, one ROCm runtime: the HIP / HSA / rocFFT / RCCL libraries are each mapped
    once, from the installation libmdx.so was built for — the stack bench.py and smoke() run on."""
    if "torch" in sys.modules:
        return ["`torch` was imported into the pytest process (its wheel bundles another ROCm runtime)"]
    from mdhelper_amd import _lib
    if _lib._lib is None:                   # nothing loaded the library (yet): nothing to check
        return []
    return _lib.runtime()["problems"]


def pytest_collection_finish(session):
    # collection imports every test module: none of them may have brought a second runtime in
    problems = _runtime_problems()
    if problems:
        pytest.exit("ROCm runtime check failed after collection: " + "; ".join(problems), returncode=3)


@pytest.fixture(autouse=True)
def _one_rocm_runtime(request):
    yield
    if request.node.get_closest_marker("gpu") is not None:
        problems = _runtime_problems()
        assert not problems, "ROCm runtime check: " + "; ".join(problems)


def pytest_terminal_summary(terminalreporter):
    from mdhelper_amd import _lib
    if _lib._lib is not None:
        r = _lib.runtime_summary()
        terminalreporter.write_line("libmdx runtime: " + ", ".join(f"{k}={v}" for k, v in r.items()))
