import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    # torch.jit is only used by test_tensor_loader.py to WRITE archives in the reference's format
    config.addinivalue_line("filterwarnings", "ignore:.*torch.jit.*:DeprecationWarning")
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _library_build_ids():
    """(expected id of the tree, {library path: id baked into it}) for the in-tree HIP libraries that exist."""
    import ctypes
    import importlib.util
    spec = importlib.util.spec_from_file_location("build_id", os.path.join(ROOT, "tools", "build_id.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = {}
    try:
        import torch  # noqa: F401  same load order as dorado_amd.capi.lib(): the library shares torch's libamdhip64 (same SONAME)
    except Exception:  # pragma: no cover
        pass
    for name, suffix in (("libmibc.so", ""), ("libmibc_dbg.so", "-dbg")):
        path = os.path.join(ROOT, "dorado_amd", name)
        if not os.path.exists(path):
            continue
        try:
            lib = ctypes.CDLL(path)
            lib.mibc_build_id.restype = ctypes.c_char_p
            got[path] = (lib.mibc_build_id().decode(), suffix)
        except (OSError, AttributeError) as exc:      # not loadable here / predates the build id: stale by definition
            got[path] = (f"<no build id: {exc}>", suffix)
    return mod.build_id(), got


def pytest_sessionstart(session):
    """built artefacts travel with the tree (git-ignored, shipped to the GPU box), so a stale libmibc.so
    would pass silently if mtimes ever mis-ordered.  The library carries a hash of the sources it was compiled from
    (mibc_build_id, tools/build_id.py); a library built from other sources ends the session before any test runs."""
    want, got = _library_build_ids()
    stale = {p: i for p, (i, suf) in got.items() if i != want + suf}
    if stale:
        pytest.exit(f"stale HIP library (sources hash to {want}): {stale} — rebuild with "
                    "`python -c 'import __graft_entry__ as g; g.build()'`", returncode=3)
