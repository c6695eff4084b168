"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/ica_hip.h declares.
No compute entry point is called here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ica_hip.h")


@pytest.fixture(scope="module")
def lib():
    from inverse_compositional_algorithm_amd import _lib, build
    build.build()  # hipcc cross-compiles gfx950 on a CPU-only box
    return _lib.load()


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ica_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from inverse_compositional_algorithm_amd import _lib
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ica_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, "ctypes binding and header disagree"


def test_struct_layout_matches_header(lib):
    from inverse_compositional_algorithm_amd._lib import Problem, Stats
    assert C.sizeof(Problem) == 12 * 4 + 6 * 8
    assert C.sizeof(Stats) == 16 * 8 + 8 + 16 * 8 + 16 * 8 + 16 * 8 + 2 * 16 * 4 + 3 * 16 * 8 + 8
    assert Stats.shader_clock_ghz.offset == C.sizeof(Stats) - 8
    assert Problem.tol.offset == 48 and Problem.lambda_ratio.offset == 88


def test_host_only_entry_points(lib):
    assert lib.ica_version() == 4
    assert [lib.ica_nparams(t) for t in (1, 2, 3, 4, 5)] == [2, 3, 4, 6, 8]  # transformation.py:16-32
    assert lib.ica_nparams(0) < 0 and lib.ica_nparams(6) < 0
    w, h = C.c_int(), C.c_int()
    for (W, H, nu, want) in [(584, 388, 0.5, (292, 194)), (135, 1080, 0.5, (68, 540)), (1920, 1080, 0.5, (960, 540)),
                             (5, 3, 0.5, (2, 2))]:  # np.round: half to even (zoom.py:20-21)
        assert lib.ica_zoom_size(W, H, nu, C.byref(w), C.byref(h)) == 0
        assert (w.value, h.value) == want


def test_status_strings_are_the_reference_messages(lib):
    # inverse_compositional_algorithm.py:49, :56, :60
    assert lib.ica_strerror(-1).decode() == "I1 and I2 must be RGB images with channels in the last dimension"
    assert lib.ica_strerror(-2).decode() == "I1 and I2 must have the same dimensions"
    assert lib.ica_strerror(-3).decode() == "TOL must be positive and very small (less than 0.01)"
    assert lib.ica_strerror(0).decode() == "ok"


def test_no_device_is_an_error_not_a_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.ica_device_count() == 0
    ctx = C.c_void_p()
    assert lib.ica_create(0, C.byref(ctx)) == -9  # ICA_ENODEVICE
    from inverse_compositional_algorithm_amd.engine import IcaEngine
    with pytest.raises(RuntimeError, match="no CPU path"):
        IcaEngine(0)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from inverse_compositional_algorithm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
