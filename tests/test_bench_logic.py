"""bench.py's host-side logic that needs no GPU: the traffic figure is reported only for the library it was measured on."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_traffic_is_null_unless_measured_on_the_loaded_library(tmp_path, monkeypatch):
    bench = load_bench()
    cfg = bench.CONFIGS[4]
    f = tmp_path / "traffic.json"
    f.write_text(json.dumps({"library_fingerprint": "abc", "kernels": {
        bench.kernel_key(cfg): {"fetch_bytes_over_algorithmic_bytes": 1.12, "source": "somewhere"}}}))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    monkeypatch.setattr(bench, "library_fingerprint", lambda: "abc")
    assert bench.measured_traffic(cfg) == (1.12, "somewhere")
    assert bench.measured_traffic(cfg, "u8")[0] is None                # the uint8-storage kernel was not measured: no figure
    monkeypatch.setattr(bench, "library_fingerprint", lambda: "another build")
    ratio, why = bench.measured_traffic(cfg)
    assert ratio is None and "stale" in why
    monkeypatch.setattr(bench, "library_fingerprint", lambda: None)   # an experiment build (ICA_LIB): no stamp
    assert bench.measured_traffic(cfg)[0] is None
    f.write_text(json.dumps({"kernels": {}}))                          # a file without a fingerprint vouches for nothing
    monkeypatch.setattr(bench, "library_fingerprint", lambda: "abc")
    assert bench.measured_traffic(cfg)[0] is None


def test_the_loaded_library_has_a_fingerprint(monkeypatch):
    bench = load_bench()
    monkeypatch.delenv("ICA_LIB", raising=False)
    from inverse_compositional_algorithm_amd import build
    if not os.path.isfile(build.STAMP):
        return  # nothing built yet in this checkout
    assert bench.library_fingerprint() == build.read_stamp()[0]
    monkeypatch.setenv("ICA_LIB", "/somewhere/lib_variant.so")
    assert bench.library_fingerprint() is None
