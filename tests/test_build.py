"""The in-tree build of libica_hip.so can never ship a binary that is older than its sources: the rebuild decision
hashes every file the library is compiled from (no GPU needed)."""
import os
import re
import shutil

from inverse_compositional_algorithm_amd import build as b


def test_every_include_of_the_library_is_a_tracked_dependency():
    names = {os.path.basename(d) for d in b.deps()}
    assert {"ica_api.hip", "ica_kernels.hpp", "ica_math.hpp", "ica_pyramid.hpp", "ica_hip.h"} <= names
    for d in b.deps():
        for inc in re.findall(r'#include\s+"([^"]+)"', open(d).read()):
            assert os.path.basename(inc) in names, f"{d} includes {inc}, which the build does not track"


def test_an_edit_to_any_source_or_flag_makes_the_library_stale(tmp_path, monkeypatch):
    # a private copy of the source tree, so that the real sources are never touched
    here = tmp_path / "pkg"
    shutil.copytree(b.CSRC, here / "csrc")
    shutil.copytree(b.INCLUDE, tmp_path / "include")
    monkeypatch.setattr(b, "HERE", str(here))
    monkeypatch.setattr(b, "CSRC", str(here / "csrc"))
    monkeypatch.setattr(b, "INCLUDE", str(tmp_path / "include"))
    monkeypatch.setattr(b, "LIB", str(here / "libica_hip.so"))
    monkeypatch.setattr(b, "STAMP", str(here / "libica_hip.so.stamp"))
    monkeypatch.setattr(b, "SRC", str(here / "csrc" / "ica_api.hip"))
    monkeypatch.delenv("ICA_EXTRA_FLAGS", raising=False)
    assert b.is_stale()  # no library yet
    (here / "libica_hip.so").write_bytes(b"not a real library")
    (here / "libica_hip.so.stamp").write_text(b.fingerprint() + "\n" + b.lib_digest() + "\n")
    assert not b.is_stale()
    # a library that some other tool overwrote (a partial link, a truncated copy) is stale although no source changed
    (here / "libica_hip.so").write_bytes(b"not a real libr")
    assert b.is_stale()
    (here / "libica_hip.so").write_bytes(b"a foreign file of another size")
    assert b.is_stale()
    (here / "libica_hip.so").write_bytes(b"not a real library")
    assert not b.is_stale()
    # a stamp from before the digest was recorded (one line) does not vouch for the binary
    (here / "libica_hip.so.stamp").write_text(b.fingerprint() + "\n")
    assert b.is_stale()
    (here / "libica_hip.so.stamp").write_text(b.fingerprint() + "\n" + b.lib_digest() + "\n")
    for name in sorted(os.listdir(here / "csrc")) + ["../../include/ica_hip.h"]:
        path = (here / "csrc" / name).resolve()
        old = path.read_bytes()
        path.write_bytes(old + b"\n// edited\n")
        assert b.is_stale(), f"an edit to {name} does not trigger a rebuild"
        path.write_bytes(old)
        assert not b.is_stale()
    monkeypatch.setenv("ICA_EXTRA_FLAGS", "-DICA_ABLATE=1")
    assert b.is_stale()  # other flags, other binary


def test_no_tool_names_the_product_library():
    """Experiment and analysis scripts build into tools/abl/ or a scratch directory; none of them may write to (or even
    name) the product library: a partial link left there once shipped a broken binary to the GPU box."""
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(b.__file__))), "tools")
    for root, _, files in os.walk(tools):
        for name in files:
            if not name.endswith((".py", ".sh", ".hip")):
                continue
            with open(os.path.join(root, name), errors="replace") as fh:
                assert "libica_hip.so" not in fh.read(), f"tools/{name} names the product library"
