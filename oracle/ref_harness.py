"""Import the reference's own Python modules in the BUILD CONTAINER  --  test infrastructure.

Only usable where /root/reference exists (never on the GPU box).  Used to (a) compare every
oracle function with the reference function it restates and (b) generate the golden
vectors committed under tests/golden/ (tests/golden/make_golden.py).

scikit-image, numba and imageio are not installed in this image and cannot be installed.
The reference modules import them at module scope, so inert placeholders are registered
first: ``numba.jit`` returns the undecorated function (numba never changes results, only
speed) and every scikit-image / imageio attribute raises if it is ever called.  The two
scikit-image calls that sit on the hot path are replaced by the oracle's restatements:

  bicubic_interpolation.bicubic_interpolation_skimage -> oracle.keys_warp  (ref :154-206)
  inverse_compositional_algorithm.rescale             -> oracle.rescale_restated (ref :333-336)

Those two restatements are pinned separately against the traces stored in the reference's
notebooks (tests/test_oracle_kat.py).  Everything else that runs through this harness is
the reference's own code, unmodified, read from /root/reference/src at import time.
"""
from __future__ import annotations

import os
import sys
import types

REF_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isfile(os.path.join(REF_SRC, "inverse_compositional_algorithm.py"))


def _passthrough_jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


def _absent(name):
    def _raise(*a, **k):
        raise RuntimeError(f"{name} is not installed in this image; the harness must not reach it")
    return _raise


_loaded = None


def load():
    """Returns a namespace with the reference modules (cached)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference sources not present")
    from . import ica_oracle as oc

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    mod("numba", jit=_passthrough_jit, njit=_passthrough_jit, prange=range)
    tnames = ["warp", "rescale", "AffineTransform", "SimilarityTransform", "EuclideanTransform",
              "ProjectiveTransform"]
    skt = mod("skimage.transform", **{n: _absent("skimage.transform." + n) for n in tnames})
    mod("skimage.filters", prewitt_h=_absent("prewitt_h"), prewitt_v=_absent("prewitt_v"))
    mod("skimage.exposure", rescale_intensity=_absent("rescale_intensity"))
    sk = mod("skimage", img_as_ubyte=_absent("img_as_ubyte"))
    sk.transform = skt
    mod("imageio")

    old_flag = sys.dont_write_bytecode
    sys.dont_write_bytecode = True  # /root/reference is read-only
    sys.path.insert(0, REF_SRC)
    try:
        import bicubic_interpolation as r_bi
        import configuration_handler as r_cfg
        import constants as r_cts
        import derivatives as r_de
        import image_optimisation as r_io
        import inverse_compositional_algorithm as r_ica
        import transformation as r_tr
        import zoom as r_zm
    finally:
        sys.path.remove(REF_SRC)
        sys.dont_write_bytecode = old_flag

    def _warp(image, params, transformation_type, nanifoutside, delta):
        return oc.warp_params(image, params, transformation_type.value)

    def _rescale(image, scale, **kw):
        assert kw.get("order") == 3 and kw.get("anti_aliasing") is True and kw.get("mode") == "constant"
        return oc.rescale_restated(image, scale)

    r_bi.bicubic_interpolation_skimage = _warp
    r_ica.rescale = _rescale
    _loaded = types.SimpleNamespace(bi=r_bi, cfg=r_cfg, cts=r_cts, de=r_de, io=r_io, ica=r_ica,
                                    tr=r_tr, zm=r_zm)
    return _loaded
