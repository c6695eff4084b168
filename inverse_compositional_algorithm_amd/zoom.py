"""Host-side mirror of the reference's ``zoom`` helpers that the drivers use (reference: src/zoom.py)."""
from __future__ import annotations

import numpy as np

from .transformation import _tt


def zoom_size(nx: int, ny: int, factor: float):
    """reference: zoom.py:8-22 (np.round: half to even)."""
    return int(np.round(nx * factor)), int(np.round(ny * factor))


def zoom_in_parameters(p, transformation_type, nx, ny, nxx, nyy) -> np.ndarray:
    """reference: zoom.py:62-125"""
    nu = max(nxx / nx, nyy / ny)
    q = np.array(p, dtype=np.float64)
    if _tt(transformation_type) == 5:
        q[2] *= nu
        q[5] *= nu
        q[6] /= nu
        q[7] /= nu
    else:
        q[0] *= nu
        q[1] *= nu
    return q
