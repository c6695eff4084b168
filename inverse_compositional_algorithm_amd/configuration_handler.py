"""config.ini reader / writer with the reference's schema (reference: src/configuration_handler.py:4-67,
config.ini:1-21).  Inline ``# ...`` comments are accepted, so the reference's root config.ini -- which its own
reader cannot parse -- loads too."""
from __future__ import annotations

import configparser

from .image_optimisation import RobustErrorFunctionType
from .transformation import TransformType

_SECTIONS = ("InverseCompositionalAlgorithm", "RobustInverseCompositionalAlgorithm",
             "PyramidalInverseCompositionalAlgorithm")


def create_config_file(filename):
    """reference: configuration_handler.py:4-33 (same defaults)."""
    config = configparser.ConfigParser()
    config[_SECTIONS[0]] = {"TOL": "1e-3", "transform_type": "EUCLIDEAN", "verbose": "False"}
    config[_SECTIONS[1]] = {"TOL": "1e-3", "transform_type": "EUCLIDEAN", "robust_type": "CHARBONNIER",
                            "lambda": "0.0", "verbose": "False"}
    config[_SECTIONS[2]] = {"TOL": "1e-3", "transform_type": "EUCLIDEAN", "pyramid_levels": "2", "nu": "0.5",
                            "robust_type": "QUADRATIC", "lambda": "0.0", "verbose": "False"}
    with open(filename, "w") as fh:
        config.write(fh)


def read_config_file(filename):
    """reference: configuration_handler.py:35-67 -> dict of three dicts, same keys."""
    config = configparser.ConfigParser(inline_comment_prefixes=("#",))
    config.read(filename)
    ica, rica, pica = (config[s] for s in _SECTIONS)
    return {
        "inverse_compositional_algorithm": {
            "TOL": float(ica["TOL"]),
            "transform_type": TransformType[ica["transform_type"].strip()],
            "verbose": ica.getboolean("verbose"),
        },
        "robust_inverse_compositional_algorithm": {
            "TOL": float(rica["TOL"]),
            "transform_type": TransformType[rica["transform_type"].strip()],
            "robust_type": RobustErrorFunctionType[rica["robust_type"].strip()],
            "lambda": float(rica["lambda"]),
            "verbose": rica.getboolean("verbose"),
        },
        "pyramidal_inverse_compositional_algorithm": {
            "TOL": float(pica["TOL"]),
            "transform_type": TransformType[pica["transform_type"].strip()],
            "pyramid_levels": int(pica["pyramid_levels"]),
            "nu": float(pica["nu"]),
            "robust_type": RobustErrorFunctionType[pica["robust_type"].strip()],
            "lambda": float(pica["lambda"]),
            "verbose": pica.getboolean("verbose"),
        },
    }
