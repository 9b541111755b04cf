"""Coordinate-holding stand-in for shapely (see ../README.md). Not GEOS."""
from . import geometry, ops, prepared, strtree, validation, vectorized  # noqa: F401

__version__ = "2.0.0-standin"
