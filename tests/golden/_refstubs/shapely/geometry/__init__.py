from .base import (  # noqa: F401
    BaseGeometry,
    LinearRing,
    LineString,
    MultiPolygon,
    Point,
    Polygon,
)
