from . import etree  # noqa: F401
