"""Stand-in for lxml.etree on top of the stdlib parser (adds getchildren())."""
import xml.etree.ElementTree as _ET


class Element(_ET.Element):
    def getchildren(self):
        return list(self)


class _Tree:
    def __init__(self, root):
        self._root = root

    def getroot(self):
        return self._root

    def find(self, path):
        return self._root.find(path)

    def findall(self, path):
        return self._root.findall(path)

    def iterfind(self, path):
        return self._root.iterfind(path)


def parse(source):
    parser = _ET.XMLParser(target=_ET.TreeBuilder(element_factory=Element))
    with open(source, "rb") as f:
        parser.feed(f.read())
    return _Tree(parser.close())


def tostring(el, **kw):
    return _ET.tostring(el)
