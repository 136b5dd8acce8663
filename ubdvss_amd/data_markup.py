"""Plain records returned by the postprocess.

API-compatible with the reference's ``ObjectMarkup`` / ``ClassifiedObjectMarkup``
(semantic_segmentation/data_markup.py:9-36): attribute ``bbox`` (8 ints x1,y1,...,x4,y4 of the rotated
quadrilateral), attribute ``object_type`` (int class id) on the classified variant, and
``create_same_markup(new_bbox)`` which clones the record around another box.
"""
import numpy as np


class ObjectMarkup(object):
    __slots__ = ("bbox",)

    def __init__(self, bbox):
        self.bbox = bbox

    def _clone_args(self):
        return ()

    def create_same_markup(self, new_bbox):
        """Same kind of record (and same class id, if any) around ``new_bbox``."""
        return type(self)(new_bbox, *self._clone_args())

    def as_points(self):
        """The quadrilateral as a (4, 2) integer array."""
        return np.asarray(self.bbox).reshape(-1, 2)

    def __repr__(self):
        extra = "".join(f", {a}" for a in self._clone_args())
        return f"{type(self).__name__}({list(np.asarray(self.bbox).tolist())}{extra})"


class ClassifiedObjectMarkup(ObjectMarkup):
    __slots__ = ("object_type",)

    def __init__(self, bbox, object_type):
        ObjectMarkup.__init__(self, bbox)
        self.object_type = int(object_type)          # plain int: numpy integers upset some consumers

    def _clone_args(self):
        return (self.object_type,)
