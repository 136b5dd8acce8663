"""Return types of the postprocess -- mirrors semantic_segmentation/data_markup.py:9-36."""


class ObjectMarkup:
    """One found object: ``bbox`` = 8 ints x1,y1,...,x4,y4 (rotated quadrilateral)."""
    __slots__ = ['bbox']

    def __init__(self, bbox):
        self.bbox = bbox

    def create_same_markup(self, new_bbox):
        return ObjectMarkup(new_bbox)


class ClassifiedObjectMarkup(ObjectMarkup):
    """Object with its voted class id (data_markup.py:24-36)."""
    __slots__ = ['object_type']

    def __init__(self, bbox, object_type):
        super().__init__(bbox)
        self.object_type = int(object_type)

    def create_same_markup(self, new_bbox):
        return ClassifiedObjectMarkup(new_bbox, self.object_type)
