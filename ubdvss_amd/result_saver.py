"""Result output of predict.py -- host mirror of semantic_segmentation/model_runner.py:215-228 (CSV of the found
quadrilaterals) and of the rescale step model_runner.py:140-148 / utils.py:67-69."""
from .data_markup import ClassifiedObjectMarkup


def markup_csv_string(markups):
    """One line per object: x1,y1,...,x4,y4,""[,type]  (model_runner.py:215-228)."""
    out = ''
    for markup in markups:
        out += '{:d},{:d},{:d},{:d},{:d},{:d},{:d},{:d},""'.format(*[int(xy) for xy in markup.bbox])
        if isinstance(markup, ClassifiedObjectMarkup):
            out += f',{markup.object_type}\n'
        else:
            out += '\n'
    return out


def save_markup_csv(filename, markups):
    with open(filename, "w") as text_file:
        text_file.write(markup_csv_string(markups))
