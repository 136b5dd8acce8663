"""CPU: result CSV format (model_runner.py:215-228) and box rescale (utils.py:67-69, model_runner.py:140-148)."""
import numpy as np

from ubdvss_amd import ObjectMarkup, ClassifiedObjectMarkup, ModelRunner
from ubdvss_amd.result_saver import markup_csv_string, save_markup_csv
from ubdvss_amd.utils import rescale_bbox


def test_csv_format(tmp_path):
    ms = [ObjectMarkup(np.array([1, 2, 3, 4, 5, 6, 7, 8])), ClassifiedObjectMarkup(np.array([10, 20, 30, 40, 50, 60, 70, 80]), 3)]
    s = markup_csv_string(ms)
    assert s == '1,2,3,4,5,6,7,8,""\n10,20,30,40,50,60,70,80,"",3\n'
    save_markup_csv(tmp_path / "a.csv", ms)
    assert (tmp_path / "a.csv").read_text() == s


def test_rescale_truncates_toward_zero():
    class Meta:
        xscale, yscale = 1.5, 0.5
    assert rescale_bbox(np.array([3, 5, 7, 9, 1, 1, 2, 2]), 1.5, 0.5).tolist() == [4, 2, 10, 4, 1, 0, 3, 1]
    out = ModelRunner.rescale([[ClassifiedObjectMarkup(np.array([3, 5, 7, 9, 1, 1, 2, 2]), 2)]], [Meta()])
    assert out[0][0].bbox.tolist() == [4, 2, 10, 4, 1, 0, 3, 1] and out[0][0].object_type == 2


def test_label_map_builder():
    """f1: SegmapManager.build_segmentation_map / _proper_round (segmap_manager.py:81-133), PIL polygon fill."""
    from PIL import Image
    from ubdvss_amd import SegmapManager
    r = SegmapManager._proper_round(np.array([10.2, 4.7, 30.5, 4.2, 30.9, 20.1, 10.6, 20.8]))
    assert r.tolist() == [10, 4, 31, 4, 31, 21, 10, 21]             # grows outward
    img = Image.new("L", (64, 32))
    ms = [ClassifiedObjectMarkup(np.array([8, 4, 40, 4, 40, 24, 8, 24]), 2), ObjectMarkup(np.array([44, 8, 60, 8, 60, 16, 44, 16]))]
    seg = np.asarray(SegmapManager.build_segmentation_map(img, ms, scale=4))
    assert seg.shape == (8, 16) and seg[3, 5] == 3 and seg[3, 12] == 1 and seg[0, 0] == 0
    assert seg[1:7, 2:11].min() == 3
