"""CPU: the C-ABI library loads, exports every symbol include/ubd.h declares, and refuses to run
without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ubd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ubd_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ubdvss_amd import _lib
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ubd.h but not exported"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.ubd_abi_version() == 3


def test_no_cpu_fallback():
    import torch
    from ubdvss_amd import _lib, NetConfig
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    cfg = _lib.UbdConfig(3, 0, 1, 0)
    h = ctypes.c_void_p()
    assert lib.ubd_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"no CPU fallback" in lib.ubd_last_error()
    import ubdvss_amd
    with pytest.raises(RuntimeError):
        ubdvss_amd.Model(NetConfig(grey=False))
    with pytest.raises(RuntimeError):
        ubdvss_amd.SegmapManager.postprocess([[0, 1], [1, 1]])


def test_product_does_not_import_oracle():
    """The shipped package must never touch oracle/ (the checker)."""
    pkg = os.path.join(ROOT, "ubdvss_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "oracle/_build" not in text and "libubd_oracle" not in text, f


def test_netconfig_mirror():
    from ubdvss_amd import NetConfig, PreprocessingType
    c = NetConfig()
    assert (c.is_grey(), c.get_scale(), c.get_min_pixels_for_detection(), c.get_side_multiple(), c.get_max_side(),
            c.is_fml_compatible(), c.is_classification_supported()) == (True, 4, 5, 64, 512, True, False)
    c2 = NetConfig.from_others(c, max_image_side=1024, min_pixels_for_detection=7)
    assert c2.get_max_side() == 1024 and c2.get_min_pixels_for_detection() == 7 and c.get_max_side() == 512
    c3 = NetConfig(class_names=["ean13", "qr"], grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE)
    assert c3.get_n_classes() == 2 and c3.get_class_id("qr") == 1 and c3.is_classification_supported()
    assert c3.get_preprocessing_fn()(255.0) == 1.0


def test_host_staging_copy_is_exact_for_every_size_and_thread_count():
    """ubd_host_memcpy_mt (the staging helper of ModelRunner.predict_stream) is host code: it runs here.  Sizes around the one-thread
    threshold (1 MiB), sizes that are no multiple of the 4 KiB piece rounding, thread counts outside 1..16 (clamped), zero bytes."""
    import numpy as np
    from ubdvss_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for n in (0, 1, 4095, 4096, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 3 * (1 << 20) + 12345, 25165824):
        src = rng.integers(0, 256, n, dtype=np.uint8)
        for threads in (-3, 1, 2, 3, 7, 8, 16, 99):
            dst = np.full(n + 64, 0xAB, dtype=np.uint8)
            assert lib.ubd_host_memcpy_mt(dst.ctypes.data + 32, src.ctypes.data, n, threads) == 0
            assert np.array_equal(dst[32:32 + n], src), (n, threads)
            assert (dst[:32] == 0xAB).all() and (dst[32 + n:] == 0xAB).all(), (n, threads)     # nothing outside [dst, dst + n)
