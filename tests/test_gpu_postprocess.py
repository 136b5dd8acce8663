"""GPU parity: device postprocess (threshold -> external components -> contourArea filter ->
minAreaRect -> boxPoints -> rounded quads -> class vote) vs the C oracle, bit-exact on the integer
outputs, through the C ABI (Model.postprocess_on_device / SegmapManager.postprocess / ModelRunner)."""
import os

import numpy as np
import pytest
import torch
from scipy import ndimage as ndi

from oracle import cv_post as ocv, net_numpy as onet
from ubdvss_amd import NetConfig, Model, ModelRunner, SegmapManager, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["lds", "lds_512", "lds_split", "global"], autouse=True)
def front_end(request, monkeypatch):
    """Every case runs through every code path: "lds" = the whole postprocess of a batch as ONE launch (maps of <= 16384
    pixels: front end, boxes, class vote and emit in one block per image), "lds_split" = the one-launch front end followed by
    the separate boxes / vote / emit kernels (UBD_PP_SPLIT; what maps too tall for the in-block box scratch take) with the
    ONE-LANE form of the box fit (UBD_PP_SERIAL_TAIL: hull clean-up and rotating calipers on lane 0 over LDS arrays, what hulls
    of more than 64 vertices take; the other two variants run the wave-uniform register form), "global" = the multi-launch
    global-memory path that larger maps take (UBD_PP_GLOBAL forces it at any size); "lds_512" = the one-launch form in 512-thread
    blocks (UBD_PP_THREADS_512): the block shape the job has when it rides inside the stem kernel."""
    from ubdvss_amd import segmap_manager
    segmap_manager._reset_handles()          # the switches are read when a handle is created
    monkeypatch.delenv("UBD_PP_GLOBAL", raising=False)
    monkeypatch.delenv("UBD_PP_SPLIT", raising=False)
    monkeypatch.delenv("UBD_PP_SERIAL_TAIL", raising=False)
    monkeypatch.delenv("UBD_PP_THREADS_512", raising=False)
    if request.param == "lds_512":
        monkeypatch.setenv("UBD_PP_THREADS_512", "1")
    elif request.param == "global":
        monkeypatch.setenv("UBD_PP_GLOBAL", "1")
    elif request.param == "lds_split":
        monkeypatch.setenv("UBD_PP_SPLIT", "1")
        monkeypatch.setenv("UBD_PP_SERIAL_TAIL", "1")
    yield request.param
    segmap_manager._reset_handles()          # no handle created under a switch outlives the test


def _model(ncls=0, cin=3):
    cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)] if ncls else None, grey=(cin == 1))
    return Model(cfg, seed=0)


def _run(model, logits, thr=0.0, scale=4, min_area=5, cap=256):
    lt = torch.from_numpy(np.ascontiguousarray(logits, dtype=np.float32)).cuda()
    bmap, quads, classes, counts = model.postprocess_on_device(lt, thr, scale, min_area, cap=cap)
    counts = counts.cpu().numpy(); quads = quads.cpu().numpy()
    classes = classes.cpu().numpy() if classes is not None else None
    out = []
    for i in range(len(counts)):
        n = min(int(counts[i]), cap)
        out.append((quads[i, :n], classes[i, :n] if classes is not None else None))
    return bmap.cpu().numpy(), out, counts


def _compare(model, logits, n_cls, thr=0.0, min_area=5, cap=256):
    bmap, out, counts = _run(model, logits, thr, 4, min_area, cap)
    det = (logits[..., 0] > thr).astype(np.int32)
    assert np.array_equal(bmap, det)
    for i in range(logits.shape[0]):
        q, c = ocv.postprocess(det[i], logits[i, ..., 1:] if n_cls else None, 4, min_area)
        assert int(counts[i]) == len(q), (i, int(counts[i]), len(q))
        assert np.array_equal(out[i][0], q), (i, out[i][0], q)
        if n_cls:
            assert np.array_equal(out[i][1], c), (i, out[i][1], c)


def test_golden_rectangles(golden_dir, manifest):
    maps = np.load(os.path.join(golden_dir, "post_rect.npz"))["maps"].astype(np.int32)
    lg = synthetic.logits_from_maps(maps, 4, seed=5, noise=0.0)
    bmap, out, counts = _run(_model(4), lg, thr=0.0)
    for (q, c), gold in zip(out, manifest["post_rect"]):
        assert q.tolist() == gold["quads"] and c.tolist() == gold["classes"]


def test_golden_stress_maps(golden_dir, manifest):
    stress = np.load(os.path.join(golden_dir, "post_stress_maps.npy"))
    lg = np.where(stress[..., None] > 0, 1.0, -1.0).astype(np.float32)
    _, out, _ = _run(_model(0), lg, cap=2048)
    for (q, _), gold in zip(out, manifest["post_stress"]):
        assert q.tolist() == gold


@pytest.mark.parametrize("n_cls", [0, 3])
def test_rectangles_vs_oracle(n_cls):
    maps = synthetic.rectangle_maps(21, 16, 128, 128, n_classes=n_cls)
    lg = synthetic.logits_from_maps(maps, n_cls, seed=22)
    _compare(_model(n_cls), lg, n_cls)


def test_random_noise_maps_vs_oracle():
    """Pathological maps: thousands of tiny components, holes, nesting, frame contact."""
    rng = np.random.default_rng(5)
    for p, hw in [(0.5, (64, 64)), (0.62, (48, 80)), (0.8, (40, 40)), (0.3, (64, 32))]:
        m = (rng.random((6,) + hw) < p)
        m[1] = ndi.binary_dilation(m[1]); m[2] = ndi.binary_erosion(m[2]); m[3] = ndi.binary_closing(m[3])
        lg = np.where(m[..., None], 2.0, -2.0).astype(np.float32)
        _compare(_model(0), lg, 0, cap=2048)
        _compare(_model(0), lg, 0, min_area=0, cap=2048)


def test_ragged_and_tiny_maps_vs_oracle():
    """Map sizes that are not multiples of the 64-pixel wave (ballot tails), single rows / columns, a 1 x 1 map, and the
    largest map the one-launch front end takes (128 x 128) filled with noise; with and without classes."""
    rng = np.random.default_rng(11)
    for hw in [(1, 1), (1, 37), (29, 1), (3, 5), (17, 33), (50, 130), (128, 128)]:
        m = rng.random((3,) + hw) < 0.55
        if hw[0] >= 3 and hw[1] >= 3:
            m[1] = ndi.binary_closing(m[1])
        lg = np.where(m[..., None], 1.5, -1.5).astype(np.float32)
        _compare(_model(0), lg, 0, min_area=0, cap=4200)
        lg2 = np.concatenate([lg, rng.normal(0, 1, (3,) + hw + (2,)).astype(np.float32)], axis=-1)
        _compare(_model(2), lg2, 2, min_area=1, cap=4200)


def test_random_shape_soak_vs_oracle():
    """Random map sizes (1 .. 160 per side: below, at and above the 16384-pixel limit of the one-launch form), random density, random
    morphology (noise, closed / opened noise, blobs with holes), min_area 0 / 5, with and without classes: every case bit-exact against
    the oracle through the fixture's three code paths.  UBD_PP_SOAK_CASES scales it (default 24 cases per path)."""
    rng = np.random.default_rng(2024)
    for case in range(int(os.environ.get("UBD_PP_SOAK_CASES", "24"))):
        h, w = int(rng.integers(1, 161)), int(rng.integers(1, 161))
        n = int(rng.integers(1, 5))
        p = float(rng.uniform(0.05, 0.95))
        m = rng.random((n, h, w)) < p
        kind = int(rng.integers(0, 4))
        if h >= 3 and w >= 3:
            for i in range(n):
                if kind == 1: m[i] = ndi.binary_closing(m[i])
                elif kind == 2: m[i] = ndi.binary_opening(m[i])
                elif kind == 3: m[i] = ndi.binary_dilation(ndi.binary_erosion(m[i], iterations=2), iterations=3) & ~(rng.random((h, w)) < 0.02)
        n_cls = int(rng.choice([0, 0, 3]))
        lg = np.where(m[..., None], 1.0, -1.0).astype(np.float32)
        if n_cls:
            lg = np.concatenate([lg, rng.normal(0, 1, (n, h, w, n_cls)).astype(np.float32)], axis=-1)
        _compare(_model(n_cls), lg, n_cls, min_area=int(rng.choice([0, 5])), cap=8192)


def test_isolated_pixel_grids_on_odd_sizes():
    """Most components a map can hold: isolated pixels on every other row and column = ceil(h/2) * ceil(w/2) external
    components, which exceeds h*w/4 when a side is odd (127 x 127: 4096).  Every one has contourArea 0, so min_area -1
    keeps them all; the image after a full grid must be untouched (per-image root slices do not overlap)."""
    for hw in [(127, 127), (128, 127), (127, 128), (1, 255), (5, 3)]:
        m = np.zeros((3,) + hw, bool)
        m[0, ::2, ::2] = 1
        m[1, 1::2, 1::2] = 1
        m[2, hw[0] // 3:hw[0] // 3 + 1, :] = 1
        lg = np.where(m[..., None], 1.0, -1.0).astype(np.float32)
        _compare(_model(0), lg, 0, min_area=-1, cap=4200)
        bmap, out, counts = _run(_model(0), lg, min_area=-1, cap=4200)
        assert counts[0] == ((hw[0] + 1) // 2) * ((hw[1] + 1) // 2)


def test_nested_rings_and_class_vote():
    m = np.zeros((1, 64, 64), bool)
    m[0, 4:60, 4:60] = 1; m[0, 10:54, 10:54] = 0; m[0, 16:48, 16:48] = 1; m[0, 22:42, 22:42] = 0; m[0, 28:36, 28:36] = 1
    rng = np.random.default_rng(3)
    lg = np.concatenate([np.where(m[..., None], 3.0, -3.0), rng.normal(0, 1, (1, 64, 64, 3))], axis=-1).astype(np.float32)
    lg[0, 20:44, 20:44, 2] += 4.0          # enclosed pixels vote too (filled contour includes holes)
    _compare(_model(3), lg, 3)


def test_threshold_strictness_and_probability():
    model = _model(0)
    lg = np.full((1, 16, 16, 1), -1.0, np.float32)
    lg[0, 4:12, 4:12, 0] = 0.0               # exactly at the p=0.5 threshold: NOT positive (strict >)
    _, out, counts = _run(model, lg, thr=float(onet.logit_threshold(0.5)))
    assert counts[0] == 0
    lg[0, 4:12, 4:12, 0] = 2.5
    thr = float(onet.logit_threshold(0.9))   # log(9) = 2.197
    _, out, counts = _run(model, lg, thr=thr)
    assert counts[0] == 1 and sorted(map(tuple, out[0][0].reshape(4, 2).tolist())) == [(16, 16), (16, 44), (44, 16), (44, 44)]


def test_capacity_overflow_is_reported():
    rng = np.random.default_rng(1)
    m = np.zeros((1, 64, 64), bool)
    for by in range(0, 60, 6):
        for bx in range(0, 60, 6):
            m[0, by:by + 4, bx:bx + 4] = 1          # 100 blocks of area 9
    lg = np.where(m[..., None], 1.0, -1.0).astype(np.float32)
    _, _, counts = _run(_model(0), lg, cap=16)
    assert counts[0] == 100
    cfg = NetConfig(grey=False)
    runner = ModelRunner(cfg, max_objects_per_image=16)
    assert rng is not None and runner.logit_threshold == 0


def test_segmap_manager_static_api():
    m = np.zeros((32, 32, 1), np.int64)
    m[5:12, 8:20, 0] = 1
    objs = SegmapManager.postprocess(m, None, scale=4, min_area_threshold=5)
    assert len(objs) == 1 and sorted(map(tuple, np.asarray(objs[0].bbox).reshape(4, 2).tolist())) == [(32, 20), (32, 44), (76, 20), (76, 44)]
    cl = np.zeros((32, 32, 3), np.float32); cl[..., 1] = 2.0
    objs = SegmapManager.postprocess(m, cl, scale=4)
    assert objs[0].object_type == 1
    ref_q, ref_c = ocv.postprocess(m[..., 0], cl, 4, 5)
    assert np.array_equal(np.asarray(objs[0].bbox), ref_q[0]) and objs[0].object_type == ref_c[0]


def test_model_runner_predict_end_to_end():
    """image -> boxes with the reference's return triple (model_runner.py:105-138); logits compared
    with tolerance, then the postprocess of the GPU's own map compared bit-exact with the oracle."""
    cfg = NetConfig(class_names=["a", "b"], grey=False)
    model = Model(cfg, seed=3)
    w = onet.init_weights(31, 3, 2, bias_scale=0.3)
    model.set_weights(w)
    labels = synthetic.rectangle_maps(8, 3, 32, 32)
    x = synthetic.textured_images(4, labels, 4, 3).astype(np.float32) / 127.5 - 1.0
    runner = ModelRunner(cfg, pixel_threshold=0.5)
    det, cls_logits, found = runner.predict(model, x)
    ref = onet.forward(x.astype(np.float64), w)
    assert det.shape == (3, 32, 32, 1) and cls_logits.shape == (3, 32, 32, 2)
    far = np.abs(ref[..., 0]) > 1e-3
    assert np.array_equal(det[..., 0][far], (ref[..., 0] > 0)[far].astype(det.dtype))
    for i in range(3):
        q, c = ocv.postprocess(det[i], cls_logits[i], 4, 5)
        assert len(found[i]) == len(q)
        for o, qq, cc in zip(found[i], q, c):
            assert np.array_equal(np.asarray(o.bbox), qq) and o.object_type == cc


def test_large_map_1024_input_shape():
    """cfg5 shape (1024^2 input -> 256x256 map) and idempotence of the quads under re-run."""
    maps = synthetic.rectangle_maps(33, 4, 256, 256)
    lg = synthetic.logits_from_maps(maps, 0, seed=1)
    model = _model(0)
    _compare(model, lg, 0)
    a = _run(model, lg)[1]; b = _run(model, lg)[1]
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(a, b))


def _same_results(got, ref):
    """(logits, binary map, quads, classes or None, counts) of two runs: equal, lists compared up to their counts (result
    buffers of the pipelined runner are reused, so entries behind a list's end may be stale)."""
    lg, bm, q, c, cnt = got
    rlg, rbm, rq, rc, rcnt = ref
    if not (torch.equal(lg, rlg) and torch.equal(bm, rbm) and torch.equal(cnt, rcnt)):
        return False
    live = torch.arange(q.shape[1], device=q.device)[None, :] < rcnt[:, None]
    ok = bool(((q == rq) | ~live[..., None]).all())
    if c is not None:
        ok = ok and bool(((c == rc) | ~live).all())
    return ok


def test_pipelined_runner_matches_serial():
    """Pipeline (the postprocess of batch k rides in the stem kernel of batch k+1's forward pass, ubd_forward_postprocess) gives
    the same results as the serial path, batch after batch; big batches so that the one-kernel stem -- and with it the in-kernel
    postprocess -- really runs (small ones take the back-to-back fallback, covered by the second loop)."""
    for n_cls, n_img, side in ((0, 4, 128), (0, 34, 256), (3, 34, 256)):       # 34 x 16 strips = 544 >= 2 x 256 CUs: fused stem
        cfg = NetConfig(class_names=[f"c{i}" for i in range(n_cls)] if n_cls else None, grey=False)
        model = Model(cfg, seed=5)
        model.set_weights(onet.init_weights(41, 3, n_cls, bias_scale=0.3))
        serial, piped = ModelRunner(cfg), ModelRunner(cfg, pipelined=True)
        batches = []
        for k in range(5):
            labels = synthetic.rectangle_maps(70 + k, n_img, side // 4, side // 4)
            batches.append(torch.from_numpy(synthetic.textured_images(80 + k, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda())
        ref = [[t.clone() if t is not None else None for t in serial.predict_on_device(model, b)] for b in batches]
        got = []
        for k, b in enumerate(batches):
            got.append(piped.predict_on_device(model, b))
            if k >= 1:                                            # batch k-1 is complete once this call has run
                torch.cuda.synchronize()
                assert _same_results(got[k - 1], ref[k - 1]), (n_cls, n_img, k - 1)
        piped.synchronize()                                       # flushes the last batch's postprocess
        assert _same_results(got[4], ref[4]), (n_cls, n_img, 4)


# ---------------------------------------------------------------------------------------------- stress / determinism
STRESS_LAUNCHES = int(os.environ.get("UBD_PP_STRESS_LAUNCHES", "500"))


def _stress_inputs(golden_dir):
    """The batch that exposed the round-2 flatten race, the golden rectangles and 128 x 128 noise maps."""
    rng = np.random.default_rng(77)
    cases = []
    maps = synthetic.rectangle_maps(21, 16, 128, 128)
    cases.append(("rect21", synthetic.logits_from_maps(maps, 0, seed=22), 256))
    gmaps = np.load(os.path.join(golden_dir, "post_rect.npz"))["maps"].astype(np.int32)
    cases.append(("golden", synthetic.logits_from_maps(gmaps, 0, seed=5, noise=0.0), 256))
    noise = rng.random((4, 128, 128)) < 0.55
    noise[1] = ndi.binary_closing(noise[1]); noise[2] = rng.random((128, 128)) < 0.5
    cases.append(("noise", np.where(noise[..., None], 1.5, -1.5).astype(np.float32), 4200))
    return cases


def _oracle_lists(lg, min_area, cap):
    det = (lg[..., 0] > 0).astype(np.int32)
    quads = np.zeros((lg.shape[0], cap, 8), np.int32)
    counts = np.zeros((lg.shape[0],), np.int32)
    for i in range(lg.shape[0]):
        q, _ = ocv.postprocess(det[i], None, 4, min_area)
        counts[i] = len(q)
        if len(q):
            quads[i, :len(q)] = np.asarray(q).reshape(-1, 8)
    return quads, counts


@pytest.mark.parametrize("mode", ["alone", "concurrent_forward", "poison"])
def test_postprocess_stress_deterministic(golden_dir, front_end, mode, monkeypatch):
    """>= 500 launches per input through each front end; EVERY launch must reproduce the oracle's lists (a rare race in the
    union-find phases shows up as one wrong quad in one launch -- round 2's driver run).  `concurrent_forward`: a forward
    pass of another batch runs on a second stream the whole time, as in the pipelined runner (blocks of both kernels share
    CUs, LDS contents of earlier blocks differ).  `poison`: the one-launch front end starts from poisoned LDS and checks that the
    forest is flat and every slot it reads was written (an integrity failure makes counts impossible)."""
    if mode == "poison":
        if front_end != "lds":
            pytest.skip("the poison hook belongs to the one-launch LDS front end")
        monkeypatch.setenv("UBD_PP_POISON", "1")
    model = _model(0)
    side = torch.cuda.Stream()
    fwd_in = torch.from_numpy(synthetic.noise_images(3, 8, 512, 512, 3)).cuda() if mode == "concurrent_forward" else None
    for name, lg, cap in _stress_inputs(golden_dir):
        ref_q, ref_c = _oracle_lists(lg, 5, cap)
        ref_q, ref_c = torch.from_numpy(ref_q).cuda(), torch.from_numpy(ref_c).cuda()
        lt = torch.from_numpy(lg).cuda()
        outs = model.alloc_postprocess_outputs(lg.shape[0], lg.shape[1], lg.shape[2], cap)
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        first_bad = None
        for it in range(STRESS_LAUNCHES):
            if fwd_in is not None and it % 4 == 0:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    model.predict_on_device(fwd_in)
            outs[1].fill_(-7)                                            # stale results of the previous launch cannot pass
            _, quads, _, counts = model.postprocess_on_device(lt, 0.0, 4, 5, cap=cap, outputs=outs)
            ok = torch.equal(counts, ref_c)
            if ok:
                live = torch.arange(cap, device="cuda")[None, :, None] < ref_c[:, None, None]
                ok = bool(((quads == ref_q) | ~live).all())
            if not ok and first_bad is None:
                first_bad = (it, counts.cpu().numpy().copy(), quads.cpu().numpy().copy())
            bad += 0 if ok else 1
        torch.cuda.synchronize()
        assert int(bad) == 0, (name, front_end, mode, int(bad), first_bad[0], first_bad[1], ref_c.cpu().numpy())


def test_postprocess_stress_in_pipelined_runner():
    """The same check inside ModelRunner(pipelined=True) -- the bench's headline path: the postprocess of batch k runs in the
    first blocks of the stem kernel of batch k+1 (512-thread blocks, LDS shared with the stem, other blocks of the kernel
    already convolving).  300 steps over a ring of batches at the headline shape (32 x 512 x 512); every step's lists equal the
    serial runner's lists of that batch, and -- directly, not through the serial HIP path -- the ORACLE's lists (oracle/cv_post.c
    on the host-thresholded logits of the batch)."""
    cfg = NetConfig(grey=False)
    model = Model(cfg, seed=5)
    w = onet.init_weights(41, 3, 0, bias_scale=0.0)               # (with bias_scale 0.3 these weights give all-negative maps: nothing to find)
    model.set_weights(w)
    serial, piped = ModelRunner(cfg, max_objects_per_image=4200), ModelRunner(cfg, pipelined=True, max_objects_per_image=4200)
    batches, ref, oracle = [], [], []
    for k in range(4):
        labels = synthetic.rectangle_maps(170 + k, 32, 128, 128)
        b = torch.from_numpy(synthetic.textured_images(180 + k, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
        batches.append(b)
        lg, bmap, quads, _, counts = serial.predict_on_device(model, b)
        ref.append((bmap.clone(), quads.clone(), counts.clone()))
        oq, oc = _oracle_lists(np.where(lg.cpu().numpy() > serial.logit_threshold, 1.0, -1.0).astype(np.float32), cfg.get_min_pixels_for_detection(), 4200)
        oracle.append((torch.from_numpy(oq).cuda(), torch.from_numpy(oc).cuda()))
    torch.cuda.synchronize()
    assert sum(int(o[1].sum()) for o in oracle) > 100              # the oracle found objects on these maps (several per image)
    assert model._lib.ubd_num_cus(model._h) * 2 <= 32 * 32        # 1024 strips: the one-kernel stem (and the in-kernel postprocess) is in use
    pending = None
    bad, bad_oracle = [], []
    for it in range(300):
        k = it % 4
        out = piped.predict_on_device(model, batches[k])
        if pending is not None:                                   # the previous step's lists: complete now that this call is enqueued
            pk, pout = pending
            torch.cuda.synchronize()
            _, bmap, quads, _, counts = pout
            live = torch.arange(quads.shape[1], device="cuda")[None, :, None] < ref[pk][2][:, None, None]
            if not (torch.equal(bmap, ref[pk][0]) and torch.equal(counts, ref[pk][2]) and bool(((quads == ref[pk][1]) | ~live).all())):
                bad.append(it - 1)
            if not (torch.equal(counts, oracle[pk][1]) and bool(((quads == oracle[pk][0]) | ~live).all())):     # the pipelined lists vs the oracle's
                bad_oracle.append(it - 1)
        pending = (k, out)
    piped.synchronize()
    assert not bad, bad[:10]
    assert not bad_oracle, bad_oracle[:10]


@pytest.mark.parametrize("grey,u8", [(False, False), (False, True), (True, False), (True, True)])
def test_postprocess_inside_every_stem_kernel_variant(grey, u8, monkeypatch):
    """ubd_forward_postprocess through every instantiation of the one-kernel stem (grey / RGB, fp32 fed as it is / uint8 with the
    fused preprocessing), with MORE maps than the kernel has blocks (UBD_TEST_NUM_CUS=2: each block does 17 maps in turn, then
    joins the strip queue), maps of another shape than the pass's own, classes, and the LDS poison / integrity mode: the lists
    equal the stand-alone postprocess's, the logits equal the plain forward pass's."""
    from ubdvss_amd import PreprocessingType
    monkeypatch.setenv("UBD_TEST_NUM_CUS", "2")
    monkeypatch.setenv("UBD_STEM", "fused123")
    monkeypatch.setenv("UBD_PP_POISON", "1")
    n_cls = 2
    cfg = NetConfig(class_names=["a", "b"], grey=grey, preprocessing=PreprocessingType.MOBILENET_LIKE if u8 else PreprocessingType.NONE)
    cin = 1 if grey else 3
    model = Model(cfg, seed=2)
    assert model.num_cus == 2
    model.set_weights(onet.init_weights(7, cin, n_cls, bias_scale=0.2))
    x = synthetic.noise_images(3, 3, 96, 160, cin, as_float=not u8)
    xt = torch.from_numpy(x).cuda()
    plain = model.predict_on_device(xt).clone()
    maps = synthetic.rectangle_maps(91, 34, 48, 80, n_classes=n_cls)              # 34 maps for 2 blocks, not the pass's map shape
    lg = torch.from_numpy(synthetic.logits_from_maps(maps, n_cls, seed=92)).cuda()
    ref = model.postprocess_on_device(lg, 0.0, 4, 5, cap=64)
    ref = [t.clone() for t in ref]
    outs = model.alloc_postprocess_outputs(34, 48, 80, 64)
    job = {"logits": lg, "logit_threshold": 0.0, "scale": 4, "min_area": 5, "cap": 64, "outputs": outs}
    for _ in range(3):                                                           # repeated: the strip counters reset themselves
        outs[1].fill_(-3)
        got = model.predict_on_device(xt, postprocess=job)
        torch.cuda.synchronize()
        assert torch.equal(got, plain)
        assert _same_results((got,) + tuple(outs), (plain,) + tuple(ref))
        assert int(outs[3].max()) <= 64                                          # no integrity flag in the counts


def test_pipelined_runner_16bit_model_falls_back_to_back_to_back_calls():
    """A 16-bit handle has no one-kernel stem: ubd_forward_postprocess makes the two calls one after the other; same results."""
    cfg = NetConfig(grey=False)
    model = Model(cfg, dtype="float16", seed=5)
    model.set_weights(onet.init_weights(41, 3, 0, bias_scale=0.3))
    serial, piped = ModelRunner(cfg), ModelRunner(cfg, pipelined=True)
    batches = []
    for k in range(3):
        labels = synthetic.rectangle_maps(70 + k, 34, 64, 64)
        batches.append(torch.from_numpy(synthetic.textured_images(80 + k, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda())
    ref = [[t.clone() if t is not None else None for t in serial.predict_on_device(model, b)] for b in batches]
    got = [piped.predict_on_device(model, b) for b in batches]
    piped.synchronize()
    assert _same_results(got[2], ref[2]) and _same_results(got[1], ref[1])


def test_crowded_maps_vs_oracle():
    """About fifty separate objects per 128 x 128 map (synthetic.crowded_maps: the bench's many-object leg): counts and quads
    bit-exact against the oracle on every map, in every code path."""
    maps = synthetic.crowded_maps(77, 6, 128, 128)
    lg = synthetic.logits_from_maps(maps, 0, seed=78, noise=0.0)
    model = _model(0)
    _compare(model, lg, 0, cap=256)
    _, out, counts = _run(model, lg, cap=256)
    assert 40 <= int(counts.min()) and int(counts.max()) <= 64


def test_pipelined_runner_random_shapes_changing_between_calls():
    """The pipelined runner fed batches whose shape CHANGES from call to call (batch 1..40, sides 32..256 in steps of 4, maps wider /
    narrower than the stem's tiles): the postprocess job that rides in a stem kernel then belongs to logits of another shape than the
    pass's own, small batches take the back-to-back fallback, big ones the one-kernel stem.  Every batch's results equal the serial
    runner's (logits bit for bit, maps, lists up to their counts)."""
    rng = np.random.default_rng(31)
    for n_cls in (0, 2):
        cfg = NetConfig(class_names=[f"c{i}" for i in range(n_cls)] if n_cls else None, grey=False)
        model = Model(cfg, seed=5)
        model.set_weights(onet.init_weights(41, 3, n_cls, bias_scale=0.3))
        serial, piped = ModelRunner(cfg), ModelRunner(cfg, pipelined=True)
        prev = None
        for k in range(int(os.environ.get("UBD_PIPE_SOAK_STEPS", "14"))):
            n = int(rng.choice([1, 2, 3, 8, 20, 40])) if k % 3 else int(rng.integers(1, 6))
            hh, ww = 4 * int(rng.integers(8, 65)), 4 * int(rng.integers(8, 65))
            labels = synthetic.rectangle_maps(300 + k, n, hh // 4, ww // 4, n_classes=n_cls)
            b = torch.from_numpy(synthetic.textured_images(400 + k, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
            ref = [t.clone() if t is not None else None for t in serial.predict_on_device(model, b)]
            got = piped.predict_on_device(model, b)
            if prev is not None:                                  # the previous batch is complete once this call has been enqueued
                torch.cuda.synchronize()
                assert _same_results(prev[0], prev[1]), (n_cls, k - 1, prev[2])
            prev = (got, ref, (n, hh, ww))
        piped.synchronize()
        assert _same_results(prev[0], prev[1]), (n_cls, "last", prev[2])


def test_discs_ellipses_and_tall_objects_vs_oracle():
    """The box fit's hull stage beyond rectangles (round 6: both chains are wrapped at once, one per half wave, the right chain staged in the
    second half of the point slots): discs and rotated ellipses -- dozens of hull vertices per chain, polygons of more than 64 vertices (the
    one-lane form with the right chain moved behind the left one) --, objects of more than 64 and more than 128 rows (a lane takes several
    rows of a chain), thin tall slivers, one pixel columns; 128 x 128 maps (one-launch front end) and 400 x 96 / 96 x 400 maps (the
    global-memory path at any setting).  Bit-exact against the oracle."""
    rng = np.random.default_rng(17)
    def shapes(h, w, k):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        m = np.zeros((h, w), bool)
        for _ in range(k):
            cy, cx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w
            a, b, t = rng.uniform(0.05, 0.45) * h, rng.uniform(0.05, 0.45) * w, rng.uniform(0, np.pi)
            u, v = (yy - cy) * np.cos(t) + (xx - cx) * np.sin(t), -(yy - cy) * np.sin(t) + (xx - cx) * np.cos(t)
            m |= (u / a) ** 2 + (v / b) ** 2 <= 1.0
        return m
    maps128 = np.stack([shapes(128, 128, 1 + i % 3) for i in range(8)])
    maps128[6] = False; maps128[6, 2:126, 60] = True; maps128[6, 5:120, 90:93] = True          # a one-pixel column, a sliver
    yy, xx = np.mgrid[0:128, 0:128]
    maps128[7] = (yy - 64) ** 2 + (xx - 64) ** 2 <= 61 ** 2                                    # a big disc: > 64 hull vertices
    yy, xx = np.mgrid[0:400, 0:400]
    big = np.stack([(yy - 200) ** 2 + (xx - 200) ** 2 <= 190 ** 2, ((yy - 200) / 195.0) ** 2 + ((xx - 200) / 120.0) ** 2 <= 1.0])   # ~110 hull vertices: the one-lane form
    for maps in (maps128, np.stack([shapes(400, 96, 2) for _ in range(3)]), np.stack([shapes(96, 400, 2) for _ in range(3)]), big):
        lg = np.where(maps[..., None], 1.0, -1.0).astype(np.float32)
        _compare(_model(0), lg, 0, min_area=3, cap=64)
