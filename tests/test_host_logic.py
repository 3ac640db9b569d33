"""CPU (-m "not gpu"): host-side logic of the product package and the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from apse_uav_amd import _lib
    lib = _lib.load()
    with open(os.path.join(ROOT, "include", "apse_hip.h")) as f:
        text = f.read()
    declared = set(re.findall(r"\b(apse_[a-z0-9_]+)\s*\(", text))
    declared -= {"apse_ctx", "apse_config"}
    assert len(declared) >= 35
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export " + name
    assert set(_lib.EXPORTS) <= declared
    assert ctypes.sizeof(_lib.Config) == 27 * 4      # apse_config: 27 four-byte fields


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: creating a context without a HIP device must raise, not degrade."""
    from apse_uav_amd import _lib
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.networks.track_rcnn import TrackRCNN
    from apse_uav_amd.weights import synthetic_detector_state
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = TrackRCNN(setup_cfg())
    m.load_state_dict(synthetic_detector_state(0, (1, 1, 1, 1)))
    with pytest.raises(_lib.ApseError):
        m._ensure_ctx((270, 480), (252, 448))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "apse_uav_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn


def _det(n, frame=(100, 200)):
    from apse_uav_amd.structures.instances import Boxes, Instances
    from apse_uav_amd.structures.window_mask import MaskList, WindowMask
    inst = Instances(frame)
    inst.pred_boxes = Boxes(torch.arange(n * 4, dtype=torch.float32).view(n, 4))
    inst.scores = torch.linspace(0.9, 0.6, n)
    inst.pred_classes = torch.arange(n) % 4
    inst.pred_masks = MaskList(WindowMask(None, (0, 0, 0, 0), frame, (k + 1, k + 2), 5) for k in range(n))
    return inst


def test_object_instances_semantics():
    """object_instances.py: ids from 1, never reused; associate keeps the score; ageing and deletion."""
    from apse_uav_amd.structures.object_instances import ObjectInstances
    objs = ObjectInstances((100, 200))
    d = _det(3)
    emb = torch.eye(3, 128)
    for k in range(3):
        objs.add_new_object(k, d, emb)
    assert objs.ids == [1, 2, 3] and len(objs) == 3
    objs.finish_association()
    assert objs.frames_since_detected == [0, 0, 0] and objs.detected_this_frame == [False] * 3
    d2 = _det(2)
    objs.associate_detection(1, 0, d2, torch.ones(2, 128))
    assert objs.detected_this_frame == [True, False, False]
    assert float(objs.scores[0]) == float(d.scores[0])                 # score frozen at first sighting
    assert torch.equal(objs.pred_boxes[0].tensor, d2.pred_boxes[1].tensor)
    rec = objs.get_recent_objects()
    assert rec.ids == [1] and len(rec) == 1
    objs.finish_association()
    assert objs.frames_since_detected == [0, 1, 1]
    objs._fields["frames_since_detected"] = [0, 101, 100]
    objs.delete_undetected_objects(100)
    assert objs.ids == [1, 3]
    objs.add_new_object(0, d2, torch.ones(2, 128))
    assert objs.ids == [1, 3, 4]                                        # 2 is never reused


def test_csv_formats_and_consumer_roundtrip(tmp_path, golden_dir):
    from apse_uav_amd.utils import csv_log
    lines = ["0,1911.0,966.0,1911.0,966.0,192.0,1380.0,365.0,1338.0,3388.0,1020.0,3156.0,1014.0,269.0,498.0,468.0,543.0",
             "1,1911.0,966.0,1911.0,966.0,,,,,3388.0,1020.0,3156.0,1014.0,269.0,498.0,468.0,543.0",
             "2,1911.0,966.0,nan,nan,189.0,1389.0,nan,nan"]
    raw = tmp_path / "raw.csv"
    csv_log.write_raw_csv(str(raw), lines, 1, 4)
    txt = raw.read_text().split("\n")
    assert txt[0] == "Ford id: 1" and txt[1].startswith("frame,id_1 cent_x,id_1 cent_y,id_1 clos_x,id_1 clos_y,id_2 cent_x")
    out = tmp_path / "consumer.csv"
    csv_log.write_consumer_csv(str(out), lines, host_id=2, vehicle_ids=[4, 1, 3])
    got = out.read_text()
    with open(os.path.join(golden_dir, "static_dcnn_data_head.csv")) as f:
        ref = f.read().split("\n")
    g = got.split("\n")
    assert g[0] == ref[0] and g[1] == ref[1]                           # shipped header lines, byte for byte
    assert got.endswith("\n") and all(len(r.split(",")) == 17 for r in g[:-1])
    data = csv_log.read_centroid_data(str(out))
    assert data[0] == [0, 192, 1380, 365, 1338, 269, 498, 468, 543, 1911, 966, 1911, 966, 3388, 1020, 3156, 1014]
    assert data[1][1:5] == [0, 0, 0, 0]                                 # blank cells -> 0 like readCentroidData
    for name in ("static", "dynamic"):
        ref_rows = csv_log.read_centroid_data(os.path.join(golden_dir, name + "_dcnn_data_head.csv"))
        assert all(len(r) == 17 for r in ref_rows) and len(ref_rows) >= 5


def test_log_line_matches_reference_shape():
    from apse_uav_amd.structures.object_instances import ObjectInstances
    from apse_uav_amd.utils import csv_log
    objs = ObjectInstances((100, 200))
    d = _det(3)
    for k in range(3):
        objs.add_new_object(k, d, torch.eye(3, 128))
    objs.delete_undetected_objects(100)
    rec = objs.get_recent_objects()
    line, hi = csv_log.generate_log_oneline(rec, 2, 7, closest_lookup=lambda k, h: (10.0 + k, 20.0 + h))
    assert hi == 3
    assert line == "7,1.0,2.0,10.0,21.0,2.0,3.0,11.0,21.0,3.0,4.0,12.0,21.0"
    line, _ = csv_log.generate_log_oneline(rec, 9, 8, closest_lookup=lambda k, h: (0.0, 0.0))
    assert line == "8,1.0,2.0,nan,nan,2.0,3.0,nan,nan,3.0,4.0,nan,nan"


def test_config_matches_reference_yaml_values():
    from apse_uav_amd.config import get_cfg, setup_cfg
    cfg = setup_cfg()
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 4 and cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST == 0.5
    assert cfg.MODEL.RPN.PRE_NMS_TOPK_TEST == 1000 and cfg.MODEL.RPN.POST_NMS_TOPK_TEST == 1000
    assert cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION == 7 and cfg.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION == 14
    assert tuple(cfg.MODEL.ROI_HEADS.IN_FEATURES) == ("p2", "p3", "p4", "p5")
    c2 = cfg.clone()
    c2.freeze()
    with pytest.raises(AttributeError):
        c2.MODEL.DEVICE = "cpu"
    assert get_cfg().INPUT.MIN_SIZE_TEST == 800


def test_window_mask_dense_roundtrip():
    from apse_uav_amd.structures.window_mask import WindowMask
    rng = np.random.default_rng(0)
    H, W = 120, 300
    x0, y0, x1, y1 = 70, 10, 205, 50
    win = rng.random((y1 - y0, x1 - x0)) > 0.5
    w0, w1 = x0 >> 6, (x1 + 63) >> 6
    bits = np.zeros((y1 - y0, w1 - w0), np.uint64)
    for yy in range(y1 - y0):
        for xx in range(x1 - x0):
            if win[yy, xx]:
                X = x0 + xx
                bits[yy, (X >> 6) - w0] |= np.uint64(1) << np.uint64(X & 63)
    m = WindowMask(torch.from_numpy(bits.view(np.int64)), (x0, y0, x1, y1), (H, W), (1, 1), int(win.sum()))
    dense = m.dense().numpy()
    assert dense[y0:y1, x0:x1].tolist() == win.tolist() and dense.sum() == win.sum()


def test_synthetic_sequence_deterministic():
    from apse_uav_amd.synthetic import SyntheticSequence
    a = SyntheticSequence("dynamic", 108, 192).frame(3)
    b = SyntheticSequence("dynamic", 108, 192).frame(3)
    assert a.dtype == np.uint8 and a.shape == (108, 192, 3) and np.array_equal(a, b)
    s = SyntheticSequence("dynamic", 108, 192)
    assert len(s.boxes(0)) == 4 and len(s.boxes(25)) == 3


def test_consumer_distance_step(golden_dir):
    """aruco_detect.py:483-492 on rows of the shipped CSV: the pixel->metre step downstream of the log."""
    from apse_uav_amd.utils import csv_log
    rows = csv_log.read_centroid_data(os.path.join(golden_dir, "static_dcnn_data_head.csv"))
    host, _ = csv_log.dcnn_points(rows[0], 0)
    cen, clo = csv_log.dcnn_points(rows[0], 1)
    assert host == (1911, 966) and cen == (192, 1380) and clo == (365, 1338)
    da, db = csv_log.calculate_distance([host], [cen], [clo], 0.8, 40.0, 44.0)
    assert abs(da - np.hypot(1911 - 192, 966 - 1380) * 0.8 / 42.0) < 1e-9
    assert abs(db - np.hypot(1911 - 365, 966 - 1338) * 0.8 / 42.0) < 1e-9 and db < da


# ---------------------------------------------------------------- weight formats
def test_detector_checkpoint_formats_roundtrip(tmp_path):
    """`.pth` = {"model": state_dict, ...} (dcnn/scripts/train/finetune_uav.py:273-283) through cfg.MODEL.WEIGHTS, a
    bare state_dict, detectron2's non-weight buffers dropped, dtypes normalised to f32; loaders are weights_only."""
    import torch
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.engines.track_predictor import TrackPredictor
    from apse_uav_amd.weights import blocks_from_state, load_detector_file, synthetic_detector_state
    sd = synthetic_detector_state(3, (1, 1, 1, 1))
    extra = dict(sd)
    extra["proposal_generator.anchor_generator.cell_anchors.0"] = torch.zeros(3, 4)     # buffers a detectron2 0.1.2 checkpoint holds
    extra["pixel_mean"] = torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1)
    extra["pixel_std"] = torch.ones(3, 1, 1)
    extra["roi_heads.box_head.fc2.bias"] = extra["roi_heads.box_head.fc2.bias"].double()   # a non-f32 tensor
    wrapped, bare = str(tmp_path / "model_final.pth"), str(tmp_path / "bare.pth")
    torch.save({"model": extra, "iteration": 1234, "optimizer": {"state": {}}}, wrapped)
    torch.save(extra, bare)
    for path in (wrapped, bare):
        got = load_detector_file(path)
        assert set(got) == set(sd)
        assert all(v.dtype == torch.float32 for v in got.values())
        assert all(torch.equal(got[k], sd[k]) for k in sd)
        assert blocks_from_state(got) == (1, 1, 1, 1)
    cfg = setup_cfg(device="cpu")
    cfg.MODEL.WEIGHTS = wrapped
    pr = TrackPredictor(cfg)                                   # the constructor path the reference uses (track_predictor.py:20-25)
    assert set(pr.model._state) == set(sd) and torch.equal(pr.model._state["backbone.fpn_output2.weight"], sd["backbone.fpn_output2.weight"])


def test_association_checkpoint_and_tracker_ctor(tmp_path):
    """Association head: plain state_dict {fc.weight 128x25600, fc.bias 128} (association_head.py:13), loaded from the
    path given as RcnnTracker's `weights` argument (rcnn_tracker.py:56)."""
    import torch
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.weights import load_association_file, synthetic_association_state, synthetic_detector_state
    asd = synthetic_association_state(5)
    path = str(tmp_path / "association_head.pth")
    torch.save(asd, path)
    got = load_association_file(path)
    assert set(got) == {"fc.weight", "fc.bias"} and got["fc.weight"].shape == (128, 25600)
    assert all(torch.equal(got[k], asd[k]) for k in asd)
    tr = RcnnTracker(setup_cfg(device="cpu"), (270, 480), path, detector_state=synthetic_detector_state(0, (1, 1, 1, 1)))
    assert torch.equal(tr.association_head.fc.weight.detach().cpu(), asd["fc.weight"])
    assert tr.predictor.model._assoc is tr.association_head


def test_roi_features_generator_takes_backbone_only_checkpoints(tmp_path):
    """PartialCheckpointer (dcnn/utils/partial_checkpointer.py:11-20) strips `backbone.` from every key and loads the
    backbone alone: a stripped dict, a full dict and a `.pth` holding either must all yield the same backbone tensors."""
    import torch
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.engines.roi_features_generator import RoiFeaturesGenerator
    from apse_uav_amd.weights import synthetic_detector_state
    sd = synthetic_detector_state(2, (1, 1, 1, 1))
    stripped = {k.split("backbone.")[-1]: v for k, v in sd.items() if k.startswith("backbone.")}
    path = str(tmp_path / "backbone_only.pth")
    torch.save({"model": stripped}, path)
    cfg = setup_cfg(device="cpu")
    a = RoiFeaturesGenerator(cfg, roi_size=8, state_dict=stripped).model._state
    b = RoiFeaturesGenerator(cfg, roi_size=8, state_dict=sd).model._state
    cfg2 = cfg.clone()
    cfg2.MODEL.WEIGHTS = path
    c = RoiFeaturesGenerator(cfg2, roi_size=8).model._state
    for st in (a, b, c):
        for k, v in sd.items():
            if k.startswith("backbone."):
                assert torch.equal(st[k], v), k
            else:
                assert k in st and st[k].shape == v.shape
    for st in (a, b, c):                                       # heads are zero-filled: this generator never runs them
        assert all(torch.count_nonzero(st[k]) == 0 for k in st if not k.startswith("backbone."))


def test_model_zoo_pickle_is_refused(tmp_path):
    """Model-zoo `.pkl` files need pickle (add_mask_head_to_frcnn.py:53-55): the loader never unpickles and says so."""
    import pickle
    import pytest
    from apse_uav_amd.weights import load_detector_file
    p = tmp_path / "model_final_a3ec72.pkl"
    with open(p, "wb") as f:
        pickle.dump({"model": {"backbone.bottom_up.stem.conv1.weight": np.zeros((64, 3, 7, 7), np.float32)},
                     "__author__": "Detectron2 Model Zoo"}, f)              # numpy arrays, as the zoo files hold
    with pytest.raises(Exception) as ei:
        load_detector_file(str(p))
    assert "pickle" in str(ei.value).lower() or "weights_only" in str(ei.value).lower() or "unsupported" in str(ei.value).lower()


def test_lab_tables_of_the_library_equal_the_oracles():
    """The Lab step of preprocess_img (visualize_uav.py:62-69) is integer arithmetic on tables (csrc/preproc_pixel.h): the
    library builds them on the host in C++ (libm, double), the oracle with numpy.  Every entry must agree -- this is what makes
    the GPU bytes equal to the oracle's by construction; it runs on whatever host the tests run on (no GPU call)."""
    import ctypes as C
    from apse_uav_amd import _lib
    from oracle import preproc as op

    class LabTables(C.Structure):
        _fields_ = [("lin", C.c_uint16 * 256), ("cbrt", C.c_uint16 * 3072), ("c", C.c_int32 * 9), ("lut", C.c_uint8 * 256),
                    ("fy", C.c_uint16 * 256), ("y", C.c_int32 * 256), ("at", C.c_int32 * 256), ("bt", C.c_int32 * 256),
                    ("ci", C.c_int32 * 9), ("inv", C.c_uint8 * 4097), ("pad_", C.c_uint8 * 3)]
    lib = _lib.load()
    for gamma in (2.0, 1.0, 0.5):
        lut = op.gamma_lut(gamma)
        t = LabTables()
        assert lib.apse_lab_tables_host(_lib.ptr(lut), None, 0) == C.sizeof(LabTables)
        assert lib.apse_lab_tables_host(_lib.ptr(lut), C.byref(t), C.sizeof(t)) == C.sizeof(LabTables)
        ref = op.lab_tables(lut)
        for name in ("lin", "cbrt", "lut", "fy", "y", "at", "bt", "inv"):
            assert np.array_equal(np.asarray(getattr(t, name), np.int64), ref[name]), name
        assert np.array_equal(np.asarray(t.c, np.int64).reshape(3, 3), ref["c"])
        assert np.array_equal(np.asarray(t.ci, np.int64).reshape(3, 3), ref["ci"])
    # table sanity: end points and monotonicity
    assert ref["lin"][0] == 0 and ref["lin"][255] == 2040 and np.all(np.diff(ref["lin"]) >= 0)
    assert ref["inv"][0] == 0 and ref["inv"][4096] == 255 and np.all(np.diff(ref["inv"]) >= 0)
    assert ref["cbrt"][2040] == 32768 and np.all(np.diff(ref["cbrt"]) >= 0)


def test_integer_lab_gamma_against_the_f32_formulas():
    """The integer Lab step against the published CIE formulas in f32 (the form rounds 1-2 used): they must describe the same
    transform -- identity LUT returns the input up to the 8-bit Lab quantisation, gamma 2 agrees within that quantisation."""
    from oracle import preproc as op
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (256, 384, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:256, 0:384]
    smooth = np.stack([xx * 255 // 383, yy, (xx + yy) * 255 // 638], -1).astype(np.uint8)
    for im in (img, smooth):
        a = op.lab_gamma(im, op.gamma_lut()).astype(np.int32)
        b = op.lab_gamma_f32(im, op.gamma_lut()).astype(np.int32)
        d = np.abs(a - b)
        assert d.mean() < 0.5 and (d > 2).mean() < 0.04, (d.mean(), (d > 2).mean(), d.max())
        ident = np.arange(256, dtype=np.uint8)
        ri = np.abs(op.lab_gamma(im, ident).astype(np.int32) - im.astype(np.int32))
        rf = np.abs(op.lab_gamma_f32(im, ident).astype(np.int32) - im.astype(np.int32))
        assert ri.mean() < rf.mean() + 0.25, (ri.mean(), rf.mean())         # no worse a round trip than the f32 form's
    grey = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    out = op.lab_gamma(grey, op.gamma_lut())
    assert np.all(np.diff(out[0, :, 0].astype(np.int32)) >= 0) and out[0, 255, 0] >= 254 and out[0, 0, 0] == 0      # grey ramp stays monotone
    assert np.abs(out[..., 0].astype(np.int32) - out[..., 1]).max() <= 1 and np.abs(out[..., 1].astype(np.int32) - out[..., 2]).max() <= 1


def test_host_copy_pool_equals_memcpy():
    """csrc/host_stage.hip: the ingest path's staging copy on the library's thread pool (FrameUploader) -- sizes below the
    threading floor, odd sizes, a 4K frame, repeated calls with different thread counts."""
    from apse_uav_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    for nbytes, threads in ((0, 4), (17, 4), ((1 << 20) + 13, 3), (2160 * 3840 * 3, 8), (2160 * 3840 * 3, 1), (5 << 20, 32), (3 << 20, 2)):
        src = rng.integers(0, 256, nbytes, dtype=np.uint8)
        dst = np.zeros(nbytes + 64, np.uint8)
        assert lib.apse_host_copy(dst.ctypes.data, src.ctypes.data, nbytes, threads) == 0
        assert np.array_equal(dst[:nbytes], src) and not dst[nbytes:].any()
    assert lib.apse_host_copy(None, None, 5, 2) < 0
