"""Segmentation metrics of the pixel-task eval loops (SURVEY.md 8 f-1 anchors utils/avss_utils.py:8-96, 379-435).

CPU: oracle/metrics_oracle.py against tests/golden/seg_metrics.npz (outputs of the reference's own functions, make_golden.py metrics).
GPU: crab_amd.avss_utils (crab_mask_iou / crab_fmeasure / crab_miou_fscore through the C-ABI) against the fixture and the oracle.

Bar: every pixel count bit-exact; fp32 values formed per image / per class from the counts bit-exact (one rounding per operation on both
sides); values that end in a torch.sum over images / classes within REL = 1e-6 (the summation order of torch.sum is not specified; the
restatements add in index order).  The threshold sweep of Eval_Fmeasure compares fp32 sigmoid values against fp32 thresholds: both are
host-dependent in the reference at the last bit (vectorised expf, vectorised linspace), so counts are bit-exact HIP vs oracle (same fp64
sigmoid, same table) and the reference fixture allows one pixel per threshold-adjacent value - none occurs on the committed fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

REL = 1e-6


def _fx():
    A = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seg_metrics.npz")))
    A.pop("meta")
    return A


def _close(a, b, rel=REL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.all(np.abs(a[m] - b[m]) <= rel * np.maximum(1.0, np.abs(b[m]))), (a, b)


# ------------------------------------------------------------------ CPU: the oracle is pinned to the reference
def test_oracle_binary_metrics_match_the_reference_fixture():
    A = _fx()
    pred, gt = A["bin_pred"], A["bin_gt"]
    N = pred.shape[0]
    _close(MO.mask_iou(pred, gt), A["iou_all"])
    for n in range(N):
        assert MO.mask_iou(pred[n:n + 1], gt[n:n + 1]) == A["iou_each"][n]                # one image: no sum involved
        assert MO.metric_s_for_null(pred[n:n + 1]) == A["s_each"][n]
    assert MO.mask_iou(pred[1:2], gt[1:2]) == np.float32((pred[1] <= 0).sum()) / np.float32(pred[1].size)   # empty target: its empty pixels over all


@pytest.mark.parametrize("table", ["own", "torch"])
def test_oracle_fmeasure_matches_the_reference_fixture(table):
    A = _fx()
    pred, gt = A["bin_pred"], A["bin_gt"]
    th = None if table == "own" else A["thlist"]
    val, d = MO.eval_fmeasure(pred, gt, 255, th=th, details=True)
    assert np.array_equal(d["ge"][:, 0], A["ge_tp"]) and np.array_equal(d["ge"][:, 1], A["ge_cnt"])
    assert d["images"] == 3                                                                # image 1 has a black ground truth
    _close(val, float(A["f_all"]))
    for n in range(pred.shape[0]):
        assert MO.eval_fmeasure(pred[n:n + 1], gt[n:n + 1], 255, th=th) == float(A["f_each"][n])
    assert MO.eval_fmeasure(pred[1:2], gt[1:2]) == 0.0 == float(A["f_black_only"])
    sp = MO.sigmoid32(pred)
    for n in (0, 2):
        prec, rec, _, _ = MO.eval_pr(sp[n], gt[n], MO.thresholds(255) if th is None else th)
        assert np.array_equal(prec, A["prec"][n]) and np.array_equal(rec, A["recall"][n])


def test_threshold_table_is_within_one_ulp_of_torch_linspace():
    from crab_amd.avss_utils import fmeasure_thresholds
    A = _fx()
    own = fmeasure_thresholds(255)                                                         # r06: torch.linspace itself, the reference's own tensor on this host
    assert np.array_equal(own, torch.linspace(0, 1 - 1e-10, 255).numpy())
    assert own[0] == 0.0 and own[-1] == 1.0 and np.all(np.diff(own) > 0)
    recorded = A["thlist"]                                                                 # torch.linspace(0, 1 - 1e-10, 255) on the fixture's host
    for other in (recorded, MO.thresholds(255)):                                           # another host's vector width / the fp64-rounded table: <= 1 ulp
        assert np.all(np.abs(own - other) <= np.spacing(np.maximum(own, other)))
    assert np.array_equal(fmeasure_thresholds(1), np.zeros(1, np.float32))


def test_oracle_class_metrics_match_the_reference_fixture():
    A = _fx()
    cp, ct = A["cls_pred"], A["cls_tgt"]
    mi, fs, cc, vid, iou_fc = MO.batch_miou_fscore(cp, ct)
    assert np.array_equal(mi, A["cls_miou"]) and np.array_equal(fs, A["cls_fscore"]) and np.array_equal(cc, A["cls_count"])
    assert np.array_equal(iou_fc, A["cls_iou_fc"])
    _close(vid, A["cls_vid"])
    for f in range(cp.shape[0]):
        _mi, fs1, _cc, _v, _i = MO.batch_miou_fscore(cp[f:f + 1], ct[f:f + 1])
        assert np.array_equal(fs1, A["cls_fs_fc"][f])
    areas = MO.class_areas(cp, ct)
    assert areas[1, 2].sum() == (ct[1] != 255).sum() and areas[1, 2, 5] == 0              # out-of-range labels counted nowhere; class 5 absent
    assert areas[2, 1].sum() == (ct[2] >= 0).sum() and areas[0, 1, 6] == 0                # negative labels remove the prediction as well
    assert np.all(areas[:, 0] <= np.minimum(areas[:, 1], areas[:, 2]))


def test_oracle_palette_and_colour_map_match_the_reference_fixture():
    """dataset/quick_start_dataset.py:35-73: the palette of get_v2_pallete (= harness.default_palette: the reference's table is code, not data)
    and color_mask_to_label on a map with colours outside the table."""
    from crab_amd import harness
    A = _fx()
    assert np.array_equal(MO.get_v2_pallete(71), A["v2_pallete"]) and np.array_equal(harness.default_palette(71), A["v2_pallete"])
    assert harness.get_v2_pallete is harness.default_palette
    lab = MO.color_mask_to_label(A["color_mask"], A["v2_pallete"])
    assert np.array_equal(lab, A["color_label"]) and (lab[0, :4] == 0).all() and len(np.unique(lab)) == 71


def test_host_tensors_and_bad_shapes_are_refused_without_a_gpu():
    from crab_amd import _lib, avss_utils as AU
    p, t = torch.zeros(1, 4, 4), torch.zeros(1, 4, 4)
    for call in (lambda: AU.mask_iou(p, t), lambda: AU.Eval_Fmeasure(p, t), lambda: AU.metric_s_for_null(p),
                 lambda: AU.calc_color_miou_fscore(torch.zeros(1, 3, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))):
        with pytest.raises(_lib.CrabHipError):
            call()
    lib = _lib.load()
    assert lib.crab_color_to_label(None, None, None, 16, None, 71, None) == -1
    from crab_amd import harness
    with pytest.raises(_lib.CrabHipError):
        harness.color_mask_to_label(np.zeros((4, 4), np.uint8))                            # not [H, W, 3]
    with pytest.raises(_lib.CrabHipError):
        harness.color_mask_to_label(np.zeros((4, 4, 3), np.uint8), np.zeros((300, 3)))     # more than 256 colours
    assert lib.crab_mask_iou(None, None, None, None, 1, 16, 1e-7, None, None) == -1       # CRAB_E_INVALID before any HIP call
    assert lib.crab_fmeasure(None, None, None, None, 1, 16, None, 255, 0.3, None, None, None, None, None) == -1
    assert lib.crab_miou_fscore(None, None, None, None, 1, 3, 16, 0.3, None, None, None, None, None, None) == -1


# ------------------------------------------------------------------ GPU: the HIP path
def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.gpu
def test_hip_binary_metrics_match_fixture_and_oracle():
    from crab_amd import avss_utils as AU
    A = _fx()
    pred, gt = A["bin_pred"], A["bin_gt"]
    P, G = _dev(pred), _dev(gt)
    N = pred.shape[0]
    iou, counts = AU.mask_iou(pred=P, target=G, details=True)
    assert np.array_equal(counts[:, :5].numpy(), MO.mask_counts(pred, gt)) and int(counts[:, 5].sum()) == 0
    assert iou.dim() == 0 and iou.dtype == torch.float32
    _close(iou.item(), A["iou_all"])
    from tests.util import record_parity
    record_parity("seg metrics: mask_iou (4 masks) vs the reference's value", abs(iou.item() - float(A["iou_all"])), float(A["iou_all"]), tol=REL, counts_exact=True)
    assert iou.item() == float(MO.mask_iou(pred, gt))                                      # both add in image order
    for n in range(N):
        assert AU.mask_iou(P[n:n + 1], G[n:n + 1]).item() == float(A["iou_each"][n])
        assert AU.metric_s_for_null(P[n:n + 1]).item() == float(A["s_each"][n])
    val, d = AU.Eval_Fmeasure(pred=P, gt=G, details=True)
    assert np.array_equal(d["ge"][:, 0].numpy(), A["ge_tp"]) and np.array_equal(d["ge"][:, 1].numpy(), A["ge_cnt"])
    assert d["images"] == 3 and np.array_equal(d["ysum"][:, 0].numpy(), (gt != 0).sum((1, 2)))
    oval, od = MO.eval_fmeasure(pred, gt, details=True)
    assert np.array_equal(d["fscore"].numpy(), od["fscore"]) and np.array_equal(d["score"].numpy(), od["score"]) and val == oval
    _close(val, float(A["f_all"]))
    record_parity("seg metrics: Eval_Fmeasure (4 masks, 255 thresholds) vs the reference's value", abs(val - float(A["f_all"])), float(A["f_all"]), tol=REL,
                  threshold_counts_exact=True)
    for n in range(N):
        assert AU.Eval_Fmeasure(P[n:n + 1], G[n:n + 1]) == float(A["f_each"][n])
    assert AU.Eval_Fmeasure(P[1:2], G[1:2]) == 0.0


@pytest.mark.gpu
def test_hip_class_metrics_match_fixture_and_oracle():
    from crab_amd import avss_utils as AU
    A = _fx()
    cp, ct = A["cls_pred"], A["cls_tgt"]
    mi, fs, cc, vid, d = AU.calc_color_miou_fscore(pred=_dev(cp), target=_dev(ct), T=1, details=True)
    assert np.array_equal(d["areas"].numpy(), MO.class_areas(cp, ct))
    assert np.array_equal(mi.cpu().numpy(), A["cls_miou"]) and np.array_equal(fs.cpu().numpy(), A["cls_fscore"])
    assert np.array_equal(cc.cpu().numpy(), A["cls_count"]) and np.array_equal(d["iou_fc"].numpy(), A["cls_iou_fc"])
    assert isinstance(vid, list) and len(vid) == cp.shape[0] and vid[0].dim() == 0
    _close(torch.stack(vid).cpu().numpy(), A["cls_vid"])
    from tests.util import record_parity
    record_parity("seg metrics: calc_color_miou_fscore per-frame mean IoU vs the reference's values (per-class sums bit-equal)",
                  float(np.abs(torch.stack(vid).cpu().numpy() - A["cls_vid"]).max()), float(np.abs(A["cls_vid"]).max()), tol=REL, areas_exact=True)
    assert np.array_equal(torch.stack(vid).cpu().numpy(), MO.batch_miou_fscore(cp, ct)[3])
    # int32 / uint8 label maps are accepted (cast to the reference's int64), float ones are not
    mi2, *_ = AU.calc_color_miou_fscore(_dev(cp), _dev(np.where(ct < 0, 200, ct).astype(np.uint8)))
    assert mi2.shape == mi.shape
    from crab_amd import _lib
    with pytest.raises(_lib.CrabHipError):
        AU.calc_color_miou_fscore(_dev(cp), _dev(ct.astype(np.float32)))


@pytest.mark.gpu
def test_hip_metrics_full_size_and_edges_vs_oracle():
    """The eval shapes (224 x 224 masks, 71 AVSS classes) and the edges: one image, odd plane sizes, empty / full ground truths, all-negative
    and all-positive predictions, a NaN logit, pr_num 1 and 1024, bf16 masks, a single class."""
    from crab_amd import _lib, avss_utils as AU
    rng = np.random.default_rng(5)
    for (N, H, W) in ((5, 224, 224), (1, 224, 224), (3, 7, 9), (2, 1, 1)):
        pred = (rng.standard_normal((N, H, W)) * 4).astype(np.float32)
        gt = (rng.random((N, H, W)) > 0.6).astype(np.float32)
        if N > 1:
            gt[1] = 0
        if N > 2:
            gt[2] = 1
            pred[2] = -np.abs(pred[2])                                                    # nothing predicted on a full ground truth
        pred.flat[0] = 0.0                                                                 # sigmoid(0) = 0.5 is not > 0.5
        P, G = _dev(pred), _dev(gt)
        iou, counts = AU.mask_iou(P, G, details=True)
        assert np.array_equal(counts[:, :5].numpy(), MO.mask_counts(pred, gt))
        assert iou.item() == float(MO.mask_iou(pred, gt))
        assert AU.metric_s_for_null(P).item() == float(np.sqrt(np.float32((pred > 0).sum()) / np.float32(pred.size), dtype=np.float32))
        for T in ((255, 1, 1024) if H == 7 else (255,)):
            val, d = AU.Eval_Fmeasure(P, G, pr_num=T, details=True)
            oval, od = MO.eval_fmeasure(pred, gt, T, details=True)
            assert np.array_equal(d["ge"].numpy(), od["ge"]) and np.array_equal(d["fscore"].numpy(), od["fscore"])
            assert val == oval and d["images"] == od["images"]
            assert np.all(np.diff(d["ge"][:, 1].numpy(), axis=1) <= 0)                     # counts fall as the threshold rises
            assert np.array_equal(d["ge"][:, 1, 0].numpy(), np.full(N, H * W))             # th_0 = 0: every pixel
        # a perfect prediction scores 1 on both (up to union + 1e-7 at a one-pixel union; P = R = 1 at the thresholds inside sigmoid's gap)
        pf = np.where(gt > 0, 3.0, -3.0).astype(np.float32)
        if gt.sum((1, 2)).min() > 0:
            assert AU.mask_iou(_dev(pf), G).item() == float(MO.mask_iou(pf, gt)) >= 0.9999998
            assert AU.Eval_Fmeasure(_dev(pf), G) == MO.eval_fmeasure(pf, gt) >= 0.9999998
    nan = np.zeros((1, 4, 4), np.float32)
    nan[0, 0, 0] = np.nan
    nan[0, 1] = 2.0
    g1 = np.ones((1, 4, 4), np.float32)
    val, d = AU.Eval_Fmeasure(_dev(nan), _dev(g1), details=True)
    oval, od = MO.eval_fmeasure(nan, g1, details=True)
    assert np.array_equal(d["ge"].numpy(), od["ge"]) and val == oval and d["ge"][0, 1, 0].item() == 15    # NaN >= 0 is false
    with pytest.raises(_lib.CrabHipError):
        AU.mask_iou(_dev(nan), _dev(g1 * 0.5))                                            # a soft ground truth is not what the reference feeds
    with pytest.raises(_lib.CrabHipError):
        AU.Eval_Fmeasure(_dev(nan), _dev(g1 * 255))
    pb = torch.randn(2, 32, 32, device="cuda").to(torch.bfloat16)
    gb = (torch.rand(2, 32, 32, device="cuda") > 0.5)
    assert AU.mask_iou(pb, gb.to(torch.bfloat16)).item() == float(MO.mask_iou(pb.float().cpu().numpy(), gb.float().cpu().numpy()))
    # AVSS: 71 classes at 224 x 224, two frames; then the running sums of the eval loop
    BF, C, H, W = 2, 71, 224, 224
    cp = rng.standard_normal((BF, C, H, W)).astype(np.float32)
    ct = rng.integers(0, C, (BF, H, W)).astype(np.int64)
    ct[0, :40] = cp[0, :, :40].argmax(0)
    ct[1][ct[1] > 60] = 0
    cp[1, 0, 0, :2] = cp[1, 1, 0, :2] = 9.0                                                # a tie: the first maximum wins, as in torch.argmax
    mi, fs, cc, vid, d = AU.calc_color_miou_fscore(_dev(cp), _dev(ct), T=1, details=True)
    omi, ofs, occ, ovid, oiou = MO.batch_miou_fscore(cp, ct)
    assert np.array_equal(d["areas"].numpy(), MO.class_areas(cp, ct)) and d["areas"][:, 1].sum().item() == BF * H * W
    assert np.array_equal(mi.cpu().numpy(), omi) and np.array_equal(fs.cpu().numpy(), ofs) and np.array_equal(cc.cpu().numpy(), occ)
    assert np.array_equal(d["iou_fc"].numpy(), oiou) and np.array_equal(torch.stack(vid).cpu().numpy(), ovid)
    for (bf, c, hh, ww) in ((2, 5, 7, 9), (1, 20, 3, 4), (3, 9, 1, 1), (70, 71, 4, 4), (1100, 2, 2, 2)):   # the last two: frames beyond one LDS chunk of the finish                   # odd planes (scalar loads), C across the 8-deep unroll
        xp = rng.standard_normal((bf, c, hh, ww)).astype(np.float32)
        xt = rng.integers(-1, c + 1, (bf, hh, ww)).astype(np.int64)
        r = AU.calc_color_miou_fscore(_dev(xp), _dev(xt), details=True)
        o = MO.batch_miou_fscore(xp, xt)
        assert np.array_equal(r[4]["areas"].numpy(), MO.class_areas(xp, xt))
        for got, want in zip((r[0], r[1], r[2], torch.stack(r[3])), o[:4]):
            assert np.array_equal(got.cpu().numpy(), want, equal_nan=True)
    meter = AU.AVSSMeter(C)
    sums = [np.zeros(C, np.float32) for _ in range(3)]
    for f in range(BF):
        meter.update(_dev(cp[f]), _dev(ct[f:f + 1]))
        for s, v in zip(sums, MO.batch_miou_fscore(cp[f:f + 1], ct[f:f + 1])[:3]):
            s += v
    want = MO.avss_final(*sums)
    got = meter.result()
    assert set(got) == set(want)
    for k in want:
        _close(got[k], want[k])
    one = AU.calc_color_miou_fscore(_dev(cp[:1, :1]), _dev(np.zeros((1, H, W), np.int64)))
    assert one[0].item() == 1.0 and one[2].item() == 1.0                                   # a single class: everything is that class


@pytest.mark.gpu
def test_hip_colour_map_to_labels_matches_fixture_and_oracle():
    from PIL import Image
    from crab_amd import harness
    A = _fx()
    got = harness.color_mask_to_label(A["color_mask"], A["v2_pallete"])
    assert got.dtype == torch.int64 and got.is_cuda and np.array_equal(got.cpu().numpy(), A["color_label"])
    assert np.array_equal(harness.color_mask_to_label(Image.fromarray(A["color_mask"], "RGB")).cpu().numpy(), A["color_label"])   # PIL in, default table
    rng = np.random.default_rng(2)
    pal = harness.default_palette(71)
    for (h, w) in ((224, 224), (7, 9), (1, 1)):
        cls = rng.integers(0, 71, (h, w))
        rgb = pal[cls].copy()
        off = rng.random((h, w)) < 0.2
        rgb[off] = rng.integers(0, 256, (int(off.sum()), 3)).astype(np.uint8)
        want = MO.color_mask_to_label(rgb, pal)
        assert np.array_equal(harness.color_mask_to_label(torch.from_numpy(rgb).cuda(), pal).cpu().numpy(), want)
        # the label map is what the AVSS metric takes: a prediction that is the map itself scores IoU 1 on every class present
        if h == 224:
            from crab_amd import avss_utils as AU
            lab = harness.color_mask_to_label(rgb, pal)
            onehot = torch.nn.functional.one_hot(lab, 71).permute(2, 0, 1).float()[None]
            mi, fs, cc, vid = AU.calc_color_miou_fscore(onehot, lab[None])
            assert torch.equal(mi, cc) and vid[0].item() == 1.0
    dup = np.array([[5, 5, 5], [9, 9, 9], [5, 5, 5]], np.uint8)                            # a repeated colour: the first index wins (argmax)
    img = np.array([[[5, 5, 5], [9, 9, 9], [1, 2, 3]]], np.uint8)
    assert harness.color_mask_to_label(img, dup).cpu().tolist() == [[0, 1, 0]]
    assert harness.color_mask_to_label(img, dup[1:2]).cpu().tolist() == [[0, 0, 0]]        # one colour: index 0 whether it matches or not
