"""GPU parity, the whole hot path (predictor.py:46-60 with NMS(300)): images -> proposals."""
import numpy as np
import pytest
import torch

from oracle import bbox_oracle as bo
from oracle import c_oracle as co
from oracle import conv_oracle as cv
from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backbone,img,fm", [("vgg16", 160, 10), ("mobilenet_v2", 160, 10)])
def test_propose_against_oracle_pipeline(backbone, img, fm):
    hp = bo.get_hyper_params(backbone, img_size=img, feature_map_shape=fm)
    weights = synthetic_weights(backbone, hp, seed=3)
    prop = Proposer(backbone, hyper_params=dict(hp), weights=weights, max_batch=2, iou_threshold=0.7)
    imgs = np.random.RandomState(1).uniform(0, 1, size=(2, img, img, 3)).astype(np.float32)
    x = torch.from_numpy(imgs).cuda()
    boxes, scores, valid, idx = [t.cpu().numpy() for t in prop.propose(x)]
    deltas, obj = [t.cpu().numpy() for t in prop.forward(x)]
    # conv stack vs oracle: 1e-4 on deltas and objectness
    reg_ref, cls_ref = cv.rpn_forward(backbone, imgs, weights)
    assert np.abs(deltas.reshape(reg_ref.shape) - reg_ref).max() <= 1e-4
    assert np.abs(obj.reshape(cls_ref.shape) - cls_ref).max() <= 1e-4
    # box stage vs oracle fed the GPU head outputs: integer outputs bit-exact, floats <= 1e-4
    anchors = bo.generate_anchors(hp)
    assert np.array_equal(prop.anchors.cpu().numpy(), anchors)
    dec = bo.get_bboxes_from_deltas(anchors, bo.scale_deltas(deltas, hp["variances"]))
    rb, rs, _rc, rv, ri = co.combined_nms(dec[:, :, None, :], obj[:, :, None], 300, 300, iou_threshold=0.7)
    assert np.array_equal(valid, rv) and np.array_equal(idx, ri)
    assert np.abs(boxes - rb).max() <= 1e-4 and np.array_equal(scores, rs)
    # the unfused route (reference's separate calls) agrees with the fused kernel
    ub, us, uv, ui = [t.cpu().numpy() for t in prop.propose_unfused(x)]
    assert np.array_equal(uv, valid) and np.array_equal(ui, idx) and np.array_equal(ub, boxes)
    # the reference's own selector: top-10 by objectness (predictor.py:58-60)
    tb, order = prop.top_k(x, 10)
    assert np.array_equal(order.cpu().numpy(), bo.top_k_indices(obj, 10))


def test_propose_full_size_properties():
    """C2 (B=8, VGG16, 500x500): output contract and size-independent properties."""
    prop = Proposer("vgg16", max_batch=8, iou_threshold=0.7)
    imgs = torch.rand((8, 500, 500, 3), generator=torch.Generator().manual_seed(0)).cuda()
    boxes, scores, valid, idx = [t.clone() for t in prop.propose(imgs)]
    assert boxes.shape == (8, 300, 4) and scores.shape == (8, 300) and valid.dtype == torch.int32
    assert (valid > 0).all() and (valid <= 300).all()
    assert boxes.min() >= 0 and boxes.max() <= 1                      # clip_boxes
    for b in range(8):
        v = int(valid[b])
        s = scores[b, :v]
        assert (s[:-1] >= s[1:]).all() and (scores[b, v:] == 0).all() and (idx[b, v:] == -1).all()
        assert len(set(idx[b, :v].tolist())) == v
    again = [t.clone() for t in prop.propose(imgs)]                    # deterministic
    assert torch.equal(again[0], boxes) and torch.equal(again[3], idx)
    rec = prop.pack_records(boxes, scores, valid)
    b2, s2, v2 = prop.unpack_records(rec, 300)
    assert torch.equal(b2, boxes) and torch.equal(v2, valid)


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
def test_propose_split_precision_full_size(precision):
    """C2 with the split-precision conv stack: proposals stay within the parity bound of the exact-f32 path.
    Integer outputs are compared through the oracle fed the SAME head outputs (bit-exact), floats <= 1e-4."""
    hp = dict(bo.get_hyper_params("vgg16"))
    weights = synthetic_weights("vgg16", hp, seed=1)
    imgs = torch.rand((2, 500, 500, 3), generator=torch.Generator().manual_seed(0)).cuda()
    p32 = Proposer("vgg16", hyper_params=dict(hp), weights=weights, precision="f32", max_batch=2)
    d32, s32 = [t.clone() for t in p32.forward(imgs)]
    del p32
    prop = Proposer("vgg16", hyper_params=dict(hp), weights=weights, precision=precision, max_batch=2)
    boxes, scores, valid, idx = [t.cpu().numpy() for t in prop.propose(imgs)]
    deltas, obj = prop.forward(imgs)
    assert (deltas - d32).abs().max().item() <= 1e-4 and (obj - s32).abs().max().item() <= 1e-4
    anchors = bo.generate_anchors(hp)
    dec = co.decode(anchors, deltas.cpu().numpy(), np.float32(hp["variances"]))
    rb, rs, _rc, rv, ri = co.combined_nms(dec[:, :, None, :], obj.cpu().numpy()[:, :, None], 300, 300, iou_threshold=0.7)
    assert np.array_equal(valid, rv) and np.array_equal(idx, ri)
    assert np.abs(boxes - rb).max() <= 1e-4 and np.array_equal(scores, rs)


def test_pipelined_propose_matches_serial():
    """overlap_nms=True (NMS on a side stream, double-buffered) returns exactly what the serial path returns,
    also when batches alternate and are consumed late."""
    hp = dict(bo.get_hyper_params("vgg16", img_size=160, feature_map_shape=10))
    weights = synthetic_weights("vgg16", hp, seed=3)
    serial = Proposer("vgg16", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3")
    piped = Proposer("vgg16", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3", overlap_nms=True)
    gen = torch.Generator().manual_seed(5)
    batches = [torch.rand((2, 160, 160, 3), generator=gen).cuda() for _ in range(5)]
    want = [[t.clone() for t in serial.propose(x)] for x in batches]
    got = []
    for x in batches:                               # issue everything back to back, read afterwards
        out = piped.propose(x)
        piped.wait()
        got.append([t.clone() for t in out])
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for a, b in zip(w, g):
            assert torch.equal(a, b)


def test_propose_is_graph_capturable():
    """The whole step (conv stack + fused decode/NMS) performs no allocation or synchronisation, so it can be
    captured into a HIP graph and replayed; replays give the eager result bit for bit."""
    hp = dict(bo.get_hyper_params("vgg16", img_size=160, feature_map_shape=10))
    weights = synthetic_weights("vgg16", hp, seed=3)
    prop = Proposer("vgg16", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3")
    gen = torch.Generator().manual_seed(6)
    a, b = [torch.rand((2, 160, 160, 3), generator=gen).cuda() for _ in range(2)]
    want_a = [t.clone() for t in prop.propose(a)]
    want_b = [t.clone() for t in prop.propose(b)]
    static_in = a.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up on the capture stream
        prop.propose(static_in)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = prop.propose(static_in)
    for src, want in ((b, want_b), (a, want_a), (b, want_b)):
        static_in.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        for o, w in zip(outs, want):
            assert torch.equal(o, w)


def test_side_stream_runs_beside_the_conv_stream_whatever_was_created_before():
    """HIP maps streams onto four hardware queues in creation order (the default stream's included): a Proposer created after a
    few other streams used to get an NMS stream that SHARED the conv stream's queue -- same results, no overlap, 35 % slower at
    one image.  The Proposer now tests its side stream (rpn_stream_spin: one sleeping wave per stream) and draws another."""
    from tf_rpn_amd import predictor as P
    cur = torch.cuda.current_stream()
    scrambled = [torch.cuda.Stream() for _ in range(7)]              # whatever the process created before
    for s in scrambled:
        P._streams_overlap(cur, s)
    shared = [s for s in scrambled if not P._streams_overlap(cur, s)]
    hp = dict(bo.get_hyper_params("mobilenet_v2", img_size=160, feature_map_shape=10))
    weights = synthetic_weights("mobilenet_v2", hp, seed=5)
    for _ in range(3):                                               # three in a row: each draws from torch's pool where the last stopped
        prop = Proposer("mobilenet_v2", hyper_params=dict(hp), weights=weights, max_batch=1, precision="f16x3", overlap_nms=True)
        assert P._streams_overlap(cur, prop._nms_stream)
    if shared:                                                       # the probe itself tells the two cases apart
        assert not P._streams_overlap(cur, shared[0])
    with pytest.raises(ValueError):
        P.L.check(P.L.lib().rpn_stream_spin(P.L.vp(cur.cuda_stream), 20000), "rpn_stream_spin")


def test_proposer_pool_matches_a_single_proposer():
    """Two pipelines in flight (ProposerPool): every batch of a sequence gives the proposals a single Proposer gives, bit for bit
    (each pipeline is a Proposer of its own; only the overlap on the device differs)."""
    from tf_rpn_amd.predictor import ProposerPool
    hp = dict(bo.get_hyper_params("mobilenet_v2", img_size=160, feature_map_shape=10))
    weights = synthetic_weights("mobilenet_v2", hp, seed=5)
    single = Proposer("mobilenet_v2", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3")
    pool = ProposerPool(2, "mobilenet_v2", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3")
    gen = torch.Generator().manual_seed(9)
    seq = [torch.rand((2, 160, 160, 3), generator=gen).cuda() for _ in range(4)]
    want = [[t.clone() for t in single.propose(x)] for x in seq]
    got = [pool.propose_async(x) for x in seq]         # two pipelines x two output slots each: all four stay intact
    pool.wait()
    torch.cuda.synchronize()
    for k in range(len(seq)):
        for g, w in zip(got[k], want[k]):
            assert torch.equal(g, w), k
    for g, w in zip(pool.propose(seq[1]), want[1]):
        assert torch.equal(g, w)


def test_proposer_pool_input_freed_and_reallocated_between_calls():
    """The caller drops each batch right after handing it over and immediately allocates + fills same-sized tensors on ITS stream
    (what ``pool.propose_async(next_batch())`` does): the pipeline stream must still read the original pixels -- ProposerPool tells
    the caching allocator that the pipeline's stream uses the block (``record_stream``)."""
    from tf_rpn_amd.predictor import ProposerPool
    hp = dict(bo.get_hyper_params("mobilenet_v2", img_size=160, feature_map_shape=10))
    weights = synthetic_weights("mobilenet_v2", hp, seed=5)
    single = Proposer("mobilenet_v2", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3")
    pool = ProposerPool(2, "mobilenet_v2", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3")
    gen = torch.Generator().manual_seed(11)
    host = [torch.rand((2, 160, 160, 3), generator=gen) for _ in range(6)]
    want = [[t.clone() for t in single.propose(h.cuda())] for h in host]
    torch.cuda.synchronize()
    got = []
    for h in host:
        x = h.cuda()
        out = pool.propose_async(x)
        got.append([t for t in out])
        del x, out
        junk = [torch.full((2, 160, 160, 3), float("nan"), device="cuda") for _ in range(3)]   # would land in the freed block
        del junk
        if len(got) % 2 == 0:                       # the pool's two pipelines x two output slots: collect before they are reused
            pool.wait()
            torch.cuda.synchronize()
            for k in (len(got) - 2, len(got) - 1):
                got[k] = [t.clone() for t in got[k]]
    for k in range(len(host)):
        for g, w in zip(got[k], want[k]):
            assert torch.equal(g, w), k


def test_pipelined_distributed_path_world1():
    """The N > 1 code path (record packing + all-gather one step behind the convs) on a world-size-1 RCCL group:
    gathered records must equal the serial proposals, in order, including the flushed last batch."""
    import socket
    import torch.distributed as dist
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        hp = dict(bo.get_hyper_params("vgg16", img_size=160, feature_map_shape=10))
        weights = synthetic_weights("vgg16", hp, seed=3)
        serial = Proposer("vgg16", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3")
        # (created AFTER the RCCL group on purpose: the communicator's streams took hardware queues first; the Proposer tests its
        # side stream against the conv stream and draws another until the two run beside each other)
        from tf_rpn_amd import predictor as P
        piped = Proposer("vgg16", hyper_params=dict(hp), weights=weights, max_batch=2, precision="f16x3", overlap_nms=True)
        assert P._streams_overlap(torch.cuda.current_stream(), piped._nms_stream)
        M = piped.topn
        gen = torch.Generator().manual_seed(8)
        batches = [torch.rand((2, 160, 160, 3), generator=gen).cuda() for _ in range(4)]
        want = []
        for x in batches:
            b, s, v, _ = serial.propose(x)
            want.append(serial.pack_records(b, s, v).clone())
        bufs = [torch.empty((2, M * 5 + 1), device="cuda") for _ in range(2)]
        got = []
        for x in batches:
            out = piped.propose_distributed_pipelined(x, bufs)
            if out is not None:
                got.append(out.clone())
        got.append(piped.flush_distributed(bufs).clone())
        torch.cuda.synchronize()
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert torch.equal(g, w)
        rec = serial.propose_distributed(batches[0])                      # unpipelined helper, world size 1
        assert torch.equal(rec, want[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_predictor_cli_on_a_folder_of_images(tmp_path):
    """``python -m tf_rpn_amd.predictor`` = the reference script's custom-image branch (predictor.py:8-60) minus the
    drawing: folder -> PIL/Lanczos -> model -> decode -> top-k (or NMS) -> pixel boxes, ragged last batch included."""
    import json
    from PIL import Image
    from tf_rpn_amd import predictor
    rng = np.random.RandomState(1)
    for i, (h, w) in enumerate(((60, 80), (90, 70), (50, 50))):
        Image.fromarray(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(tmp_path / ("img%d.png" % i))
    imgdir = tmp_path
    outdir = tmp_path.parent / (tmp_path.name + "_out")
    outdir.mkdir()
    out = outdir / "p.json"                       # (like the reference, every file of the folder is opened as an image)
    res = predictor.main(["--backbone", "mobilenet_v2", "--images", str(imgdir), "--synthetic-weights", "--batch-size", "2",
                          "--top-k", "7", "--out", str(out)])
    assert len(res) == 3 and json.load(open(out)) == res
    for r in res:
        assert len(r["boxes_y1x1y2x2"]) == 7 and len(r["scores"]) == 7 and r["img_size"] == 500
        assert all(r["scores"][i] >= r["scores"][i + 1] for i in range(6))          # top_k: descending
    nms = predictor.main(["--backbone", "mobilenet_v2", "--images", str(tmp_path), "--synthetic-weights", "--nms",
                          "--out", str(out)])
    assert len(nms) == 3 and all(0 < len(r["boxes_y1x1y2x2"]) <= 300 for r in nms)
    # same images through the library calls the script is made of
    from tf_rpn_amd.utils import data_utils, train_utils
    hp = train_utils.get_hyper_params("mobilenet_v2")
    paths = sorted(data_utils.get_custom_imgs(str(tmp_path)))
    paths = [p for p in paths if p.endswith(".png")]
    imgs = np.stack([im for im, _, _ in data_utils.custom_data_generator(paths, 500, 500)])
    prop = predictor.Proposer("mobilenet_v2", hyper_params=hp, precision="f16x3", max_batch=3)
    boxes, _idx, scores = prop.top_k(torch.from_numpy(imgs).cuda(), 7, return_scores=True)
    for i, r in enumerate(res):
        np.testing.assert_allclose(r["scores"], scores[i].cpu().numpy(), rtol=0, atol=1e-6)
        np.testing.assert_array_equal(np.array(r["boxes_y1x1y2x2"]), np.round(boxes[i].cpu().numpy() * np.float32(500)).astype(int))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--force-dist", "--no-extra-legs"], ["--serial-nms", "--steps", "4", "--no-extra-legs"]])
def test_bench_prints_one_json_line(extra):
    """The driver's contract: rank 0 prints ONE JSON line on stdout (RCCL's version banner must not land there), with the
    roofline object; `--force-dist` runs the N > 1 code path on a world-size-1 RCCL group."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "16", "--warmup", "1"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["value"] > 100
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "conv3x3_split16_dma" in rf["kernel"] and 0.2 < rf["frac"] < 1.0
    assert d["rccl_ranks"] == 1 and "traffic_source" in rf
    if "--force-dist" in extra:
        assert d["allgather_ms"] is not None and d["allgather_ms"] < 5.0
        # BASELINE.json's two 8-GPU configs at their per-GPU shapes, through record packing + the all-gather (world-size-1 RCCL here)
        mg = d["multi_gpu_configs"]
        assert set(mg) == {"c4", "c5"}
        assert "vgg16" in mg["c4"]["workload"] and "batch 32 per GPU" in mg["c4"]["workload"] and mg["c4"]["global_batch"] == 32
        assert "mobilenet_v2, 1024x1024, 15 anchors" in mg["c5"]["workload"] and "batch 1 per GPU" in mg["c5"]["workload"]
        for leg in mg.values():
            assert leg["rccl_ranks"] == 1 and leg["images_per_s"] > 100 and 0 < leg["allgather_ms"] < 5.0
            assert len(leg["per_rank"]["images_per_s"]) == 1 and leg["record_bytes_per_rank"] % 6004 == 0
            ck = leg["checks"]
            assert ck["valid_min"] >= 1 and ck["proposals_finite"] and ck["own_rows_match"] and not ck["any_rank_failed"] and not ck["f16_range"]
    else:
        assert "multi_gpu_configs" not in d
    if "--no-extra-legs" in extra:
        assert "c3" not in d and d["nms_boxes_per_sec"] > 1e7
        return
    # the N = 1 line carries the box-path leg (configs[2], the metric's "NMS boxes/sec") and the exact-f32 leg
    c3 = d["c3"]
    assert c3["B"] == 64 and c3["iou_map"]["GBps"] > 100 and c3["decode"]["GBps"] > 100
    assert d["nms_boxes_per_sec"] == c3["decode_nms_iou0.7"]["boxes_per_sec"] > 1e8
    f32 = d["exact_f32"]
    assert f32["dtype"] == "f32" and 100 < f32["value"] < d["value"] and 0.2 < f32["roofline"]["frac"] < 1.0
    # ... and one figure per other BASELINE.json config (per-GPU shapes)
    oc = d["other_configs"]
    assert set(oc) == {"c1", "c4", "c5", "mobilenet_v2_b8"}
    assert oc["c4"]["value"] > 100 and "batch 32" in oc["c4"]["workload"] and "1024x1024, 15 anchors" in oc["c5"]["workload"]
    assert all(v["unit"] == "images/s" and v["value"] > 0 and v["conv_launches_per_step"] <= 20 for v in oc.values())
