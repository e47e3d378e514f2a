"""How deep into the score-sorted candidate list does NMS(300) walk, and what does the kernel cost?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases
from oracle import bbox_oracle as bo
from tf_rpn_amd.utils import bbox_utils
from tf_rpn_amd.predictor import Proposer

def depth(scores, idx, valid):
    out = []
    for b in range(scores.shape[0]):
        order = np.argsort(-scores[b], kind="stable")
        rank = np.empty_like(order); rank[order] = np.arange(len(order))
        out.append(int(rank[idx[b, valid[b] - 1]]) + 1 if valid[b] else 0)
    return out

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

VAR = np.float32([0.1, 0.1, 0.2, 0.2])
anchors = bo.generate_anchors(bo.get_hyper_params("vgg16"))
A = len(anchors)
for B in (8, 64):
    deltas = np.random.RandomState(2).standard_normal((B, A, 4)).astype(np.float32)
    scores = cases.permutation_scores(np.random.RandomState(3), B, A)
    d, s, a = torch.from_numpy(deltas).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(anchors).cuda()
    for thr in (0.5, 0.7):
        fb, fs, fi, fv = bbox_utils.decode_and_nms(a, d, s, VAR, 300, iou_threshold=thr)
        us = timeit(lambda: bbox_utils.decode_and_nms(a, d, s, VAR, 300, iou_threshold=thr))
        dp = depth(scores, fi.cpu().numpy(), fv.cpu().numpy())
        print("C3 synthetic B=%d iou=%.1f: %.1f us (incl. python alloc), walk depth min/mean/max = %d/%.0f/%d"
              % (B, thr, us, min(dp), np.mean(dp), max(dp)), flush=True)
prop = Proposer("vgg16", max_batch=8, iou_threshold=0.7, precision="f16x3")
imgs = torch.rand((8, 500, 500, 3), generator=torch.Generator().manual_seed(0)).cuda()
boxes, sc, valid, idx = prop.propose(imgs)
dl, obj = prop.forward(imgs)
dp = depth(obj.cpu().numpy(), idx.cpu().numpy(), valid.cpu().numpy())
print("model outputs B=8 iou=0.7: valid", valid.cpu().numpy().tolist(), "walk depth", dp, flush=True)
o = obj.cpu().numpy()
print("objectness: min %.4f max %.4f, distinct values per image %d of %d" % (o.min(), o.max(), len(np.unique(o[0])), o.shape[1]))
