"""Which HIP streams run concurrently?  One sleeping wave per stream (rpn_stream_spin), pairwise: '+' beside each other, '.' one behind
the other (shared hardware queue).  Stream 0 is the default stream, 1..N are torch.cuda.Stream() in creation order.
usage: [GPU_MAX_HW_QUEUES=n] python scripts/queue_probe.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd import predictor as P

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 9
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n)]
for s in streams:                      # first launches
    P._streams_overlap(streams[0], s)
print("   " + " ".join("%2d" % j for j in range(len(streams))))
for i, a in enumerate(streams):
    print("%2d " % i + " ".join(" -" if i == j else (" +" if P._streams_overlap(a, b) else " .") for j, b in enumerate(streams)), flush=True)

if "--pool" in sys.argv:               # where a ProposerPool(2) created NOW puts its four streams
    from tf_rpn_amd.models._rpn_model import synthetic_weights
    from tf_rpn_amd.utils import train_utils
    hp = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=160, feature_map_shape=10))
    pool = P.ProposerPool(2, "mobilenet_v2", hyper_params=hp, weights=synthetic_weights("mobilenet_v2", hp, seed=1), max_batch=1, precision="f16x3")
    ss = [("default", torch.cuda.current_stream())]
    for i, (p, s) in enumerate(zip(pool.pipelines, pool.streams)):
        ss += [("conv%d" % i, s), ("nms%d" % i, p._nms_stream)]
    print("         " + " ".join("%7s" % n for n, _ in ss))
    for n, a in ss:
        print("%8s " % n + " ".join("%7s" % ("-" if a is b else ("+" if P._streams_overlap(a, b) else ".")) for _, b in ss), flush=True)
