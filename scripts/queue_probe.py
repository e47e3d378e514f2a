"""Which HIP streams run concurrently?  One sleeping wave per stream (rpn_stream_spin), pairwise: '+' beside each other, '.' one behind
the other (shared hardware queue).  Stream 0 is the default stream, 1..N are torch.cuda.Stream() in creation order.
usage: [GPU_MAX_HW_QUEUES=n] python scripts/queue_probe.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd import predictor as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n)]
for s in streams:                      # first launches
    P._streams_overlap(streams[0], s)
print("   " + " ".join("%2d" % j for j in range(len(streams))))
for i, a in enumerate(streams):
    print("%2d " % i + " ".join(" -" if i == j else (" +" if P._streams_overlap(a, b) else " .") for j, b in enumerate(streams)), flush=True)
