#!/bin/bash
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from tf_rpn_amd.utils import train_utils
print(json.dumps(bench.c3_leg(dict(train_utils.get_hyper_params("vgg16")))["decode"]))
PY
