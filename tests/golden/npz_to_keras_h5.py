"""Write a flat .npz of layer weights (keys "<layer>/<param>", as RPNModel.save_weights) as a Keras-layout .h5 weights
file with the real h5py -- the way ``model.save_weights("x.h5")`` lays it out (hdf5_format.save_weights_to_hdf5_group).
Run with an interpreter that has h5py:  /opt/conda/bin/python3.9 npz_to_keras_h5.py in.npz out.h5 [fixed|vlen]"""
import sys

import h5py
import numpy as np

KERAS_NAME = {"kernel": "kernel", "bias": "bias", "gamma": "gamma", "beta": "beta", "mean": "moving_mean",
              "var": "moving_variance"}
ORDER = ["kernel", "bias", "gamma", "beta", "mean", "var"]


def main(src, dst, strings="fixed"):
    data = np.load(src)
    layers = {}
    for key in data.files:
        layer, param = key.rsplit("/", 1)
        layers.setdefault(layer, {})[param] = data[key]

    def put(obj, name, values):
        if strings == "fixed":
            obj.attrs[name] = np.array([v.encode() for v in values], dtype="S") if values else np.zeros((0,), "S1")
        else:
            obj.attrs[name] = [v.encode() for v in values]
    with h5py.File(dst, "w") as f:
        put(f, "layer_names", list(layers))
        f.attrs["backend"] = np.bytes_(b"tensorflow")
        f.attrs["keras_version"] = np.bytes_(b"2.2.4-tf")
        for lname, params in layers.items():
            g = f.create_group(lname)
            names = []
            for p in ORDER:
                if p in params:
                    arr = params[p]
                    kn = KERAS_NAME[p]
                    if p == "kernel" and arr.ndim == 4 and arr.shape[3] == 1 and "depthwise" in lname:
                        kn = "depthwise_kernel"
                    names.append("%s/%s:0" % (lname, kn))
                    g.create_dataset(names[-1], data=np.asarray(arr, dtype=np.float32))
            put(g, "weight_names", names)


if __name__ == "__main__":
    main(*sys.argv[1:])
