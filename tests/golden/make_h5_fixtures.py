"""Keras-layout HDF5 weight files written by the REAL h5py / libhdf5, used to pin tf_rpn_amd/utils/h5_weights.py.

Run with the interpreter that has h5py (here: /opt/conda/bin/python3.9, h5py 3.3.0 / HDF5 1.10.6):
    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py
TensorFlow / Keras themselves are not available in this container, so the files are written with h5py following
tensorflow/python/keras/saving/hdf5_format.py (save_weights_to_hdf5_group / save_attributes_to_hdf5_group):
root attrs layer_names / backend / keras_version, one group per layer with attr weight_names, one dataset per weight
named "<layer>/<weight>:0".  Every array's contents are a function of its path (expected_array below), so the tests
need no companion file."""
import os
import zlib

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def expected_array(path, shape):
    """Deterministic float32 contents for the dataset at ``path``."""
    n = int(np.prod(shape)) if len(shape) else 1
    k = zlib.crc32(path.encode()) % 97
    return ((np.arange(n, dtype=np.float64) * 0.25 + k) * (-1.0) ** k).astype(np.float32).reshape(shape)


VGG_TINY = [   # (layer, [(weight, shape)])
    ("input_1", []),
    ("block1_conv1", [("kernel", (3, 3, 3, 4)), ("bias", (4,))]),
    ("block1_pool", []),
    ("rpn_conv", [("kernel", (3, 3, 4, 8)), ("bias", (8,))]),
    ("rpn_cls", [("kernel", (1, 1, 8, 9)), ("bias", (9,))]),
    ("rpn_reg", [("kernel", (1, 1, 8, 36)), ("bias", (36,))]),
]


def mnv2_like(n_blocks=16):
    layers = [("input_1", []), ("Conv1", [("kernel", (3, 3, 3, 8))]),
              ("bn_Conv1", [("gamma", (8,)), ("beta", (8,)), ("moving_mean", (8,)), ("moving_variance", (8,))])]
    for i in range(n_blocks):                         # ~50 layer groups: the root group's B-tree points to several symbol-table nodes
        layers.append(("block_%d_expand" % i, [("kernel", (1, 1, 8, 16))]))
        layers.append(("block_%d_depthwise" % i, [("depthwise_kernel", (3, 3, 16, 1))]))
        layers.append(("block_%d_depthwise_BN" % i, [("gamma", (16,)), ("beta", (16,)), ("moving_mean", (16,)),
                                                      ("moving_variance", (16,))]))
    return layers


def write(path, layers, strings="fixed", prefix="", libver=None, chunked=False, extra_root_attrs=None):
    kw = {} if libver is None else {"libver": libver}
    with h5py.File(path, "w", **kw) as f:
        root = f.create_group(prefix) if prefix else f

        def set_strings(obj, name, values):
            if strings == "fixed":                    # h5py 2.x (TF 2.0 era): numpy S array -> fixed-length strings
                obj.attrs[name] = np.array([v.encode() for v in values], dtype="S") if values else np.zeros((0,), "S1")
            else:                                     # h5py 3.x: list of bytes -> variable-length strings
                obj.attrs[name] = [v.encode() for v in values]
        set_strings(root, "layer_names", [l for l, _ in layers])
        if strings == "fixed":
            root.attrs["backend"] = np.bytes_(b"tensorflow")
            root.attrs["keras_version"] = np.bytes_(b"2.2.4-tf")
        else:
            root.attrs["backend"] = "tensorflow"
            root.attrs["keras_version"] = "2.2.4-tf"
        for k, v in (extra_root_attrs or {}).items():
            f.attrs[k] = v
        for lname, ws in layers:
            g = root.create_group(lname)
            set_strings(g, "weight_names", ["%s/%s:0" % (lname, w) for w, _ in ws])
            for w, shape in ws:
                name = "%s/%s:0" % (lname, w)
                data = expected_array("%s/%s" % (lname, name), shape)
                if chunked:
                    g.create_dataset(name, data=data, chunks=True, compression="gzip")
                else:
                    g.create_dataset(name, data=data)


if __name__ == "__main__":
    write(os.path.join(HERE, "keras_weights_fixed_strings.h5"), VGG_TINY, strings="fixed")
    write(os.path.join(HERE, "keras_weights_vlen_strings.h5"), VGG_TINY, strings="vlen")
    write(os.path.join(HERE, "keras_full_model.h5"), VGG_TINY, strings="vlen", prefix="model_weights",
          extra_root_attrs={"model_config": '{"class_name": "Model", "config": {"name": "rpn", "layers": []}}' + " " * 700,
                            "keras_version": "2.2.4-tf", "backend": "tensorflow"})
    write(os.path.join(HERE, "keras_weights_many_layers.h5"), mnv2_like(), strings="fixed")
    write(os.path.join(HERE, "keras_weights_latest_libver.h5"), VGG_TINY, strings="vlen", libver="latest")
    write(os.path.join(HERE, "keras_weights_chunked_gzip.h5"), VGG_TINY[:2], strings="fixed", chunked=True)
    # Keras splits attributes above 64512 bytes into <name>0, <name>1, ... (save_attributes_to_hdf5_group): same naming,
    # small contents
    path = os.path.join(HERE, "keras_weights_chunked_attrs.h5")
    write(path, VGG_TINY, strings="fixed")
    with h5py.File(path, "r+") as f:
        names = f.attrs["layer_names"]
        del f.attrs["layer_names"]
        f.attrs["layer_names0"] = names[:4]
        f.attrs["layer_names1"] = names[4:]
        g = f["rpn_reg"]
        wn = g.attrs["weight_names"]
        del g.attrs["weight_names"]
        g.attrs["weight_names0"] = wn[:1]
        g.attrs["weight_names1"] = wn[1:]
    for n in sorted(os.listdir(HERE)):
        if n.endswith(".h5"):
            print(n, os.path.getsize(os.path.join(HERE, n)))
