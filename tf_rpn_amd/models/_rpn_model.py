"""Python handle over the native ``rpn_model`` of librpn_hip.so -- the object that stands where
the reference's Keras ``rpn_model`` stands (models/rpn_vgg16.py:21, predictor.py:41-50).

Only the inference surface the proposal path touches is mirrored: ``predict_on_batch``,
``__call__`` and ``load_weights`` (Keras ``.h5`` checkpoints through ``utils/h5_weights.py`` -- no h5py
needed -- or a flat ``.npz``).
"""
import ctypes

import numpy as np
import torch

from .. import _lib as L


class FeatureExtractor(object):
    """Stand-in for the Keras layer handle ``get_model`` returns second (models/rpn_vgg16.py:17,22):
    names the tap layer and can fetch its activation after a forward pass."""

    def __init__(self, model, name):
        self._model = model
        self.name = name

    @property
    def output_shape(self):
        return self._model.activation_shape(self.name)

    def output(self):
        return self._model.get_activation(self.name)


class RPNModel(object):
    def __init__(self, backbone, hyper_params, precision="f32", max_batch=8, keep_activations=False):
        if backbone not in L.BACKBONES:
            raise ValueError("unknown backbone %r" % (backbone,))
        if precision not in L.PRECISIONS:
            raise ValueError("unknown precision %r (choose from %s)" % (precision, sorted(L.PRECISIONS)))
        self.backbone = backbone
        self.precision = precision
        self.img_size = int(hyper_params["img_size"])
        self.anchor_count = int(hyper_params["anchor_count"])
        self.max_batch = int(max_batch)
        self._h = L.vp(0)
        lib = L.lib()
        st = lib.rpn_model_create(L.BACKBONES[backbone], self.img_size, self.anchor_count,
                                  L.PRECISIONS[precision], self.max_batch, ctypes.byref(self._h))
        L.check(st, "rpn_model_create")
        if keep_activations:
            L.check(lib.rpn_model_keep_activations(self._h, 1), "rpn_model_keep_activations")
        self.feature_map_shape = int(lib.rpn_model_feature_map_shape(self._h))
        self.flops_per_image = float(lib.rpn_model_flops_per_image(self._h))
        self.layers = self._enumerate_layers()
        self.tap_layer = "block5_conv3" if backbone == "vgg16" else "block_13_expand"

    # ---- introspection ----------------------------------------------------------------
    def _enumerate_layers(self):
        lib = L.lib()
        out = []
        buf = ctypes.create_string_buffer(128)
        shape = (ctypes.c_int * 4)()
        kind = ctypes.c_int(0)
        for i in range(lib.rpn_model_num_layers(self._h)):
            L.check(lib.rpn_model_layer_info(self._h, i, buf, 128, shape, ctypes.byref(kind)), "rpn_model_layer_info")
            name = buf.value.decode()
            L.check(lib.rpn_model_layer_bn_name(self._h, i, buf, 128), "rpn_model_layer_bn_name")
            out.append({"name": name, "bn_name": buf.value.decode(), "shape": tuple(shape), "kind": int(kind.value)})
        return out

    def memory_bytes(self):
        w, a = ctypes.c_size_t(0), ctypes.c_size_t(0)
        L.check(L.lib().rpn_model_memory_bytes(self._h, ctypes.byref(w), ctypes.byref(a)), "rpn_model_memory_bytes")
        return int(w.value), int(a.value)

    # ---- weights ------------------------------------------------------------------------
    def set_weights(self, weights, partial=False):
        """weights: {layer_name: {"kernel": HWIO, "bias": (Cout,)}} and, for layers followed by
        BatchNorm, {bn_name: {"gamma", "beta", "mean", "var"}} -- Keras layer names.  ``partial``: layers absent
        from ``weights`` are left as they are (Keras ``by_name`` loading); forward still refuses to run until every
        layer has been set once.  Returns the names of the layers set."""
        lib = L.lib()
        fp = lambda a: (np.ascontiguousarray(a, dtype=np.float32))
        done = []
        for layer in self.layers:
            name, bn = layer["name"], layer["bn_name"]
            if name not in weights:
                if partial:
                    continue
                raise KeyError("weights for layer %r are missing" % name)
            kernel = fp(weights[name]["kernel"])
            if tuple(kernel.shape) != layer["shape"]:
                raise ValueError("layer %r: kernel shape %s, expected %s" % (name, kernel.shape, layer["shape"]))
            bias = weights[name].get("bias")
            bias = fp(bias) if bias is not None else None
            args = [kernel.ctypes.data_as(L.c_float_p), bias.ctypes.data_as(L.c_float_p) if bias is not None else None]
            keep = [kernel, bias]
            if bn:
                if bn not in weights:
                    raise KeyError("BatchNorm parameters %r (after %r) are missing" % (bn, name))
                for key in ("gamma", "beta", "mean", "var"):
                    arr = fp(weights[bn][key])
                    keep.append(arr)
                    args.append(arr.ctypes.data_as(L.c_float_p))
            else:
                args += [None, None, None, None]
            L.check(lib.rpn_model_set_layer(self._h, name.encode(), *args), "rpn_model_set_layer(%s)" % name)
            done.append(name)
        return done

    def load_weights(self, path, by_name=True):
        """``model.load_weights(path, by_name=True)`` of the reference (predictor.py:43-44).

        ``path``: a Keras ``.h5`` / ``.hdf5`` weights file (what the reference's trainer checkpoints, or the
        ``model_weights`` of a full-model file; read by ``utils/h5_weights.py``, no h5py needed), or a flat ``.npz``
        written by ``save_weights`` (keys ``<layer>/<param>``).  Layers are matched by their Keras names; BatchNorm is
        folded into the preceding conv at load time.  ``by_name=True``: layers of this model that the file does not
        contain are left untouched, layers of the file that this model lacks are ignored.  Returns the layers set."""
        with open(path, "rb") as f:
            is_h5 = f.read(8) == b"\x89HDF\r\n\x1a\n"
        if is_h5:
            from ..utils import h5_weights
            weights = h5_weights.to_layer_arrays(h5_weights.read_keras_weights(path)[0])
        else:
            data = np.load(path)
            weights = {}
            for key in data.files:
                layer, param = key.rsplit("/", 1)
                weights.setdefault(layer, {})[param] = data[key]
        return self.set_weights(weights, partial=bool(by_name))

    @staticmethod
    def save_weights(weights, path):
        np.savez(path, **{"%s/%s" % (layer, p): v for layer, d in weights.items() for p, v in d.items()})

    # ---- forward ------------------------------------------------------------------------
    def predict_on_batch(self, imgs):
        """imgs (B, img_size, img_size, 3) float32 in [0,1] -> [rpn_reg (B,F,F,4K), rpn_cls (B,F,F,K)]
        (the reference's output order, models/rpn_vgg16.py:21)."""
        x, was_np = L.to_device(imgs)
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.img_size, self.img_size, 3):
            raise ValueError("imgs must be (B,%d,%d,3) NHWC, got %s" % (self.img_size, self.img_size, tuple(x.shape)))
        B = int(x.shape[0])
        F, K = self.feature_map_shape, self.anchor_count
        reg = torch.empty((B, F, F, 4 * K), dtype=torch.float32, device="cuda")
        cls = torch.empty((B, F, F, K), dtype=torch.float32, device="cuda")
        self.forward_into(x, reg, cls)
        if was_np and self.precision in ("f16x3", "fp16x3"):
            # numpy in / numpy out synchronises anyway: never hand back silently wrong outputs.  CUDA-tensor callers
            # (the hot path) poll ``status()`` themselves, a forward never reads the flag back.
            self.raise_on_range_error()
        return [L.from_device(reg, was_np), L.from_device(cls, was_np)]

    __call__ = predict_on_batch

    def forward_into(self, x, reg, cls):
        """Forward on preallocated CUDA tensors (no allocation: graph-capturable).  The C side sees raw pointers, so
        dtype / device / layout / shape are checked here."""
        B = int(x.shape[0]) if isinstance(x, torch.Tensor) and x.dim() == 4 else -1
        F, K = self.feature_map_shape, self.anchor_count
        for t, shape, what in ((x, (B, self.img_size, self.img_size, 3), "imgs"), (reg, (B, F, F, 4 * K), "reg"),
                               (cls, (B, F, F, K), "cls")):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                    and tuple(t.shape) == shape):
                raise ValueError("%s must be a contiguous CUDA float32 tensor of shape %s, got %s"
                                 % (what, shape, (tuple(t.shape), t.dtype, t.device) if isinstance(t, torch.Tensor) else type(t)))
        if not 1 <= B <= self.max_batch:
            raise ValueError("batch %d outside [1, %d]" % (B, self.max_batch))
        st = L.lib().rpn_model_forward(self._h, L.ptr(x), int(x.shape[0]), L.ptr(reg), L.ptr(cls), L.stream_ptr())
        L.check(st, "rpn_model_forward")

    # ---- float16 range status (precision "f16x3") ---------------------------------------------------
    def status(self, reset=False):
        """Sticky flags raised on the device by the forwards so far -> {"f16_range": bool}.  ``f16_range``: some
        activation did not fit float16 when it was written in split form (|x| > 65504 or non-finite), the outputs of
        that forward are invalid.  Synchronises the current stream (not part of the hot path)."""
        flags = ctypes.c_uint(0)
        L.check(L.lib().rpn_model_status(self._h, ctypes.byref(flags), 1 if reset else 0, L.stream_ptr()),
                "rpn_model_status")
        return {"f16_range": bool(flags.value & L.STATUS_F16_RANGE)}

    def raise_on_range_error(self):
        if self.status(reset=True)["f16_range"]:
            raise FloatingPointError("precision 'f16x3': an activation left the float16 range (|x| > 65504); the outputs "
                                     "are invalid -- use precision='bf16x3' (float32 range, same speed) or 'f32' for "
                                     "these weights")

    # ---- per-op timing (HIP events on the launch stream) ---------------------------------------
    def set_profiling(self, n_forwards=1):
        """Keep HIP-event timings of the last ``n_forwards`` forwards (0 switches profiling off)."""
        L.check(L.lib().rpn_model_set_profiling(self._h, int(n_forwards)), "rpn_model_set_profiling")

    def set_profiling_mask(self, mask=None):
        """Time only the ops with a true entry in ``mask`` (one per op; None: every op)."""
        if mask is None:
            L.check(L.lib().rpn_model_set_profiling_mask(self._h, None, 0), "rpn_model_set_profiling_mask")
            return
        buf = (ctypes.c_ubyte * len(mask))(*[1 if v else 0 for v in mask])
        L.check(L.lib().rpn_model_set_profiling_mask(self._h, buf, len(mask)), "rpn_model_set_profiling_mask")

    def set_profiling_rotate(self, on=True):
        """With a mask: time one marked op per forward, round robin (2 events per forward)."""
        L.check(L.lib().rpn_model_set_profiling_rotate(self._h, 1 if on else 0), "rpn_model_set_profiling_rotate")

    def ops(self):
        """[{name, kernel, flops_per_image, bytes_per_image, arith, launches}] in graph order.  `arith` is the arithmetic the
        op's matrix work runs in ("f32" | "bf16x3" | "f16x3"); `launches` is 0 for a max-pool that runs inside the previous
        conv's epilogue (kernel "fused:maxpool_split": no launch, no bytes of its own), else 1."""
        lib = L.lib()
        out = []
        nb, kb = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
        fl, by = ctypes.c_double(0), ctypes.c_double(0)
        for i in range(lib.rpn_model_num_ops(self._h)):
            L.check(lib.rpn_model_op_info(self._h, i, nb, 128, kb, 128, ctypes.byref(fl), ctypes.byref(by)),
                    "rpn_model_op_info")
            kernel = kb.value.decode()
            out.append({"name": nb.value.decode(), "kernel": kernel, "flops_per_image": fl.value,
                        "bytes_per_image": by.value,
                        "arith": {0: "f32", 1: "bf16x3", 2: "f16x3", 3: "f32w"}[lib.rpn_model_op_arith(self._h, i)],
                        "launches": 0 if kernel.startswith("fused:") else 1})
        return out

    def profile_ms(self):
        """Mean milliseconds per op over the kept forwards -> (list, n_forwards)."""
        n = L.lib().rpn_model_num_ops(self._h)
        buf = (ctypes.c_float * n)()
        kept = ctypes.c_int(0)
        L.check(L.lib().rpn_model_get_profile(self._h, buf, n, ctypes.byref(kept)), "rpn_model_get_profile")
        return [float(v) for v in buf], int(kept.value)

    def activation_shape(self, name):
        shape = (ctypes.c_int * 4)()
        L.check(L.lib().rpn_model_get_activation(self._h, name.encode(), None, 0, shape, None),
                "rpn_model_get_activation")
        return tuple(shape)

    def get_activation(self, name, batch=None):
        shape = self.activation_shape(name)
        out = torch.empty(shape, dtype=torch.float32, device="cuda")
        L.check(L.lib().rpn_model_get_activation(self._h, name.encode(), L.ptr(out), out.numel() * 4, None,
                                                 L.stream_ptr()), "rpn_model_get_activation")
        return out if batch is None else out[:batch]

    def __del__(self):
        try:
            if self._h:
                L.lib().rpn_model_destroy(self._h)
                self._h = L.vp(0)
        except Exception:
            pass


def synthetic_weights(backbone, hyper_params, seed=1):
    """Seeded random-init weights of the right architecture (there is no network access for
    ImageNet / trained checkpoints): He-normal kernels (std = sqrt(2 / fan_in)), small uniform
    biases, ``rpn_reg`` scaled by 0.1 so that |dh|,|dw| stay small after ``x variances``
    (SURVEY.md 8d, H6).  MobileNetV2 BatchNorm gets seeded non-trivial (gamma, beta, mean, var).
    Keys are Keras layer names.  The layer table comes from the native graph builder."""
    probe = RPNModel(backbone, hyper_params, max_batch=1)
    rng = np.random.RandomState(seed)
    weights = {}
    for layer in probe.layers:
        R, S, Cin, Cout = layer["shape"]
        fan_in = R * S * (1 if layer["kind"] == 2 else Cin)
        std = np.sqrt(2.0 / fan_in)
        kernel = (rng.standard_normal(layer["shape"]) * std).astype(np.float32)
        ch = Cin if layer["kind"] == 2 else Cout
        entry = {"kernel": kernel}
        if layer["kind"] == 0:
            entry["bias"] = rng.uniform(-0.05, 0.05, size=(ch,)).astype(np.float32)
        if layer["name"] == "rpn_reg":
            entry["kernel"] = (kernel * 0.1).astype(np.float32)
        weights[layer["name"]] = entry
        if layer["bn_name"]:
            weights[layer["bn_name"]] = {
                "gamma": rng.uniform(0.8, 1.2, size=(ch,)).astype(np.float32),
                "beta": rng.uniform(-0.1, 0.1, size=(ch,)).astype(np.float32),
                "mean": rng.uniform(-0.1, 0.1, size=(ch,)).astype(np.float32),
                "var": rng.uniform(0.8, 1.2, size=(ch,)).astype(np.float32),
            }
    del probe
    return weights
