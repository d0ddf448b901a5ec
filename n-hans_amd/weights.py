"""Checkpoint weights for the N-HANS inference path: seeded synthetic generator and the
loader for real TensorFlow bundles.  (Folding/packing for the HIP kernels is in fold.py.)

The reference's own initialisers (trunc-normal sigma=0.01, and sigma=0.0 for `*_dense3`, `*_emb`,
`last_dense`: SN/main.py:136,142,146,238) give a degenerate out==0 network, and its trained
weights are git-LFS pointers in the tree, so measured/parity runs use the seeded recipe below
unless a real bundle is supplied.
"""
import zlib
from collections import OrderedDict

import numpy as np

from . import spec, tfbundle


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("utf-8"))])


def synthetic_weights(kind=spec.DENOISER, seed=7):
    """Seeded float32 weights with the exact names/shapes of the checkpoint inventory.

    Recipe (variance-preserving so activations stay O(1)-O(10) through the stack):
      conv / dense `w`      N(0, 2/fan_in)   (fan_in = kh*kw*cin or in_dim)
      1x1 `_transform/w`    N(0, 1/cin)
      convs reading the 1-channel log-magnitude image: std x 0.25
      conv / dense `b`      N(0, 0.05^2)
      BN gamma U(0.5,1.5), beta N(0,0.1^2), pop_mean N(0,0.1^2), pop_variance U(0.5,1.5);
         `_addition` BNs use pop_variance U(2.0,3.5) (sum of two paths)
      `*_emb/w`             N(0, 1/512),  `*_emb/b` N(0, 0.05^2)
      position MLPs         dense1 N(0, 0.02^2) (inputs are 0..200), dense2 N(0, 2/50),
                            dense3 N(0, 1/50)
      last_dense/w          N(0, 1/13312)
    Each tensor draws from its own stream keyed by (seed, crc32(name)), so same-named tensors of
    the denoiser and separator are identical and the result does not depend on iteration order.
    """
    out = OrderedDict()
    for name, shape in spec.variable_shapes(kind).items():
        r = _rng(seed, name)
        scope, leaf = name.rsplit("/", 1)
        if leaf == "w":
            if len(shape) == 4:
                fan_in = shape[0] * shape[1] * shape[2]
                std = np.sqrt((1.0 if scope.endswith("_transform") else 2.0) / fan_in)
                if shape[2] == 1:
                    std *= 0.25           # single-channel log-magnitude input is not unit-scale
            elif scope.endswith("_emb"):
                std = np.sqrt(1.0 / spec.EMB)
            elif scope.endswith("_dense1"):
                std = 0.02
            elif scope.endswith("_dense2"):
                std = np.sqrt(2.0 / 50)
            elif scope.endswith("_dense3"):
                std = np.sqrt(1.0 / 50)
            elif scope == "last_dense":
                std = np.sqrt(1.0 / shape[0])
            else:
                std = np.sqrt(2.0 / shape[0])
            a = r.normal(0.0, std, size=shape)
        elif leaf == "b":
            a = r.normal(0.0, 0.05, size=shape)
        elif leaf == "gamma":
            a = r.uniform(0.5, 1.5, size=shape)
        elif leaf in ("beta", "pop_mean"):
            a = r.normal(0.0, 0.1, size=shape)
        elif leaf == "pop_variance":
            a = (r.uniform(2.0, 3.5, size=shape) if scope.endswith("_addition")
                 else r.uniform(0.5, 1.5, size=shape))
        else:
            raise KeyError(name)
        out[name] = a.astype(np.float32)
    return out


def load_checkpoint(prefix, kind, verify_crc=True):
    """Load a real TF bundle (user-supplied LFS blob) and check it against the inventory and the
    per-tensor CRCs of its index."""
    raw = tfbundle.load_checkpoint(prefix, verify_crc)
    out = OrderedDict()
    for name, shape in spec.variable_shapes(kind).items():
        if name not in raw:
            raise KeyError("checkpoint %s lacks tensor %s" % (prefix, name))
        a = raw[name]
        if tuple(a.shape) != tuple(shape) or a.dtype != np.float32:
            raise ValueError("tensor %s: got %s %s, want float32 %s"
                             % (name, a.dtype, a.shape, shape))
        out[name] = a
    return out
