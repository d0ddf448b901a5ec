"""Weight sets for the parity tests beyond `weights.synthetic_weights(kind, 7)` (tests/test_gpu_recipes.py).

The trained checkpoint of the reference is a git-LFS pointer in the tree, so every golden vector and fuzz batch uses
seeded synthetic weights; what the split-f16 arithmetic of the HIP path depends on -- the hi/lo split of weights whose
columns were scaled by a power of two, the Winograd re-split of transformed activations, the activation exponents --
depends on the weight DISTRIBUTION.  These recipes span what a trained or a freshly initialised network looks like:

  seed        the standard recipe with another seed
  heavy       Student-t (3 degrees of freedom) conv / dense weights whose input channels carry a log-uniform scale
              over three decades -- 10^3 of dynamic range inside every weight column --, per-tensor RMS as in the
              standard recipe so that activations stay O(1)
  trained_bn  BatchNorm statistics of a trained net: pop_variance log-uniform 1e-6 .. 10 (BN scale gamma/sqrt(var+1e-3)
              up to 31 gamma, SN/blocks.py:104-108), the convolution in front of each BN scaled per channel by
              sqrt(pop_variance) and pop_mean likewise (in a trained net the statistics ARE those of the activations),
              gamma with 10 % exact zeros (dead channels), beta uniform in [-3, 3]
  tf_init     the reference's own initialisers (SN/main.py:136,142,146,238; SN/blocks.py): truncated normal sigma =
              0.01 for conv / dense weights, sigma = 0 for `*_dense3`, `*_emb` and `last_dense`, zero biases, BatchNorm
              at its TF defaults (gamma 1, beta 0, pop_mean 0, pop_variance 1): `out` is exactly 0 and `denoised` is
              exactly the centre frame of the window.
"""
import zlib

import numpy as np

import nhans_amd  # noqa: F401
from nhans_amd import spec, weights


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("utf-8")), 77])


def seed(kind, s):
    return weights.synthetic_weights(kind, s)


def heavy(kind, s=31):
    W = dict(weights.synthetic_weights(kind, s))
    for name, a in W.items():
        scope, leaf = name.rsplit("/", 1)
        if leaf != "w" or a.ndim not in (2, 4) or a.shape[-2] < 16:
            continue                                     # (the one-channel convs and the position MLPs keep their recipe)
        r = _rng(s, name)
        rms = float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))
        t = r.standard_t(3, size=a.shape)
        cin_scale = 10.0 ** r.uniform(-1.5, 1.5, size=a.shape[-2])
        t = t * cin_scale.reshape((1,) * (a.ndim - 2) + (-1, 1))
        t *= rms / np.sqrt(np.mean(t ** 2))
        W[name] = t.astype(np.float32)
    return W


def trained_bn(kind, s=41):
    W = dict(weights.synthetic_weights(kind, s))
    for name in list(W):
        scope, leaf = name.rsplit("/", 1)
        if leaf != "pop_variance":
            continue
        r = _rng(s, name)
        n = W[name].size
        shape = W[name].shape
        old_var = W[name].astype(np.float64).reshape(-1)
        addition = scope.endswith("_addition")
        # the `_addition` BatchNorm sees conv2 + shortcut: its statistics follow two producers, keep them within a decade
        var = 10.0 ** (r.uniform(-1.0, 1.0, n) if addition else r.uniform(-6.0, 1.0, n))
        gain = np.sqrt((var + spec.BN_EPS) / (old_var + spec.BN_EPS))          # BN scale changes by 1 / gain ...
        W[name] = var.reshape(shape).astype(np.float32)
        W[scope + "/pop_mean"] = (W[scope + "/pop_mean"].astype(np.float64).reshape(-1) * gain).reshape(shape).astype(np.float32)
        gamma = W[scope + "/gamma"].astype(np.float64).reshape(-1).copy()
        gamma[r.random(n) < 0.10] = 0.0
        W[scope + "/gamma"] = gamma.reshape(shape).astype(np.float32)
        W[scope + "/beta"] = r.uniform(-3.0, 3.0, n).reshape(shape).astype(np.float32)
        # ... and what feeds the BN is scaled by gain per channel (conv1 for `_conv1`; conv2, its bias, the shortcut and
        # the projections for `_addition`; the dense layer for the position MLPs and `last_conv`)
        ea, eb = spec.emb_scopes(kind)
        q = (scope[:-len("_addition")] + "_conv2") if addition else scope
        # everything that is summed in front of this BN: the conv (and its bias), for the conditioned blocks the two
        # projections and the last layer of the two position MLPs, for `_addition` the `_transform` shortcut as well
        feeders = [q + "/w", q + "/b", q + ea + "/w", q + ea + "/b", q + eb + "/w", q + eb + "/b",
                   q + "_temb_dense3/w", q + "_femb_dense3/w"]
        if addition:
            feeders += [scope[:-len("_addition")] + "_transform/w", scope[:-len("_addition")] + "_transform/b"]
        if scope[-7:-1] == "_dense":                                          # doubled BN scopes of the position MLPs: S + S + "_denseN"
            feeders = [scope[(len(scope) - 7) // 2:] + "/w"]
        for f in feeders:
            if f in W:
                W[f] = (W[f].astype(np.float64) * gain.reshape((1,) * (W[f].ndim - 1) + (-1,))).astype(np.float32)
    return W


def tf_init(kind, s=51):
    W = {}
    for name, shape in spec.variable_shapes(kind).items():
        scope, leaf = name.rsplit("/", 1)
        r = _rng(s, name)
        if leaf == "w":
            if scope.endswith("_dense3") or scope.endswith("_emb") or scope == "last_dense":
                a = np.zeros(shape)
            else:
                a = r.normal(0.0, 0.01, size=shape)
                bad = np.abs(a) > 0.02                                        # truncated normal: redraw beyond 2 sigma
                while bad.any():
                    a[bad] = r.normal(0.0, 0.01, size=int(bad.sum()))
                    bad = np.abs(a) > 0.02
        elif leaf in ("b", "beta", "pop_mean"):
            a = np.zeros(shape)
        elif leaf in ("gamma", "pop_variance"):
            a = np.ones(shape)
        else:
            raise KeyError(name)
        W[name] = a.astype(np.float32)
    return W


RECIPES = {
    "seed11": lambda kind: seed(kind, 11),
    "seed23": lambda kind: seed(kind, 23),
    "heavy": heavy,
    "trained_bn": trained_bn,
    "tf_init": tf_init,
}
