"""Static description of the N-HANS inference path: signal constants, layer geometry and the
checkpoint variable inventory.

Follows the reference (citations relative to /root/reference):
  * constants            N_HANS___Selective_Noise/apply.py:37-38,368-369, reader.py:37
  * embedding tower      N_HANS___Selective_Noise/main.py:102-124,190-216
  * conditioned blocks   N_HANS___Selective_Noise/main.py:126-187,219-229
  * head                 N_HANS___Selective_Noise/main.py:232-242
  * separator naming     N_HANS___Source_Separation/main.py:157-163
"""
from collections import OrderedDict

FS = 16000
WIN = 400          # int(Fs * 0.025)
HOP = 160          # int(Fs * 0.010)
BINS = WIN // 2 + 1
MIX_WIN = 35       # Mix_Win
NOISE_WIN = 200    # Noise_Win
CENTER = MIX_WIN // 2
EMB = 512
BN_EPS = 1e-3
LOG_EPS = 1e-5
MIN_CTX_SAMPLES = WIN + HOP * (NOISE_WIN - 1)   # 32240: shortest context giving 200 frames

DENOISER = "denoiser"
SEPARATOR = "separator"

# (name, (kh, kw), (sh, sw), cout) -- embedding tower, main.py:194-198
TOWER_BLOCKS = [
    ("noise_resblock1_1", (8, 4), (3, 2), 64),
    ("noise_resblock2_1", (8, 4), (3, 2), 128),
    ("noise_resblock3_1", (4, 4), (1, 1), 256),
    ("noise_resblock4_1", (4, 4), (1, 2), 512),
]
# (name, k, s, cout) -- conditioned residual stack, main.py:221-229
MAIN_BLOCKS = [
    ("resblock1_1", 4, 1, 64),
    ("resblock1_2", 4, 1, 64),
    ("resblock2_1", 4, 2, 128),
    ("resblock2_2", 4, 1, 128),
    ("resblock3_1", 3, 2, 256),
    ("resblock3_2", 3, 1, 256),
    ("resblock4_1", 3, 2, 512),
    ("resblock4_2", 3, 1, 512),
]


def emb_scopes(kind):
    """Scope suffixes of the two conditioning projections, in resnet_block argument order
    (first, second).  Denoiser: (pos, neg) main.py:142,146.  Separator: (noise, clean)
    SS/main.py:157,161 -- `noise` is fed from --neg, `clean` from --pos."""
    if kind == DENOISER:
        return ("_noise_pos_emb", "_noise_neg_emb")
    if kind == SEPARATOR:
        return ("_noise_emb", "_clean_emb")
    raise ValueError("unknown model kind %r" % (kind,))


def same_out(n, s):
    return -(-n // s)


def same_pad(n, k, s):
    """TF 'SAME' padding split (before, after) -- asymmetric for even kernels."""
    out = same_out(n, s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def tower_geometry():
    """Per tower block: dict(name, kh, kw, sh, sw, cin, cout, hin, win, hout, wout)."""
    h, w, c = NOISE_WIN, BINS, 1
    out = []
    for name, (kh, kw), (sh, sw), cout in TOWER_BLOCKS:
        ho, wo = same_out(h, sh), same_out(w, sw)
        out.append(dict(name=name, kh=kh, kw=kw, sh=sh, sw=sw, cin=c, cout=cout,
                        hin=h, win=w, hout=ho, wout=wo))
        h, w, c = ho, wo, cout
    return out


def main_geometry():
    h, w, c = MIX_WIN, BINS, 1
    out = []
    for name, k, s, cout in MAIN_BLOCKS:
        ho, wo = same_out(h, s), same_out(w, s)
        out.append(dict(name=name, kh=k, kw=k, sh=s, sw=s, cin=c, cout=cout,
                        hin=h, win=w, hout=ho, wout=wo))
        h, w, c = ho, wo, cout
    return out


def head_geometry():
    g = main_geometry()[-1]
    return dict(hin=g["hout"], win=g["wout"], cin=g["cout"], cout=512,
                flat=g["wout"] * 512, nout=BINS)


def variable_shapes(kind):
    """Ordered {checkpoint variable name: shape} of every float tensor the inference graph
    reads (verified against the shipped .index files in tests)."""
    v = OrderedDict()

    def bn(scope, c, rank4=True):
        shp = (1, 1, 1, c) if rank4 else (1, c)
        for n in ("beta", "gamma", "pop_mean", "pop_variance"):
            v["%s/%s" % (scope, n)] = shp

    for g in tower_geometry():
        p = "embedding/" + g["name"]
        v[p + "_conv1/w"] = (g["kh"], g["kw"], g["cin"], g["cout"])
        bn(p + "_conv1", g["cout"])
        v[p + "_conv2/w"] = (g["kh"], g["kw"], g["cout"], g["cout"])
        v[p + "_conv2/b"] = (1, 1, 1, g["cout"])
        v[p + "_transform/w"] = (1, 1, g["cin"], g["cout"])
        v[p + "_transform/b"] = (1, 1, 1, g["cout"])
        bn(p + "_addition", g["cout"])

    ea, eb = emb_scopes(kind)
    for g in main_geometry():
        p, c = g["name"], g["cout"]
        v[p + "_conv1/w"] = (g["kh"], g["kw"], g["cin"], c)
        bn(p + "_conv1", c)
        v[p + "_conv2/w"] = (g["kh"], g["kw"], c, c)
        v[p + "_conv2/b"] = (1, 1, 1, c)
        if g["cin"] != c:
            v[p + "_transform/w"] = (1, 1, g["cin"], c)
            v[p + "_transform/b"] = (1, 1, 1, c)
        bn(p + "_addition", c)
        for i in (1, 2):
            q = "%s_conv%d" % (p, i)
            for e in (ea, eb):
                v[q + e + "/w"] = (EMB, c)
                v[q + e + "/b"] = (1, c)
            for tf_ in ("_temb", "_femb"):
                s = q + tf_
                v[s + "_dense1/w"] = (1, 50)
                v[s + "_dense2/w"] = (50, 50)
                v[s + "_dense3/w"] = (50, c)
                bn(s + s + "_dense1", 50, rank4=False)   # doubled scope, main.py:131
                bn(s + s + "_dense2", 50, rank4=False)   # main.py:134
    v["last_conv/w"] = (5, 1, 512, 512)
    bn("last_conv", 512)
    v["last_dense/w"] = (head_geometry()["flat"], BINS)
    v["last_dense/b"] = (1, BINS)
    return v


def frames_for_samples(n):
    """Trim rule + frame count (apply.py:158-161, tf.signal.stft pad_end=False)."""
    if n < WIN:
        return 0, 0
    kept = n - ((n - WIN) % HOP)
    return kept, 1 + (kept - WIN) // HOP
