"""Host-side constant folding of an N-HANS checkpoint into the blob libnhans_hip.so consumes.

Everything that depends only on the weights is evaluated once here, in float64, and rounded to
float32 at the end:
  * inference BatchNorm (SN/blocks.py:104-108) -> per-channel scale/shift; the scale is multiplied
    into the convolution / projection weights and the position tables, the shift into the bias;
  * the time/frequency position MLPs `cont_embed` (SN/main.py:127-137) -> tables t[Ho,C], f[Wo,C];
  * conv bias, `_transform` bias and the two projection biases -> one additive row per conv;
  * the 16 conditioning projections (SN/main.py:139-148) -> one [1024, 3840] matrix;
  * HWIO kernels -> the MFMA fragment order of conv_igemm.hip (`pack_igemm`).

Blob layout (little endian): header {char magic[8]="NHANSFW1"; u32 version=BLOB_VERSION; u32 n_entries;
u64 total_bytes}, then n_entries x {char name[48]; u64 offset; u64 nfloats}, then float32 arrays
at 256-byte-aligned offsets.
"""
import struct
from fractions import Fraction

import numpy as np

from . import spec

F64 = np.float64
# Version of the packing conventions below; nhans_create refuses any other (a blob folded by an older tree would load
# and compute wrong results: version 1 walked K as (filter row, chunk, column) and, later, (chunk, row, column) under
# the same number; 2 = (chunk, row, column) for the direct kernels and the 8-channel / two-row k-steps of pack_wino).
BLOB_VERSION = 2


def _bn(W, scope):
    g = W[scope + "/gamma"].astype(F64).reshape(-1)
    b = W[scope + "/beta"].astype(F64).reshape(-1)
    m = W[scope + "/pop_mean"].astype(F64).reshape(-1)
    v = W[scope + "/pop_variance"].astype(F64).reshape(-1)
    scale = g / np.sqrt(v + spec.BN_EPS)
    return scale, b - m * scale


def _cont_embed(W, n, scope):
    """Position table of `n` rows (SN/main.py:127-137); BN scopes are doubled (`scope+scope`)."""
    z = np.arange(n, dtype=F64).reshape(n, 1)
    for i in (1, 2):
        s, sh = _bn(W, "%s%s_dense%d" % (scope, scope, i))
        z = np.maximum((z @ W["%s_dense%d/w" % (scope, i)].astype(F64)) * s + sh, 0.0)
    return z @ W[scope + "_dense3/w"].astype(F64)


def kmat(w4):
    """HWIO conv weight [KH, KW, C, N] -> [K, N] in the K order the conv kernels walk: 32-channel chunk, filter
    row, filter column, channel within the chunk.  The KW taps of one (chunk, row) are adjacent so that the halo
    kernel (conv_igemm_halo.hip) streams its weights strictly sequentially while one staged activation image serves
    all KW of them, and the KH images of a chunk -- the same pixels shifted by one image row -- are staged back to
    back, so that all but the first come out of the L2.  (C = 1 convs are not GEMMs: plain order.)"""
    kh, kw, c, n = w4.shape
    if c % 32:
        return w4.reshape(-1, n)
    return w4.reshape(kh, kw, c // 32, 32, n).transpose(2, 0, 1, 3, 4).reshape(-1, n)


def pack_igemm(wkn, npad=None):
    """[K, N] (K in kmat() order, multiple of 32) -> float32 [K/32][Npad/32][4][64][4]
    so that lane l of MFMA (q, e) of chunk c, n-tile t reads W[32c + 8q + 4(l>>5) + e][32t + (l&31)]
    as element e of one 16-byte vector (conv_igemm.hip)."""
    k, n = wkn.shape
    assert k % 32 == 0, k
    npad = n if npad is None else npad
    assert npad % 32 == 0 and npad >= n
    wp = np.zeros((k, npad), dtype=F64)
    wp[:, :n] = wkn
    a = wp.reshape(k // 32, 4, 2, 4, npad // 32, 32)       # chunk, q, half, e, nt, col
    a = a.transpose(0, 4, 1, 2, 5, 3)                       # chunk, nt, q, half, col, e
    return np.ascontiguousarray(a).reshape(-1).astype(np.float32)


def col_scale(*mats):
    """Power-of-two per output column that brings max|w| over all given [K_i, N] matrices into
    [32, 64): the f16 `lo` parts of the scaled weights are then normal numbers (no subnormal
    precision loss); the kernel epilogue multiplies the accumulator by 1/scale (exact)."""
    cmax = np.max(np.stack([np.abs(m).max(axis=0) for m in mats]), axis=0)
    e = np.floor(np.log2(np.where(cmax > 0, cmax, 1.0)))
    return np.where(cmax > 0, 2.0 ** (5 - e), 1.0)


def pack_igemm_h3(wkn, scale, npad=None):
    """Split-f16 packing for conv_igemm PREC 1: w*scale = hi + lo (both f16), laid out
    [K/32][Npad/32][s=2][h=2][64 lanes][8 halfs]; lane l, element e of k-step s holds
    k = 32c + 16s + 8(l>>5) + e, column 32t + (l&31).  Returned as a float32 view of the half bits."""
    k, n = wkn.shape
    assert k % 32 == 0, k
    npad = n if npad is None else npad
    assert npad % 32 == 0 and npad >= n
    wp = np.zeros((k, npad), dtype=F64)
    wp[:, :n] = wkn * scale[None, :n]
    hi = wp.astype(np.float16)
    lo = (wp - hi.astype(F64)).astype(np.float16)
    assert np.isfinite(hi).all()
    parts = []
    for a in (hi, lo):
        a = a.reshape(k // 32, 2, 2, 8, npad // 32, 32)     # chunk, s, g8, e, nt, col
        parts.append(a.transpose(0, 4, 1, 2, 5, 3))          # chunk, nt, s, g8, col, e
    p = np.stack(parts, axis=3)                               # chunk, nt, s, h, g8, col, e
    return np.ascontiguousarray(p).reshape(-1).view(np.float32)


# ------------------------------------------------------------------------------ 1-D Winograd along W
# conv_wino.hip runs the stride-1 4 x 4 convs of the stack as F(m, k) along the image width, plain
# accumulation over the filter rows and the channels: per tile of m output columns the 8 transformed
# positions V_p = sum_x BT[p][x] d[x] of its 8 input columns are multiplied with the transformed
# filter U_p[kh] = sum_kw G[p][kw] w[kh][kw] -- 8*kh instead of m*k*kh products per channel pair
# (m = 5, k = 4: 2.5 x fewer; m = 6, k = 3: 2.25 x fewer) -- and Y = AT M gives the m outputs.
# Interpolation points 0, +-1, +-2, +-1/2 and infinity (8 positions for either filter width, so that
# each of the 8 MFMA waves of a workgroup owns one position); tests/winograd_probe.py is the accuracy gate.
WINO_POINTS = (0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2))
WINO_N = 8


def _solve_exact(rows, rhs):
    """Gauss-Jordan over Fractions; the (consistent, full column rank) system rows . x = rhs."""
    n = len(rows[0])
    a = [list(map(Fraction, r)) + [Fraction(b)] for r, b in zip(rows, rhs)]
    piv = 0
    for col in range(n):
        k = next((i for i in range(piv, len(a)) if a[i][col] != 0), None)
        assert k is not None
        a[piv], a[k] = a[k], a[piv]
        a[piv] = [v / a[piv][col] for v in a[piv]]
        for i in range(len(a)):
            if i != piv and a[i][col] != 0:
                f = a[i][col]
                a[i] = [vi - f * vp for vi, vp in zip(a[i], a[piv])]
        piv += 1
    assert all(all(v == 0 for v in r) for r in a[piv:])
    return [a[i][n] for i in range(n)]


def wino_matrices(m, r):
    """Cook-Toom matrices of F(m, r) on WINO_POINTS + infinity (m + r - 1 == 8), exact rationals as
    float64: (AT [m, 8], G [8, r], BT [8, 8]) with y = AT ((G g) * (BT d)) for the m-output,
    r-tap correlation y[i] = sum_k d[i + k] g[k]."""
    n = m + r - 1
    assert n == WINO_N
    pts = [Fraction(p) for p in WINO_POINTS]
    AT = [[(pts[j] ** i if j < n - 1 else Fraction(1 if i == m - 1 else 0)) for j in range(n)] for i in range(m)]
    norm = []
    for j in range(n - 1):
        v = Fraction(1)
        for k in range(n - 1):
            if k != j:
                v *= pts[j] - pts[k]
        norm.append(v)
    G = [[(pts[j] ** i / norm[j] if j < n - 1 else Fraction(1 if i == r - 1 else 0)) for i in range(r)] for j in range(n)]
    BT = [[Fraction(0)] * n for _ in range(n)]
    for col in range(n):
        rows, rhs = [], []
        for i in range(m):
            for k in range(r):
                rows.append([AT[i][j] * G[j][k] for j in range(n)])
                rhs.append(1 if col == i + k else 0)
        sol = _solve_exact(rows, rhs)
        for j in range(n):
            BT[j][col] = sol[j]
    f = lambda M: np.array([[float(v) for v in row] for row in M], dtype=F64)
    return f(AT), f(G), f(BT)


# the input transform is hard-coded in conv_wino.hip (structured: 24 fused multiply-adds instead of 64)
WINO_BT = np.array([[-1, 0, 5.25, 0, -5.25, 0, 1, 0],
                    [0, 1, 1, -4.25, -4.25, 1, 1, 0],
                    [0, -1, 1, 4.25, -4.25, -1, 1, 0],
                    [0, 0.5, 0.25, -2.5, -1.25, 2, 1, 0],
                    [0, -0.5, 0.25, 2.5, -1.25, -2, 1, 0],
                    [0, 2, 4, -2.5, -5, 0.5, 1, 0],
                    [0, -2, 4, 2.5, -5, -0.5, 1, 0],
                    [0, -1, 0, 5.25, 0, -5.25, 0, 1]], dtype=F64)


def wino_outputs(kw):
    """outputs per tile for a kw-tap filter row (8 positions): 5 for 4 taps, 6 for 3."""
    return WINO_N + 1 - kw


def wino_eligible(kh, kw, sh, sw, cin, cout):
    """What conv_wino.hip runs: stride-1 4x4 convs (the F(6,3) form of the 3x3 convs is not built -- their 9x51 / 5x26
    images fill 63 % of a tile-pixel block -- so their 94 MB of packs are not emitted either)."""
    return (sh, sw) == (1, 1) and (kh, kw) == (4, 4) and cin % 16 == 0 and cout % 64 == 0


def pack_wino(w4):
    """HWIO weights [KH, KW, C, N] (BatchNorm scale folded in) -> (packed U as a float32 view of f16 bits,
    per-channel unscale vector).  U_p[kh][c][n] = sum_kw G[p][kw] w[kh][kw][c][n], scaled per output channel by a
    power of two into [32, 64), split hi/lo.  conv_wino.hip's MFMA k-step is 8 channels x TWO filter rows:
    [N/64][p 8][C/8][s KH/2][nt 2][h 2][lane 64][e 8] -- lane l, element e of k-step s of 8-channel chunk c8 of
    (64-channel block nb, position p) holds filter row 2 s + (l >> 5), channel 8 c8 + e, column 64 nb + 32 nt +
    (l & 31); the wave that owns position p streams its fragments strictly sequentially."""
    kh, kw, c, n = w4.shape
    assert c % 16 == 0 and n % 64 == 0 and kh % 2 == 0
    _, G, BT = wino_matrices(wino_outputs(kw), kw)
    assert np.array_equal(BT, WINO_BT)
    U = np.einsum("pk,hkcn->phcn", G, w4.astype(F64))              # [8, KH, C, N]
    sc = col_scale(np.abs(U).reshape(-1, n))
    Us = U * sc
    hi = Us.astype(np.float16)
    lo = (Us - hi.astype(F64)).astype(np.float16)
    parts = []
    for a in (hi, lo):
        a = a.reshape(WINO_N, kh // 2, 2, c // 8, 8, n // 64, 2, 32)  # p, s, g8, c8, e, nb, nt, col
        parts.append(a.transpose(5, 0, 3, 1, 6, 2, 7, 4))            # nb, p, c8, s, nt, g8, col, e
    pk = np.stack(parts, axis=5)                                     # nb, p, c8, s, nt, h, g8, col, e
    assert np.isfinite(hi).all()
    return np.ascontiguousarray(pk).reshape(-1).view(np.float32), 1.0 / sc


def _pad(v, npad):
    o = np.ones(npad, dtype=F64)
    o[:len(v)] = v
    return o


def fold_arrays(W, kind, split_f16=True):
    """-> dict name -> float32 1-D array (see nhans_api.hip for the consumer).  With split_f16 the
    blob also carries every packed conv weight in split-f16 form (`*_h`) and the per-channel
    unscale vectors (`<conv>.ws`) for the f16x3 mode."""
    out = {}
    w64 = lambda n: W[n].astype(F64)

    def emit(conv, mats, npad=None):
        """mats: [(array name, [K,N] matrix)] sharing one accumulator -> f32 pack (+ split pack)."""
        for name, m in mats:
            out[name] = pack_igemm(m, npad)
        if split_f16:
            sc = col_scale(*[m for _, m in mats])
            for name, m in mats:
                out[name + "_h"] = pack_igemm_h3(m, sc, npad)
            out[conv + ".ws"] = _pad(1.0 / sc, npad or len(sc))

    # --- transform constants (tf.signal.stft / inverse_stft_window_fn)
    j = np.arange(spec.WIN, dtype=F64)
    ang = -2.0 * np.pi * j / spec.WIN
    out["tw400"] = np.stack([np.cos(ang), np.sin(ang)], 1).reshape(-1)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * j / spec.WIN)
    out["window"] = win
    den = np.zeros(spec.HOP, dtype=F64)
    for q in range(-(-spec.WIN // spec.HOP)):
        seg = win[q * spec.HOP:(q + 1) * spec.HOP] ** 2
        den[:len(seg)] += seg
    out["wsyn"] = win / np.tile(den, -(-spec.WIN // spec.HOP))[:spec.WIN]
    out["zero"] = np.zeros(16384)      # zero page: padded taps read their channels from here

    # --- embedding tower (SN/main.py:102-124,190-216)
    for i, g in enumerate(spec.tower_geometry()):
        p, s = "t%d" % i, "embedding/" + g["name"]
        s1, h1 = _bn(W, s + "_conv1")
        w1 = kmat(w64(s + "_conv1/w")) * s1
        if g["cin"] == 1:
            out[p + ".c1.w"] = w1.reshape(-1)
        else:
            emit(p + ".c1", [(p + ".c1.wpk", w1)])
        out[p + ".c1.cb"] = h1
        sa, ha = _bn(W, s + "_addition")
        w2 = kmat(w64(s + "_conv2/w")) * sa
        wt = w64(s + "_transform/w").reshape(g["cin"], g["cout"]) * sa
        if g["cin"] == 1:
            out[p + ".c2.idw"] = wt.reshape(-1)
            emit(p + ".c2", [(p + ".c2.wpk", w2)])
        else:
            emit(p + ".c2", [(p + ".c2.wpk", w2), (p + ".c2.wpk_t", wt)])
        out[p + ".c2.cb"] = sa * (w64(s + "_conv2/b").reshape(-1) + w64(s + "_transform/b").reshape(-1)) + ha

    # --- conditioned stack (SN/main.py:126-187,219-229)
    ea, eb = spec.emb_scopes(kind)
    cond_w, cond_b = [], []
    for i, g in enumerate(spec.main_geometry()):
        p, s, c = "m%d" % i, g["name"], g["cout"]
        s1, h1 = _bn(W, s + "_conv1")
        sa, ha = _bn(W, s + "_addition")
        w1 = kmat(w64(s + "_conv1/w")) * s1
        if g["cin"] == 1:
            out[p + ".c1.w"] = w1.reshape(-1)
        else:
            emit(p + ".c1", [(p + ".c1.wpk", w1)])
        w2 = kmat(w64(s + "_conv2/w")) * sa
        extra_bias = w64(s + "_conv2/b").reshape(-1)
        if g["cin"] == 1:
            out[p + ".c2.idw"] = w64(s + "_transform/w").reshape(-1) * sa
            extra_bias = extra_bias + w64(s + "_transform/b").reshape(-1)
            emit(p + ".c2", [(p + ".c2.wpk", w2)])
        elif g["cin"] != c:
            emit(p + ".c2", [(p + ".c2.wpk", w2),
                             (p + ".c2.wpk_t", w64(s + "_transform/w").reshape(g["cin"], c) * sa)])
            out[p + ".c2.idw"] = np.zeros(c)          # unused: the transform rides in the K loop
            extra_bias = extra_bias + w64(s + "_transform/b").reshape(-1)
        else:
            out[p + ".c2.idw"] = sa
            emit(p + ".c2", [(p + ".c2.wpk", w2)])
        if split_f16:
            # 1-D Winograd form of the stride-1 convs without a `_transform` segment (conv_wino.hip)
            for cv, scl, cin_, st in ((1, s1, g["cin"], (g["sh"], g["sw"])), (2, sa, c, (1, 1))):
                w4 = w64("%s_conv%d/w" % (s, cv)) * scl
                if cin_ == c and wino_eligible(g["kh"], g["kw"], st[0], st[1], cin_, c):
                    out["%s.c%d.wino" % (p, cv)], out["%s.c%d.wino.ws" % (p, cv)] = pack_wino(w4)
        for cv, sc, sh, bias in ((1, s1, h1, 0.0), (2, sa, ha, extra_bias)):
            q = "%s_conv%d" % (s, cv)
            # time + frequency position terms in one [Ho*Wo, C] table (one coalesced read per output)
            tt = _cont_embed(W, g["hout"], q + "_temb") * sc
            ff = _cont_embed(W, g["wout"], q + "_femb") * sc
            out["%s.c%d.tf" % (p, cv)] = (tt[:, None, :] + ff[None, :, :]).reshape(-1)
            # the two terms on their own (conv_wino.hip, direct_conv64: the big layers, whose [Ho*Wo, C] table -- 1.8 MB
            # for 35 x 201 x 64 -- does not survive in a 4 MB L2 beside the streaming tensors and was re-fetched from
            # HBM almost once per frame: 3.8 GB per launch)
            out["%s.c%d.tt" % (p, cv)] = tt.reshape(-1)
            out["%s.c%d.ff" % (p, cv)] = ff.reshape(-1)
            cond_w.append(np.concatenate([w64(q + ea + "/w"), w64(q + eb + "/w")], 0) * sc)
            cond_b.append(sc * (w64(q + ea + "/b").reshape(-1) + w64(q + eb + "/b").reshape(-1) + bias) + sh)
    out["cond.w"] = np.concatenate(cond_w, 1).reshape(-1)       # [1024, 3840]
    out["cond.base"] = np.concatenate(cond_b, 0)

    # --- head (SN/main.py:232-242)
    s, h = _bn(W, "last_conv")
    emit("head.conv", [("head.conv.wpk", kmat(w64("last_conv/w")) * s)])
    out["head.conv.cb"] = h
    emit("head.dense", [("head.dense.wpk", w64("last_dense/w"))], 256)
    cb = np.zeros(256)
    cb[:spec.BINS] = w64("last_dense/b").reshape(-1)
    out["head.dense.cb"] = cb
    out["head.dense.idw"] = np.ones(256)
    return {k: np.ascontiguousarray(v, dtype=np.float32).reshape(-1) for k, v in out.items()}


def write_blob(arrays):
    names = sorted(arrays)
    head = 24 + 64 * len(names)
    off = (head + 255) & ~255
    entries, chunks = [], []
    for n in names:
        a = arrays[n]
        nb = n.encode("ascii")
        assert len(nb) < 48, n
        entries.append(struct.pack("<48sQQ", nb, off, a.size))
        chunks.append((off, a))
        off = (off + a.nbytes + 255) & ~255
    blob = bytearray(off)
    blob[0:24] = struct.pack("<8sIIQ", b"NHANSFW1", BLOB_VERSION, len(names), off)
    for i, e in enumerate(entries):
        blob[24 + 64 * i:24 + 64 * (i + 1)] = e
    for o, a in chunks:
        blob[o:o + a.nbytes] = a.tobytes()
    return bytes(blob)


def fold_weights(W, kind):
    """Checkpoint dict -> blob bytes for nhans_create()."""
    return write_blob(fold_arrays(W, kind))


def blob_header_ok(blob):
    """Does this byte buffer (bytes / uint8 array) start with a header of THIS packing version that describes its own
    length?  (blobcache.py: a cached blob of another version, or a truncated file, is rebuilt.)"""
    if len(blob) < 24:
        return False
    magic, version, _, total = struct.unpack("<8sIIQ", bytes(blob[:24]))
    return magic == b"NHANSFW1" and version == BLOB_VERSION and total == len(blob)
