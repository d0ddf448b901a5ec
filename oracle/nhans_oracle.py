"""CPU ORACLE (test infrastructure, NOT product code) -- float64 numpy restatement of the
N-HANS per-frame inference path.

PARITY UNPINNED: the reference's arithmetic lives in TensorFlow (`tensorflow_gpu>=1.14.0`,
setup.py:25 -- an unpinned floor, not vendored), which is absent from this image; the trained
weights are git-LFS pointers; and the reference ships no tests or golden vectors.  This file
restates the reference's call sites op for op using the published semantics of the TF ops
(tf.signal.stft / inverse_stft / inverse_stft_window_fn, tf.nn.conv2d NHWC/HWIO with SAME/VALID
padding, tf.nn.batch_normalization, tf.matmul, tf.nn.avg_pool2d).  What IS pinned: the
trim/frame/OLA geometry against the shipped input/output wav pairs (tests/golden/geometry.json),
the variable inventory against the shipped .index files, and this restatement against an
independent second implementation built from torch CPU ops (oracle/torch_ref.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Citations are relative to /root/reference; SN = N_HANS___Selective_Noise, SS = N_HANS___Source_Separation.
"""
import numpy as np
from scipy.io import wavfile

FS = 16000
WIN = 400        # int(FLAGS.Fs * 0.025)   SN/apply.py:368
HOP = 160        # int(FLAGS.Fs * 0.010)   SN/apply.py:369
BINS = 201
MIX_WIN = 35     # SN/apply.py:38
NOISE_WIN = 200  # SN/apply.py:37
BN_EPS = 0.001   # SN/blocks.py:108
F64 = np.float64


# ----------------------------------------------------------------------------- wav front end
def read_wav(path):
    """SN/apply.py:46-53 -- 16 kHz int16 only; stereo -> float64 mean over channels."""
    rate, samples = wavfile.read(path)
    assert rate == FS
    assert samples.dtype == np.int16
    if samples.ndim > 1:
        samples = samples.mean(axis=1)
    assert samples.ndim == 1
    return samples


def normalise(samples):
    """SN/apply.py:150-155 -- x / (max(abs(x)) + 1e-6) in float64, then float32.
    abs() of an int16 array keeps int16 (abs(-32768) wraps), exactly as in the reference."""
    with np.errstate(over="ignore"):
        peak = np.max(np.abs(samples)) if len(samples) else 0
    return (samples / (peak + 0.000001)).astype(np.float32)


def trim_to_frames(x):
    """SN/apply.py:158-161 -- drop the tail so (len - 400) % 160 == 0 (mixture only)."""
    r = (len(x) - WIN) % HOP
    return x[:-r] if r != 0 else x


def handle_signals(mixedpath, ctx1path, ctx2path):
    """SN/apply.py:142-167 (SS/apply.py:111-136): returns (ctx1, ctx2, mixed) float32."""
    mixed = trim_to_frames(normalise(read_wav(mixedpath)))
    return normalise(read_wav(ctx1path)), normalise(read_wav(ctx2path)), mixed


# ----------------------------------------------------------------------------- STFT features
def hann_periodic(n=WIN):
    """tf.signal.hann_window(periodic=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=F64) / n)


def stft(x):
    """tf.signal.stft(x, 400, 160, fft_length=400), pad_end=False (SN/apply.py:370)."""
    x = np.asarray(x, dtype=F64)
    if len(x) < WIN:
        return np.zeros((0, BINS), dtype=np.complex128)
    t = 1 + (len(x) - WIN) // HOP
    idx = np.arange(WIN)[None, :] + HOP * np.arange(t)[:, None]
    return np.fft.rfft(x[idx] * hann_periodic()[None, :], n=WIN, axis=1)


def logmag_phase(spec):
    """SN/apply.py:373-375 -- log(|X| + 1e-5), angle(X)."""
    return np.log(np.abs(spec) + 1e-5), np.angle(spec)


def strided_crop(feat, length):
    """SN/apply.py:170-186 -- zero-pad ((L+1)//2 - 1, L//2) rows, all length-L windows, stride 1."""
    before, after = ((length + 1) // 2) - 1, length // 2
    padded = np.pad(feat, [[before, after], [0, 0]])
    t = feat.shape[0]
    idx = np.arange(length)[None, :] + np.arange(t)[:, None]
    return padded[idx]                      # [T, length, F]


def context(feat):
    """SN/apply.py:381-382 -- first 200 frames, reshape [200, 201] (needs >= 200 frames)."""
    c = feat[:NOISE_WIN]
    return c.reshape(NOISE_WIN, feat.shape[1])


# ----------------------------------------------------------------------------- NN primitives
def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def conv2d(x, w, strides=(1, 1), padding="SAME"):
    """tf.nn.conv2d, NHWC input, HWIO kernel, cross-correlation (SN/blocks.py:38-48)."""
    b, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    sh, sw = strides
    if padding == "SAME":
        ho, pt, pb = same_pad(h, kh, sh)
        wo, pl, pr = same_pad(wd, kw, sw)
        x = np.pad(x, [[0, 0], [pt, pb], [pl, pr], [0, 0]])
    else:
        ho, wo = (h - kh) // sh + 1, (wd - kw) // sw + 1
    out = np.zeros((b * ho * wo, cout), dtype=F64)
    for i in range(kh):
        for j in range(kw):
            patch = x[:, i:i + (ho - 1) * sh + 1:sh, j:j + (wo - 1) * sw + 1:sw, :]
            out += patch.reshape(-1, cin) @ w[i, j]
    return out.reshape(b, ho, wo, cout)


def batch_norm(x, W, scope):
    """Inference branch of SN/blocks.py:104-108."""
    g = W[scope + "/gamma"].astype(F64)
    bt = W[scope + "/beta"].astype(F64)
    m = W[scope + "/pop_mean"].astype(F64)
    v = W[scope + "/pop_variance"].astype(F64)
    return (x - m) * (g / np.sqrt(v + BN_EPS)) + bt


def relu(x):
    return np.maximum(x, 0.0)


def dense(x, W, scope, with_bias):
    """SN/blocks.py:23-35."""
    out = x @ W[scope + "/w"].astype(F64)
    if with_bias:
        out = out + W[scope + "/b"].astype(F64)
    return out


def conv(x, W, scope, strides, padding, with_bias):
    out = conv2d(x, W[scope + "/w"].astype(F64), strides, padding)
    if with_bias:
        out = out + W[scope + "/b"].astype(F64)
    return out


# ----------------------------------------------------------------------------- network
TOWER = [("noise_resblock1_1", (3, 2)), ("noise_resblock2_1", (3, 2)),
         ("noise_resblock3_1", (1, 1)), ("noise_resblock4_1", (1, 2))]     # SN/main.py:194-198
STACK = [("resblock1_1", 1), ("resblock1_2", 1), ("resblock2_1", 2), ("resblock2_2", 1),
         ("resblock3_1", 2), ("resblock3_2", 1), ("resblock4_1", 2), ("resblock4_2", 1)]  # :221-229


def noise_resnet_block(x, W, scope, stride):
    """SN/main.py:102-124 (channels always change in the tower, so the identity is a 1x1 conv)."""
    p1 = conv(x, W, scope + "_conv1", stride, "SAME", False)
    p1 = relu(batch_norm(p1, W, scope + "_conv1"))
    p1 = conv(p1, W, scope + "_conv2", (1, 1), "SAME", True)
    if x.shape[3] == p1.shape[3]:
        p2 = x
    else:
        p2 = conv(x, W, scope + "_transform", stride, "SAME", True)
    return relu(batch_norm(p1 + p2, W, scope + "_addition"))


def embed_tower(ctx, W, taps=None):
    """SN/main.py:190-216 -- ctx [B,200,201] -> [B,512]; variables under 'embedding/'."""
    x = np.asarray(ctx, dtype=F64)[..., None]
    for name, stride in TOWER:
        x = noise_resnet_block(x, W, "embedding/" + name, stride)
        if taps is not None:
            taps[name] = x
    return x.mean(axis=(1, 2))              # avg_pool over the whole map, VALID


def cont_embed(n, W, scope):
    """SN/main.py:127-137 -- MLP of the integer positions 0..n-1; BN scopes are doubled."""
    z = np.arange(n, dtype=F64).reshape(n, 1)
    z = relu(batch_norm(dense(z, W, scope + "_dense1", False), W, scope + scope + "_dense1"))
    z = relu(batch_norm(dense(z, W, scope + "_dense2", False), W, scope + scope + "_dense2"))
    return dense(z, W, scope + "_dense3", False)


def emb_scopes(kind):
    """Projection scope suffixes in resnet_block argument order: SN/main.py:142,146;
    SS/main.py:157,161."""
    return ("_noise_pos_emb", "_noise_neg_emb") if kind == "denoiser" else ("_noise_emb", "_clean_emb")


def resnet_block(x, emb_a, emb_b, W, scope, stride, kind):
    """SN/main.py:126-187."""
    sa, sb = emb_scopes(kind)

    def cond(match, s):                                   # process_noise_t_f, :139-159
        pa = dense(emb_a, W, s + sa, True)[:, None, None, :]
        pb = dense(emb_b, W, s + sb, True)[:, None, None, :]
        t = cont_embed(match.shape[1], W, s + "_temb")[None, :, None, :]
        f = cont_embed(match.shape[2], W, s + "_femb")[None, None, :, :]
        return pa, pb, t, f

    p1 = conv(x, W, scope + "_conv1", (stride, stride), "SAME", False)
    pa, pb, t, f = cond(p1, scope + "_conv1")
    p1 = p1 + pa + pb + t + f
    p1 = relu(batch_norm(p1, W, scope + "_conv1"))
    p1 = conv(p1, W, scope + "_conv2", (1, 1), "SAME", True)
    pa, pb, t, f = cond(p1, scope + "_conv2")
    p1 = p1 + pa + pb + t + f
    if x.shape[3] == p1.shape[3]:
        p2 = x
    else:
        p2 = conv(x, W, scope + "_transform", (stride, stride), "SAME", True)
    return relu(batch_norm(p1 + p2, W, scope + "_addition"))


def mask_net(mixed, emb_a, emb_b, W, kind="denoiser", taps=None):
    """SN/main.py:219-242 given the two embeddings.  mixed [B,35,201]; emb_* [B,512].
    Returns (out, denoised): `out` = last_dense output (the "mask logits"), denoised = add_72."""
    mixed = np.asarray(mixed, dtype=F64)
    x = mixed[..., None]
    for name, stride in STACK:
        x = resnet_block(x, emb_a, emb_b, W, name, stride, kind)
        if taps is not None:
            taps[name] = x
    x = conv(x, W, "last_conv", (1, 1), "VALID", False)
    x = relu(batch_norm(x, W, "last_conv"))
    if taps is not None:
        taps["last_conv"] = x
    x = x.reshape(x.shape[0], -1)                          # flatten, SN/blocks.py:64-69
    out = dense(x, W, "last_dense", True)
    return out, mixed[:, MIX_WIN // 2, :] + out


def model(mixed, ctx_a, ctx_b, W, kind="denoiser"):
    """Whole graph as the reference feeds it (contexts tiled per frame): SN/main.py:98-256."""
    return mask_net(mixed, embed_tower(ctx_a, W), embed_tower(ctx_b, W), W, kind)


# ----------------------------------------------------------------------------- reconstruction
def istft_window():
    """tf.signal.inverse_stft_window_fn(160, hann periodic)(400): w[n] / sum_q w^2[n%160 + 160 q]."""
    w = hann_periodic()
    den = np.zeros(HOP, dtype=F64)
    for q in range(-(-WIN // HOP)):
        seg = w[q * HOP:(q + 1) * HOP] ** 2
        den[:len(seg)] += seg
    return w / np.tile(den, -(-WIN // HOP))[:WIN]


def inverse_stft(spec):
    """tf.signal.inverse_stft(S, 400, 160, 400, window_fn=inverse_stft_window_fn(...)):
    irfft per frame, synthesis window, overlap-add at hop 160 (SN/apply.py:199-201)."""
    t = spec.shape[0]
    if t == 0:
        return np.zeros(0, dtype=F64)
    frames = np.fft.irfft(spec, n=WIN, axis=1) * istft_window()[None, :]
    out = np.zeros((t - 1) * HOP + WIN, dtype=F64)
    for i in range(t):
        out[i * HOP:i * HOP + WIN] += frames[i]
    return out


def recover_samples(logspec, phase):
    """SN/apply.py:189-204 -- exp, polar, complex64 feed, inverse STFT; float32 result."""
    spec = np.exp(logspec) * np.exp(1j * phase)
    return inverse_stft(spec)


# ----------------------------------------------------------------------------- whole apply path
def enhance(mixed_wav, ctx_a_wav, ctx_b_wav, W, kind="denoiser", batch=8, frames=None,
            compensate=0.0, ac=False):
    """apply_snc (SN/apply.py:339-472) / apply_separator (SS/apply.py:288-397) on normalised
    float32 waveforms (mixture already trimmed).  ctx_a / ctx_b are in resnet_block argument
    order: denoiser (pos, neg); separator (noise=--neg, clean=--pos).
    `frames`: optional subset of frame indices for the network (others left undenoised).
    Returns a dict of float64 arrays."""
    lm, ph = logmag_phase(stft(mixed_wav))
    la, _ = logmag_phase(stft(ctx_a_wav))
    lb, _ = logmag_phase(stft(ctx_b_wav))
    emb_a = embed_tower(context(la)[None], W)
    emb_b = embed_tower(context(lb)[None], W)
    win = strided_crop(lm, MIX_WIN)
    t = lm.shape[0]
    sel = np.arange(t) if frames is None else np.asarray(frames)
    logits = np.zeros((t, BINS), dtype=F64)
    for i in range(0, len(sel), batch):
        idx = sel[i:i + batch]
        o, _ = mask_net(win[idx], np.repeat(emb_a, len(idx), 0), np.repeat(emb_b, len(idx), 0), W, kind)
        logits[idx] = o
    denoised = lm + logits
    res = dict(logmag=lm, phase=ph, emb_a=emb_a[0], emb_b=emb_b[0], logits=logits, denoised=denoised)
    res["denoised_wav"] = recover_samples(denoised, ph)
    res["mixed_wav"] = recover_samples(lm, ph)             # *mixed_processed.wav, :457-458
    if kind == "denoiser":                                  # SN/apply.py:459-472
        removed = res["mixed_wav"] - res["denoised_wav"]
        res["removed_wav"] = removed
        with np.errstate(divide="ignore", invalid="ignore"):
            snr_est = np.mean(np.square(res["denoised_wav"])) / np.mean(np.square(removed))
        res["snr_est"] = snr_est
        factor = snr_est / 20 if ac else compensate
        res["compensated_wav"] = res["denoised_wav"] + removed * factor
    return res


# ----------------------------------------------------------------------------- demo / eval mode
def _fit(noise, n):
    """SN/apply.py:58-72 -- repeat the noise while shorter than the speech, cut if longer."""
    nse = noise
    while n - len(nse) > 0:
        nse = np.concatenate([nse, noise[:n - len(nse)]], axis=0)
    return noise[:n] if n - len(noise) < 0 else nse


def power(x):
    """`sum(abs(x) * abs(x)) / x.shape[0]` (SN/apply.py:75-77) as the reference's pinned stack
    (TF 1.14, NumPy 1.x) evaluates it: builtin sum() starts from the int 0, `0 + np.float32` is a
    float64 there, so the float32 squares are accumulated sequentially in float64."""
    acc = 0.0
    for v in (abs(x) * abs(x)).tolist():
        acc += v
    return np.float64(acc) / x.shape[0]


def domixing(clean, pos, neg, snr_pos, snr_neg):
    """SN/apply.py:56-104 (== SN/reader.py:128-176), including the re-use of the normalised `mixed`
    when "normalising" target and the noise signals (:98-102)."""
    nse_pos, nse_neg = _fit(pos, len(clean)), _fit(neg, len(clean))
    ps, pp, pn = power(clean), power(nse_pos), power(nse_neg)
    k_pos = 1 if pp == 0 else np.sqrt((ps / pp) * pow(10, -snr_pos / 10.0))
    k_neg = 1 if pn == 0 else np.sqrt((ps / pn) * pow(10, -snr_neg / 10.0))
    # (NumPy 1.x: a float64 scalar times a float32 array is a float32 product)
    pos_s, neg_s = nse_pos.dtype.type(k_pos) * nse_pos, nse_neg.dtype.type(k_neg) * nse_neg
    mixed = clean + pos_s + neg_s
    mixed = mixed / (max(abs(mixed)) + 0.000001)
    peak = max(abs(mixed)) + 0.000001
    return mixed, (clean + pos_s) / peak, k_pos, k_neg, pos_s / peak, neg_s / peak


def demo_signals(clean_i16, pos_i16, neg_i16, snr_pos=0, snr_neg=0):
    """combine_signals of SN/apply.py:107-135 on already-read int16 arrays."""
    clean = trim_to_frames(normalise(clean_i16))
    mixed, target, _, _, pos_sig, neg_sig = domixing(clean, normalise(pos_i16), normalise(neg_i16), snr_pos, snr_neg)
    return target, pos_sig, neg_sig, mixed


def enhance_after_context(mixed, ctx_a_wav, ctx_b_wav, W, kind="denoiser", frames=None):
    """apply_demo / eval reader (SN/apply.py:247-266,318-336; SN/reader.py:398-409): contexts = first
    200 frames of the conditioning signals, network on mix frames from 200 on, windowed on their own."""
    lm, ph = logmag_phase(stft(mixed))
    la, _ = logmag_phase(stft(ctx_a_wav))
    lb, _ = logmag_phase(stft(ctx_b_wav))
    ea, eb = embed_tower(context(la)[None], W), embed_tower(context(lb)[None], W)
    lm_s, ph_s = lm[NOISE_WIN:], ph[NOISE_WIN:]
    win = strided_crop(lm_s, MIX_WIN)
    sel = np.arange(lm_s.shape[0]) if frames is None else np.asarray(frames)
    logits = np.zeros_like(lm_s)
    for i in range(0, len(sel), 8):
        idx = sel[i:i + 8]
        o, _ = mask_net(win[idx], np.repeat(ea, len(idx), 0), np.repeat(eb, len(idx), 0), W, kind)
        logits[idx] = o
    den = lm_s + logits
    return dict(logmag=lm_s, phase=ph_s, logits=logits, denoised=den, denoised_wav=recover_samples(den, ph_s),
                mixed_wav=recover_samples(lm_s, ph_s))
