"""Drop-in API surface of N-HANS inference on MI355X.

Mirrors the reference's entry points -- same names, argument meaning, file outputs and error
behaviour -- with the TensorFlow graph calls replaced by the HIP library (engine.Engine):

    apply_snc(mixedpath, pospath, negpath, save_to)        SN/apply.py:339-472
    apply_denoiser(mixedpath, negpath, save_to)            SN/apply.py:478-481
    apply_separator(mixedpath, cleanpath, noisepath, save_to)   SS/apply.py:288-397
    main() for the `nhans_denoiser` / `nhans_separator` console scripts   setup.py:44-50,
        flags --input --neg --pos --output --compensate --ac            SN/apply.py:29-35

Host code here is I/O only: wav reading/normalising/trimming (numpy, float64 like the reference),
writing float32 wavs, and the denoiser's removed/compensated side outputs.

Deviations that the tree forces (documented in DESIGN.md):
  * Conditioning recordings shorter than 200 frames (32,240 samples) crash the in-tree reference
    (`tf.reshape(..., [200, 201])`, SN/apply.py:381-382) although its own examples are 1 s long.
    They are repeated to the needed length with the reference's own repeat rule for short noise
    (`domixing`, SN/apply.py:60-66).
  * `./audio_examples/Silent.wav` (default --pos) is all zeros; if the file is absent an all-zero
    recording is used.
  * Weights: `./trained_model/<bundle>` relative to the CWD as in the reference (or
    $NHANS_MODEL_DIR).  The bundles in the reference tree are git-LFS pointers; loading one raises
    unless synthetic weights are requested explicitly (--weights synthetic / NHANS_WEIGHTS=synthetic).
"""
import argparse
import os
import sys

import numpy as np
from scipy.io.wavfile import read as wavread
from scipy.io.wavfile import write as wavwrite

from . import spec

Fs = spec.FS
Noise_Win = spec.NOISE_WIN
Mix_Win = spec.MIX_WIN

DENOISER_BUNDLE = "81448_0-1000000"      # SN/apply.py:431-432
SEPARATOR_BUNDLE = "81457_2-545000"      # SS/apply.py:371-372

_engines = {}
_lite = {}               # torch-free engines of the single-process command line (lite.py)
_device_index = 0
TIMING = None            # --timing: dict of the seconds each start-up step took (printed as JSON on stderr)


class Flags(object):
    """Stand-in for the reference's absl FLAGS (SN/apply.py:29-35)."""
    input = "./audio_examples/mixed.wav"
    neg = "./audio_examples/game_noise.wav"
    pos = "./audio_examples/Silent.wav"
    output = "./audio_examples/denoised.wav"
    compensate = 0.0
    ac = False
    convert = False      # library calls keep the in-tree asserts; the CLI converts like the packaged tool
    Fs = Fs
    weights = os.environ.get("NHANS_WEIGHTS", "checkpoint")
    model_dir = os.environ.get("NHANS_MODEL_DIR", "./trained_model")
    cache = True         # folded-blob cache (blobcache.py); --no-cache folds the weights in this process


FLAGS = Flags()


# ------------------------------------------------------------------------------ wav front end
def read_wav(in_path):
    """Contract of the reference's reader (SN/apply.py:46-53): 16 kHz int16 PCM only, anything else
    is an AssertionError; a multi-channel file comes back as the float64 mean of its channels."""
    rate, pcm = wavread(in_path)
    if rate != FLAGS.Fs:
        raise AssertionError("%s: sample rate %d Hz, expected %d" % (in_path, rate, FLAGS.Fs))
    if pcm.dtype != np.int16:
        raise AssertionError("%s: sample format %s, expected int16" % (in_path, pcm.dtype))
    return pcm if pcm.ndim == 1 else pcm.mean(axis=1)


def read_wav_any(in_path):
    """Format converter of the packaged tool (README.md:42: other formats are "automatically
    converted" to 16 kHz / 16-bit PCM with sox; sox is not available, scipy does the same job):
    any PCM/float wav at any rate -> what read_wav() would have returned for the converted file."""
    rate, samples = wavread(in_path)
    if rate == FLAGS.Fs and samples.dtype == np.int16:        # already what read_wav() accepts
        return samples if samples.ndim == 1 else samples.mean(axis=1)
    x = samples.astype(np.float64)
    if samples.dtype == np.uint8:
        x = (x - 128.0) * 256.0
    elif samples.dtype == np.int32:
        x = x / 65536.0
    elif samples.dtype.kind == 'f':
        x = x * 32768.0
    if rate != FLAGS.Fs:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(FLAGS.Fs), int(rate))
        x = resample_poly(x, FLAGS.Fs // g, rate // g, axis=0)
    x = np.clip(np.round(x), -32768, 32767).astype(np.int16)
    if x.ndim > 1:
        x = x.mean(axis=1)
    return x


def _reader():
    return read_wav_any if getattr(FLAGS, 'convert', False) else read_wav


def normalise(samples):
    """x / (max(abs(x)) + 1e-6) in float64 -> float32 (SN/apply.py:150-155); int16 abs wraps like
    the reference's."""
    with np.errstate(over="ignore"):
        peak = np.max(np.abs(samples)) if len(samples) else 0
    return (samples / (peak + 0.000001)).astype(np.float32)


def trim_to_frames(samples):
    """Drop the tail that does not fill a hop, so that the STFT covers every kept sample
    (SN/apply.py:158-161).  Python's non-negative modulo applies to recordings shorter than one
    window too, exactly as in the reference (300 samples -> 240, which then has no frame at all)."""
    spare = (len(samples) - spec.WIN) % spec.HOP
    return samples[:len(samples) - spare] if spare else samples


def extend_context(samples):
    """Short-context policy: repeat the recording until it yields Noise_Win frames, with the
    reference's repeat rule for noise shorter than speech (SN/apply.py:60-66)."""
    need = spec.MIN_CTX_SAMPLES
    if len(samples) == 0:
        return np.zeros(need, dtype=samples.dtype)
    out = samples
    while need - len(out) > 0:
        out = np.concatenate([out, samples[:need - len(out)]], axis=0)
    return out


def handle_signals(mixedpath, noisepospath, noisenegpath):
    """SN/apply.py:142-167: returns (pos, neg, mixed) float32; wav problems print
    'error in threads' and yield None, like the reference's bare except."""
    try:
        mixedsamples = _reader()(mixedpath)
        noisepossamples = _read_context(noisepospath)
        noisenegsamples = _read_context(noisenegpath)
        mixedsamples = trim_to_frames(normalise(mixedsamples))
        if len(mixedsamples) < spec.WIN:
            raise ValueError("%s: shorter than one %d-sample STFT window" % (mixedpath, spec.WIN))
        return normalise(noisepossamples), normalise(noisenegsamples), mixedsamples
    except Exception:
        print('error in threads')
        print(mixedpath, noisepospath, noisenegpath)
        return None


def _read_context(path):
    if path is None or (os.path.basename(path) == "Silent.wav" and not os.path.exists(path)):
        return np.zeros(spec.MIN_CTX_SAMPLES, dtype=np.int16)
    return extend_context(_reader()(path))


# ------------------------------------------------------------------------------ engine / weights
def _load_weights(kind):
    from . import weights
    if FLAGS.weights == "synthetic":
        return weights.synthetic_weights(kind, 7)
    bundle = DENOISER_BUNDLE if kind == spec.DENOISER else SEPARATOR_BUNDLE
    return weights.load_checkpoint(os.path.join(FLAGS.model_dir, bundle), kind)


def get_engine(kind):
    """The full engine (stage-level entry points, streams, torch tensors): demo / evaluation mode, the sharded
    multi-GPU command line, tests."""
    if kind not in _engines:
        if _lite:
            raise RuntimeError("this process runs the torch-free command-line engine (lite.py); the full engine needs "
                               "PyTorch's HIP runtime loaded first -- use one or the other in a process")
        from . import engine
        _engines[kind] = engine.Engine(kind, _load_weights(kind), device=_device_index)
    return _engines[kind]


def _cache_key(kind):
    from . import blobcache
    if FLAGS.weights == "synthetic":
        return blobcache.key_for_synthetic(kind, 7)
    bundle = DENOISER_BUNDLE if kind == spec.DENOISER else SEPARATOR_BUNDLE
    return blobcache.key_for_checkpoint(os.path.join(FLAGS.model_dir, bundle), kind)


def get_enhancer(kind):
    """What the file / directory command line runs on: an engine with .enhance(mixes, ctx_a, ctx_b, want_mixed).  A
    pre-installed or already built full engine if there is one (set_engine: tests, drivers) or if PyTorch is in the
    process anyway; otherwise -- a fresh `nhans_denoiser` process, the reference's normal use (SN/apply.py:478-527) --
    the torch-free LiteEngine on the cached folded blob: no `import torch` (1.5 s), no folding (3 s), no calibration
    pass."""
    if kind in _engines:
        return _engines[kind]
    if "torch" in sys.modules or int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("NHANS_FORCE_TORCH_ENGINE") == "1":
        return get_engine(kind)
    if kind not in _lite:
        os.environ["NHANS_NO_TORCH"] = "1"               # (hip.load: do not pull torch in for its HIP runtime)
        from . import lite
        _lite[kind] = lite.cached_engine(kind, _cache_key(kind) if FLAGS.cache else None, lambda: _load_weights(kind),
                                         device=_device_index, use_cache=FLAGS.cache, timing=TIMING)
    return _lite[kind]


def set_engine(kind, eng):
    """Install a pre-built engine (tests, benchmarks, multi-GPU drivers)."""
    _engines[kind] = eng


# ------------------------------------------------------------------------------ apply_*
def _enhance_files(kind, mixedpath, ctx_a_path, ctx_b_path):
    """-> (denoised_samples, mixed_processed_samples) float32 for one file triple; ctx order is
    resnet_block argument order."""
    import time
    t0 = time.perf_counter()
    sig = handle_signals(mixedpath, ctx_a_path, ctx_b_path)
    if sig is None:
        raise RuntimeError("could not read %s / %s / %s" % (mixedpath, ctx_a_path, ctx_b_path))
    ca, cb, mixed = sig
    t1 = time.perf_counter()
    eng = get_enhancer(kind)
    t2 = time.perf_counter()
    res = eng.enhance([mixed], [ca], [cb], want_mixed=True)
    if TIMING is not None:
        TIMING.update(read_wavs_s=t1 - t0, engine_s=t2 - t1, enhance_s=time.perf_counter() - t2,
                      engine=type(eng).__name__, audio_s=len(mixed) / float(FLAGS.Fs))
    return res["denoised_wav"][0], res["mixed_wav"][0]


def side_prefix(save_to):
    """Prefix of the side files written next to `save_to`.  The reference cuts 12 characters off the
    output path (`save_to[:-12]`, SN/apply.py:457-470) -- right for its default `.../denoised.wav`
    and for any `<prefix>denoised.wav`, wrong for everything else (for `out/a.wav` the cut eats the
    directory, and in directory mode every clip would overwrite the same three files).  So: the
    reference's cut when the name ends in `denoised.wav`, else `<output without .wav>_`."""
    if os.path.basename(save_to).endswith('denoised.wav'):
        return save_to[:-12]
    return os.path.splitext(save_to)[0] + '_'


def write_snc_outputs(save_to, denoised_samples, mixed_samples):
    """The four files of SN/apply.py:456-472: denoised, mixed_processed (STFT->iSTFT round trip of
    the input), removed = mixed - denoised, compensated = denoised + removed * factor with the
    factor FLAGS.compensate or, under --ac, snr_est / 20.  Returns (snr_est, factor)."""
    pre = side_prefix(save_to)
    wavwrite(save_to, FLAGS.Fs, denoised_samples)
    wavwrite(pre + 'mixed_processed.wav', FLAGS.Fs, mixed_samples)
    removed_samples = mixed_samples - denoised_samples
    wavwrite(pre + 'removed.wav', FLAGS.Fs, removed_samples)
    with np.errstate(divide="ignore", invalid="ignore"):
        snr_est = np.mean(np.square(denoised_samples)) / np.mean(np.square(removed_samples))
    print(snr_est)
    print('---------------------------')
    factor = snr_est / 20 if FLAGS.ac else FLAGS.compensate
    compensated_samples = denoised_samples + removed_samples * factor
    wavwrite(pre + 'compensated.wav', FLAGS.Fs, compensated_samples.astype(np.float32))
    return snr_est, factor


def apply_snc(mixedpath, pospath, negpath, save_to):
    """Selective noise suppression: keep `pos`-like noise, remove `neg`-like noise
    (SN/apply.py:339-472).  Writes save_to plus the *mixed_processed / *removed / *compensated
    side files (naming: side_prefix)."""
    denoised_samples, mixed_samples = _enhance_files(spec.DENOISER, mixedpath, pospath, negpath)
    write_snc_outputs(save_to, denoised_samples, mixed_samples)


def apply_denoiser(mixedpath, negpath, save_to):
    """SN/apply.py:478-481: plain denoising = selective suppression with a silent positive."""
    dir = './audio_examples/'
    pospath = dir + 'Silent.wav'
    apply_snc(mixedpath, pospath, negpath, save_to)


def apply_separator(mixedpath, cleanpath, noisepath, save_to):
    """SS/apply.py:288-397: keep the `cleanpath` (target, --pos) speaker, remove the `noisepath`
    (interferer, --neg) speaker.  resnet_block order is (noise, clean), SS/main.py:205-242."""
    denoised_samples, mixed_samples = _enhance_files(spec.SEPARATOR, mixedpath, noisepath, cleanpath)
    write_separator_outputs(save_to, denoised_samples, mixed_samples)


def write_separator_outputs(save_to, denoised_samples, mixed_samples):
    """SS/apply.py:391-397: the separated target and the round trip of the input."""
    wavwrite(save_to, FLAGS.Fs, denoised_samples)
    wavwrite(side_prefix(save_to) + 'mixed_processed.wav', FLAGS.Fs, mixed_samples)


# Audio per nhans_enhance_clips call in directory mode: 4,096 s = 410 ten-second clips = 1.4 GB of
# spectra and waveforms beside the fixed 20 GB stack workspace; a longer job list runs as several calls.
MAX_CALL_SAMPLES = 4096 * Fs

_owns_process_group = False


def _enhance_in_calls(eng, mixes, ca, cb):
    """-> [(denoised, mixed_processed)] per clip, the clips going through the hot path in as few
    ragged calls as MAX_CALL_SAMPLES allows (one for any ordinary directory)."""
    outs, i = [], 0
    while i < len(mixes):
        j, tot = i, 0
        while j < len(mixes) and (j == i or tot + len(mixes[j]) <= MAX_CALL_SAMPLES):
            tot += len(mixes[j])
            j += 1
        res = eng.enhance(mixes[i:j], ca[i:j], cb[i:j], want_mixed=True)
        outs.extend(zip(res["denoised_wav"], res["mixed_wav"]))
        i = j
    return outs


def apply_batch(kind, jobs):
    """Directory mode (README.md:59-66) as ONE batch: `jobs` is a list of (mixedpath, pospath,
    negpath, save_to).  The recordings go through the hot path in ragged `nhans_enhance_clips`
    calls; under `python -m torch.distributed.run` (WORLD_SIZE > 1) the JOB LIST is sharded over the
    ranks (job i -> rank floor(i*G/N), dist.py): each rank reads, converts and normalises only its
    own block of files and runs it on its own GPU, one all-gather reassembles the waveforms (a job
    whose files could not be read travels as length -1, dist.gather_ragged) and rank 0 writes the files.
    Returns the number of clips written by this rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        from . import dist as nd
    lo, hi = nd.shard_bounds(len(jobs), world, rank) if world > 1 else (0, len(jobs))
    sigs = []
    for mixedpath, pospath, negpath, _ in jobs[lo:hi]:
        a_path, b_path = (pospath, negpath) if kind == spec.DENOISER else (negpath, pospath)
        sigs.append(handle_signals(mixedpath, a_path, b_path))      # None: unreadable triple, reported there, skipped
    good = [s for s in sigs if s is not None]
    if world == 1 and not good:
        return 0
    eng = get_enhancer(kind)
    done = iter(_enhance_in_calls(eng, [s[2] for s in good], [s[0] for s in good], [s[1] for s in good]))
    outs = [None if s is None else next(done) for s in sigs]
    if world > 1:
        import torch.distributed as tdist
        if not tdist.is_initialized():
            global _owns_process_group
            tdist.init_process_group("nccl" if os.environ.get("NHANS_DIST_BACKEND", "nccl") == "nccl" else "gloo")
            _owns_process_group = True
        gather_dev = eng.device if tdist.get_backend() == "nccl" else torch.device("cpu")
        # denoised and round trip travel in one tensor per clip: ONE data all-gather
        local = [None if o is None else torch.from_numpy(np.concatenate(o)).to(gather_dev) for o in outs]
        both = nd.gather_ragged(local, len(jobs), gather_dev)
        if rank != 0:
            return 0
        outs = [None if t is None else (t[:t.numel() // 2].cpu().numpy(), t[t.numel() // 2:].cpu().numpy())
                for t in both]
    written = 0
    for (_, _, _, save_to), o in zip(jobs, outs):
        if o is None:
            continue
        if kind == spec.DENOISER:
            write_snc_outputs(save_to, o[0], o[1])
        else:
            write_separator_outputs(save_to, o[0], o[1])
        written += 1
    return written


def finish_distributed():
    """End of a sharded CLI run: the other ranks wait until rank 0 has written the files, then the
    process group this module created is torn down (no RCCL teardown warnings from early exits)."""
    global _owns_process_group
    if not _owns_process_group:
        return
    import torch.distributed as tdist
    if tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()
    _owns_process_group = False


# ------------------------------------------------------------------------------ demo / eval mode
def _enhance_after_context(eng, mix_wav, ctx_a_wav, ctx_b_wav, extra_wavs=()):
    """Shared by apply_demo and the evaluation reader (SN/apply.py:247-266, SN/reader.py:398-409):
    the first Noise_Win frames of the two conditioning signals are the contexts, and the network
    runs on the mixture's frames FROM Noise_Win ON, windowed as a clip of its own (zero rows before
    frame Noise_Win, not the preceding frames).  Uses the stage-level C ABI.  Returns
    (denoised_samples, mixed_samples, [iSTFT of each extra signal's frames from Noise_Win on],
    logits, denoised_logmag)."""
    import torch
    dev = eng.device
    wavs = [mix_wav] + list(extra_wavs)
    flat = torch.from_numpy(np.concatenate([np.asarray(w, dtype=np.float32) for w in wavs])).to(dev)
    off = [0]
    for w in wavs:
        off.append(off[-1] + len(w))
    lm, ph = eng.stft_features(flat, off)
    nfr = [int(eng.lib.nhans_num_frames(len(w))) for w in wavs]
    t = nfr[0]
    if t <= Noise_Win:
        raise ValueError("recording has %d frames; more than %d are needed" % (t, Noise_Win))
    ctx = torch.from_numpy(np.concatenate([ctx_a_wav, ctx_b_wav]).astype(np.float32)).to(dev)
    cl, _ = eng.stft_features(ctx, [0, len(ctx_a_wav), len(ctx_a_wav) + len(ctx_b_wav)], Noise_Win, False)
    emb = eng.embed(cl.view(2, Noise_Win, spec.BINS))
    lm_s, ph_s = lm[Noise_Win:t].contiguous(), ph[Noise_Win:t].contiguous()
    n = t - Noise_Win
    logits, den = eng.mask_net(lm_s, [0, n], emb[0:1], emb[1:2])
    den_wav, _ = eng.istft(den, ph_s, [0, n])
    mix_rt, _ = eng.istft(lm_s, ph_s, [0, n])
    extras, f0 = [], t
    for k in nfr[1:]:
        w, _ = eng.istft(lm[f0 + Noise_Win:f0 + k].contiguous(), ph[f0 + Noise_Win:f0 + k].contiguous(), [0, k - Noise_Win])
        extras.append(w.cpu().numpy())
        f0 += k
    torch.cuda.synchronize(dev)
    return den_wav.cpu().numpy(), mix_rt.cpu().numpy(), extras, logits.cpu().numpy(), den.cpu().numpy(), lm_s.cpu().numpy()


def apply_demo(speechpath, pospath, negpath, save_to):
    """Denoiser demo (SN/apply.py:212-336): mix clean speech with a positive and a negative noise
    at 0 dB each, condition on the first 200 frames of the two (scaled) noises, enhance the rest.
    Writes save_to and save_to[:-15] + 'mixed_demo.wav'."""
    from . import mixing
    _, pos_sig, neg_sig, mixed, _, _ = mixing.combine_signals(read_wav, speechpath, pospath, negpath)
    den, mix_rt, _, _, _, _ = _enhance_after_context(get_engine(spec.DENOISER), mixed, pos_sig, neg_sig)
    wavwrite(save_to, FLAGS.Fs, den)
    wavwrite(save_to[:-15] + 'mixed_demo.wav', FLAGS.Fs, mix_rt)


def apply_demo_separator(cleanpath, noisepath, save_to):
    """Separator demo (SS/apply.py:179-285): mix two speakers at 0 dB; contexts are the first 200
    frames of the interferer (noise*K -> noisecontextph) and of the target (clean -> cleancontextph)."""
    from . import mixing
    clean, noise_k, mixed, _ = mixing.combine_signals_separator(read_wav, cleanpath, noisepath)
    den, mix_rt, _, _, _, _ = _enhance_after_context(get_engine(spec.SEPARATOR), mixed, noise_k, clean)
    wavwrite(save_to, FLAGS.Fs, den)
    wavwrite(save_to[:-15] + 'mixed_demo.wav', FLAGS.Fs, mix_rt)


def evaluate_utterance(cleanpath, noisepospath, noisenegpath, dump_dir, modelname='nhans', step=0):
    """Evaluation-reader pipeline for one (speech, pos noise, neg noise) triple: SN/reader.py:310-420
    (eval branch, eval_stride 1) + SN/main.py:264-353.  Mixes at the reference's md5-derived SNRs,
    runs the network on frames from context_frames on, writes the reference's five wavs
    `<model>_<step>_<clean>_<pos>_<neg>_<snrp>_<snrn>_{mixed,denoised,target,posNoise,negNoise}.wav`
    and returns the loss of SN/main.py:245-248 for this utterance."""
    from . import mixing
    target, pos_sig, neg_sig, mixed, snr_pos, snr_neg = mixing.combine_signals(
        read_wav, cleanpath, noisepospath, noisenegpath, snrs=mixing.eval_snrs(cleanpath))
    eng = get_engine(spec.DENOISER)
    den, mix_rt, extras, _, den_lm, _ = _enhance_after_context(eng, mixed, pos_sig, neg_sig,
                                                               extra_wavs=(target, pos_sig, neg_sig))
    import torch
    tgt_lm, _ = eng.stft_features(torch.from_numpy(np.asarray(target, dtype=np.float32)).to(eng.device), [0, len(target)], 0, False)
    tgt_lm = tgt_lm[Noise_Win:].cpu().numpy()
    imp = np.linspace(2, 1, spec.BINS, dtype=np.float32).reshape(1, spec.BINS)
    loss = float(np.mean(np.mean(np.square(den_lm - tgt_lm) * imp, axis=1)))
    base = [os.path.basename(p)[:-4] for p in (cleanpath, noisepospath, noisenegpath)]
    os.makedirs(dump_dir, exist_ok=True)
    for kind, w in (('mixed', mix_rt), ('denoised', den), ('target', extras[0]), ('posNoise', extras[1]),
                    ('negNoise', extras[2])):
        name = '{}_{}_{}_{}_{}_{}_{}_{}.wav'.format(modelname, step, base[0], base[1], base[2], snr_pos, snr_neg, kind)
        wavwrite(os.path.join(dump_dir, name), FLAGS.Fs, w)
    return loss


# ------------------------------------------------------------------------------ CLI
def _parse(argv, prog):
    p = argparse.ArgumentParser(prog=prog)
    p.add_argument('--input', default=FLAGS.input)
    p.add_argument('--neg', default=FLAGS.neg)
    p.add_argument('--pos', default=FLAGS.pos)
    p.add_argument('--output', default=FLAGS.output)
    p.add_argument('--compensate', type=float, default=0.0)
    p.add_argument('--ac', action='store_true')
    p.add_argument('--weights', default=FLAGS.weights, choices=['checkpoint', 'synthetic'])
    p.add_argument('--model_dir', default=FLAGS.model_dir)
    p.add_argument('--no-cache', dest='cache', action='store_false', default=True,
                   help='fold the weights in this process instead of reading / writing the folded-blob cache '
                        '($NHANS_CACHE_DIR or ~/.cache/nhans_amd: blobcache.py)')
    p.add_argument('--timing', action='store_true', help='print the seconds of each start-up step as JSON on stderr')
    p.add_argument('--no-convert', dest='convert', action='store_false', default=True,
                   help='reject inputs that are not 16 kHz int16 PCM (in-tree reference behaviour) '
                        'instead of converting them (packaged-tool behaviour, README.md:42)')
    a = p.parse_args(argv)
    for k, v in vars(a).items():
        setattr(FLAGS, k, v)
    return a


def _pairs(a):
    """File mode or directory mode: in directory mode inputs and conditioning recordings are
    paired by identical file name (README.md:59-66)."""
    if os.path.isdir(a.input):
        os.makedirs(a.output, exist_ok=True)
        for name in sorted(os.listdir(a.input)):
            if not name.lower().endswith('.wav'):
                continue
            pick = lambda d: os.path.join(d, name) if os.path.isdir(d) else d
            yield os.path.join(a.input, name), pick(a.pos), pick(a.neg), os.path.join(a.output, name)
    else:
        yield a.input, a.pos, a.neg, a.output


def _run_cli(kind, a):
    global TIMING
    import time
    t0 = time.perf_counter()
    TIMING = {} if getattr(a, "timing", False) else None
    try:
        _run_cli_jobs(kind, a)
    finally:
        if TIMING is not None:
            import json
            TIMING["run_cli_s"] = time.perf_counter() - t0
            TIMING["torch_imported"] = "torch" in sys.modules
            sys.stderr.write("nhans timing: " + json.dumps(TIMING) + "\n")
            TIMING = None


def _run_cli_jobs(kind, a):
    jobs = list(_pairs(a))
    if os.path.isdir(a.input) or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        _bind_rank_device()
        try:
            apply_batch(kind, jobs)
        finally:
            finish_distributed()
    else:
        for mixed, pos, neg, out in jobs:
            (apply_snc if kind == spec.DENOISER else apply_separator)(mixed, pos, neg, out)


def _bind_rank_device():
    """One process per GPU under torch.distributed.run: LOCAL_RANK picks the device.  Runs before
    anything has touched the GPU."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if local and all(k not in _engines for k in (spec.DENOISER, spec.SEPARATOR)):
        import torch
        torch.cuda.set_device(local)
        global _device_index
        _device_index = local


def main(argv=None):
    """`nhans_denoiser` (setup.py:46).  `--input/--output` (and optionally `--pos/--neg`) may be
    directories: files are paired by name and processed as one batch; launched through
    `python -m torch.distributed.run --nproc-per-node N -m ...` the batch is sharded over N GPUs."""
    _run_cli(spec.DENOISER, _parse(argv, 'nhans_denoiser'))


def main_separator(argv=None):
    """`nhans_separator` (setup.py:48)."""
    _run_cli(spec.SEPARATOR, _parse(argv, 'nhans_separator'))


if __name__ == '__main__':
    main(sys.argv[1:])
