"""On-disk cache of the folded weight blob (fold.py: 3-4 s of numpy for 240 MB) and of the activation exponents the
library calibrated for it -- what stands between `nhans_denoiser one.wav` and its 21 ms of GPU work (SN/apply.py:478-527
is one process per file).  An entry is keyed by
    sha256( what the weights are | fold.BLOB_VERSION | C-ABI version | model kind )
where "what the weights are" is, for a TensorFlow bundle, the bytes of its .index file (every tensor's shape, offset and
masked crc32c) + the size and modification time of its data file -- content-addressed without hashing 126 MB on every
start -- and for the
seeded synthetic weights the seed and the text of the recipe (weights.py).  Files: <dir>/<key>.blob (the exact bytes
fold_weights returns) and <dir>/<key>.json ({"exponents": [...], "blob_sha256": ...}), written through a temporary name
and renamed, so a reader never sees half a file; a blob whose size or BLOB_VERSION header does not match is ignored and
rebuilt.  <dir> = $NHANS_CACHE_DIR or ~/.cache/nhans_amd.  `--no-cache` (apply.py) bypasses it."""
import hashlib
import json
import os

import numpy as np

from . import fold, hip


def cache_dir():
    return os.environ.get("NHANS_CACHE_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "nhans_amd")


def _key(parts):
    h = hashlib.sha256()
    for p in parts:
        b = p if isinstance(p, bytes) else str(p).encode()
        h.update(len(b).to_bytes(8, "little"))
        h.update(b)
    return h.hexdigest()[:40]


def key_for_checkpoint(prefix, kind):
    """prefix = path of the bundle without .index / .data-00000-of-00001"""
    with open(prefix + ".index", "rb") as f:
        index = f.read()
    st = os.stat(prefix + ".data-00000-of-00001")
    # (size AND modification time of the data file: a data file that was replaced under an unchanged .index -- which the
    # per-tensor CRCs would catch at load time -- must not be served from the cache unread)
    return _key([b"tf-bundle", index, st.st_size, st.st_mtime_ns, fold.BLOB_VERSION, hip.ABI_VERSION, kind])


def key_for_synthetic(kind, seed):
    from . import spec, weights
    src = b""
    for mod in (weights, spec):
        with open(mod.__file__.replace(".pyc", ".py"), "rb") as f:
            src += f.read()
    return _key([b"synthetic", seed, hashlib.sha256(src).digest(), fold.BLOB_VERSION, hip.ABI_VERSION, kind])


def load(key):
    """-> (blob as a read-only numpy uint8 array, exponents list or None) or None"""
    base = os.path.join(cache_dir(), key)
    try:
        blob = np.fromfile(base + ".blob", dtype=np.uint8)
        with open(base + ".json") as f:
            meta = json.load(f)
    except (OSError, ValueError):
        return None
    if meta.get("nbytes") != blob.size or meta.get("blob_version") != fold.BLOB_VERSION or not fold.blob_header_ok(blob):
        return None
    exps = meta.get("exponents")
    if exps is not None and len(exps) != hip.NUM_ACTIVATIONS:
        exps = None
    return blob, exps


def store(key, blob, exponents=None):
    """blob: bytes / bytearray / uint8 array.  Failures to write (read-only home, full disk) are not errors."""
    d = cache_dir()
    try:
        os.makedirs(d, exist_ok=True)
        base = os.path.join(d, key)
        arr = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        tmp = "%s.%d.tmp" % (base, os.getpid())
        arr.tofile(tmp + ".blob")
        os.replace(tmp + ".blob", base + ".blob")
        meta = {"nbytes": int(arr.size), "blob_version": fold.BLOB_VERSION, "abi_version": hip.ABI_VERSION,
                "exponents": None if exponents is None else [int(e) for e in exponents],
                "blob_sha256": hashlib.sha256(memoryview(np.ascontiguousarray(arr))).hexdigest()}      # (the whole blob; on a miss only)
        with open(tmp + ".json", "w") as f:
            json.dump(meta, f)
        os.replace(tmp + ".json", base + ".json")
        return True
    except OSError:
        return False
