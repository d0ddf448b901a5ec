"""`load_denoiser` / `load_separator` console entry points (reference setup.py:47,49 point at a
`load_model:main` that is absent from the tree; the README says they download the trained models).
There is no network egress to rely on here, so these commands VERIFY a user-supplied checkpoint
instead: the bundle's data shard must match the size and sha256 recorded in the reference's git-LFS
pointer files, its index must match the variable inventory, and every tensor must load.
"""
import argparse
import hashlib
import os
import sys

from . import apply, spec, tfbundle, weights

# sha256 / size of `<bundle>.data-00000-of-00001`, from the LFS pointers shipped in the reference
# (N_HANS___Selective_Noise/trained_model/, N_HANS___Source_Separation/trained_model/)
BUNDLES = {
    spec.DENOISER: (apply.DENOISER_BUNDLE, 115999524,
                    "6bff37f367362cadf2433439d0096a1896461b954c9b2331552853c317e4f026"),
    spec.SEPARATOR: (apply.SEPARATOR_BUNDLE, 115999528,
                     "68ee8a6e4ae9948ed8d9188c93d8b7599e1dad1f51ad6b93d1cc11fcea6b90d9"),
}


def verify(kind, model_dir="./trained_model", check_hash=True):
    """Returns the loaded weights dict; raises with an actionable message otherwise."""
    name, size, sha = BUNDLES[kind]
    prefix = os.path.join(model_dir, name)
    data = tfbundle.data_path(prefix)
    if not os.path.exists(prefix + ".index"):
        raise FileNotFoundError("%s.index not found: copy the reference's trained_model/ directory here" % prefix)
    if not os.path.exists(data) or tfbundle.is_lfs_pointer(data):
        raise FileNotFoundError("%s is missing or still a git-LFS pointer: fetch the %d-byte blob "
                                "(`git lfs pull` in the N-HANS repository)" % (data, size))
    if os.path.getsize(data) != size:
        raise ValueError("%s has %d bytes, expected %d" % (data, os.path.getsize(data), size))
    if check_hash:
        h = hashlib.sha256()
        with open(data, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 22), b""):
                h.update(chunk)
        if h.hexdigest() != sha:
            raise ValueError("%s: sha256 %s does not match the reference's %s" % (data, h.hexdigest(), sha))
    # (--no-hash skips both integrity checks: the file hash and the per-tensor crc32c of the index)
    return weights.load_checkpoint(prefix, kind, verify_crc=check_hash)


def _main(kind, argv):
    p = argparse.ArgumentParser(prog="load_" + kind)
    p.add_argument("--model_dir", default="./trained_model")
    p.add_argument("--no-hash", action="store_true")
    a = p.parse_args(argv)
    w = verify(kind, a.model_dir, not a.no_hash)
    print("%s checkpoint OK: %d tensors, %d parameters" % (kind, len(w), sum(v.size for v in w.values())))


def main(argv=None):
    _main(spec.DENOISER, sys.argv[1:] if argv is None else argv)


def main_separator(argv=None):
    _main(spec.SEPARATOR, sys.argv[1:] if argv is None else argv)
