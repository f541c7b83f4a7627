"""Build libdimx_hip.so (hipcc, gfx950 only) in-tree next to the sources.

    python dyadic-interaction-modeling_amd/build.py [--force]

The shared library has no torch dependency; it is loaded with ctypes (dimx.lib).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdimx_hip.so")
SOURCES = ["gemm.hip", "gemm_x3.hip", "gemm256.hip", "norm.hip", "attention.hip", "attention_tr.hip", "mlp_fused.hip", "decode_attn.hip", "vq.hip", "elementwise.hip", "chain.hip", "model.hip",
           "train_kernels.hip", "train_attn.hip", "train.hip"]
HEADERS = ["common.hpp", "model.hpp", "train.hpp", os.path.join("..", "..", "include", "dimx.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=on"]
# per-source additions: beside MFMAs the SLP vectoriser's v_pk_add_f32 / v_pk_mul_f32 cost more than the two scalar instructions
# they replace (MI355X_MICROARCH.md, price of one filler beside MFMAs)
SRC_FLAGS = {"attention_tr.hip": ["-fno-slp-vectorize"]}
if os.environ.get("DIMX_TUNING"):   # also instantiate the measured-dead-end GEMM configurations (A/B runs; use --force)
    FLAGS.append("-DDIMX_GEMM_TUNING")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _file_hash(paths, extra=""):
    import hashlib
    hsh = hashlib.sha256(extra.encode())
    for f in paths:
        with open(f, "rb") as fh:
            hsh.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return hsh.hexdigest()


def _compile(src):
    """Compile one source unless its object was built from exactly these bytes (source + headers + flags): a content
    hash next to the object, not mtimes -- the copy of the tree that travels to a GPU box does not keep timestamps, and a
    stamp must never be written for an object that was not rebuilt from the hashed sources."""
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    flags = FLAGS + SRC_FLAGS.get(src, [])
    want = _file_hash(deps, " ".join(flags))
    stamp = obj + ".srchash"
    if os.path.exists(obj) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == want:
                return obj, False
    tmp = "%s.tmp.%d" % (obj, os.getpid())
    cmd = [_hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    os.replace(tmp, obj)
    with open(stamp, "w") as fh:
        fh.write(want + "\n")
    return obj, True


def have_hipcc():
    import shutil
    c = _hipcc()
    return (os.path.isabs(c) and os.path.exists(c)) or shutil.which(c) is not None


STAMP = LIB + ".srchash"   # sha256 of the sources / headers / flags the shipped library was built from


def _src_hash():
    import hashlib
    hsh = hashlib.sha256((" ".join(FLAGS) + repr(sorted(SRC_FLAGS.items()))).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            hsh.update(f.encode() + b"\0" + fh.read())
    return hsh.hexdigest()


def stale():
    """True when libdimx_hip.so is missing or was built from other sources than the ones in the tree (content hash,
    not mtime: the copy that travels to a GPU box does not keep timestamps)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != _src_hash()


def build(force=False):
    """Build (or complete) the in-tree library.  Serialised across processes with an exclusive lock on csrc/_obj/.lock:
    under torchrun every rank imports the package at once, and two hipcc runs into the same objects / the same .so could
    hand a rank a half-written library.  The link goes to a temporary file that is renamed into place."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():
                return LIB   # another process finished the same build while this one waited for the lock
            if force:
                for f in os.listdir(OBJ):
                    if f != ".lock" and os.path.isfile(os.path.join(OBJ, f)):
                        os.remove(os.path.join(OBJ, f))
            with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
                res = list(ex.map(_compile, SOURCES))
            objs = [o for o, _ in res]
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
            os.replace(tmp, LIB)
            with open(STAMP, "w") as fh:
                fh.write(_src_hash() + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
