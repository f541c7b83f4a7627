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
SOURCES = ["gemm.hip", "gemm256.hip", "norm.hip", "attention.hip", "decode_attn.hip", "vq.hip", "elementwise.hip", "chain.hip", "model.hip"]
HEADERS = ["common.hpp", "model.hpp", os.path.join("..", "..", "include", "dimx.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=on"]
if os.environ.get("DIMX_TUNING"):   # also instantiate the measured-dead-end GEMM configurations (A/B runs; use --force)
    FLAGS.append("-DDIMX_GEMM_TUNING")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if not _stale(obj, deps):
        return obj, False
    cmd = [_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def have_hipcc():
    import shutil
    c = _hipcc()
    return (os.path.isabs(c) and os.path.exists(c)) or shutil.which(c) is not None


STAMP = LIB + ".srchash"   # sha256 of the sources / headers / flags the shipped library was built from


def _src_hash():
    import hashlib
    hsh = hashlib.sha256(" ".join(FLAGS).encode())
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
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            if os.path.isfile(os.path.join(OBJ, f)):
                os.remove(os.path.join(OBJ, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(STAMP, "w") as fh:
        fh.write(_src_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
