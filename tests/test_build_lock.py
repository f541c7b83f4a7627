"""dimx.build: concurrent builders (every rank of a torchrun job imports the package at once) are serialised by a file
lock, objects are rebuilt by content hash (not mtime), and a stale library without a compiler is refused."""
import os
import shutil
import stat
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dyadic-interaction-modeling_amd")

FAKE_HIPCC = textwrap.dedent("""\
    #!/usr/bin/env python3
    import os, sys, time
    log = os.environ["FAKE_HIPCC_LOG"]
    out = sys.argv[sys.argv.index("-o") + 1]
    with open(log, "a") as fh:
        fh.write("start %d %s\\n" % (os.getpid(), os.path.basename(out)))
    time.sleep(0.15)
    with open(out, "w") as fh:
        fh.write("obj built by %d\\n" % os.getpid())
    with open(log, "a") as fh:
        fh.write("end %d %s\\n" % (os.getpid(), os.path.basename(out)))
""")


@pytest.fixture()
def fake_tree(tmp_path):
    pkg = tmp_path / "pkg"
    (pkg / "csrc").mkdir(parents=True)
    (tmp_path / "include").mkdir()
    shutil.copy(os.path.join(PKG, "build.py"), pkg / "build.py")
    sys.path.insert(0, str(pkg))
    import importlib.util
    spec = importlib.util.spec_from_file_location("fake_build_probe", str(pkg / "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.path.pop(0)
    for src in mod.SOURCES:
        (pkg / "csrc" / src).write_text("// %s\n" % src)
    for hdr in mod.HEADERS:
        path = (pkg / "csrc" / hdr).resolve()
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text("// header\n")
    hipcc = tmp_path / "fake_hipcc"
    hipcc.write_text(FAKE_HIPCC)
    hipcc.chmod(hipcc.stat().st_mode | stat.S_IEXEC)
    env = dict(os.environ, HIPCC=str(hipcc), FAKE_HIPCC_LOG=str(tmp_path / "log.txt"))
    return pkg, env, tmp_path / "log.txt", len(mod.SOURCES)


def _builder(pkg, env, code="import build; build.build()"):
    return subprocess.Popen([sys.executable, "-c", code], cwd=str(pkg), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def test_concurrent_builds_are_serialised_and_done_once(fake_tree):
    pkg, env, log, nsrc = fake_tree
    procs = [_builder(pkg, env) for _ in range(3)]
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()
    lines = log.read_text().splitlines()
    starts = [ln for ln in lines if ln.startswith("start")]
    # one process compiled every source once and linked once; the others found a current library after the lock
    assert len(starts) == nsrc + 1, lines
    # the temporary output names carry the BUILDER's pid
    assert len({ln.split()[2].rsplit(".", 1)[1] for ln in starts}) == 1, "two processes ran the compiler: %s" % lines
    assert (pkg / "libdimx_hip.so").exists() and not list(pkg.glob("libdimx_hip.so.tmp.*"))


def test_rebuild_is_by_content_not_mtime(fake_tree):
    pkg, env, log, nsrc = fake_tree
    p = _builder(pkg, env)
    p.communicate(timeout=120)
    assert p.returncode == 0
    n0 = len([ln for ln in log.read_text().splitlines() if ln.startswith("start")])
    # a changed source with an OLD timestamp (what a copied tree looks like) must still be recompiled, and only it
    src = pkg / "csrc" / "vq.hip"
    src.write_text("// vq.hip changed\n")
    os.utime(src, (1, 1))
    code = "import build; assert build.stale(); build.build(); assert not build.stale()"
    p = _builder(pkg, env, code)
    out, err = p.communicate(timeout=120)
    assert p.returncode == 0, err.decode()
    new = [ln for ln in log.read_text().splitlines() if ln.startswith("start")][n0:]
    assert sorted(ln.split()[2].rsplit(".tmp.", 1)[0] for ln in new) == ["libdimx_hip.so", "vq.o"], new


def test_stale_library_without_compiler_is_refused(monkeypatch):
    import dimx  # noqa: F401
    from dimx import build as B
    from dimx import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(B, "stale", lambda: True)
    monkeypatch.setattr(B, "have_hipcc", lambda: False)
    monkeypatch.delenv("DIMX_ALLOW_STALE_LIB", raising=False)
    monkeypatch.delenv("DIMX_LIB", raising=False)
    with pytest.raises(L.DimxError, match="not built from the sources"):
        L.load()
    monkeypatch.setattr(L, "_lib", None)
