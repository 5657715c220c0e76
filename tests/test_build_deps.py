"""csrc/build.sh is incremental over hipcc's own depfiles: editing ANY header rebuilds exactly the objects that include
it (the old script only looked at the source, common.h and the public header, so an edit of
rowpiece.h could leave stale objects in the .so that the GPU tests then blessed).  CPU only: the real tree's depfiles are
read with a recording stand-in for the compiler; one tiny file goes through the real hipcc."""
import os
import shutil
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "biomedkg_amd", "csrc")


def _run(cwd, **env):
    out = subprocess.run(["bash", os.path.join(cwd, "build.sh")], cwd=cwd, env=dict(os.environ, **env), check=True,
                         capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("rebuilt:")][0]
    names = line[len("rebuilt:"):].split()
    return [] if names == ["(nothing)"] else sorted(names)


def _includers(header):
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") and f'#include "{header}"' in open(os.path.join(CSRC, f)).read():
            out.append(f[:-4])
    return out


@pytest.fixture()
def tree(tmp_path):
    """A copy of csrc/ (+ the public header at its relative place) with the objects and depfiles of the last real build,
    and a stand-in compiler that records what it is asked to build and touches the outputs."""
    if not os.path.exists(os.path.join(CSRC, "obj", "spmm.d")):
        pytest.skip("no depfiles yet: run __graft_entry__.build() first")
    dst = tmp_path / "biomedkg_amd" / "csrc"
    shutil.copytree(CSRC, dst, ignore=shutil.ignore_patterns("*.so"))
    os.makedirs(tmp_path / "include")
    shutil.copy(os.path.join(ROOT, "include", "biomedkg_hip.h"), tmp_path / "include" / "biomedkg_hip.h")
    fake = tmp_path / "fakecc"
    fake.write_text("#!/bin/bash\n"
                    "# records the compile, leaves the depfile as it is, refreshes the object\n"
                    "out=; while [ $# -gt 0 ]; do case \"$1\" in -o) out=$2; shift;; esac; shift; done\n"
                    "touch \"$out\"\n")
    fake.chmod(0o755)
    now = time.time()
    for d, _, files in os.walk(tmp_path):              # sources and headers older than every object
        for f in files:
            p = os.path.join(d, f)
            os.utime(p, (now - 100, now - 100) if not f.endswith((".o", ".d", ".flags")) else (now - 50, now - 50))
    return str(dst), str(fake)


def test_nothing_is_rebuilt_when_nothing_changed(tree):
    cwd, fake = tree
    assert _run(cwd, HIPCC=fake) == []


@pytest.mark.parametrize("header", ["rowpiece.h", "gemm_epilogue.h", "gemm_tile_bf16.h", "common.h"])
def test_a_header_edit_rebuilds_every_includer(tree, header):
    cwd, fake = tree
    os.utime(os.path.join(cwd, header))
    rebuilt = _run(cwd, HIPCC=fake)
    direct = _includers(header)
    assert direct, header
    assert set(direct) <= set(rebuilt), (header, direct, rebuilt)
    if header == "rowpiece.h":
        assert "spmm" in rebuilt and "kge_fused" in rebuilt
    if header != "common.h":
        assert len(rebuilt) < len([f for f in os.listdir(cwd) if f.endswith(".hip")])      # ... and nothing else
    assert _run(cwd, HIPCC=fake) == []


def test_the_public_header_and_flags_count_too(tree):
    cwd, fake = tree
    os.utime(os.path.join(cwd, "..", "..", "include", "biomedkg_hip.h"))
    assert "common" in _run(cwd, HIPCC=fake)
    every = sorted(f[:-4] for f in os.listdir(cwd) if f.endswith(".hip"))
    # the recorded flags differ (the stand-in does not rewrite them -- only a finished real compile does)
    assert _run(cwd, HIPCC=fake, EXTRA_FLAGS="-DSOMETHING=1") == every


def test_real_hipcc_writes_a_depfile_that_names_nested_headers(tmp_path):
    """One tiny translation unit through the real compiler: a.hip -> h1.h -> h2.h."""
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    shutil.copy(os.path.join(CSRC, "build.sh"), tmp_path / "build.sh")
    (tmp_path / "h2.h").write_text("#define TWO 2\n")
    (tmp_path / "h1.h").write_text('#include "h2.h"\n')
    (tmp_path / "a.hip").write_text('#include "h1.h"\nextern "C" int two() { return TWO; }\n')
    assert _run(str(tmp_path)) == ["a"]
    assert _run(str(tmp_path)) == []
    later = time.time() + 5
    os.utime(tmp_path / "h2.h", (later, later))
    assert _run(str(tmp_path)) == ["a"]
