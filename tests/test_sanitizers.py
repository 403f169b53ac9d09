"""AddressSanitizer + UndefinedBehaviorSanitizer builds of everything on the CPU side (the reference's CI runs its Rust
under the compiler's checks; GPU sanitizers are not available on this pool): the oracle, the host's header-only code, the span solver of the
raster path compiled for the host, and the image-order walk model.  Any report aborts the run (-fno-sanitize-recover)."""
import os
import subprocess

from conftest import ROOT

SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, env=ENV, timeout=900, **kw)
    assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_the_oracle_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    run(["gcc", "-std=c11", *SAN, "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tests", "cpp", "oracle_san_main.c"), "-o", exe, "-lm"])
    assert "oracle under ASan + UBSan: ok" in run([exe])


def test_the_host_headers_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_san")
    run(["g++", "-std=c++17", *SAN, os.path.join(ROOT, "tests", "cpp", "host_san_main.cpp"), "-o", exe, "-lz", "-lpthread"])
    assert "host headers under ASan + UBSan: ok" in run([exe, str(tmp_path)])


def test_the_span_solver_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "raster_fuzz_san")
    run(["g++", "-std=c++17", *SAN, "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "raster_fuzz.cpp"), "-o", exe])
    out = run([exe, "20000", "20250816"])
    assert "failures 0" in [l for l in out.splitlines() if l.startswith("triangles")][0]
