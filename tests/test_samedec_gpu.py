"""samedec_gpu, the command-line decoder, against the reference's
integration fixtures: each sample recording must print exactly the lines of sample/<name>.txt
-- the decoded headers from the decoder, the "+OK" lines from the alert command it spawns
(sample/test.sh:22-33, 48-62).  The header of long_message is only produced by the end-of-file
flush (crates/samedec/src/app.rs:118)."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

SAMPLES = ["npt.22050.s16le", "two_and_two.22050.s16le", "long_message.22050.s16le"]
CHECK = os.path.join(ROOT, "tests", "helpers", "samedec_child_check.py")


@pytest.fixture(scope="module")
def exe():
    from sameold_amd import build as sbuild
    sbuild.build()
    assert os.path.exists(sbuild.SAMEDEC)
    return sbuild.SAMEDEC


def run(exe, args, stdin=None):
    p = subprocess.run([exe] + args, stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


@pytest.mark.parametrize("name", SAMPLES)
def test_sample_with_child(exe, name, tmp_path):
    counts = tmp_path / "counts"
    rc, out, err = run(exe, ["--rate", "22050", "--file", os.path.join(GOLDEN, name + ".bin"), "--",
                             sys.executable, CHECK, name, str(counts)])
    assert rc == 0, err
    with open(os.path.join(GOLDEN, name + ".txt")) as f:
        assert out == f.read(), err
    # the child got audio: from the sample after the header event to the next message or EOF
    got = [int(x) for x in counts.read_text().split()]
    size = os.path.getsize(os.path.join(GOLDEN, name + ".bin"))
    assert len(got) == 1 and got[0] % 2 == 0 and got[0] <= size
    if name == "two_and_two.22050.s16le":
        assert got[0] > 0


@pytest.mark.parametrize("name", SAMPLES)
def test_sample_without_child_and_stdin(exe, name):
    with open(os.path.join(GOLDEN, name + ".txt")) as f:
        want = "".join(line for line in f if not line.startswith("+OK"))
    rc, out, err = run(exe, ["-r", "22050", "--file", os.path.join(GOLDEN, name + ".bin")])
    assert (rc, out) == (0, want), err
    with open(os.path.join(GOLDEN, name + ".bin"), "rb") as f:
        rc, out, err = run(exe, ["--rate=22050"], stdin=f)
    assert (rc, out) == (0, want), err
    rc, out, err = run(exe, ["--quiet", "--file", os.path.join(GOLDEN, name + ".bin")])
    assert (rc, out) == (0, "")


def test_cli_errors_and_demo(exe, tmp_path):
    rc, out, err = run(exe, ["--preamble-max-errors", "9"])
    assert rc == 2 and "preamble-max-errors" in err
    rc, out, err = run(exe, ["--file", str(tmp_path / "missing.bin")])
    assert rc == 1 and "Unable to open --file" in err
    rc, out, err = run(exe, ["--bogus"])
    assert rc == 2
    # a command that cannot be started is reported and decoding goes on (app.rs:154-163)
    rc, out, err = run(exe, ["--file", os.path.join(GOLDEN, "npt.22050.s16le.bin"), "--", "/nonexistent/alert-cmd"])
    assert rc == 0 and out == "ZCZC-PEP-NPT-000000+0030-2771820-TEST    -\n" and "unable to spawn child process" in err
    # --demo: a DMO header stamped with the current UTC time, then three end-of-message lines
    silence = tmp_path / "silence.bin"
    silence.write_bytes(b"\0\0" * 22050)
    rc, out, err = run(exe, ["--demo", "--file", str(silence)])
    lines = out.splitlines()
    assert rc == 0 and len(lines) == 4 and lines[1:] == ["NNNN"] * 3
    assert lines[0].startswith("ZCZC-EAS-DMO-999000+0015-") and lines[0].endswith("-N0 CALL -")
