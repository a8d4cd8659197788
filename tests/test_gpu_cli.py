"""The reference's own command line tool on libsla_hip.so (run with -m gpu on an MI355X).

oracle/_ref/sla_cli_on_hip is the UNMODIFIED reference main.c / wav.c / command_line_parser.c compiled against this
repo's public headers and linked to sla_amd/libsla_hip.so (recipe: oracle/Makefile; built in the container that has
the reference sources, the binary travels like the other built files).  It must reproduce the known answers the
reference CLI gives for its own test file (md5 of `sla -e -m {0,2,4} test/a.wav`), and decode them
back to the very bytes of a.wav -- through SLADecoder_DecodeWhole and through the streaming decoder.
Nothing here reads /root/reference."""
import hashlib
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "oracle", "_ref", "sla_cli_on_hip")
A_WAV = os.path.join(ROOT, "tests", "golden", "a.wav")

KNOWN = {0: ("48c60a59f94f70303be8207d7ea9dc03", 55982), 2: ("9739dfd1acd3eeaec7a3f4345ee8c4a4", 49454),
         4: ("9ad138cb6ad58ab8b074eae1132b3c28", 49450)}


def run(*args):
    return subprocess.run([CLI] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/sla_cli_on_hip not built (needs the reference sources at build time)")
    return CLI


@pytest.mark.parametrize("mode", sorted(KNOWN))
def test_reference_cli_encodes_the_known_answers_and_decodes_them_back(cli, tmp_path, mode):
    sla = str(tmp_path / ("a_m%d.sla" % mode))
    r = run("-e", "-m", str(mode), A_WAV, sla)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    data = open(sla, "rb").read()
    assert (hashlib.md5(data).hexdigest(), len(data)) == KNOWN[mode]
    want = open(A_WAV, "rb").read()
    for extra in ([], ["-s"], ["-c", "no"]):
        wav = str(tmp_path / ("back_m%d_%s.wav" % (mode, "".join(extra).strip("-") or "whole")))
        r = run("-d", *extra, sla, wav)
        assert r.returncode == 0, r.stdout.decode(errors="replace")
        assert open(wav, "rb").read() == want, extra


CLI_PRED = os.path.join(ROOT, "oracle", "_ref", "sla_cli_ref_codec_on_hip_predictor")


@pytest.mark.parametrize("mode", sorted(KNOWN))
def test_reference_encoder_and_decoder_on_the_hip_predictor_api(tmp_path, mode):
    """BASELINE's north star, literally: the reference's own SLAEncoder.c / SLADecoder.c (and coder, bit stream, CLI) with
    src/SLAPredictor.c replaced by libsla_hip.so's per-call API -- one block per call -- reproduce the known answers"""
    if not os.path.exists(CLI_PRED):
        pytest.skip("oracle/_ref/sla_cli_ref_codec_on_hip_predictor not built (needs the reference sources at build time)")
    sla = str(tmp_path / ("a_m%d.sla" % mode))
    r = subprocess.run([CLI_PRED, "-e", "-m", str(mode), A_WAV, sla], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    data = open(sla, "rb").read()
    assert (hashlib.md5(data).hexdigest(), len(data)) == KNOWN[mode]
    wav = str(tmp_path / ("back_m%d.wav" % mode))
    r = subprocess.run([CLI_PRED, "-d", sla, wav], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert open(wav, "rb").read() == open(A_WAV, "rb").read()


def test_reference_cli_reports_a_damaged_file(cli, tmp_path):
    sla = str(tmp_path / "a.sla")
    assert run("-e", A_WAV, sla).returncode == 0
    data = bytearray(open(sla, "rb").read())
    data[5000] ^= 0x40
    bad = str(tmp_path / "bad.sla")
    open(bad, "wb").write(bytes(data))
    r = run("-d", bad, str(tmp_path / "bad.wav"))
    assert r.returncode != 0 and b"failed to decode" in r.stdout
