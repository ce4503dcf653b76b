"""Wire compatibility of the SocketMessenger with the reference's own, compiled from the reference sources where
they lie (oracle/Makefile -> oracle/_ref/ref_socket_party): three single-party processes on loopback, any mix of the
two builds, exchange six rounds of payloads (including empty ones) and every party checks every byte it received.
Same ports, handshake and framing as utils/SocketMessenger.cpp:34-164."""
import os
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
BIN = {"ref": os.path.join(ORACLE, "_ref", "ref_socket_party"), "own": os.path.join(ORACLE, "my_socket_party")}


def run_mix(mix):
    procs = [subprocess.Popen([BIN[kind], str(party)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for party, kind in enumerate(mix)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=60)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [p.returncode for p in procs], outs


@pytest.mark.skipif(not os.path.exists(BIN["own"]), reason="oracle/my_socket_party not built (make -C oracle)")
def test_own_parties_talk_to_each_other():
    codes, outs = run_mix(("own", "own", "own"))
    assert codes == [0, 0, 0], outs


@pytest.mark.skipif(not (os.path.exists(BIN["ref"]) and os.path.exists(BIN["own"])),
                    reason="the reference build exists only where /root/reference was present at build time")
@pytest.mark.parametrize("mix", [("ref", "own", "own"), ("own", "ref", "own"), ("own", "own", "ref"),
                                 ("ref", "ref", "own"), ("own", "ref", "ref"), ("ref", "ref", "ref")])
def test_mixed_reference_and_own_parties(mix):
    codes, outs = run_mix(mix)
    assert codes == [0, 0, 0], outs
