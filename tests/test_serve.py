"""`inquistr serve` / INQ_SERVER without a GPU: the request travels, the answer (here: the loud failure of a box without a gfx950
device) comes back as the call's own exit status and message, nothing is written; no server = the call runs by itself."""
import os
import subprocess
import time

import pytest

from inquistr_amd import call
from tests.test_host_frontend import _make_case


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="the GPU twin is tests/test_gpu_end_to_end.py::test_call_through_a_resident_server_equals_call")
def test_served_call_reports_what_the_call_itself_reports(tmp_path):
    bam, bed, loci, recs = _make_case(tmp_path, 5, n_loci=8)
    sock = str(tmp_path / "s.sock")
    env = dict(os.environ)
    env.pop("INQ_SERVER", None)
    direct = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed, "-t", "2"], capture_output=True, text=True, env=env)
    assert direct.returncode == 1 and "no CPU fallback" in direct.stderr and direct.stdout == ""
    # no server behind the address: as without INQ_SERVER
    lonely = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed, "-t", "2"], capture_output=True, text=True, env=dict(env, INQ_SERVER=sock))
    assert (lonely.returncode, lonely.stdout, lonely.stderr) == (direct.returncode, direct.stdout, direct.stderr)
    server = subprocess.Popen([call.CLI_PATH, "serve", "--socket", sock, "--idle-exit", "60"], env=env, stderr=subprocess.PIPE, text=True)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        assert os.path.exists(sock) and (os.stat(sock).st_mode & 0o077) == 0
        # a second server may not take a live socket
        second = subprocess.run([call.CLI_PATH, "serve", "--socket", sock], capture_output=True, text=True, env=env)
        assert second.returncode == 1 and "in use" in second.stderr
        for _ in range(3):  # relative paths are the caller's
            served = subprocess.run([call.CLI_PATH, "call", os.path.basename(bam), "-R", os.path.basename(bed), "-t", "2"], capture_output=True,
                                    text=True, env=dict(env, INQ_SERVER=sock), cwd=os.path.dirname(bam))
            assert (served.returncode, served.stdout, served.stderr) == (direct.returncode, direct.stdout, direct.stderr)
        # usage errors never reach the server
        bad = subprocess.run([call.CLI_PATH, "call", bam, "--nope"], capture_output=True, text=True, env=dict(env, INQ_SERVER=sock))
        assert bad.returncode == 2
        assert subprocess.run([call.CLI_PATH, "serve", "--socket", sock, "--quit"], env=env).returncode == 0
        assert server.wait(timeout=30) == 0
        assert "leaving after 3 calls" in server.stderr.read() and not os.path.exists(sock)
    finally:
        if server.poll() is None:
            server.kill()
            server.wait(timeout=30)
