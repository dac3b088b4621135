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


def test_server_leaves_on_sigterm_and_on_idle(tmp_path):
    import signal

    env = dict(os.environ)
    env.pop("INQ_SERVER", None)
    for how in ("term", "idle"):
        sock = str(tmp_path / f"{how}.sock")
        args = [call.CLI_PATH, "serve", "--socket", sock] + (["--idle-exit", "0.5"] if how == "idle" else [])
        server = subprocess.Popen(args, env=env, stderr=subprocess.PIPE, text=True)
        try:
            for _ in range(200):
                if os.path.exists(sock):
                    break
                time.sleep(0.05)
            assert os.path.exists(sock)
            if how == "term":
                server.send_signal(signal.SIGTERM)
            assert server.wait(timeout=30) == 0
            assert "leaving after 0 calls" in server.stderr.read() and not os.path.exists(sock)
        finally:
            if server.poll() is None:
                server.kill()
                server.wait(timeout=30)


def test_malformed_requests_are_refused_and_the_server_stays(tmp_path):
    """Garbage on the socket (no descriptor, wrong magic, a body shorter than announced) gets an error answer or a closed
    connection; the next proper request is served."""
    import socket
    import struct

    bam, bed, loci, recs = _make_case(tmp_path, 6, n_loci=5)
    sock = str(tmp_path / "m.sock")
    env = dict(os.environ)
    env.pop("INQ_SERVER", None)
    server = subprocess.Popen([call.CLI_PATH, "serve", "--socket", sock, "--idle-exit", "60"], env=env, stderr=subprocess.PIPE, text=True)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        for payload in (b"", b"abc", struct.pack("<II", 0x12345678, 0), struct.pack("<II", 0x31514E49, 16) + b"\x00" * 4,
                        struct.pack("<II", 0x31514E49, 0xFFFFFFFF)):
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            c.settimeout(10)
            c.connect(sock)
            if payload:
                c.sendall(payload)
            c.shutdown(socket.SHUT_WR)
            got = b""
            try:
                while True:
                    chunk = c.recv(4096)
                    if not chunk:
                        break
                    got += chunk
            except (ConnectionResetError, socket.timeout):
                pass
            c.close()
            if len(got) >= 4:
                assert struct.unpack("<i", got[:4])[0] != 0
        if not _have_gpu():
            served = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed], capture_output=True, text=True, env=dict(env, INQ_SERVER=sock))
            assert served.returncode == 1 and "no CPU fallback" in served.stderr
        assert subprocess.run([call.CLI_PATH, "serve", "--socket", sock, "--quit"], env=env).returncode == 0
        assert server.wait(timeout=30) == 0
    finally:
        if server.poll() is None:
            server.kill()
            server.wait(timeout=30)


@pytest.mark.skipif(_have_gpu(), reason="the GPU twin is in tests/test_gpu_end_to_end.py")
def test_auto_server_is_started_once_and_found_again(tmp_path):
    """INQ_SERVER=auto: the first call starts this user's server for the device (detached, idle exit), the second finds it; the
    call's own status and message come back either way."""
    bam, bed, loci, recs = _make_case(tmp_path, 7, n_loci=6)
    env = dict(os.environ, INQ_SERVER="auto", XDG_RUNTIME_DIR=str(tmp_path), INQ_SERVER_IDLE="20")
    sock = tmp_path / f"inquistr-{os.getuid()}-dev0.sock"
    try:
        for _ in range(2):
            r = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed, "-t", "2"], capture_output=True, text=True, env=env, timeout=60)
            assert r.returncode == 1 and "no CPU fallback" in r.stderr and r.stdout == ""
            assert sock.exists()
    finally:
        q = subprocess.run([call.CLI_PATH, "serve", "--socket", str(sock), "--quit"], capture_output=True, text=True, timeout=60)
    assert q.returncode == 0
    for _ in range(100):
        if not sock.exists():
            break
        time.sleep(0.05)
    assert not sock.exists()


def test_auto_server_is_not_started_from_a_process_that_maps_a_gpu_runtime(tmp_path):
    """Starting the server is fork + exec, which is only safe from a process that has not initialised the GPU.  A tool library
    preloaded into `inquistr call` (a profiler's is) may have: with a GPU runtime among the process's mappings the auto-start is
    refused - no server, no socket - and the call runs in the caller's own process, with its own status and message."""
    hip = "/opt/rocm/lib/libamdhip64.so"
    if not os.path.exists(hip):
        pytest.skip("no HIP runtime library to preload")
    bam, bed, loci, recs = _make_case(tmp_path, 11, n_loci=6)
    env = dict(os.environ, INQ_SERVER="auto", XDG_RUNTIME_DIR=str(tmp_path), INQ_SERVER_IDLE="5", LD_PRELOAD=hip)
    sock = tmp_path / f"inquistr-{os.getuid()}-dev0.sock"
    r = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed, "-t", "2"], capture_output=True, text=True, env=env, timeout=60)
    try:
        assert not sock.exists(), "a server was started from a process with the HIP runtime mapped"
        # the call itself ran here: on a box without a GPU that is the loud failure, on a GPU box the rows
        assert r.returncode in (0, 1)
        if r.returncode == 1:
            assert "no CPU fallback" in r.stderr and r.stdout == ""
        else:
            assert r.stdout.startswith("chromosome\tbegin\tend\t")
    finally:
        if sock.exists():
            subprocess.run([call.CLI_PATH, "serve", "--socket", str(sock), "--quit"], capture_output=True, text=True, timeout=60)
