"""Resident server mode (reference: src/util/gpuserver.cpp): `mmgpu_server` keeps databases resident,
the patched `mmseqs` attaches through libmmgpu_client.so (LD_PRELOAD) instead of opening the device itself.  Result DBs
must equal the stock binary's byte for byte, and a second run against the same database must not upload it again.

CPU: the server's device library is the CPU stand-in of the C-ABI (oracle/_build/emu, LD_PRELOAD into the server).
GPU: the server owns the real device."""
import ctypes
import os
import signal
import subprocess
import time

import pytest

from mmseqs2_amd import workloads as wl
from tests.test_mmseqs_dropin import STOCK, MMGPU, EMU, THREADS, run, same

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERVER = os.path.join(ROOT, "mmseqs2_amd", "lib", "mmgpu_server")
CLIENT = os.path.join(ROOT, "mmseqs2_amd", "lib", "libmmgpu_client.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(STOCK) and os.path.exists(MMGPU) and os.path.exists(SERVER) and os.path.exists(CLIENT)),
                                reason="needs the mmseqs binaries (integration/build_mmseqs.sh) and mmseqs2_amd/lib/mmgpu_server")


class Server:
    def __init__(self, sock, emulate):
        env = dict(os.environ)
        env.pop("LD_PRELOAD", None)
        if emulate:
            env["LD_PRELOAD"] = EMU
        self.sock = sock
        self.p = subprocess.Popen([SERVER, "--socket", sock], env=env, stderr=subprocess.PIPE, text=True)
        t0 = time.time()
        while not os.path.exists(sock):
            assert self.p.poll() is None, "mmgpu_server exited: " + self.p.stderr.read()
            assert time.time() - t0 < 120, "mmgpu_server did not come up"
            time.sleep(0.05)

    def stats(self):
        L = ctypes.CDLL(CLIENT)
        os.environ["MMGPU_SERVER_SOCKET"] = self.sock
        ctx = ctypes.c_void_p()
        assert L.mmgpu_init(ctypes.byref(ctx), 0) == 0
        out = (ctypes.c_uint64 * 6)()
        assert L.mmgpu_client_server_stats(ctx, out) == 0
        L.mmgpu_destroy.argtypes = [ctypes.c_void_p]
        L.mmgpu_destroy(ctx)
        return dict(zip(["requests", "clients", "target_uploads", "index_uploads", "pf_batches", "sw_batches"], [int(x) for x in out]))

    def stop(self):
        self.p.send_signal(signal.SIGTERM)
        try:
            self.p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            self.p.kill()
        return self.p.stderr.read()


def _pipeline(w, emulate):
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=60, members=20, n_queries=40, seed=5)
    wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
    wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
    run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"], w)
    run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"], w)
    run(STOCK, ["prefilter", "q", "t", "pref_s", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
    run(STOCK, ["align", "q", "t", "pref_s", "aln_s", "-a", "--threads", THREADS, "-v", "2"], w)
    srv = Server(os.path.join(w, "mmgpu.sock"), emulate)
    try:
        env = {"LD_PRELOAD": CLIENT, "MMGPU_SERVER_SOCKET": srv.sock}
        for rnd in range(2):
            log = run(MMGPU, ["prefilter", "q", "t", "pref_g%d" % rnd, "-s", "5.7", "--threads", THREADS, "-v", "3"], w, extra_env=env)
            assert "mmgpu_server" in log and "using the CPU path" not in log, log[-2000:]
            same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g%d" % rnd))
            log = run(MMGPU, ["align", "q", "t", "pref_s", "aln_g%d" % rnd, "-a", "--threads", THREADS, "-v", "3"], w, extra_env=env)
            assert "mmgpu_server" in log and "using the CPU path" not in log, log[-2000:]
            same(os.path.join(w, "aln_s"), os.path.join(w, "aln_g%d" % rnd))
        st = srv.stats()
        # two modules x two rounds = four clients (+ this one); the prefilter's masked lookup and the aligner's plain
        # sequences are two resident databases, each uploaded once, the index once
        assert st["target_uploads"] == 2 and st["index_uploads"] == 1, st
        assert st["pf_batches"] >= 2 and st["sw_batches"] >= 2 and st["clients"] == 5, st
    finally:
        err = srv.stop()
    assert "target uploads" in err, err
    assert not os.path.exists(srv.sock)


def test_server_mode_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    _pipeline(str(tmp_path), emulate=True)


def test_client_without_server_fails_loudly(tmp_path):
    """no server, no silent CPU computation: the binary exits with the client library's message"""
    w = str(tmp_path)
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=10, members=5, n_queries=5, seed=1)
    wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
    run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"], w)
    env = dict(os.environ, LD_PRELOAD=CLIENT, MMGPU_SERVER_SOCKET=os.path.join(w, "nobody.sock"))
    r = subprocess.run([MMGPU, "prefilter", "q", "q", "pref", "--threads", "2", "-v", "3"], cwd=w, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "cannot connect to mmgpu_server" in r.stdout, r.stdout[-1500:]


@pytest.mark.gpu
def test_server_mode_on_device(tmp_path):
    _pipeline(str(tmp_path), emulate=False)
