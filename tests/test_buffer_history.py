"""integration/MMGpuBufferHistory.h - what the nucleotide alignment hook uses to know which letter the reference finds one
residue past the end of its per-thread buffers - against real buffers filled by memcpy (tests/buffer_history_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_history_equals_a_real_buffer(tmp_path):
    exe = str(tmp_path / "buffer_history_check")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-Wall", "-I" + os.path.join(ROOT, "integration"),
                           os.path.join(ROOT, "tests", "buffer_history_check.cpp"), "-o", exe])
    for seed in (1, 2, 3):
        out = subprocess.run([exe, str(seed)], stdout=subprocess.PIPE, text=True)
        assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout
