"""host/input_guard.hpp on the CPU: a mapped input that is cut short while it is read must not kill the process with SIGBUS
(the reference's fread() reports a short read as "Read input file failed", /root/reference/src/gpu_compressor.cpp:146-150);
the GPU CLI's own test of the same thing is tests/test_cli_gpu.py::test_gpu_cli_survives_an_input_cut_short_under_its_mapping."""
import os
import signal
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reads_behind_a_cut_see_zeros_and_a_mark_instead_of_sigbus(tmp_path):
    exe = str(tmp_path / "input_guard_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "gpuar_amd", "csrc", "host"),
                           "-o", exe, os.path.join(ROOT, "tests", "input_guard_test.cpp"), "-lpthread"])
    r = subprocess.run([exe, str(tmp_path / "mapped.dat")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "input guard ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    # a SIGBUS in a mapping nobody watches is none of the guard's business: the default action still applies
    r = subprocess.run([exe, str(tmp_path / "foreign.dat"), "foreign"], capture_output=True, text=True, timeout=60)
    assert r.returncode == -signal.SIGBUS, (r.returncode, r.stdout)
