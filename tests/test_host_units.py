"""Host-only C++ units of the product built with g++ and run on the CPU (no HIP involved)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parallel_stable_order_equals_std_stable_sort(tmp_path):
    exe = str(tmp_path / "test_hostsort")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(HERE, "cpp", "test_hostsort.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_host_pool_regions(tmp_path):
    """l3d::HostPool (persistent workers of the finishing stages): every index once per region, nested and concurrent regions,
    fresh workers in a forked child."""
    exe = str(tmp_path / "test_hostpool")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(HERE, "cpp", "test_hostpool.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
