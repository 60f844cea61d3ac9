"""examples/abi_demo.c: the C ABI used from plain C, examples/host_demo.cpp: the C++ host mirror
of the dusk interface (include/plonk_mi355x.hpp), and examples/bench_driver.cpp: the C++ bench driver over the C ABI
 -- no Python, no torch in the process.
CPU: they compile, link against the built library and fail loudly without a device.
GPU: they run their NTT / MSM / polynomial checks."""
import os
import subprocess

import pytest

from conftest import ROOT

LIBDIR = os.path.join(ROOT, "plonk-prototype_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "abi_demo")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "abi_demo.c"), "-I", os.path.join(ROOT, "include"),
           "-L", LIBDIR, "-lplonk_mi355x", f"-Wl,-rpath,{LIBDIR}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_demo_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the gpu-marked test runs the demo")
    exe = _build(tmp_path)
    r = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "pm_init" in r.stderr and "-5" in r.stderr           # PM_ERR_NO_DEVICE, no silent CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [3, 12, 18])
def test_c_demo_runs_on_the_gpu(tmp_path, log_n):
    exe = _build(tmp_path)
    r = subprocess.run([exe, str(log_n)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "abi_demo OK" in r.stdout


def _build_cpp(tmp_path):
    exe = str(tmp_path / "host_demo")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "host_demo.cpp"),
           "-I", os.path.join(ROOT, "include"), "-L", LIBDIR, "-lplonk_mi355x", f"-Wl,-rpath,{LIBDIR}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_cpp_host_mirror_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the gpu-marked test runs the demo")
    r = subprocess.run([_build_cpp(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Error -5" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_runs_on_the_gpu(tmp_path):
    r = subprocess.run([_build_cpp(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "host_demo OK" in r.stdout


def _build_driver(tmp_path):
    exe = str(tmp_path / "bench_driver")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "bench_driver.cpp"),
           "-I", os.path.join(ROOT, "include"), "-L", LIBDIR, "-lplonk_mi355x", f"-Wl,-rpath,{LIBDIR}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_cpp_bench_driver_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the gpu-marked test runs the driver")
    r = subprocess.run([_build_driver(tmp_path), "8"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "pm_init" in r.stderr and "-5" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [6, 14])
def test_cpp_bench_driver_runs_on_the_gpu(tmp_path, log_n):
    """Both checks inside the driver (round trip byte-equal; MSM == (sum s_i g^i) G over an SRS stand-in made on the device)
    pass, and its one JSON line parses."""
    import json
    r = subprocess.run([_build_driver(tmp_path), str(log_n), "5", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "bench_driver OK" in r.stdout
    line = json.loads(r.stdout.splitlines()[0])
    assert line["log_n"] == log_n and line["ntt_butterflies_per_s"] > 0 and line["msm_scalar_muls_per_s"] > 0
