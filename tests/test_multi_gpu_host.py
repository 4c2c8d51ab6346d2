"""The C++ hosts of the multi-GPU rows (BASELINE configs[3] / [4]) without a GPU: argument handling of examples/multi_gpu_host.cpp (one worker per device,
independent frames) and the whole logic of examples/stream_launcher.cpp (N encodes, one per device, aggregate fps) driven with stand-in commands."""
import json
import os
import subprocess

import pytest

import x265hip
from x265hip_pkg.binding import HERE, lib_path


def exe(name):
    x265hip.build_libraries()
    p = os.path.join(HERE, "build", name)
    assert os.path.exists(p), "%s is built by the package Makefile" % p
    return p


def run(args, **kw):
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    return subprocess.run(args, capture_output=True, text=True, env=env, timeout=120, **kw)


def test_launcher_starts_one_stream_per_device_and_adds_up_the_frames(tmp_path):
    cmd = ('echo "noise"; echo "{\\"dev\\": $X265TME_DEVICE, \\"other\\": $X265HIP_DEVICE, \\"arg\\": \\"$0\\", \\"frames\\": 1$X265TME_DEVICE, \\"seconds\\": 2.0, \\"fps\\": 5.5}"'
           ' | tee %s/out_{dev}_{k}.txt' % tmp_path)
    r = run([exe("stream_launcher"), "--devices", "0,2-3", "--", "sh", "-c", cmd, "stream{k}on{dev}"])
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["streams"] == 3 and d["frames"] == 10 + 12 + 13
    assert [s["device"] for s in d["per_stream"]] == [0, 2, 3] and all(s["fps"] == 5.5 for s in d["per_stream"])
    assert abs(d["aggregate_fps"] - d["frames"] / d["wall_seconds"]) < 0.05 * d["aggregate_fps"] + 0.5
    for k, dev in enumerate((0, 2, 3)):     # {dev} / {k} substituted in the arguments, the device in both environment variables
        line = json.loads(open(tmp_path / ("out_%d_%d.txt" % (dev, k))).read().strip().splitlines()[-1])
        assert line["dev"] == dev and line["other"] == dev and line["arg"] == "stream%don%d" % (k, dev)


def test_launcher_takes_another_environment_name():
    r = run([exe("stream_launcher"), "--devices", "5", "--env", "MY_GPU", "--", "sh", "-c", 'echo "{\\"frames\\": $MY_GPU, \\"seconds\\": 1.0}"'])
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout)["frames"] == 5


def test_launcher_reports_a_failed_stream_and_bad_arguments():
    r = run([exe("stream_launcher"), "--devices", "0,1", "--", "sh", "-c", 'if [ $X265TME_DEVICE = 1 ]; then echo broken; exit 7; fi; echo "{\\"frames\\": 3, \\"seconds\\": 1.0}"'])
    assert r.returncode == 1 and "stream 1 (device 1) failed (status 7)" in r.stderr and "broken" in r.stderr
    r = run([exe("stream_launcher"), "--devices", "0", "--", "sh", "-c", "echo no json here"])
    assert r.returncode == 1 and "printed no" in r.stderr
    r = run([exe("stream_launcher"), "--devices", "0", "--", "/nonexistent/encoder"])
    assert r.returncode == 1 and "cannot run" in r.stderr
    for bad in (["--devices", "0,,x", "--", "true"], ["--devices", "3-1", "--", "true"], ["--devices", "0"], ["--", "true"], ["--frobnicate", "1", "--", "true"]):
        r = run([exe("stream_launcher")] + bad)
        assert r.returncode == 2, bad


def test_multi_gpu_host_rejects_bad_arguments():
    h = exe("multi_gpu_host")
    lib = lib_path(10)
    for bad in (["--devices", "0,x"], ["--devices", "-1"], [], ["--devices", "0", "--width", "100"], ["--devices", "0", "--steps", "0"], ["--devices", "0", "--frames", "two"],
                ["--devices", "0", "--nope", "1"], ["--devices"]):
        r = run([h, lib] + bad)
        assert r.returncode == 2, (bad, r.stderr)
    r = run([h, "/nonexistent/libx265hip.so", "--devices", "0"])
    assert r.returncode == 2 and "dlopen" in r.stderr


@pytest.mark.parametrize("procs", [False, True])
def test_multi_gpu_host_names_a_device_it_cannot_open(procs):
    """device 99 does not exist on any box (and without a GPU no device does): x265hip_ctx_create's X265HIP_EDEVICE reaches the exit code (3) and the message names the device --
    never a silent fallback to device 0 or to the CPU"""
    r = run([exe("multi_gpu_host"), lib_path(8), "--devices", "99", "--width", "128", "--height", "128", "--frames", "1", "--steps", "1"] + (["--procs"] if procs else []))
    assert r.returncode == 3, r.stderr
    assert "x265hip_ctx_create on device 99" in r.stderr
