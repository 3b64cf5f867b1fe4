"""Race detection for the host side of the boundary: the string layer's interning, combining front-end
and reference counting (rio-rs_amd/csrc/gpu_object_placement.cpp) compiled with ThreadSanitizer against a host-memory
stub of the dense C ABI (tests/stub_rio_gp.cpp — test infrastructure, not a product path) and hammered by 12 threads."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_string_layer_under_thread_sanitizer(tmp_path):
    exe = tmp_path / "race_driver"
    srcs = [os.path.join(ROOT, "rio-rs_amd", "csrc", "gpu_object_placement.cpp"),
            os.path.join(ROOT, "tests", "stub_rio_gp.cpp"), os.path.join(ROOT, "tests", "host_layer_race_driver.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", os.path.join(ROOT, "include")]
                   + srcs + ["-o", str(exe)], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    # ThreadSanitizer's runtime occasionally refuses to start under address-space randomisation ("FATAL: ThreadSanitizer:
    # unexpected memory mapping", a property of the host kernel's mmap entropy, before main() runs): that is not a finding
    # about the code under test, so the run is repeated — without randomisation when setarch is there
    cmd = [str(exe)]
    if shutil.which("setarch"):
        cmd = ["setarch", os.uname().machine, "-R"] + cmd
    for attempt in range(4):
        r = subprocess.run(cmd if attempt < 2 else [str(exe)], capture_output=True, text=True, timeout=600, env=env)
        if "FATAL: ThreadSanitizer" not in r.stderr and not (r.returncode != 0 and not r.stdout and "setarch" in r.stderr):
            break
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
    assert "wrong=0" in r.stdout
