"""Race detection for the host side of the boundary: the string layer's interning, combining front-end
and reference counting (rio-rs_amd/csrc/gpu_object_placement.cpp) compiled with ThreadSanitizer against a host-memory
stub of the dense C ABI (tests/stub_rio_gp.cpp — test infrastructure, not a product path) and hammered by 12 threads."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_string_layer_under_thread_sanitizer(tmp_path):
    exe = tmp_path / "race_driver"
    srcs = [os.path.join(ROOT, "rio-rs_amd", "csrc", "gpu_object_placement.cpp"),
            os.path.join(ROOT, "tests", "stub_rio_gp.cpp"), os.path.join(ROOT, "tests", "host_layer_race_driver.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", os.path.join(ROOT, "include")]
                   + srcs + ["-o", str(exe)], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
    assert "wrong=0" in r.stdout
