"""The host half of the library (csrc/rm_scene.cpp: scene builder, OBJ ingest) compiled
with AddressSanitizer + UBSan and driven through the C ABI on well-formed and malformed
input.  (GPU sanitizers are not available on this pool; the device half is covered by the
parity tests.)"""
import os
import subprocess


def test_scene_builder_and_obj_loader_under_asan_ubsan(entry, cornell_path, tmp_path):
    exe = tmp_path / "host_sanitize"
    subprocess.check_call([
        "g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
        "-ffp-contract=off", "-I", os.path.join(entry.ROOT, "include"), "-I", os.path.join(entry.PKG_DIR, "csrc"),
        os.path.join(entry.PKG_DIR, "csrc", "rm_scene.cpp"),
        os.path.join(entry.ROOT, "tests", "native", "host_sanitize_main.cpp"), "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([str(exe), cornell_path, str(tmp_path)], capture_output=True, env=env)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert b"host sanitize ok" in out.stdout
