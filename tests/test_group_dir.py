"""The group directory of the one-process-per-server C layer (APUS_GROUP_DIR, apus_amd/host/apus_proxy.c): stamped control
files, stale files of an earlier run, zombies.  Plain C, no GPU: tests/group_dir_check.c includes the source."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stamped_control_files_and_liveness(tmp_path):
    from apus_amd import build as b
    lib = b.build(force=False)
    exe = str(tmp_path / "group_dir_check")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "group_dir_check.c"), "-o", exe, "-L", os.path.dirname(lib), "-lapus_gpu", "-lpthread",
                    "-Wl,-rpath," + os.path.dirname(lib)], check=True, capture_output=True, text=True)
    gdir = tmp_path / "group"
    gdir.mkdir()
    r = subprocess.run([exe, str(gdir)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
