"""apus_gpu_join's answer count on the host (apus_amd/csrc/apus_members_host.h: what every member ITSELF holds, derived
from the journal of CONFIG entries and votes; poll_config_entries src/dare/dare_server.c:2133-2187, handle_rc_syn
src/dare/dare_ibv_ud.c, oracle/apus_oracle.c:orc_join -6) -- plain C++, checked here without a GPU; the device side and
the random join schedules are tests/test_gpu_parity.py::test_random_join_traces*."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_member_views_and_join_answers(tmp_path):
    exe = str(tmp_path / "members_host_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", os.path.join(ROOT, "tests", "members_host_check.cpp"), "-o", exe],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
