"""The generated gfx950 code of the box engine's sweep kernels must not touch the registers of a load in flight before the counted
s_waitcnt that makes them valid (tools/check_box_isa.py: the compiler knows nothing about that protocol and is free to copy such a
register under pressure).  Cross-compiles the device code (no GPU needed, about a minute)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_register_of_a_load_in_flight_is_touched():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_box_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no hazard found" in r.stdout
