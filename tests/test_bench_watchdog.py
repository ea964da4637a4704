"""bench.py's safety net for the multi-rank extras: if a collective wedges, rank 0 still prints the
headline JSON line (with the unfinished legs marked) and the process exits."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_extras_watchdog_prints_the_headline_and_exits():
    code = textwrap.dedent('''
        import sys, time
        sys.argv = ["bench.py"]
        import bench
        w = bench._ExtrasWatchdog(0, {"metric": "filter_take_mrows_per_s", "value": 1.0, "hash_sum": {"rows": 8}}, 0.3)
        w.start()
        time.sleep(20)
        print("NOT REACHED")
    ''')
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and "NOT REACHED" not in r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.0 and line["hash_sum"] == {"rows": 8}
    assert "watchdog" in line["sort_indices"]["error"]


def test_extras_watchdog_cancelled_is_silent():
    code = textwrap.dedent('''
        import sys, time
        sys.argv = ["bench.py"]
        import bench
        w = bench._ExtrasWatchdog(0, {"metric": "m"}, 0.3)
        w.start(); w.cancel()
        time.sleep(1.0)
        print("REACHED")
    ''')
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "REACHED"
