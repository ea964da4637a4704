"""The undefined-name lint of scripts/lint_names.py as a test of the CPU tier: a name used in a test body or in one of
the plugin scripts and bound nowhere (the defect that turned GPUTEST_r03 / r04 red) fails here, without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_undefined_names_in_tests_bench_and_plugin_scripts():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lint_names.py")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_the_lint_sees_a_body_that_names_another_scopes_variable():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import lint_names as L

    src = "import os\nS = '''wh = 1'''\ndef test_a():\n    x = os.getcwd()\n    assert x\n\n\n    assert wh.schema and pa\n"
    assert sorted(n for _f, _l, n in L.undefined_names(src, "t")) == ["pa", "wh"]
    assert L.undefined_names("def f(a, *b, c=1, **d):\n    return [a + e for e in b if (g := e)] + [c, d, g]\n", "t") == []
