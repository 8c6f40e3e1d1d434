"""bench.py's supervisor (one-process runs measure in a child; one disclosed restart if the child dies of a signal before printing):
exercised on CPU with stand-in children -- the measuring code itself needs a GPU and is not touched here."""
import importlib.util
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_supervised(tmp_path, child_src: str):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(child_src))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import importlib.util, sys
        spec = importlib.util.spec_from_file_location("bench_under_test", {os.path.join(ROOT, 'bench.py')!r})
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.supervise.__globals__["__file__"] = {str(child)!r}
        sys.argv = ["bench.py", "--steps", "3"]
        sys.exit(b.supervise() or 0)
    """))
    return subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, cwd=str(tmp_path))


def test_result_is_relayed_with_the_attempt_count(tmp_path):
    r = run_supervised(tmp_path, """
        import json, sys
        assert sys.argv[1:] == ["--steps", "3", "--child"], sys.argv
        print("RCCL banner noise")
        print(json.dumps({"metric": "cooccurrence_pairs_per_s", "value": 1.0}))
    """)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.0 and line["attempts"] == 1 and "first_attempt" not in line
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1      # ONE JSON line on stdout


def test_one_restart_after_a_signal_and_the_line_says_so(tmp_path):
    r = run_supervised(tmp_path, """
        import json, os
        if not os.path.exists("died_once"):
            open("died_once", "w").close()
            os.abort()                       # what a GPU memory fault does to the process (SIGABRT from the HSA runtime)
        print(json.dumps({"metric": "cooccurrence_pairs_per_s", "value": 2.0}))
    """)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["attempts"] == 2 and "signal 6" in line["first_attempt"] and line["value"] == 2.0


def test_ordinary_failures_are_not_retried(tmp_path):
    r = run_supervised(tmp_path, """
        import os, sys
        open("ran_%d" % len([f for f in os.listdir(".") if f.startswith("ran_")]), "w").close()
        sys.exit("pairs differ from the oracle")
    """)
    assert r.returncode != 0 and "exited with 1" in r.stderr
    assert len([f for f in os.listdir(tmp_path) if f.startswith("ran_")]) == 1       # an assertion / mismatch is reported, not retried


def test_three_signals_fail_the_run_and_the_third_attempt_is_lean(tmp_path):
    r = run_supervised(tmp_path, """
        import os, sys
        open("args_%d" % len([f for f in os.listdir(".") if f.startswith("args_")]), "w").write(" ".join(sys.argv[1:]))
        os.abort()
    """)
    assert r.returncode != 0 and "signal 6" in r.stderr and not r.stdout.strip()
    args = [open(os.path.join(tmp_path, f)).read() for f in sorted(os.listdir(tmp_path)) if f.startswith("args_")]
    assert len(args) == 3 and "--no-extras" not in args[0] and "--no-extras" not in args[1] and args[2].endswith("--no-extras")


def test_the_lean_third_attempt_is_disclosed(tmp_path):
    r = run_supervised(tmp_path, """
        import json, os, sys
        if "--no-extras" not in sys.argv:
            os.abort()
        print(json.dumps({"metric": "cooccurrence_pairs_per_s", "value": 4.0}))
    """)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["attempts"] == 3 and "extras_skipped" in line and "signal 6" in line["first_attempt"]


def test_a_child_that_dies_after_reporting_has_reported(tmp_path):
    r = run_supervised(tmp_path, """
        import json, os
        print(json.dumps({"metric": "cooccurrence_pairs_per_s", "value": 3.0}), flush=True)
        os.abort()                           # a fault while the process tears its GPU state down
    """)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 3.0 and line["attempts"] == 1 and "signal 6" in line["child_exit"]
