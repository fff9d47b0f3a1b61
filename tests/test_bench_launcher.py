"""bench.py --gpus N without an outside launcher: the parent starts N rank processes itself.  What those processes find in their
environment, that rank 0's one result line is handed through, and that a failing rank fails the whole run."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _program(tmp_path, body):
    path = tmp_path / "rank_program.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_rank_environment():
    env = bench.rank_environment(2, 4, 29999, base={"PATH": "/bin", "WORLD_SIZE": "9"})
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]) == ("2", "2", "4", "127.0.0.1", "29999")
    assert env["PATH"] == "/bin" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launcher_starts_the_ranks_and_relays_rank_zero(tmp_path, capfd):
    prog = _program(tmp_path, """
        import json, os, sys
        rank = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], "seen%d.json" % rank), "w").write(json.dumps({k: os.environ[k] for k in
            ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[2:]}))
        print("noise that is not the result")
        print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "rank": rank}))
    """)
    rc = bench.launch_ranks(3, [str(tmp_path), "--gpus", "3", "--steps", "2"], timeout_s=60, program=prog)
    out = capfd.readouterr().out
    assert rc == 0
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 3, "rank": 0}
    seen = [json.load(open(tmp_path / ("seen%d.json" % r))) for r in range(3)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"] and [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2"]
    assert all(s["WORLD_SIZE"] == "3" and s["MASTER_ADDR"] == "127.0.0.1" for s in seen)
    assert len({s["MASTER_PORT"] for s in seen}) == 1
    assert all(s["argv"] == ["--gpus", "3", "--steps", "2"] for s in seen)


def test_a_failing_rank_fails_the_run_and_ends_the_others(tmp_path, capfd):
    prog = _program(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)       # a rank left waiting in a collective
    """)
    rc = bench.launch_ranks(2, [], timeout_s=60, program=prog)
    assert rc == 1
    assert "rank 1 exited with status 7" in capfd.readouterr().err


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "--gpus 4" in p.stderr
