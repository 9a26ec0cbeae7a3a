"""bench.py's self-launch (`python bench.py --gpus N` with no launcher around it) - the plumbing, on CPU, with stub children.

What the driver's SCALE run relies on: the parent touches no GPU, starts one process per GPU through torch.distributed.run as CHILD
processes, relays rank 0's line, and falls through resident -> rccl -> onecall in fresh children when a form's children exit non-zero.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


STUB = textwrap.dedent('''
    import json, os, sys
    a = sys.argv[1:]
    form = a[a.index("--form") + 1]
    batch = a[a.index("--batch") + 1] if "--batch" in a else None
    fail = os.environ.get("STUB_FAIL", "").split(",")
    assert os.environ.get("YCGE_BENCH_CHILD") == "1"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    tag = form + ("_batch" + batch if batch and batch != "0" else "")
    with open(os.environ["STUB_LOG"], "a") as f:
        f.write(json.dumps({"tag": tag, "rank": rank, "world": world, "pid": os.getpid(), "argv": a}) + "\\n")
    if tag in fail:
        sys.stderr.write("stub: form %s told to fail\\n" % tag)
        sys.exit(3)
    if rank == 0:
        print("noise before the line")
        print(json.dumps({"metric": "stub", "value": 100.0 * world, "unit": "Mrays/s", "ms_per_step": 1.0 / world, "n_gpus": world,
                          "rccl_world": world if form != "onecall" else None, "latency_frames": 4 if form == "resident" else 1,
                          "config": {"form": form, "parallelism": tag}}))
''')


def run_stub(tmp_path, n, fail="", argv=("--gpus", "2", "--steps", "5", "--form", "auto", "--batch", "3"), **kw):
    stub = tmp_path / "stub_bench.py"
    stub.write_text(STUB)
    log = tmp_path / "stub.log"
    if log.exists():
        log.unlink()
    env = dict(os.environ, STUB_FAIL=fail, STUB_LOG=str(log))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    line, tried = bench.launch_ranks(n, list(argv), script=stub, env=env, timeout=300, log=open(os.devnull, "w"), **kw)
    recs = [json.loads(x) for x in log.read_text().splitlines()] if log.exists() else []
    return line, tried, recs


def test_launcher_helpers():
    assert bench.strip_flag(["--gpus", "8", "--form", "auto", "--batch=4", "--steps", "5", "--batch", "2"], ("--form", "--batch")) == ["--gpus", "8", "--steps", "5"]
    assert bench.last_json_line('x\n{"a": 1}\n{"metric": "m", "value": 2}\n[trailer]\n')["value"] == 2
    assert bench.last_json_line("nothing here") is None
    cmd = bench.child_command("resident", 8, 29511, "bench.py", python="python")
    assert cmd[:3] == ["python", "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert "127.0.0.1" in cmd and cmd[-1] == "bench.py"
    assert bench.child_command("onecall", 8, 1, "bench.py", python="python") == ["python", "bench.py"]
    assert [f[0] for f in bench.FORM_CHAIN] == ["resident", "rccl", "onecall"]
    p = bench.free_port()
    assert 1024 < p < 65536


def test_first_form_is_the_headline_and_batches_are_attached(tmp_path):
    """two real ranks under torch.distributed.run (CPU stub children): the tile-resident form, frame by frame, is the headline; the batched
    launches run as a second job and never replace it"""
    line, tried, recs = run_stub(tmp_path, 2)
    assert [t["form"] for t in tried] == ["resident", "resident_batch4"] and all(t["ok"] for t in tried)
    assert line["config"]["form"] == "resident" and line["config"]["parallelism"] == "resident" and line["value"] == 200.0
    assert line["rccl_world"] == 2 and line["n_gpus"] == 2
    assert line["batched"]["value"] == 200.0 and line["batched"]["parallelism"] == "resident_batch4"
    assert line["forms_tried"] == tried and "torch.distributed.run" in line["launched_by"]
    # one process per rank, per job; the caller's own --form / --batch were replaced, everything else forwarded
    heads = [r for r in recs if r["tag"] == "resident"]
    assert sorted(r["rank"] for r in heads) == [0, 1] and all(r["world"] == 2 for r in heads)
    assert len({r["pid"] for r in recs}) == len(recs) == 4
    for r in recs:
        assert r["argv"].count("--form") == 1 and r["argv"].count("--batch") == 1 and r["argv"][:4] == ["--gpus", "2", "--steps", "5"]


def test_forms_fail_soft_in_fresh_children(tmp_path):
    line, tried, recs = run_stub(tmp_path, 2, fail="resident")
    assert [(t["form"], t["ok"]) for t in tried] == [("resident", False), ("rccl", True)]
    assert tried[0]["rc"] != 0 and "told to fail" in tried[0]["stderr_tail"]
    assert line["config"]["form"] == "rccl" and "batched" not in line and line["rccl_world"] == 2
    line, tried, recs = run_stub(tmp_path, 2, fail="resident,rccl")
    assert [(t["form"], t["ok"]) for t in tried] == [("resident", False), ("rccl", False), ("onecall", True)]
    one = [r for r in recs if r["tag"] == "onecall"]
    assert len(one) == 1 and one[0]["world"] == 1          # ONE process drives all devices in that form: no launcher around it
    assert line["rccl_world"] is None
    line, tried, recs = run_stub(tmp_path, 2, fail="resident,rccl,onecall")
    assert line is None and len(tried) == 3 and not any(t["ok"] for t in tried)
    # a failing batched leg is recorded, the headline stands
    line, tried, recs = run_stub(tmp_path, 2, fail="resident_batch4")
    assert line["config"]["form"] == "resident" and line["batched"] == {"failed": tried[-1]["rc"]} and tried[-1]["rc"] != 0


def test_plain_gpus_n_never_touches_torch_in_the_parent(tmp_path):
    """`python bench.py --gpus 2` end to end with the stub in place of the child script: the parent process must not import torch (the
    driver's rule: nothing that initialises a GPU may precede the start of the ranks) and must print exactly one line."""
    stub = tmp_path / "stub_bench.py"
    stub.write_text(STUB)
    code = textwrap.dedent(f'''
        import sys, json, builtins
        sys.path.insert(0, {str(ROOT)!r})
        real = builtins.__import__
        def guard(name, *a, **k):
            if name == "torch" or name.startswith("torch."):
                raise AssertionError("the launching parent imported " + name)
            return real(name, *a, **k)
        builtins.__import__ = guard
        import bench
        orig = bench.launch_ranks
        bench.launch_ranks = lambda n, argv, **kw: orig(n, argv, script={str(stub)!r}, log=open("/dev/null", "w"), **kw)
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"]
        bench.main()
        assert "torch" not in sys.modules
    ''')
    env = dict(os.environ, STUB_FAIL="", STUB_LOG=str(tmp_path / "stub.log"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "YCGE_BENCH_CHILD", "YCGE_BENCH_FORCE_TILED"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["form"] == "resident" and d["n_gpus"] == 2 and [t["form"] for t in d["forms_tried"]] == ["resident", "resident_batch4"]


def test_build_staleness_is_by_content(tmp_path):
    """build.py: a library is current when its stamp names the content of the sources (mtime says nothing on a copied snapshot)"""
    from yetanotherconsolegameengine_amd import build
    lib = tmp_path / "libx.so"
    assert build.is_stale(lib)
    lib.write_bytes(b"x")
    assert build.is_stale(lib)                                  # no stamp
    build.stamp_path(lib).write_text(build.source_hash() + "\n")
    assert not build.is_stale(lib)
    assert build.is_stale(lib, ["-DX=1"])                        # other flags, other build
    build.stamp_path(lib).write_text(build.source_hash() + " -DX=1\n")
    assert not build.is_stale(lib, ["-DX=1"]) and build.is_stale(lib)
    os.utime(lib, (1, 1))                                        # an old mtime changes nothing
    assert not build.is_stale(lib, ["-DX=1"])
    build.stamp_path(lib).write_text("0123456789abcdef\n")
    assert build.is_stale(lib, ["-DX=1"])


def test_a_job_that_hangs_is_ended_with_its_ranks(tmp_path):
    """a form whose ranks never finish: after the timeout the whole job - launcher and ranks, one process group - is gone and the next form runs"""
    import time
    stub = tmp_path / "stub_bench.py"
    stub.write_text(textwrap.dedent('''
        import json, os, sys, time
        a = sys.argv[1:]; form = a[a.index("--form") + 1]
        open(os.environ["STUB_LOG"], "a").write(json.dumps({"form": form, "pid": os.getpid(), "rank": int(os.environ.get("RANK", "0"))}) + "\\n")
        if form == "resident":
            time.sleep(600)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "stub", "value": 1.0, "config": {"form": form, "parallelism": form}}))
    '''))
    log = tmp_path / "stub.log"
    env = dict(os.environ, STUB_LOG=str(log))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    line, tried = bench.launch_ranks(2, ["--gpus", "2"], script=stub, env=env, timeout=20, log=open(os.devnull, "w"), extra=None)
    assert time.time() - t0 < 120
    assert [(t["form"], t["ok"]) for t in tried] == [("resident", False), ("rccl", True)] and tried[0]["rc"] == -9 and line["config"]["form"] == "rccl"
    hung = [json.loads(x)["pid"] for x in log.read_text().splitlines() if json.loads(x)["form"] == "resident"]
    assert len(hung) == 2
    time.sleep(1.0)
    for pid in hung:           # the ranks of the ended job are gone (no stray process keeps a GPU)
        gone = False
        try:
            os.kill(pid, 0)
            with open(f"/proc/{pid}/stat") as f:
                gone = f.read().split()[2] == "Z"
        except (ProcessLookupError, FileNotFoundError):
            gone = True
        assert gone, pid


def test_the_hang_detector_is_a_default_the_environment_can_change(tmp_path):
    """launch_ranks without a timeout takes FORM_TIMEOUT_S (a hang detector: minutes, not the 25 of round 4) or YCGE_BENCH_FORM_TIMEOUT"""
    import time
    assert 120 <= bench.FORM_TIMEOUT_S <= 600
    stub = tmp_path / "stub_bench.py"
    stub.write_text("import time\ntime.sleep(600)\n")
    env = dict(os.environ, YCGE_BENCH_FORM_TIMEOUT="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    old = os.environ.get("YCGE_BENCH_FORM_TIMEOUT")
    os.environ["YCGE_BENCH_FORM_TIMEOUT"] = "4"
    try:
        t0 = time.time()
        line, tried = bench.launch_ranks(1, ["--gpus", "1"], chain=(("onecall", ["--form", "onecall"]),), script=stub, env=env, log=open(os.devnull, "w"), extra=None)
        assert line is None and tried[0]["rc"] == -9 and time.time() - t0 < 60
    finally:
        if old is None:
            os.environ.pop("YCGE_BENCH_FORM_TIMEOUT", None)
        else:
            os.environ["YCGE_BENCH_FORM_TIMEOUT"] = old
