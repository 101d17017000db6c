"""bench.py's own launcher (`python bench.py --gpus N` outside torchrun): the wedge fallback, executed for real on CPU.

A stand-in rank program runs under the REAL `python -m torch.distributed.run`: in the first launch one rank writes the
wedge marker and leaves through os._exit(3) exactly like bench.py's `leave_wedged()`; the launcher turns that into exit
code 1 (ChildFailedError), which is why the signal is a marker file and not the code.  self_launch() must then start the
ranks once more with FDT_BENCH_EXCHANGE=torch, relay exactly ONE line (the second run's) and return that run's code."""
import argparse
import importlib.util
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


RANK_PROGRAM = textwrap.dedent('''
    import json, os, sys, time
    sys.path.insert(0, %(root)r)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(%(root)r, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    exch = os.environ.get("FDT_BENCH_EXCHANGE", "rccl-cabi")
    mode = sys.argv[1]
    with open(os.path.join(%(log)r, "launch_%%s_rank%%d" %% (exch, rank)), "w") as f:
        f.write(mode)
    if rank == 0:
        print(json.dumps({"exchange": exch, "n_gpus": world}))
        sys.stdout.flush()
    if mode == "wedge" and exch == "rccl-cabi":
        if rank == world - 1:
            bench.leave_wedged(3)        # the rank with a thread stuck inside the collective
        time.sleep(1.0)
    if mode == "plainfail" and rank == world - 1:
        sys.exit(5)                      # an ordinary failure: no marker, no second launch
''')


@pytest.fixture
def rank_program(tmp_path):
    p = tmp_path / "rank_program.py"
    p.write_text(RANK_PROGRAM % {"root": ROOT, "log": str(tmp_path)})
    return str(p), tmp_path


def run_launch(bench, script, mode, capsys, monkeypatch):
    monkeypatch.delenv("FDT_BENCH_EXCHANGE", raising=False)
    monkeypatch.delenv("FDT_BENCH_WEDGE_MARKER", raising=False)
    rc = bench.self_launch(argparse.Namespace(gpus=2), script=script, argv=[mode])
    out = capsys.readouterr().out
    return rc, [ln for ln in out.splitlines() if ln.strip()]


@pytest.mark.timeout(300)
def test_wedged_rank_makes_the_launcher_run_the_ranks_again_on_the_torch_exchange(rank_program, capsys, monkeypatch):
    script, log = rank_program
    bench = load_bench()
    rc, lines = run_launch(bench, script, "wedge", capsys, monkeypatch)
    assert rc == 0
    assert len(lines) == 1, lines                       # exactly one line relayed: the second run's
    assert json.loads(lines[0]) == {"exchange": "torch", "n_gpus": 2}
    made = sorted(p.name for p in log.iterdir() if p.name.startswith("launch_"))
    assert made == ["launch_rccl-cabi_rank0", "launch_rccl-cabi_rank1", "launch_torch_rank0", "launch_torch_rank1"]


@pytest.mark.timeout(300)
def test_clean_run_is_launched_once(rank_program, capsys, monkeypatch):
    script, log = rank_program
    bench = load_bench()
    rc, lines = run_launch(bench, script, "ok", capsys, monkeypatch)
    assert rc == 0 and len(lines) == 1 and json.loads(lines[0])["exchange"] == "rccl-cabi"
    assert sorted(p.name for p in log.iterdir() if p.name.startswith("launch_")) == \
        ["launch_rccl-cabi_rank0", "launch_rccl-cabi_rank1"]


@pytest.mark.timeout(300)
def test_ordinary_failure_is_not_retried_and_its_code_is_propagated(rank_program, capsys, monkeypatch):
    script, log = rank_program
    bench = load_bench()
    rc, lines = run_launch(bench, script, "plainfail", capsys, monkeypatch)
    assert rc != 0                                      # the launcher's own failure code, handed on
    assert not any(p.name.startswith("launch_torch") for p in log.iterdir())


def test_leave_wedged_writes_the_marker(tmp_path, monkeypatch):
    bench = load_bench()
    marker = tmp_path / "m"
    monkeypatch.setenv("FDT_BENCH_WEDGE_MARKER", str(marker))
    codes = []
    monkeypatch.setattr(os, "_exit", lambda c: codes.append(c))
    bench.leave_wedged(3)
    assert codes == [3] and marker.exists()
