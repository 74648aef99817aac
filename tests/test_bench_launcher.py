"""bench.py's own launcher (the driver contract): `python bench.py --gpus N` without a launcher starts N
ranks through torch.distributed.run BEFORE touching the GPU, rank 0 prints ONE JSON line with n_gpus == N, and a failing
rank makes the parent exit non-zero.  CPU only: --dry-run swaps the GPU work for a gloo all-reduce."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + list(argv), capture_output=True, text=True, env=e, timeout=240)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    r = _run("--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 4 and lines[0]["warmup"] == 1


def test_gpus_8_spawns_eight_ranks_and_prints_one_line():
    """the driver's largest configuration: eight ranks rendezvous on the loopback interface, the sum all-reduce sees all
    of them, ONE line comes out (this container has 8 cores; the GPU box gives the job 16)"""
    r = _run("--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    line = lines[0]
    assert {k: line[k] for k in ("dry_run", "n_gpus", "steps", "warmup")} == {"dry_run": True, "n_gpus": 8, "steps": 2, "warmup": 0}
    # round 4: the per-rank fields of the N > 1 line, reduced over the ranks the way the real run reduces them (rank r
    # pretends to 10 + r ms per step and r / 10 ms of exposed all-reduce)
    assert line["ms_per_step_min_over_ranks"] == 10.0 and line["ms_per_step_max_over_ranks"] == 17.0
    assert abs(line["allreduce_exposed_ms_max_over_ranks"] - 0.7) < 1e-12
    assert line["dist_world_size"] == 8 and line["rccl_version"] and line["reserved_cus"] == 0


def test_gpus_1_runs_in_process():
    r = _run("--dry-run")
    assert r.returncode == 0 and _json_lines(r.stdout) == [{"dry_run": True, "n_gpus": 1, "steps": 10, "warmup": 3}]


def test_world_size_mismatch_is_an_error():
    r = _run("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_failing_rank_fails_the_parent():
    """no GPU here: the real (non-dry) ranks die on `bench.py needs an MI355X`; the parent must not report success"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a machine without a GPU")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []


def test_two_ranks_on_one_device_is_an_error():
    """N ranks must hold N distinct devices.  Under the gloo dry run every rank reports LOCAL_RANK as its
    device; DSPN_DRY_DEVICE=0 makes both claim device 0 -- the parent exits non-zero, no result line, the reason on stderr"""
    r = _run("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", env={"DSPN_DRY_DEVICE": "0"})
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []
    assert "distinct devices" in r.stderr and "a rank failed" in r.stderr


def test_ranks_report_their_device_before_the_first_step():
    r = _run("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[bench rank 0/2] device 0" in r.stderr and "[bench rank 1/2] device 1" in r.stderr
