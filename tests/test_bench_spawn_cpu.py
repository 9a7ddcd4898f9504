"""`python bench.py --gpus N` starts its own N ranks (fresh `torch.distributed.run` children, before the parent
makes any GPU call), relays rank 0's one JSON line and fails when a rank fails.  Rehearsed here on CPU with gloo
ranks (`--selftest-spawn`: the same spawn / rendezvous / gather / relay plumbing, no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True,
                          timeout=300, env=env, cwd=ROOT)


def test_parent_spawns_ranks_and_relays_one_json_line():
    p = _run("--gpus", "2", "--selftest-spawn", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                       # the stray line rank 0 printed went to stderr
    r = json.loads(lines[0])
    assert r["rccl_ranks"] == 2 and r["n_gpus"] == 2 and r["value"] == 1.0
    assert "stray line on stdout" in p.stderr


def test_failing_rank_fails_the_parent():
    p = _run("--gpus", "2", "--selftest-spawn", "2")
    assert p.returncode != 0
    assert not p.stdout.strip()
    assert "a rank failed" in p.stderr


def test_more_gpus_than_the_node_has_is_a_clear_error_not_an_assert():
    p = _run("--gpus", "64")                               # no node has 64; here there is no GPU at all
    assert p.returncode == 2
    assert "nothing was run" in p.stderr and "AssertionError" not in p.stderr and "Traceback" not in p.stderr
    assert not p.stdout.strip()


def test_world_size_mismatch_is_a_clear_error():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr and "Traceback" not in p.stderr


def test_configs4_manifest_leg_on_two_gloo_ranks():
    """`bench.py --workload manifest` (BASELINE configs[4]: the ragged manifest LPT-sharded, decoded locally, gathered on rank 0)
    rehearsed on CPU: two gloo ranks through the same leg (manifest_sharded -> shard.convert_sharded -> shard.gather_waveforms)
    with a stand-in decode.  One JSON line; every utterance comes back with its own length; per-rank statistics are reported."""
    p = _run("--gpus", "2", "--selftest-spawn", "3", "--workload", "manifest", "--manifest", "37", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["backend"] == "gloo" and r["scaling"] == "strong"
    assert r["all_utterances_gathered_with_their_lengths"] is True
    assert sum(r["per_rank"]["utterances"]) == 37 and min(r["per_rank"]["utterances"]) >= 1
    assert abs(r["per_rank"]["samples"][0] - r["per_rank"]["samples"][1]) <= 0.15 * max(r["per_rank"]["samples"])      # LPT balance
    assert r["value"] > 0 and r["steps"] == 2 and len(r["per_rank"]["gather_ms_last_step"]) == 2
    assert "configs[4]" in r["config"]["workload"]
