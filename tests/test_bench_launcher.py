"""CPU tier: bench.py's own multi-GPU launcher (`--gpus N` without torchrun) starts N rank processes with the
torch.distributed environment, from a parent that never imports torch (so it cannot have touched the GPU), and
refuses a --gpus / WORLD_SIZE mismatch."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = '''
import json, os, sys
rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")}
rec["argv"] = sys.argv[1:]
json.dump(rec, open(os.path.join(os.environ["QD_STUB_DIR"], "rank%s.json" % rec["RANK"]), "w"))
if rec["RANK"] == "0":
    print(json.dumps({"n_gpus": int(rec["WORLD_SIZE"])}))
if os.environ.get("QD_STUB_HANG") and rec["RANK"] != "1":
    import time
    time.sleep(600)                                        # a rank stuck in a barrier its dead sibling never reaches
sys.exit(3 if rec["RANK"] == "1" and os.environ.get("QD_STUB_FAIL") else 0)
'''


def _run(tmp_path, extra_env, *flags):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    env = dict(os.environ, QD_BENCH_RANK_SCRIPT=str(stub), QD_STUB_DIR=str(tmp_path), **extra_env)
    env.pop("WORLD_SIZE", None) if "WORLD_SIZE" not in extra_env else None
    # the parent must work without torch: poison the import
    poison = tmp_path / "poison"; poison.mkdir(exist_ok=True)
    (poison / "torch.py").write_text("raise ImportError('the launcher parent must not import torch')\n")
    env["PYTHONPATH"] = str(poison)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True,
                          timeout=120)


def test_gpus_flag_launches_that_many_ranks(tmp_path):
    r = _run(tmp_path, {}, "--gpus", "4", "--steps", "2", "--warmup", "1", "--share-gpu", "--backend", "gloo")
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"n_gpus": 4}       # only rank 0's stdout is passed on
    recs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(4)]
    assert [x["RANK"] for x in recs] == ["0", "1", "2", "3"] and all(x["WORLD_SIZE"] == "4" for x in recs)
    assert all(x["LOCAL_RANK"] == "0" for x in recs)                            # --share-gpu rehearsal
    assert len({x["MASTER_PORT"] for x in recs}) == 1 and all(x["MASTER_ADDR"] == "127.0.0.1" for x in recs)
    assert recs[2]["argv"] == ["--gpus", "4", "--steps", "2", "--warmup", "1", "--share-gpu", "--backend", "gloo"]


def test_local_ranks_are_distinct_without_share_gpu_and_failures_propagate(tmp_path):
    r = _run(tmp_path, {"QD_STUB_FAIL": "1"}, "--gpus", "2")
    assert r.returncode == 3
    recs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [x["LOCAL_RANK"] for x in recs] == ["0", "1"]


def test_world_size_mismatch_is_refused(tmp_path):
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_eight_ranks_mixed_config_and_thread_caps(tmp_path):
    """the shape the driver runs on a whole node: --gpus 8 (here with the recording stub), --config mixed passed through,
    host threads capped per rank so that 8 device samplers do not oversubscribe the cores"""
    env = {k: v for k, v in os.environ.items()}
    env.pop("OMP_NUM_THREADS", None)
    stub = tmp_path / "stub.py"; stub.write_text(STUB)
    env.update(QD_BENCH_RANK_SCRIPT=str(stub), QD_STUB_DIR=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "mixed", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    recs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(8)]
    assert [x["LOCAL_RANK"] for x in recs] == [str(k) for k in range(8)] and all(x["WORLD_SIZE"] == "8" for x in recs)
    assert all(x["argv"][:4] == ["--gpus", "8", "--config", "mixed"] for x in recs)
    want = str(max(1, (os.cpu_count() or 8) // 8))
    assert all(x["OMP_NUM_THREADS"] == want for x in recs)


def test_a_dead_rank_takes_its_siblings_down_promptly(tmp_path):
    """rank 1 dies, the others would wait in a barrier; the parent polls, stops them and returns the failure"""
    import time
    t0 = time.time()
    r = _run(tmp_path, {"QD_STUB_FAIL": "1", "QD_STUB_HANG": "1"}, "--gpus", "3")
    assert r.returncode == 3 and "rank 1 exited with code 3" in r.stderr
    assert time.time() - t0 < 60


def test_local_device_index_with_narrowed_visibility():
    sys.path.insert(0, os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"))
    from qadapt_hip.shard import local_device_index
    assert [local_device_index(r, 8) for r in range(8)] == list(range(8))
    assert [local_device_index(r, 1) for r in range(8)] == [0] * 8              # HIP_VISIBLE_DEVICES = one GPU per rank
