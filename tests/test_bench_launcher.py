"""`python bench.py --gpus N` without WORLD_SIZE launches its N ranks itself: the parent starts
torch.distributed.run as a child process, relays rank 0's JSON line and the exit code, and never touches the GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "helpers", "fake_rank.py")


def _bench():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_spawn_ranks_relays_arguments_and_the_json_line(capfd):
    rc, line = _bench().spawn_ranks(2, ["--gpus", "2", "--steps", "3"], script=FAKE)
    assert rc == 0
    got = json.loads(line)
    assert got["argv"] == ["--gpus", "2", "--steps", "3"]
    assert got["world"] == 2 and got["master_addr"] == "127.0.0.1" and got["local_rank"] == 0
    assert line in capfd.readouterr().out  # relayed to the parent's stdout as well


def test_spawn_ranks_relays_a_failing_rank():
    rc, _ = _bench().spawn_ranks(2, ["--exit-code", "7"], script=FAKE)
    assert rc != 0


def test_plain_invocation_does_not_import_torch_cuda_in_the_parent():
    """The parent branch sits in front of every torch import of main(): run bench.py --gpus 2 with the rank script swapped
    for the fake through the launcher's own function, in a subprocess whose torch import is poisoned."""
    code = (
        "import sys, types, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "assert 'torch' not in sys.modules\n"
        f"rc, line = bench.spawn_ranks(2, ['--steps', '1'], script={FAKE!r})\n"
        "assert 'torch' not in sys.modules\n"
        "sys.exit(rc)\n"
    )
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
