"""A stand-in for bench.py's per-rank body (tests/test_bench_launcher.py): no GPU, no model.  Rank 0 prints one JSON line
with what torchrun gave it; `--exit-code N` makes rank 1 fail with N so that the launcher's relay can be checked."""
import json
import os
import sys

argv = sys.argv[1:]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
code = int(argv[argv.index("--exit-code") + 1]) if "--exit-code" in argv else 0
if rank == 0:
    print("some log line that is not JSON")
    print(json.dumps({"argv": argv, "world": world, "master_addr": os.environ.get("MASTER_ADDR"),
                      "local_rank": int(os.environ["LOCAL_RANK"])}), flush=True)
if rank == 1 and code:
    sys.exit(code)
