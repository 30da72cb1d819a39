"""Child script for tests/test_distributed_gloo.py::test_spawn_local_ranks: joins a gloo group from
the environment spawn_local_ranks sets, sums the ranks, and rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

if len(sys.argv) > 1 and sys.argv[1] == "--fail-rank" and os.environ["RANK"] == sys.argv[2]:
    sys.exit(7)
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
if dist.get_rank() == 0:
    print(json.dumps({"world": dist.get_world_size(), "sum": float(t[0]), "local_rank": os.environ["LOCAL_RANK"]}),
          flush=True)
else:
    print("noise from another rank")
dist.destroy_process_group()
