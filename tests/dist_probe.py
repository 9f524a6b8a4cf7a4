"""Rank body for the launcher test: comes up under torch.distributed.run, joins a gloo group, writes what it saw."""
import os
import sys

import torch.distributed as dist

dist.init_process_group("gloo")
out = sys.argv[1]
with open(os.path.join(out, "rank%d.txt" % dist.get_rank()), "w") as f:
    f.write("%d %d %s\n" % (dist.get_rank(), dist.get_world_size(), os.environ.get("LOCAL_RANK")))
dist.barrier()
dist.destroy_process_group()
