#!/usr/bin/env python3
"""dev tool: which of the processes of `python bench.py --gpus N` hold the GPU's device files open (the pool's process guard
counts them)?  Prints the /dev/kfd and /dev/dri links of this process after each step."""
import os, sys
def fds(tag):
    links = []
    for f in os.listdir("/proc/self/fd"):
        try:
            t = os.readlink(f"/proc/self/fd/{f}")
        except OSError:
            continue
        if "kfd" in t or "dri" in t:
            links.append(t)
    print(f"{tag}: {sorted(set(links))}", flush=True)
fds("start")
import torch
fds("import torch")
import torch.distributed.run as r
fds("import torch.distributed.run")
print("device_count", torch.cuda.device_count()); fds("device_count")
import torch.distributed.elastic.agent.server.local_elastic_agent as a
fds("import local_elastic_agent")
