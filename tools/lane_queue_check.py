#!/usr/bin/env python3
"""EngineLanes' hardware-queue check under a given GPU_MAX_HW_QUEUES (set it in the environment before starting this):
prints what the check changed and whether any two lane streams still take turns."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisper_amd import checkpoint as ck, _lib
from whisper_amd.engine import WhisperEngine
from whisper_amd.lanes import EngineLanes
dims = ck.dims_for("micro")
sd = ck.synthetic_state_dict(dims, seed=0)
lanes = EngineLanes.create(lambda n: WhisperEngine(dims, sd, precision="bf16", max_batch=2, n_lanes=n), 4)
lib = _lib.load()
streams = [ln.stream for ln in lanes._lanes]
with torch.cuda.device(0):
    still = [(i, j) for j in range(4) for i in range(j) if lanes._serialised(streams[i], streams[j], lib)]
print(json.dumps({"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "fixes": lanes.queue_fixes, "pairs_still_taking_turns": still}))
lanes.close()
