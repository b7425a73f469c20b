#!/usr/bin/env python3
"""Run under rocprofv3 --kernel-trace: a one-rank RCCL process group is initialised and used BEFORE the lanes are made (what
bench.py --gpus N does on every rank), then each of 4 lanes decodes a few tokens of a tiny model.  tools/lane_queues.sh-style
post-processing of the trace (Queue_Id per launching thread) shows whether the lane streams still get a hardware queue each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from whisper_amd import checkpoint as ck
from whisper_amd.engine import WhisperEngine
from whisper_amd.lanes import EngineLanes
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if os.environ.get("NO_RCCL") != "1":
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    t = torch.ones(4, device=dev); dist.all_reduce(t); dist.all_gather([torch.empty_like(t)], t); torch.cuda.synchronize()
dims = ck.dims_for("tiny")
sd = ck.synthetic_state_dict(dims, seed=0)
lanes = EngineLanes.create(lambda n: WhisperEngine(dims, sd, precision="bf16", max_batch=4, n_lanes=n), 4)
mel = torch.zeros((dims.n_mels, 4 * 3000), device=dev)
def job(eng, _):
    eng.encode(mel, 3000, mel.stride(0), 4 * 3000, 4)
    return eng.decode_greedy(4, eng.initial_tokens(0), max_new_tokens=6, use_graph=False)["steps"]
print(lanes.map(job, list(range(8))))
if dist.is_initialized():
    t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
    dist.destroy_process_group()
lanes.close()
