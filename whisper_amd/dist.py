"""Window-level data parallelism across the GPUs of one node.

Unit of work = one 30 s window.  Rank g of G owns a contiguous block of windows; weights are replicated; each rank
holds only its own PCM samples (+200 samples of halo each side for the centred STFT).  Two exchange steps exist,
both tiny: one MAX all-reduce of a single fp32 (the whole-file mel clamp), and one all-gather of the padded token
ids / lengths / log-probs at the end.  Everything else is rank-local.

The engine argument is duck-typed (``mel_stage1``, ``mel_stage2``, ``reset_max``, ``transcribe_mel``, ``dims``,
``torch_device``) so the N>1 logic is exercised on CPU with gloo in tests/test_dist_cpu.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .audio import HOP_LENGTH, N_FFT, N_FRAMES, N_SAMPLES, n_windows, shard_range

HALO = N_FFT // 2


def rank_world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _host_staged(group=None) -> bool:
    """gloo moves host memory: device tensors are staged through the host for it (RCCL takes them as they are)."""
    return dist.get_backend(group) == "gloo"


def all_reduce_max(t: torch.Tensor, group=None) -> None:
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)


def all_gather(t: torch.Tensor, group=None) -> List[torch.Tensor]:
    world = dist.get_world_size(group)
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        out = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(out, h, group=group)
        return out
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return out


def local_sample_range(n_file: int, w_lo: int, w_hi: int) -> Tuple[int, int]:
    """File samples a rank needs for windows [w_lo, w_hi): its own span plus the STFT halo, clipped to the file."""
    lo = max(0, w_lo * N_SAMPLES - HALO)
    hi = min(n_file, w_hi * N_SAMPLES + HALO)
    return lo, max(lo, hi)


def sharded_mel(engine, pcm_local: torch.Tensor, pcm_offset: int, n_file: int, w_lo: int, w_hi: int, group=None) -> torch.Tensor:
    """Normalised log-mel ``[n_mels, (w_hi - w_lo) * 3000]`` of this rank's windows, clamped with the WHOLE-file max."""
    n_total = n_file + N_SAMPLES
    n_frames = (w_hi - w_lo) * N_FRAMES
    frame0 = w_lo * N_FRAMES
    mel = torch.empty((engine.dims.n_mels, max(n_frames, 1)), dtype=torch.float32, device=engine.torch_device)
    gmax = torch.empty(1, dtype=torch.float32, device=engine.torch_device)
    engine.reset_max(gmax)
    if n_frames > 0:
        engine.mel_stage1(pcm_local, pcm_offset, n_file, n_total, frame0, n_frames, mel, gmax)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        all_reduce_max(gmax, group)      # 4 bytes
    if n_frames > 0:
        engine.mel_stage2(mel, n_frames, gmax)
    return mel


def gather_results(local: Dict[str, object], n_local_max: int, n_ctx: int, eot: int, device, group=None) -> Dict[str, object]:
    """All-gather per-window token ids (EOT-padded to ``n_ctx``), lengths, sum-logprobs and no-speech probs; returns the
    rank-ordered concatenation with the padding rows removed."""
    rank, world = rank_world(group)
    toks: List[List[int]] = local["tokens"]
    n = len(toks)
    ids = torch.full((n_local_max, n_ctx), eot, dtype=torch.int32)
    lens = torch.full((n_local_max,), -1, dtype=torch.int32)
    fl = torch.zeros((n_local_max, 2), dtype=torch.float32)
    for i, t in enumerate(toks):
        ids[i, :len(t)] = torch.tensor(t, dtype=torch.int32)
        lens[i] = len(t)
        fl[i, 0] = float(local["sum_logprobs"][i]); fl[i, 1] = float(local["no_speech_probs"][i])
    if world == 1:
        return {"tokens": toks, "sum_logprobs": list(local["sum_logprobs"]), "no_speech_probs": list(local["no_speech_probs"])}
    ids, lens, fl = ids.to(device), lens.to(device), fl.to(device)
    g_ids, g_lens, g_fl = all_gather(ids, group), all_gather(lens, group), all_gather(fl, group)
    out_t, out_s, out_n = [], [], []
    for r in range(world):
        ic, lc, fc = g_ids[r].cpu(), g_lens[r].cpu(), g_fl[r].cpu()
        for i in range(n_local_max):
            if int(lc[i]) < 0:
                continue
            out_t.append(ic[i, :int(lc[i])].tolist())
            out_s.append(float(fc[i, 0])); out_n.append(float(fc[i, 1]))
    return {"tokens": out_t, "sum_logprobs": out_s, "no_speech_probs": out_n}


def broadcast_int(value: int, device, src: int = 0, group=None) -> int:
    """One int32 from rank ``src`` to every rank (the detected language: 4 bytes)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(value)
    host = dist.get_backend(group) == "gloo"
    t = torch.tensor([int(value)], dtype=torch.int32, device="cpu" if host else device)
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def transcribe_sharded(engine, audio: np.ndarray, prompt, group=None, **decode_kw) -> Dict[str, object]:
    """Every rank passes the same host ``audio``; each uploads and processes only its own windows.  All ranks return
    the full, window-ordered result.

    ``prompt``: the initial ids shared by all windows, or a callable ``language_index -> ids`` -- then the language is
    detected first, as upstream does, on the first 30 s of the WHOLE-FILE mel (clamped with the whole-file maximum, which is
    known after the MAX all-reduce): rank 0 owns window 0, detects, and broadcasts the index (one more 4-byte exchange)."""
    rank, world = rank_world(group)
    is_pcm16 = isinstance(audio, np.ndarray) and audio.dtype == np.int16     # raw samples: converted on the device
    audio = np.ascontiguousarray(audio if is_pcm16 else np.asarray(audio, dtype=np.float32))
    n_file = int(audio.shape[0])
    nw = n_windows(n_file)
    w_lo, w_hi = shard_range(nw, rank, world)
    s_lo, s_hi = local_sample_range(n_file, w_lo, w_hi)
    local = audio[s_lo:max(s_hi, s_lo + 1)] if s_hi > s_lo else np.zeros(1, audio.dtype)
    pcm_local = engine.pcm16_to_f32(local) if is_pcm16 else torch.from_numpy(local).to(engine.torch_device)
    mel = sharded_mel(engine, pcm_local, s_lo, n_file, w_lo, w_hi, group)
    content = n_file // HOP_LENGTH
    language_index = None
    if callable(prompt):
        language_index = engine.detect_language(mel, content) if rank == 0 and w_hi > w_lo else 0
        language_index = broadcast_int(language_index, engine.torch_device, 0, group)
        prompt = prompt(language_index)
    if w_hi > w_lo:
        local = engine.transcribe_mel(mel, content - w_lo * N_FRAMES, 0, w_hi - w_lo, prompt, **decode_kw)
    else:
        local = {"tokens": [], "sum_logprobs": [], "no_speech_probs": [], "steps": 0}
    per = shard_range(nw, 0, world)[1]
    out = gather_results(local, max(per, 1), engine.dims.n_text_ctx + 1, engine.tok.eot, engine.torch_device, group)
    out["n_windows"] = nw
    out["local_windows"] = (w_lo, w_hi)
    out["steps"] = local.get("steps", 0)
    out["language_index"] = language_index
    return out
