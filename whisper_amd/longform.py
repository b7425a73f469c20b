"""Faithful sequential long-form decoding on the MI355X engine: what openai-whisper's
``Whisper.transcribe`` does with the arguments the reference passes (ref:app/transcription/speech_recognition.py:48-55 sets
only language / task / fp16, so upstream's defaults apply: timestamps on, ``condition_on_previous_text=True``, temperature
fallback ``(0.0, 0.2, ..., 1.0)``, compression-ratio 2.4 / avg-logprob -1.0 / no-speech 0.6 thresholds).

Per file the loop is inherently sequential -- every window's prompt is the text decoded before it and its start is the
last timestamp of the previous window -- so one file cannot be sharded across windows in this mode (that is the
independent-window mode of whisper_amd/dist.py, a declared divergence).  Throughput comes from running MANY FILES in
lockstep instead: every iteration takes the current window of each unfinished file, encodes them as one batch and decodes
them with per-row prompts (``wx_decode_rows``), so the GPU sees batch = number of files (BASELINE.json configs[4]).

Algorithm per window (restated from upstream, cross-checked with the HF port hf:models/whisper/generation_whisper.py
:785-903, :1243-1287, :1853-1918, :1977-2074):
  prompt  = [sot_prev] + previous tokens[-(n_ctx/2 - 1):] + [sot, lang, task]        (timestamps enabled)
  decode at temperature 0; while (compression ratio > 2.4 or avg logprob < -1.0) and not (no_speech > 0.6): next temperature
  skip the window if no_speech > 0.6 and avg logprob <= -1.0
  split ids at consecutive timestamp pairs into segments; seek += last timestamp x 2 frames, or the whole window
  prompt carry-over is reset after a window decoded at temperature > 0.5
Sampling at T > 0 uses the engine's counter-based generator (seeded per file, window and attempt), so those re-decodes
match upstream in distribution, not id for id.
"""
from __future__ import annotations

import zlib
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from .audio import HOP_LENGTH, N_FRAMES, N_SAMPLES, SAMPLE_RATE

TIME_PRECISION = 0.02          # seconds per timestamp token
INPUT_STRIDE = 2               # mel frames per encoder position


def compression_ratio(text: str) -> float:
    b = text.encode("utf-8")
    return len(b) / len(zlib.compress(b))


class _FileState:
    def __init__(self, mel: torch.Tensor, content_frames: int, initial_prompt: Sequence[int]):
        self.mel, self.content_frames = mel, content_frames
        self.seek = 0
        self.all_tokens: List[int] = list(initial_prompt)
        self.n_initial = len(self.all_tokens)
        self.prompt_reset_since = 0
        self.segments: List[Dict[str, object]] = []
        self.windows = 0

    @property
    def done(self) -> bool:
        return self.seek >= self.content_frames


def _split_segments(tokens: List[int], tb: int, eot: int, seek: int, segment_size: int, decode_text, meta: Dict[str, object]):
    """ids of one window -> (segments, next seek)."""
    time_offset = seek * HOP_LENGTH / SAMPLE_RATE
    segment_duration = segment_size * HOP_LENGTH / SAMPLE_RATE

    def new_segment(start, end, toks):
        sg = {"seek": seek, "start": start, "end": end, "tokens": list(toks), "text": decode_text([t for t in toks if t < eot])}
        sg.update(meta)
        return sg
    is_ts = [t >= tb for t in tokens]
    single_timestamp_ending = is_ts[-2:] == [False, True]
    consecutive = [i + 1 for i in range(len(tokens) - 1) if is_ts[i] and is_ts[i + 1]]
    out = []
    if consecutive:
        slices = list(consecutive)
        if single_timestamp_ending:
            slices.append(len(tokens))
        last_slice = 0
        for cur in slices:
            sl = tokens[last_slice:cur]
            out.append(new_segment(time_offset + (sl[0] - tb) * TIME_PRECISION, time_offset + (sl[-1] - tb) * TIME_PRECISION, sl))
            last_slice = cur
        if single_timestamp_ending:
            nxt = seek + segment_size          # a single timestamp at the end: no speech after it
        else:
            nxt = seek + (tokens[last_slice - 1] - tb) * INPUT_STRIDE
    else:
        duration = segment_duration
        ts = [t for t in tokens if t >= tb]
        if ts and ts[-1] != tb:
            duration = (ts[-1] - tb) * TIME_PRECISION
        out.append(new_segment(time_offset, time_offset + duration, tokens))
        nxt = seek + segment_size
    for sg in out:
        if sg["start"] == sg["end"] or sg["text"].strip() == "":
            sg["text"] = ""; sg["tokens"] = []
    return out, nxt


def transcribe_sequential(engine, audios: Sequence, decode_text: Callable[[Sequence[int]], str], language_index: Optional[int] = 0,
                          initial_prompts: Optional[Sequence[Optional[Sequence[int]]]] = None,
                          temperatures: Sequence[float] = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
                          compression_ratio_threshold: Optional[float] = 2.4, logprob_threshold: Optional[float] = -1.0,
                          no_speech_threshold: Optional[float] = 0.6, condition_on_previous_text: bool = True,
                          max_new_tokens: int = 0, seed: int = 0, use_graph: bool = True) -> List[Dict[str, object]]:
    """``audios``: float32 16 kHz arrays / tensors (or int16 PCM arrays), one per file.  Returns one dict per file:
    ``segments`` (seek, start, end, tokens, text, temperature, avg_logprob, compression_ratio, no_speech_prob), ``tokens``,
    ``text``, ``windows`` (decoded windows, skipped ones included), ``language_index``.  ``language_index=None`` detects
    the language of every file on the first 30 s of its whole-file mel, as upstream does."""
    tok = engine.tok
    dev = engine.torch_device
    n_mels = engine.dims.n_mels
    states: List[_FileState] = []
    for i, a in enumerate(audios):
        if isinstance(a, np.ndarray) and a.dtype == np.int16:
            pcm = engine.pcm16_to_f32(a)
        else:
            pcm = torch.as_tensor(a).to(dev, torch.float32).contiguous()
        mel = engine.log_mel(pcm)                                   # whole file + 30 s of padding, whole-file clamp
        ip = (initial_prompts[i] if initial_prompts else None) or []
        st = _FileState(mel, mel.shape[-1] - N_FRAMES, ip)
        st.language_index = engine.detect_language(mel, st.content_frames) if language_index is None else int(language_index)
        states.append(st)
    batch = torch.zeros((engine.max_batch, n_mels, N_FRAMES), dtype=torch.float32, device=dev)
    while True:
        active = [s for s in states if not s.done][: engine.max_batch]
        if not active:
            break
        B = len(active)
        sizes = []
        batch[:B].zero_()
        for b, s in enumerate(active):
            size = min(N_FRAMES, s.content_frames - s.seek)
            batch[b, :, :size] = s.mel[:, s.seek:s.seek + size]
            sizes.append(size)
        engine.encode(batch, n_mels * N_FRAMES, N_FRAMES, B * N_FRAMES, B)
        prompts = [engine.initial_tokens(s.language_index, without_timestamps=False, prompt_ids=s.all_tokens[s.prompt_reset_since:])
                   for s in active]
        result: List[Optional[Dict[str, object]]] = [None] * B
        pending = list(range(B))
        for attempt, t in enumerate(temperatures):
            r = engine.decode_rows(prompts, temperature=float(t), seed=(seed * 1000003 + active[0].windows * 101 + attempt) & 0x7FFFFFFF,
                                   max_new_tokens=max_new_tokens, without_timestamps=False, use_graph=use_graph)
            still = []
            for b in pending:
                toks = r["tokens"][b]
                avg_logprob = r["sum_logprobs"][b] / (len(toks) + 1)
                text = decode_text([x for x in toks if x < tok.eot]).strip()
                cr = compression_ratio(text) if text else 0.0
                nsp = r["no_speech_probs"][b]
                needs_fallback = False
                if compression_ratio_threshold is not None and cr > compression_ratio_threshold:
                    needs_fallback = True
                if logprob_threshold is not None and avg_logprob < logprob_threshold:
                    needs_fallback = True
                if no_speech_threshold is not None and nsp > no_speech_threshold:
                    needs_fallback = False
                result[b] = {"tokens": toks, "temperature": float(t), "avg_logprob": avg_logprob, "compression_ratio": cr,
                             "no_speech_prob": nsp, "needs_fallback": needs_fallback}
                if needs_fallback:
                    still.append(b)
            pending = still
            if not pending:
                break
        for b, s in enumerate(active):
            res = result[b]
            s.windows += 1
            if no_speech_threshold is not None:
                should_skip = res["no_speech_prob"] > no_speech_threshold
                if logprob_threshold is not None and res["avg_logprob"] > logprob_threshold:
                    should_skip = False
                if should_skip:
                    s.seek += sizes[b]
                    continue
            meta = {k: res[k] for k in ("temperature", "avg_logprob", "compression_ratio", "no_speech_prob", "needs_fallback")}
            segs, nxt = _split_segments(res["tokens"], tok.timestamp_begin, tok.eot, s.seek, sizes[b], decode_text, meta)
            s.segments.extend(segs)
            for sg in segs:
                s.all_tokens.extend(sg["tokens"])
            if not condition_on_previous_text or res["temperature"] > 0.5:
                s.prompt_reset_since = len(s.all_tokens)
            s.seek = nxt
    out = []
    for s in states:
        toks = s.all_tokens[s.n_initial:]
        out.append({"segments": s.segments, "tokens": toks, "text": decode_text([t for t in toks if t < tok.eot]), "windows": s.windows,
                    "language_index": s.language_index})
    return out
