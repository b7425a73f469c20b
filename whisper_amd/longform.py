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


def _avg_logprob(sum_logprob: float, n_tokens: int) -> float:
    """upstream DecodingResult.avg_logprob: the EOT step counts in the length"""
    return float(sum_logprob) / (int(n_tokens) + 1)


def _fallback_decision(cr: float, avg_logprob: float, no_speech_prob: float, cr_threshold: Optional[float],
                       logprob_threshold: Optional[float], no_speech_threshold: Optional[float]):
    """``(needs_fallback, should_skip)`` for one decoded window: upstream's ``decode_with_fallback`` test (too repetitive or too
    unlikely, unless the window looks like silence) and ``transcribe``'s no-speech skip (silence, unless the text is likely
    after all).  tests/golden/hf_sequential.json pins both against transformers' ``_need_fallback`` and records the two
    corners where upstream and transformers differ."""
    silent = no_speech_threshold is not None and no_speech_prob > no_speech_threshold
    repetitive = cr_threshold is not None and cr > cr_threshold
    unlikely = logprob_threshold is not None and avg_logprob < logprob_threshold
    likely = logprob_threshold is not None and avg_logprob > logprob_threshold
    return (repetitive or unlikely) and not silent, silent and not likely


def _cut_at_timestamp_pairs(tokens: Sequence[int], tb: int, seek: int, segment_size: int):
    """ids of one window -> ``([(start s, end s, lo, hi)], next seek)``: index ranges ``[lo, hi)`` of the segments and where the
    next window starts.  A segment ends after the first timestamp of every <|t|><|t|> pair; ids after the last pair are
    dropped and decoded again from that timestamp, unless the ids end in a lone timestamp (then they are the last segment
    and the whole window is consumed); without any pair the window is one segment."""
    ids = np.asarray(list(tokens), dtype=np.int64)
    n = ids.size
    t0 = seek * HOP_LENGTH / SAMPLE_RATE
    is_ts = ids >= tb
    ends = (np.flatnonzero(is_ts[:-1] & is_ts[1:]) + 1).tolist() if n >= 2 else []       # index after the first timestamp of a pair
    lone_end = n >= 2 and bool(is_ts[-1]) and not bool(is_ts[-2])
    if not ends:
        stamps = ids[is_ts]
        length = segment_size * HOP_LENGTH / SAMPLE_RATE
        if stamps.size and int(stamps[-1]) != tb:
            length = (int(stamps[-1]) - tb) * TIME_PRECISION
        return [(t0, t0 + length, 0, n)], seek + segment_size
    if lone_end:
        ends.append(n)
    bounds = list(zip([0] + ends[:-1], ends))
    spans = [(t0 + (int(ids[lo]) - tb) * TIME_PRECISION, t0 + (int(ids[hi - 1]) - tb) * TIME_PRECISION, lo, hi) for lo, hi in bounds]
    advance = segment_size if lone_end else (int(ids[ends[-1] - 1]) - tb) * INPUT_STRIDE
    return spans, seek + advance


def _split_segments(tokens: List[int], tb: int, eot: int, seek: int, segment_size: int, decode_text, meta: Dict[str, object]):
    """ids of one window -> (segments, next seek); zero-length and text-less segments keep their times and lose their ids
    (upstream drops them from the prompt carry-over)."""
    spans, nxt = _cut_at_timestamp_pairs(tokens, tb, seek, segment_size)
    out = []
    for start, end, lo, hi in spans:
        toks = list(tokens[lo:hi])
        sg = {"seek": seek, "start": start, "end": end, "tokens": toks, "text": decode_text([t for t in toks if t < eot])}
        sg.update(meta)
        if start == end or sg["text"].strip() == "":
            sg["text"] = ""; sg["tokens"] = []
        out.append(sg)
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
        slots = list(range(B))                 # file b of `active` whose window sits in engine slot j
        can_compact = hasattr(engine, "compact_windows")
        for attempt, t in enumerate(temperatures):
            if attempt > 0 and can_compact and len(pending) < len(slots):
                # upstream re-decodes ONLY the windows that failed: move their encoder results to the front and decode those rows
                engine.compact_windows([slots.index(b) for b in pending])
                slots = list(pending)
            r = engine.decode_rows([prompts[b] for b in slots], temperature=float(t),
                                   seed=(seed * 1000003 + active[0].windows * 101 + attempt) & 0x7FFFFFFF,
                                   max_new_tokens=max_new_tokens, without_timestamps=False, use_graph=use_graph)
            still = []
            for b in pending:
                j = slots.index(b)
                toks = r["tokens"][j]
                avg_logprob = _avg_logprob(r["sum_logprobs"][j], len(toks))
                text = decode_text([x for x in toks if x < tok.eot]).strip()
                cr = compression_ratio(text) if text else 0.0
                nsp = r["no_speech_probs"][j]
                needs_fallback, _ = _fallback_decision(cr, avg_logprob, nsp, compression_ratio_threshold, logprob_threshold, no_speech_threshold)
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
            if _fallback_decision(res["compression_ratio"], res["avg_logprob"], res["no_speech_prob"], compression_ratio_threshold,
                                  logprob_threshold, no_speech_threshold)[1]:
                s.seek += sizes[b]                     # looks like silence: nothing is emitted for this window
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
