"""``transcribe(audio, language=None, prompt=None) -> {"text", "confidence", "chunks_processed"}`` — the north-star
contract of the local backend (BASELINE.json), served by the MI355X engine.

The reference has no function of exactly this shape (SURVEY.md §8a "observations"); its real local call is
``transcribe_audio(audio_path, detailed)`` (ref:app/transcription/speech_recognition.py:29-79), mirrored in
``whisper_amd/speech_recognition.py`` on top of this module.

Algorithm = upstream ``Whisper.transcribe`` restricted to what the reference's call reaches, in the independent-window
variant (a declared divergence from upstream's sequential seek loop): whole-file log-mel with 30 s of
zero padding and a whole-file clamp, fixed 30 s windows, each decoded greedily (temperature 0, no fallback) from
``[sot, lang, transcribe, notimestamps]``.  With several ranks (torch.distributed initialised) windows are sharded
across GPUs (whisper_amd/dist.py).
"""
from __future__ import annotations

import logging
import math
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import checkpoint as ck
from .audio import N_FRAMES, N_SAMPLES, SAMPLE_RATE, load_wav_any
from .tokenizer import LANGUAGE_CODES, Tokenizer, language_index

logger = logging.getLogger(__name__)

_STATE: Dict[str, object] = {"engine": None, "tokenizer": None, "settings": None}


def configure(model: str = "base", checkpoint: Optional[str] = None, vocab: Optional[str] = None,
              precision: str = "bf16", max_batch: int = 16, device: Optional[int] = None,
              allow_synthetic_weights: bool = False, confidence_mode: str = "reference", seed: int = 0, lanes: int = 1,
              expect_beam_search: bool = False):
    """Create the process-global engine (the counterpart of the reference's module-level ``whisper.load_model``,
    ref:speech_recognition.py:20-26).  ``checkpoint`` is a LOCAL openai ``.pt`` or HF ``.safetensors`` file; without
    one, ``allow_synthetic_weights`` must be set explicitly (seeded random weights, for benchmarks and tests).
    ``lanes`` > 1 keeps that many batches of a long file in flight on the GPU (engine replicas on their own host threads and
    streams, whisper_amd/lanes.py: +24 % throughput at 2, +35 % at 4 for large-v3 at 64 windows per batch; one more
    workspace each, the weights are shared); windows mode only, the sequential loop uses the first replica.  bf16 results depend
    on ``lanes`` by rounding noise (it sets the key split of the cross-attention stream), fp32 results do not.
    ``expect_beam_search``: allocate every engine's cross K/V cache now (bf16, max_batch >= 32: 15.7 GB per engine at large-v3 x
    64 rows) instead of inside the first ``beam_size`` call, so that running out of memory shows here."""
    from .engine import WhisperEngine
    if checkpoint:
        dims, sd = ck.load_checkpoint(checkpoint)
    elif allow_synthetic_weights:
        dims = ck.dims_for(model)
        sd = ck.synthetic_state_dict(dims, seed=seed)
    else:
        raise FileNotFoundError(
            f"no local checkpoint configured for Whisper model {model!r} (this build never downloads weights); "
            "set whisper_amd.checkpoint or allow synthetic weights")
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    close()
    tokenizer = Tokenizer(dims.n_vocab, vocab)
    suppress_ids = None
    if tokenizer.has_vocab:
        # upstream derives the non-speech suppress set from the vocabulary it loads (Tokenizer.non_speech_tokens); so do we
        from .engine import default_suppress_ids
        suppress_ids = default_suppress_ids(tokenizer.special, dims.n_vocab, ck.non_speech_tokens_from_vocab(tokenizer.encode))
    def make(n_lanes=1):
        return WhisperEngine(dims, sd, precision=precision, max_batch=max_batch, device=device, suppress_ids=suppress_ids, n_lanes=n_lanes)
    if int(lanes) > 1:
        from .lanes import EngineLanes
        eng = EngineLanes.create(make, int(lanes))
    else:
        eng = make()
    if expect_beam_search:
        for e_i in (eng.engines if int(lanes) > 1 else [eng]):
            e_i.reserve_cross_kv()
    _STATE["engine"] = eng
    _STATE["tokenizer"] = tokenizer
    _STATE["settings"] = {"model": model, "confidence_mode": confidence_mode}
    return eng


def close() -> None:
    eng = _STATE.get("engine")
    if eng is not None:
        eng.close()
    _STATE["engine"] = None


def get_engine():
    if _STATE["engine"] is None:
        raise RuntimeError("Whisper model not initialized")
    return _STATE["engine"]


def get_tokenizer() -> Tokenizer:
    if _STATE["tokenizer"] is None:
        raise RuntimeError("Whisper model not initialized")
    return _STATE["tokenizer"]


def _load_audio(audio: Union[str, Path, np.ndarray, torch.Tensor]) -> np.ndarray:
    if isinstance(audio, (str, Path)):
        pcm, rate = load_wav_any(str(audio))
        if rate == SAMPLE_RATE and pcm.shape[1] == 1:
            return np.ascontiguousarray(pcm[:, 0])    # int16: uploaded as is, /32768 on the GPU
        # any other rate / channel count: the reference's ffmpeg normalisation, done on the device
        # (ref:app/audio/processor.py:136-147); the 16 kHz mono result comes back as the host array the sharding needs
        return get_engine().ingest_pcm16(pcm, rate).cpu().numpy()
    if torch.is_tensor(audio):
        audio = audio.detach().cpu().numpy()
    a = np.asarray(audio)
    if a.ndim != 1:
        raise ValueError("audio must be mono (1-D float32 at 16 kHz)")
    if a.dtype == np.int16:
        return np.ascontiguousarray(a)
    return np.ascontiguousarray(a, dtype=np.float32)


def _prompt_ids(prompt, tokenizer: Tokenizer) -> Optional[List[int]]:
    if prompt is None or (isinstance(prompt, str) and not prompt.strip()):
        return None
    if isinstance(prompt, str):
        if not tokenizer.has_vocab:
            logger.warning("text prompt ignored: no vocabulary file configured (pass token ids instead)")
            return None
        return tokenizer.encode(" " + prompt.strip())
    return [int(p) for p in prompt]


def transcribe(audio, language: Optional[str] = None, prompt: Union[None, str, Sequence[int]] = None,
               mode: str = "windows", **decode_kw) -> Dict[str, object]:
    """``mode="windows"`` (default): independent 30 s windows, batched and sharded across ranks.
    ``mode="sequential"``: upstream's faithful long-form loop (timestamps, previous-text prompt, timestamp-driven seek,
    no-speech skip, temperature fallback; whisper_amd/longform.py) -- per file sequential, so every rank decodes the file
    itself; the result additionally carries ``segments``.

    ``decode_kw``: ``beam_size`` (None = greedy, upstream's default for the reference's call), ``patience``,
    ``max_new_tokens``, ``without_timestamps``.

    Returns ``{"text", "confidence", "chunks_processed"}`` plus ``language``, ``tokens`` (per window),
    ``avg_logprob`` and ``no_speech_prob`` (per window).

    ``confidence``: in the default ``confidence_mode="reference"`` this is 0.0 — what the reference computes, because it
    reads a ``"confidence"`` key that openai-whisper segments do not carry (ref:speech_recognition.py:62-66).
    ``confidence_mode="avg_logprob"`` returns ``exp(mean avg_logprob)`` instead."""
    from . import dist as wdist
    eng = get_engine()
    tokenizer = get_tokenizer()
    pcm_host = _load_audio(audio)
    if pcm_host.shape[0] == 0:
        return {"text": "", "confidence": 0.0, "chunks_processed": 0, "language": language or "en", "tokens": [],
                "avg_logprob": [], "no_speech_prob": []}
    cmode = (_STATE["settings"] or {}).get("confidence_mode", "reference")
    lang_idx = None if language is None else language_index(language, eng.dims.n_vocab)
    if mode == "sequential":
        from .longform import transcribe_sequential
        r = transcribe_sequential(eng, [pcm_host], tokenizer.decode, language_index=lang_idx,
                                  initial_prompts=[_prompt_ids(prompt, tokenizer)], **decode_kw)[0]
        lps = [sg["avg_logprob"] for sg in r["segments"]]
        conf = 0.0 if cmode == "reference" or not lps else float(math.exp(sum(lps) / len(lps)))
        return {"text": r["text"].strip(), "confidence": conf, "chunks_processed": int(r["windows"]),
                "language": LANGUAGE_CODES[r["language_index"]], "tokens": r["tokens"], "segments": r["segments"],
                "avg_logprob": lps, "no_speech_prob": [sg["no_speech_prob"] for sg in r["segments"]]}
    if mode != "windows":
        raise ValueError("mode must be 'windows' or 'sequential'")
    # language None: detected inside the sharded pass, on the first 30 s of the whole-file mel (upstream's rule; one rank
    # detects, the index is broadcast)
    pids = _prompt_ids(prompt, tokenizer)
    init = (lambda li: eng.initial_tokens(li, prompt_ids=pids)) if lang_idx is None else eng.initial_tokens(lang_idx, prompt_ids=pids)
    res = wdist.transcribe_sharded(eng, pcm_host, init, **decode_kw)
    if lang_idx is None:
        lang_idx = int(res["language_index"])
    texts = [tokenizer.decode(t) for t in res["tokens"]]
    avg_lp = [float(s) / (len(t) + 1) for s, t in zip(res["sum_logprobs"], res["tokens"])]
    confidence = 0.0 if cmode == "reference" or not avg_lp else float(math.exp(sum(avg_lp) / len(avg_lp)))
    return {
        "text": "".join(texts).strip(), "confidence": confidence, "chunks_processed": int(res["n_windows"]),
        "language": LANGUAGE_CODES[lang_idx], "tokens": res["tokens"], "avg_logprob": avg_lp,
        "no_speech_prob": [float(x) for x in res["no_speech_probs"]],
    }
