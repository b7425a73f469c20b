"""Drop-in for the reference's ``app/transcription/speech_recognition.py`` backed by the MI355X engine.

Same module surface (ref:app/transcription/speech_recognition.py): globals ``WHISPER_MODEL``, ``LANGUAGE``, ``device``,
``whisper_model``; ``async transcribe_audio(audio_path, detailed=True) -> (text, confidence, language)``;
``get_supported_languages()``.  Same error behaviour: a failed load leaves ``whisper_model = None`` and never breaks
the import (:20-26); ``transcribe_audio`` raises ``RuntimeError("Whisper model not initialized")`` in that case (:43-44)
and turns every other failure into ``("Error during transcription", 0.0, "unknown")`` (:77-79); ``detailed=False``
replaces the text and caps the confidence at 0.5 while still running the model (:69-72).

Install: ``sys.modules["app.transcription.speech_recognition"] = whisper_amd.speech_recognition``
before ``app.transcription.openai_whisper`` is imported (it binds ``transcribe_audio`` at import, ref:openai_whisper.py:14).
Build-specific settings live in a NEW config section so existing keys keep their meaning:
    whisper_amd: {checkpoint: /path/model.pt, vocab: /path/multilingual.tiktoken, precision: bf16, max_batch: 16,
                  allow_synthetic_weights: false, confidence_mode: reference,
                  decode_mode: sequential | windows, beam_size: null}
``decode_mode`` defaults to ``sequential``: upstream's own long-form loop with upstream's defaults, which is what the
reference's call runs.  ``windows`` is the opt-in throughput mode (independent 30 s windows, batched, sharded over ranks).
"""
from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import Any, Dict, Tuple

logger = logging.getLogger(__name__)


def _parse_env_value(value: str) -> Any:
    if value.lower() == "true":
        return True
    if value.lower() == "false":
        return False
    for cast in (int, float):
        try:
            return cast(value)
        except ValueError:
            pass
    return value


def load_config() -> Dict[str, Any]:
    """The reference's loader when the service package is importable, else the same YAML + ``WHISPER_<SECTION>_<KEY>``
    rules (ref:app/utils/config.py:16-60; the env key is split with ``split("_", 2)``, :43)."""
    try:
        from app.utils.config import load_config as ref_load   # type: ignore
        return ref_load()
    except Exception:
        pass
    import yaml
    cfg: Dict[str, Any] = {}
    path = Path(os.environ.get("CONFIG_PATH") or "config.yaml")
    if path.exists():
        try:
            with open(path, "r") as f:
                cfg = yaml.safe_load(f) or {}
        except Exception as e:  # same tolerance as the reference
            logger.error(f"Failed to load configuration from {path}: {e}")
    for k, v in os.environ.items():
        if k.startswith("WHISPER_"):
            parts = k.lower().split("_", 2)[1:]
            if len(parts) == 1:
                cfg[parts[0]] = _parse_env_value(v)
            elif len(parts) == 2:
                cfg.setdefault(parts[0], {})
                if isinstance(cfg[parts[0]], dict):
                    cfg[parts[0]][parts[1]] = _parse_env_value(v)
    return cfg


config = load_config()
WHISPER_MODEL = (config.get("transcription") or {}).get("whisper_model", "base")
LANGUAGE = (config.get("transcription") or {}).get("language", None)   # None = auto-detection
_AMD = config.get("whisper_amd") or {}

try:
    import torch
    from . import transcribe as _tx
    device = "cuda" if torch.cuda.is_available() else "cpu"
    if device != "cuda":
        raise RuntimeError("no ROCm GPU visible; whisper_amd has no CPU path")
    whisper_model = _tx.configure(
        model=WHISPER_MODEL, checkpoint=_AMD.get("checkpoint"), vocab=_AMD.get("vocab"),
        precision=_AMD.get("precision", "bf16"), max_batch=int(_AMD.get("max_batch", 16)),
        allow_synthetic_weights=bool(_AMD.get("allow_synthetic_weights", False)),
        confidence_mode=_AMD.get("confidence_mode", "reference"), lanes=int(_AMD.get("lanes", 1)),
        expect_beam_search=bool(_AMD.get("beam_size")) and str(_AMD.get("decode_mode", "sequential")) == "windows")
    logger.info(f"Whisper model '{WHISPER_MODEL}' loaded on {device}")
    if str(_AMD.get("decode_mode", "sequential")) == "sequential" and (_AMD.get("beam_size") or int(_AMD.get("lanes", 1)) > 1):
        # the default mode changed from "windows" to "sequential" in round 2: say what that does to an older configuration
        logger.warning("whisper_amd: decode_mode is 'sequential' (the default: upstream's own loop, greedy with temperature fallback); "
                       "whisper_amd.beam_size and whisper_amd.lanes only apply to decode_mode 'windows' and are ignored")
except Exception as e:   # noqa: BLE001 — same as the reference: never fail the import
    logger.error(f"Failed to load Whisper model: {str(e)}")
    device = "cpu"
    whisper_model = None


async def transcribe_audio(audio_path: Path, detailed: bool = True) -> Tuple[str, float, str]:
    if whisper_model is None:
        raise RuntimeError("Whisper model not initialized")
    try:
        # decode_mode "sequential" (default): upstream's own long-form loop with upstream's defaults (timestamps,
        # previous-text prompt, timestamp-driven seek, no-speech skip, temperature fallback) -- what the reference's call
        # runs (ref:speech_recognition.py:48-52 passes only language, task, fp16).  "windows": independent 30 s windows,
        # batched and sharded across ranks (the throughput mode; a declared divergence)
        kw = {}
        mode = str(_AMD.get("decode_mode", "sequential"))
        if _AMD.get("beam_size") and mode == "windows":
            kw["beam_size"] = int(_AMD["beam_size"])
        result = _tx.transcribe(str(audio_path), language=LANGUAGE, mode=mode, **kw)
        text = result["text"].strip()
        language = result.get("language", "unknown")
        confidence = float(result.get("confidence", 0.0))
        if not detailed:
            text = "Voice authentication required for full transcription"
            confidence = min(confidence, 0.5)
        logger.info(f"Transcription completed. Language: {language}, Confidence: {confidence:.4f}")
        return text, confidence, language
    except Exception as e:   # noqa: BLE001
        logger.error(f"Error transcribing audio: {str(e)}")
        return "Error during transcription", 0.0, "unknown"


def get_supported_languages() -> Dict[str, str]:
    from .tokenizer import LANGUAGES, supported_languages
    if whisper_model is not None:
        return supported_languages(whisper_model.dims.n_vocab)
    return dict(LANGUAGES)
