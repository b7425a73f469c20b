"""Mirror of the reference's large-file path (ref:app/transcription/openai_whisper.py:215-337) on the MI355X backend.

``chunk_large_audio`` and ``transcribe_large_audio`` keep the reference's names, arguments, return shapes and arithmetic:
files up to 25 MB go through one call (:286-290); larger files are cut into pieces of at most ``chunk_size_mb`` by
DURATION, ``chunk_ms = int(chunk_size_mb / file_size_mb * total_ms)`` (:241-242), written as ``<stem>_chunks/chunk_NNN<suffix>``
(:245-255), transcribed IN ORDER with the last 20 words of the previous piece appended to the prompt as
``"<prompt> Previous context: <words>"`` (:300-310), then joined with spaces, mean confidence, modal language, source
``"chunked"`` (:317-321); the piece files and the directory are removed afterwards (:323-337).

Differences, both declared: (1) the reference cuts with pydub (absent here, and ffmpeg with it); this module cuts the
format the reference's own front end produces -- 16 kHz mono PCM16 WAV (ref:app/audio/processor.py:141-143) -- with the
standard-library ``wave`` module, at the same millisecond boundaries; (2) the reference needs this path only because
the OpenAI API refuses uploads over 25 MB; the local engine has no such limit (one call shards any file into 30 s
windows across the GPUs, whisper_amd/dist.py), so ``transcribe_large_audio`` exists for drop-in compatibility, and each
piece is itself decoded as a batch of windows.
"""
from __future__ import annotations

import logging
import wave
from pathlib import Path
from typing import List, Optional, Tuple

logger = logging.getLogger(__name__)

MAX_SINGLE_CALL_MB = 25          # ref:openai_whisper.py:288
# how a piece is decoded: "sequential" = upstream's own long-form loop (what the reference's local call runs), "windows" =
# independent 30 s windows, batched and sharded (whisper_amd.transcribe)
DECODE_MODE = "sequential"


async def chunk_large_audio(audio_path: Path, chunk_size_mb: float = 20.0) -> List[Path]:
    audio_path = Path(audio_path)
    try:
        file_size_mb = audio_path.stat().st_size / (1024 * 1024)
        if file_size_mb <= chunk_size_mb:
            return [audio_path]                                   # no need to split (:237-238)
        with wave.open(str(audio_path), "rb") as w:
            params = w.getparams()
            rate, n_frames = w.getframerate(), w.getnframes()
            total_duration_ms = int(n_frames * 1000 / rate)       # pydub's len(audio): whole milliseconds
            chunk_duration_ms = int((chunk_size_mb / file_size_mb) * total_duration_ms)
            if chunk_duration_ms <= 0:
                raise ValueError("chunk duration rounds to zero")
            chunk_dir = audio_path.parent / f"{audio_path.stem}_chunks"
            chunk_dir.mkdir(exist_ok=True)
            chunks: List[Path] = []
            for i, start_ms in enumerate(range(0, total_duration_ms, chunk_duration_ms)):
                end_ms = min(start_ms + chunk_duration_ms, total_duration_ms)
                f0, f1 = start_ms * rate // 1000, end_ms * rate // 1000
                w.setpos(f0)
                data = w.readframes(f1 - f0)
                chunk_path = chunk_dir / f"chunk_{i:03d}{audio_path.suffix}"
                with wave.open(str(chunk_path), "wb") as o:
                    o.setnchannels(params.nchannels); o.setsampwidth(params.sampwidth); o.setframerate(rate)
                    o.writeframes(data)
                chunks.append(chunk_path)
        logger.info(f"Split audio into {len(chunks)} chunks")
        return chunks
    except Exception as e:   # same as the reference: log and re-raise (:262-265)
        logger.error(f"Audio chunking failed: {str(e)}")
        raise


async def transcribe_audio_local(audio_path: Path, detailed: bool = True, language: Optional[str] = None,
                                 prompt: Optional[str] = None, **decode_kw) -> Tuple[str, float, str, str]:
    """The local branch of the reference's ``transcribe_audio_hybrid`` (ref:openai_whisper.py:138-153): source ``"local"`` on
    success, ``("", 0.0, "unknown", "failed")`` if the local call raises.  Unlike the reference, which drops ``language`` and
    ``prompt`` before the local call (:142), both reach the engine here."""
    from . import transcribe as tx
    try:
        decode_kw.setdefault("mode", DECODE_MODE)
        r = tx.transcribe(str(audio_path), language=language, prompt=prompt, **decode_kw)
        text, conf = r["text"].strip(), float(r["confidence"])
        if not detailed:                                          # ref:speech_recognition.py:69-72
            text, conf = "Voice authentication required for full transcription", min(conf, 0.5)
        return text, conf, r.get("language", "unknown"), "local"
    except Exception as e:   # noqa: BLE001
        logger.error(f"Local Whisper fallback also failed: {str(e)}")
        return "", 0.0, "unknown", "failed"


def _carry_prompt(prompt: Optional[str], previous_text: Optional[str]) -> Optional[str]:
    """Prompt of a piece: the caller's prompt, plus the last 20 words of the piece before it (ref :300-310)."""
    if previous_text is None:
        return prompt
    tail = " ".join(previous_text.split()[-20:])
    return f"{prompt or ''} Previous context: {tail}"


def _remove_pieces(pieces: List[Path], original: Path) -> None:
    """Delete the piece files and their (then empty) directory; never the caller's file (ref :323-337)."""
    own = [p for p in pieces if p != original]
    for p in own:
        try:
            p.unlink()
        except OSError:
            pass
    if own and own[0].parent.exists():
        try:
            own[0].parent.rmdir()
        except OSError:
            pass


async def transcribe_large_audio(audio_path: Path, detailed: bool = True, language: Optional[str] = None,
                                 prompt: Optional[str] = None, **decode_kw) -> Tuple[str, float, str, str]:
    """``decode_kw`` (not in the reference's signature): passed to ``whisper_amd.transcribe.transcribe`` for every piece
    (``mode``, ``max_new_tokens``, ``temperatures`` ...)."""
    audio_path = Path(audio_path)
    if audio_path.stat().st_size / (1024 * 1024) <= MAX_SINGLE_CALL_MB:
        return await transcribe_audio_local(audio_path, detailed, language, prompt, **decode_kw)
    pieces = await chunk_large_audio(audio_path)
    results: List[Tuple[str, float, str]] = []
    try:
        for piece in pieces:
            carried = _carry_prompt(prompt, results[-1][0] if results else None)
            text, conf, lang, _ = await transcribe_audio_local(piece, detailed, language, carried, **decode_kw)
            results.append((text, conf, lang))
    finally:
        _remove_pieces(pieces, audio_path)
    if not results:
        return "", 0.0, "unknown", "chunked"
    langs = [r[2] for r in results]
    return (" ".join(r[0] for r in results), sum(r[1] for r in results) / len(results),
            max(set(langs), key=langs.count), "chunked")
