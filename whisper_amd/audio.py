"""Host-side audio helpers: format contract, filterbank, windowing arithmetic, synthetic signals.

Format contract from the reference's normaliser (ref:app/audio/processor.py:22-23,141-143): 16 kHz, mono, PCM16.
"""
from __future__ import annotations

import math
import wave
from typing import Tuple

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE     # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH         # 3000
N_FREQ = N_FFT // 2 + 1


def slaney_mel_filterbank(n_mels: int) -> np.ndarray:
    """[n_mels, 201] float32 triangular filters on the Slaney mel scale with Slaney area normalisation
    (what openai-whisper ships as assets/mel_filters.npz; algorithm: hf:audio_utils.py:638-729)."""
    def hz_to_mel(f):
        f = np.asarray(f, np.float64)
        lin = f * 3.0 / 200.0
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), lin)

    def mel_to_hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * 200.0 / 3.0)

    bins = np.linspace(0.0, SAMPLE_RATE // 2, N_FREQ)
    edges = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(8000.0), n_mels + 2))
    out = np.zeros((n_mels, N_FREQ), np.float64)
    for m in range(n_mels):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        rise = (bins - lo) / (ce - lo)
        fall = (hi - bins) / (hi - ce)
        out[m] = np.maximum(0.0, np.minimum(rise, fall)) * (2.0 / (hi - lo))
    return out.astype(np.float32)


def n_mel_frames(n_samples: int) -> int:
    """Frames of the padded file: upstream appends 30 s of zeros and drops the last STFT frame."""
    return (n_samples + N_SAMPLES) // HOP_LENGTH


def n_windows(n_samples: int) -> int:
    """Independent 30 s windows covering the content frames."""
    content = n_samples // HOP_LENGTH
    return max(1, math.ceil(content / N_FRAMES))


def synthetic_audio(seed: int, n_samples: int = N_SAMPLES) -> np.ndarray:
    """SURVEY.md §8d benchmark signal: 220 Hz + 1330 Hz sines plus N(0, 0.05²) noise, float32."""
    n = np.arange(n_samples, dtype=np.float64)
    rng = np.random.default_rng(seed)
    x = 0.3 * np.sin(2 * np.pi * 220.0 * n / SAMPLE_RATE) + 0.2 * np.sin(2 * np.pi * 1330.0 * n / SAMPLE_RATE)
    return (x + 0.05 * rng.standard_normal(n_samples)).astype(np.float32)


def load_wav_pcm16(path: str) -> np.ndarray:
    """Read a 16 kHz mono PCM16 WAV (the output format of ref:app/audio/processor.py:123-150) as float32 in
    [-1, 1) — the same /32768 scaling upstream ``load_audio`` applies after ffmpeg."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError(f"{path}: expected 16-bit PCM, got sample width {w.getsampwidth()}")
        if w.getframerate() != SAMPLE_RATE:
            raise ValueError(f"{path}: expected {SAMPLE_RATE} Hz, got {w.getframerate()} (normalise with app/audio first)")
        ch = w.getnchannels()
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    if ch > 1:
        raw = raw.reshape(-1, ch).astype(np.float32).mean(axis=1)
        return (raw / 32768.0).astype(np.float32)
    return raw.astype(np.float32) / 32768.0


def load_wav_pcm16_raw(path: str) -> np.ndarray:
    """Same file contract as :func:`load_wav_pcm16`, but returns the int16 samples untouched (mono files only) so the
    /32768 conversion can run on the GPU (``WhisperEngine.pcm16_to_f32``)."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError(f"{path}: expected 16-bit PCM, got sample width {w.getsampwidth()}")
        if w.getframerate() != SAMPLE_RATE:
            raise ValueError(f"{path}: expected {SAMPLE_RATE} Hz, got {w.getframerate()} (normalise with app/audio first)")
        if w.getnchannels() != 1:
            raise ValueError(f"{path}: expected mono")
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")


def load_wav_any(path: str) -> Tuple[np.ndarray, int]:
    """Read ANY 16-bit PCM WAV -- the input side of the reference's normaliser (ref:app/audio/processor.py:123-150) -- as
    ``(int16 [n_frames, n_channels], sample_rate)``.  Rate conversion and the mono mix then run on the GPU
    (``WhisperEngine.ingest_pcm16``), not in an ffmpeg subprocess."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError(f"{path}: expected 16-bit PCM, got sample width {w.getsampwidth()}")
        ch, rate = w.getnchannels(), w.getframerate()
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return raw.reshape(-1, ch), rate


def resample_ratio(src_rate: int) -> Tuple[int, int]:
    """(up, down) with up/down = 16000/src_rate in lowest terms (44100 Hz -> 160/441, 48000 Hz -> 1/3, 8000 Hz -> 2/1)."""
    if src_rate <= 0:
        raise ValueError("sample rate must be positive")
    g = math.gcd(SAMPLE_RATE, int(src_rate))
    return SAMPLE_RATE // g, int(src_rate) // g


def resample_out_len(n_frames: int, up: int, down: int) -> int:
    return -(-n_frames * up // down)


def resample_filter(up: int, down: int) -> Tuple[np.ndarray, int]:
    """Centred low-pass taps of the polyphase resampler, ``(h[2*half_len + 1] float32, half_len)``: the design
    scipy.signal.resample_poly uses by default -- a Kaiser(beta = 5) windowed sinc with cut-off 1/max(up, down) of the
    up-sampled Nyquist frequency, half_len = 10*max(up, down), unit DC gain, times ``up`` -- written out with numpy so
    the product does not depend on scipy (tests/test_host_logic.py checks it against scipy.signal.firwin)."""
    q = max(up, down)
    half_len = 10 * q
    k = np.arange(-half_len, half_len + 1, dtype=np.float64)
    fc = 1.0 / q
    h = fc * np.sinc(fc * k) * np.kaiser(2 * half_len + 1, 5.0)
    h /= h.sum()
    return (h * up).astype(np.float32), half_len


def save_wav_pcm16(path: str, audio: np.ndarray) -> None:
    pcm = np.clip(np.round(np.asarray(audio, np.float64) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(SAMPLE_RATE)
        w.writeframes(pcm.tobytes())


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: rank g of G gets [g*ceil(W/G), ...)."""
    per = math.ceil(n_items / world) if n_items > 0 else 0
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
