"""Host-side logic that needs no GPU: checkpoints, token tables, tokenizer, audio helpers, the drop-in module contract
(ref:app/transcription/speech_recognition.py:29-89; contract shape pinned by ref:tests/test_hybrid_stt.py:55,130-136)."""
import asyncio
import base64
import importlib
import os
import sys

from pathlib import Path

import numpy as np
import pytest
import torch

from whisper_amd import audio as wa
from whisper_amd import checkpoint as ck
from whisper_amd.tokenizer import LANGUAGES, Tokenizer, language_index, supported_languages


def test_dims_and_special_tokens():
    d = ck.dims_for("large-v3")
    assert (d.n_audio_state, d.n_audio_head, d.n_audio_layer, d.n_text_layer, d.n_mels, d.n_vocab) == (1280, 20, 32, 32, 128, 51866)
    assert ck.dims_for("turbo").n_text_layer == 4 and ck.dims_for("tiny").n_audio_state == 384
    t = ck.special_tokens_for(51865)
    assert (t.eot, t.sot, t.translate, t.transcribe, t.sot_lm, t.sot_prev, t.no_speech, t.no_timestamps, t.timestamp_begin) == \
        (50257, 50258, 50358, 50359, 50360, 50361, 50362, 50363, 50364)
    t3 = ck.special_tokens_for(51866)
    assert (t3.no_timestamps, t3.timestamp_begin, t3.n_langs) == (50364, 50365, 100)
    assert t.lang(0) == 50259 and t3.lang(99) == 50358
    with pytest.raises(ValueError):
        t.lang(99)
    with pytest.raises(KeyError):
        ck.dims_for("gigantic")


def test_english_only_layout_and_suppress_tables():
    """The three upstream vocabulary layouts and their non-speech suppress sets.  The tables must equal transformers'
    generation defaults for the original checkpoints minus the control ids (hf:models/whisper/configuration_whisper.py:23-44)
    -- the multilingual table for 51865 / 51866, the GPT-2 table only for English-only checkpoints."""
    te = ck.special_tokens_for(51864)
    assert (te.eot, te.sot, te.translate, te.transcribe, te.sot_lm, te.sot_prev, te.no_speech, te.no_timestamps, te.timestamp_begin) == \
        (50256, 50257, 50357, 50358, 50359, 50360, 50361, 50362, 50363)
    assert not ck.is_multilingual(51864) and ck.is_multilingual(51865) and ck.is_multilingual(51866)
    with pytest.raises(ValueError):
        ck.special_tokens_for(32000)
    multi, en = ck.non_speech_suppress_ids(51865), ck.non_speech_suppress_ids(51864)
    assert ck.non_speech_suppress_ids(51866) == multi and multi != en
    assert {359, 503, 522, 50254} <= set(multi) and not {357, 366, 438} & set(multi)
    cw = pytest.importorskip("transformers.models.whisper.configuration_whisper")
    assert multi == [i for i in cw.NON_SPEECH_TOKENS_MULTI if i < ck.special_tokens_for(51865).eot]
    assert en == [i for i in cw.NON_SPEECH_TOKENS if i < te.eot]
    # upstream's "-1" set = the table + the six control tokens; HF's list holds four of them already
    from whisper_amd.engine import default_suppress_ids
    t = ck.special_tokens_for(51865)
    full = default_suppress_ids(t, 51865)
    assert set(cw.NON_SPEECH_TOKENS_MULTI) <= set(full)
    assert set(full) - set(cw.NON_SPEECH_TOKENS_MULTI) == {t.translate, t.transcribe}
    assert set(default_suppress_ids(te, 51864)) - set(cw.NON_SPEECH_TOKENS) == {te.translate, te.transcribe}


def test_non_speech_tokens_from_a_vocabulary(tmp_path):
    """Upstream's derivation rule on a toy tiktoken vocabulary: single-token symbols (alone or with a leading space) and the
    first token of every musical-note character are suppressed, " -" and " '" always, plain "-" and "'" never."""
    words = [b"\xe2\x99", b" -", b" '", b"[", b" [", b"((", b" ((", "♪".encode(), " ♪".encode(), b"<<", "♪♪".encode(), b"he"]
    lines = [base64.b64encode(bytes([i])) + b" " + str(i).encode() for i in range(256)]
    lines += [base64.b64encode(w) + b" " + str(256 + i).encode() for i, w in enumerate(words)]
    vp = tmp_path / "sym.tiktoken"
    vp.write_bytes(b"\n".join(lines) + b"\n")
    tk = Tokenizer(51865, str(vp))
    got = ck.non_speech_tokens_from_vocab(tk.encode)
    ids = {w: 256 + i for i, w in enumerate(words)}
    for w in (b" -", b" '", b"[", b" [", b"((", b" ((", "♪".encode(), " ♪".encode(), b"<<", "♪♪".encode()):
        assert ids[w] in got, w
    assert ids[b"he"] not in got and ord("-") not in got and ord("'") not in got
    assert ord("#") in got and ord("{") in got                    # one-byte symbols are single tokens of the byte alphabet
    assert ids[b"\xe2\x99"] in got                                # a note without a token of its own: its FIRST token
    assert got == sorted(set(got))


def test_beam_max_candidates_is_upstreams_round():
    """max_candidates = round(beam_size * patience) with Python's round (ties to even, double arithmetic) -- the value the
    host now computes once and hands to wx_decode_beam."""
    assert [int(round(n * p)) for n, p in ((3, 1.5), (5, 1.5), (7, 1.5), (5, 1.7), (5, 1.0), (8, 2.0))] == [4, 8, 10, 8, 5, 16]


def test_name_mapping_roundtrip_and_checkpoint_io(tmp_path):
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=3)
    assert set(sd) == set(ck.tensor_shapes(dims))
    for k in sd:
        assert ck.hf_to_canonical_name(ck.canonical_to_hf_name(k)) == k
    assert ck.hf_to_canonical_name("proj_out.weight") is None
    assert ck.dims_from_state_dict(sd) == dims
    p = str(tmp_path / "m.pt")
    ck.save_checkpoint(p, dims, sd)
    d2, sd2 = ck.load_checkpoint(p)
    assert d2 == dims and all(torch.equal(sd[k], sd2[k]) for k in sd)
    from safetensors.torch import save_file
    hf = {ck.canonical_to_hf_name(k): v.contiguous() for k, v in sd.items()}
    ps = str(tmp_path / "model.safetensors")
    save_file(hf, ps)
    d3, sd3 = ck.load_checkpoint(ps)
    assert d3 == dims and all(torch.equal(sd[k], sd3[k]) for k in sd)
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint(str(tmp_path / "missing.pt"))


def test_synthetic_weights_are_seeded():
    dims = ck.dims_for("micro")
    a, b, c = ck.synthetic_state_dict(dims, 0), ck.synthetic_state_dict(dims, 0), ck.synthetic_state_dict(dims, 1)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["decoder.token_embedding.weight"], c["decoder.token_embedding.weight"])
    assert torch.equal(a["encoder.positional_embedding"], ck.sinusoids(1500, 128))


def test_languages_and_tokenizer(tmp_path):
    assert len(LANGUAGES) == 100 and list(LANGUAGES)[:3] == ["en", "zh", "de"] and list(LANGUAGES)[-1] == "yue"
    assert len(supported_languages(51865)) == 99 and "yue" in supported_languages(51866)
    assert language_index("ru", 51865) == 4 and language_index("Russian", 51865) == 4
    with pytest.raises(ValueError):
        language_index("yue", 51865)
    tk = Tokenizer(51865)
    assert tk.decode([10, 50257, 11, 50400]) == "<|10|><|11|>"       # ids >= eot are dropped, placeholder rendering
    # a toy tiktoken-format vocabulary: 256 bytes + two merges
    lines = [base64.b64encode(bytes([i])) + b" " + str(i).encode() for i in range(256)]
    lines += [base64.b64encode(b"he") + b" 256", base64.b64encode(b" w") + b" 257", base64.b64encode(b"hel") + b" 258"]
    vp = tmp_path / "toy.tiktoken"
    vp.write_bytes(b"\n".join(lines) + b"\n")
    tk = Tokenizer(51865, str(vp))
    ids = tk.encode("hello world")
    assert tk.decode(ids) == "hello world" and 258 in ids and 257 in ids
    assert tk.decode([0xE2, 0x82, 0xAC]) == "€"


def test_audio_helpers(tmp_path):
    assert wa.n_windows(1) == 1 and wa.n_windows(480000) == 1 and wa.n_windows(480160) == 2 and wa.n_windows(16000 * 3600) == 120
    assert wa.n_mel_frames(480000) == 6000
    assert [wa.shard_range(120, r, 8) for r in (0, 7)] == [(0, 15), (105, 120)]
    assert [wa.shard_range(3, r, 2) for r in (0, 1)] == [(0, 2), (2, 3)]
    assert wa.shard_range(1, 3, 4) == (1, 1)
    x = wa.synthetic_audio(0, 16000)
    p = str(tmp_path / "a.wav")
    wa.save_wav_pcm16(p, x)
    y = wa.load_wav_pcm16(p)
    assert y.shape == x.shape and np.abs(x - y).max() <= 1.0 / 32768 + 1e-7
    fb = wa.slaney_mel_filterbank(80)
    assert fb.shape == (80, 201) and (fb >= 0).all() and (fb.sum(axis=1) > 0).all()


def test_resampler_host_side(tmp_path):
    """Rate ratios, output lengths, the numpy filter design against scipy.signal.firwin (the design scipy.signal.
    resample_poly uses by default), and the any-WAV reader."""
    import wave
    import scipy.signal as ss
    assert wa.resample_ratio(44100) == (160, 441) and wa.resample_ratio(48000) == (1, 3)
    assert wa.resample_ratio(8000) == (2, 1) and wa.resample_ratio(16000) == (1, 1)
    assert wa.resample_out_len(441, 160, 441) == 160 and wa.resample_out_len(442, 160, 441) == 161
    for up, down in ((160, 441), (1, 3), (2, 1), (320, 441), (2, 3)):
        h, half_len = wa.resample_filter(up, down)
        ref = ss.firwin(2 * half_len + 1, 1.0 / max(up, down), window=("kaiser", 5.0)) * up
        assert half_len == 10 * max(up, down) and h.dtype == np.float32
        assert np.abs(h - ref).max() < 1e-7
    h, half_len = wa.resample_filter(1, 1)            # same rate: the filter degenerates to a unit impulse
    assert h[half_len] == 1.0 and np.abs(np.delete(h, half_len)).max() < 1e-7
    pcm = np.random.default_rng(0).integers(-3000, 3000, size=(1234, 2), dtype=np.int16)
    p = str(tmp_path / "s.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
    got, rate = wa.load_wav_any(p)
    assert rate == 44100 and np.array_equal(got, pcm)


def _fresh_dropin(monkeypatch, **env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sys.modules.pop("whisper_amd.speech_recognition", None)
    return importlib.import_module("whisper_amd.speech_recognition")


def test_dropin_contract_without_model(monkeypatch, tmp_path):
    monkeypatch.chdir(tmp_path)     # no config.yaml
    sr = _fresh_dropin(monkeypatch, WHISPER_TRANSCRIPTION_WHISPER_MODEL="tiny", WHISPER_TRANSCRIPTION_LANGUAGE="ru")
    assert sr.WHISPER_MODEL == "tiny" and sr.LANGUAGE == "ru"       # WHISPER_<SECTION>_<KEY> split("_", 2) rule
    assert sr.whisper_model is None                                  # load failed (no GPU / no checkpoint): import survives
    with pytest.raises(RuntimeError, match="Whisper model not initialized"):
        asyncio.run(sr.transcribe_audio(tmp_path / "x.wav"))
    assert sr.get_supported_languages()["en"] == "english"


def test_dropin_result_shaping(monkeypatch, tmp_path):
    monkeypatch.chdir(tmp_path)
    sr = _fresh_dropin(monkeypatch)
    calls = []

    seen = {}

    def fake_transcribe(path, language=None, prompt=None, mode="windows", **kw):
        seen["mode"] = mode
        calls.append((path, language))
        return {"text": "  тестовый текст ", "confidence": 0.95, "chunks_processed": 1, "language": "ru"}

    monkeypatch.setattr(sr, "whisper_model", object())
    monkeypatch.setattr(sr._tx, "transcribe", fake_transcribe)
    out = asyncio.run(sr.transcribe_audio(tmp_path / "a.wav", detailed=True))
    assert out == ("тестовый текст", 0.95, "ru") and isinstance(out[1], float)
    assert seen["mode"] == "sequential"      # default decode_mode: upstream's own long-form loop, what the reference's call runs
    out = asyncio.run(sr.transcribe_audio(tmp_path / "a.wav", detailed=False))       # privacy branch still runs the model
    assert out == ("Voice authentication required for full transcription", 0.5, "ru") and len(calls) == 2

    def boom(*a, **k):
        raise ValueError("decode failed")

    monkeypatch.setattr(sr._tx, "transcribe", boom)
    assert asyncio.run(sr.transcribe_audio(tmp_path / "a.wav")) == ("Error during transcription", 0.0, "unknown")


def test_transcribe_requires_configuration():
    from whisper_amd import transcribe as tx
    tx.close()
    with pytest.raises(RuntimeError, match="Whisper model not initialized"):
        tx.transcribe(np.zeros(16000, np.float32))
    with pytest.raises(FileNotFoundError):
        tx.configure(model="base")       # no checkpoint and synthetic weights not allowed: never downloads


def test_chunk_large_audio_matches_reference_arithmetic(tmp_path):
    """ref:app/transcription/openai_whisper.py:236-255: no split at or under chunk_size_mb; otherwise pieces of
    int(chunk_size_mb / file_size_mb * total_ms) milliseconds, named chunk_NNN<suffix> under <stem>_chunks/, covering the
    file exactly once and in order."""
    import asyncio
    import wave
    from whisper_amd.audio import load_wav_pcm16_raw
    from whisper_amd.large_audio import chunk_large_audio
    rng = np.random.default_rng(0)
    pcm = rng.integers(-2000, 2000, size=16000 * 75, dtype=np.int16)      # 75 s = 2.29 MiB
    p = tmp_path / "long.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    assert asyncio.run(chunk_large_audio(p, chunk_size_mb=20.0)) == [p]
    chunks = asyncio.run(chunk_large_audio(p, chunk_size_mb=1.0))
    size_mb = p.stat().st_size / (1024 * 1024)
    chunk_ms = int(1.0 / size_mb * 75000)
    assert len(chunks) == -(-75000 // chunk_ms)
    assert [c.name for c in chunks] == [f"chunk_{i:03d}.wav" for i in range(len(chunks))]
    assert chunks[0].parent == tmp_path / "long_chunks"
    back = np.concatenate([load_wav_pcm16_raw(str(c)) for c in chunks])
    assert np.array_equal(back, pcm)
    for c in chunks[:-1]:
        assert len(load_wav_pcm16_raw(str(c))) == chunk_ms * 16


def test_transcribe_large_audio_prompt_carry_and_cleanup(tmp_path, monkeypatch):
    """ref:openai_whisper.py:268-337 with the engine stubbed: one call under 25 MB; above it, pieces in order, the last 20
    words of the previous piece in the prompt, text joined with spaces, mean confidence, modal language, source "chunked",
    piece files and directory removed."""
    import asyncio
    import wave
    from whisper_amd import large_audio as la
    from whisper_amd import transcribe as tx
    p = tmp_path / "big.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(np.zeros(16000 * 40, np.int16).tobytes())
    calls = []

    def fake(audio, language=None, prompt=None, **kw):
        calls.append((Path(audio).name, language, prompt))
        n = len(calls)
        return {"text": " ".join(f"w{n}_{i}" for i in range(30)), "confidence": 0.2 * n, "language": "ru" if n != 2 else "en",
                "chunks_processed": 1}
    monkeypatch.setattr(tx, "transcribe", fake)
    out = asyncio.run(la.transcribe_large_audio(p, True, "ru", "ctx"))
    assert out[3] == "local" and calls == [("big.wav", "ru", "ctx")]
    calls.clear()
    monkeypatch.setattr(la, "MAX_SINGLE_CALL_MB", 0.5)

    async def small_chunks(path, chunk_size_mb=20.0):
        return await orig(path, 0.5)
    orig = la.chunk_large_audio
    monkeypatch.setattr(la, "chunk_large_audio", small_chunks)
    text, conf, lang, source = asyncio.run(la.transcribe_large_audio(p, True, None, "ctx"))
    assert source == "chunked" and len(calls) == 3 and lang == "ru"
    assert [c[0] for c in calls] == ["chunk_000.wav", "chunk_001.wav", "chunk_002.wav"]
    assert calls[0][2] == "ctx"
    assert calls[1][2] == "ctx Previous context: " + " ".join(f"w1_{i}" for i in range(10, 30))
    assert abs(conf - (0.2 + 0.4 + 0.6) / 3) < 1e-9
    assert text.split()[0] == "w1_0" and text.split()[-1] == "w3_29" and len(text.split()) == 90
    assert not (tmp_path / "big_chunks").exists()
    # detailed=False: fixed text, confidence capped (ref:speech_recognition.py:69-72); a failing engine -> "failed"
    calls.clear()
    monkeypatch.setattr(la, "MAX_SINGLE_CALL_MB", 25)
    assert asyncio.run(la.transcribe_large_audio(p, False))[:2] == ("Voice authentication required for full transcription", 0.2)
    monkeypatch.setattr(tx, "transcribe", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("boom")))
    assert asyncio.run(la.transcribe_large_audio(p)) == ("", 0.0, "unknown", "failed")


def test_engine_lanes_deal_batches_in_order_and_pass_errors_on():
    """whisper_amd/lanes.py with duck-typed engines: batch i of a file goes to lane i mod n, a lane runs its batches in
    submission order, results come back in window order, attribute access falls through to lane 0, an exception inside a lane
    is raised from result(), replicas of different configurations are refused."""
    import threading
    import time
    import torch
    from whisper_amd.lanes import EngineLanes

    class Fake:
        def __init__(self, name, max_batch=3, delay=0.0):
            self.name, self.max_batch, self.delay = name, max_batch, delay
            self.dims, self.precision, self.torch_device = "dims", "fp32", torch.device("cpu")
            self.calls, self.closed = [], False

        def transcribe_mel(self, mel, n_content, w_lo, w_hi, prompt, beam_size=None, patience=1.0, **kw):
            time.sleep(self.delay)
            self.calls.append((w_lo, w_hi, threading.current_thread().name))
            if kw.get("boom") and w_lo == kw["boom"]:
                raise RuntimeError("lane failure")
            return {"tokens": [[w, len(prompt)] for w in range(w_lo, w_hi)], "sum_logprobs": [float(w) for w in range(w_lo, w_hi)],
                    "no_speech_probs": [0.5] * (w_hi - w_lo), "steps": 7}

        def close(self):
            self.closed = True

    a, b = Fake("a", delay=0.05), Fake("b")
    with EngineLanes([a, b]) as lanes:
        assert lanes.n_lanes == 2 and lanes.name == "a" and lanes.max_batch == 3
        out = lanes.transcribe_mel(None, 0, 2, 13, [1, 2, 3, 4])
        assert out["tokens"] == [[w, 4] for w in range(2, 13)] and out["steps"] == 28 and len(out["no_speech_probs"]) == 11
        assert [c[:2] for c in a.calls] == [(2, 5), (8, 11)] and [c[:2] for c in b.calls] == [(5, 8), (11, 13)]
        assert {c[2] for c in a.calls} == {"whisper-amd-lane-0"} and {c[2] for c in b.calls} == {"whisper-amd-lane-1"}
        out = lanes.transcribe_mel(None, 0, 0, 4, [1], beam_size=2)           # beam search: max_batch // beam_size windows per batch
        assert [c[:2] for c in a.calls[2:]] == [(0, 1), (2, 3)] and len(out["tokens"]) == 4
        with pytest.raises(RuntimeError, match="lane failure"):
            lanes.transcribe_mel(None, 0, 0, 9, [1], boom=3)
        assert lanes.map(lambda e, x: (e.name, x), [10, 11, 12]) == [("a", 10), ("b", 11), ("a", 12)]
        assert lanes.submit(lambda e: e.name).result() in ("a", "b")
    assert a.closed and b.closed
    with pytest.raises(ValueError):
        EngineLanes([Fake("a"), Fake("b", max_batch=4)])
    with pytest.raises(ValueError):
        EngineLanes([])


def test_engine_lanes_share_one_encode_gate():
    """The lanes of a GPU run their encoders one at a time: EngineLanes hands every engine that has an ``encode_gate`` slot the SAME
    lock (engines take it around ``encode()``, whisper_amd/engine.py); stand-ins without the slot are left alone, and
    ``one_encoder_at_a_time=False`` leaves real engines ungated.  Two lanes that encode at once overlap without the gate and never
    with it."""
    import threading
    import time
    import torch
    from whisper_amd.lanes import EngineLanes

    class Fake:
        def __init__(self):
            self.max_batch, self.dims, self.precision, self.torch_device = 2, "dims", "fp32", torch.device("cpu")
            self.encode_gate = None
            self.inside = []

        def encode(self, log):
            gate = self.encode_gate
            if gate is not None:
                gate.acquire()
            try:
                log.append(("in", time.perf_counter())); time.sleep(0.05); log.append(("out", time.perf_counter()))
            finally:
                if gate is not None:
                    gate.release()

        def close(self):
            pass

    def overlapped(lanes):
        log = []
        lanes.map(lambda e, _: e.encode(log), [0, 1])
        log.sort(key=lambda x: x[1])
        return [k for k, _ in log] != ["in", "out", "in", "out"]

    with EngineLanes([Fake(), Fake()]) as lanes:
        gates = {id(e.encode_gate) for e in lanes.engines}
        assert len(gates) == 1 and lanes.engines[0].encode_gate is not None
        assert not overlapped(lanes)
    with EngineLanes([Fake(), Fake()], one_encoder_at_a_time=False) as lanes:
        assert all(e.encode_gate is None for e in lanes.engines)
        assert overlapped(lanes)
    with EngineLanes([Fake()]) as lanes:                      # one lane: nothing to keep apart
        assert lanes.engines[0].encode_gate is None
