"""GPU tests of the public surface: golden fixtures through the C-ABI, transcribe(), the drop-in module, and
size-independent properties at benchmark-like sizes."""
import asyncio
import os

import numpy as np
import pytest
import torch

from whisper_amd import checkpoint as ck

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _mels(dims, seeds):
    from oracle import whisper_oracle as wo
    return torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in seeds])


@pytest.mark.parametrize("n_mels,name", [(80, "micro"), (128, "micro128")])
def test_mel_against_hf_golden(n_mels, name):
    """HF features (reflect padding at the clip end) == engine mel with padding 0."""
    from oracle import whisper_oracle as wo
    from whisper_amd.engine import WhisperEngine
    g = np.load(os.path.join(G, f"hf_mel_{n_mels}.npz"))
    dims = ck.dims_for(name)
    eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, 0), precision="fp32", max_batch=1)
    try:
        mel = eng.log_mel(torch.from_numpy(wo.varied_audio(int(g["audio_seed"]))), padding=0).cpu().numpy()
    finally:
        eng.close()
    assert mel.shape == (n_mels, 3000)
    assert np.abs(mel[:, ::7] - g["frames_stride7"]).max() < 1e-4 * np.abs(g["frames_stride7"]).max()
    np.testing.assert_allclose(mel.astype(np.float64).sum(axis=1), g["row_sums"], rtol=1e-5)


@pytest.mark.parametrize("name", ["micro", "tiny"])
def test_model_against_hf_golden_fp32(name):
    from whisper_amd.engine import WhisperEngine
    g = np.load(os.path.join(G, f"hf_model_{name}.npz"))
    dims = ck.dims_for(name)
    eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, 0), precision="fp32", max_batch=2)
    try:
        eng.encode_windows(_mels(dims, (0, 1)))
        xa = eng.encoder_output(2).cpu()
        lg = eng.decoder_logits(torch.from_numpy(g["tokens"])).cpu()
    finally:
        eng.close()
    assert np.abs(xa[:, ::50].numpy() - g["enc_rows_stride50"]).max() < 2e-4 * np.abs(g["enc_rows_stride50"]).max()
    lscale = np.abs(g["logits_stride97"]).max()
    assert np.abs(lg[..., ::97].numpy() - g["logits_stride97"]).max() < 2e-4 * lscale
    assert np.array_equal(lg.argmax(-1).numpy(), g["logits_top_indices"][..., 0])


@pytest.mark.parametrize("name", ["micro", "tiny"])
def test_greedy_against_golden_fp32(name):
    from whisper_amd.engine import WhisperEngine
    g = np.load(os.path.join(G, f"oracle_greedy_{name}.npz"))
    dims = ck.dims_for(name)
    eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, 0, eot_boost=3.0), precision="fp32", max_batch=4,
                        suppress_ids=[int(i) for i in g["suppress_ids"]])
    try:
        eng.encode_windows(_mels(dims, range(4)))
        for mode, wts, n_new in (("plain", True, 40), ("ts", False, 48)):
            init = [int(i) for i in g[f"{mode}_init"]]
            r = eng.decode_greedy(4, init, max_new_tokens=n_new, without_timestamps=wts)
            t, n = r["tokens"].cpu(), r["n_tokens"].cpu()
            for b in range(4):
                want = [int(v) for v in g[f"{mode}_ids"][b] if v >= 0]
                assert t[b, len(init):len(init) + int(n[b])].tolist() == want
            np.testing.assert_allclose(r["sum_logprobs"].cpu().numpy(), g[f"{mode}_sum_logprobs"], rtol=1e-4, atol=1e-3)
            np.testing.assert_allclose(r["no_speech_probs"].cpu().numpy(), g[f"{mode}_no_speech"], rtol=1e-3)
    finally:
        eng.close()


def test_transcribe_contract_and_dropin(tmp_path, monkeypatch):
    from oracle import whisper_oracle as wo
    from whisper_amd import audio as wa
    from whisper_amd import transcribe as tx
    eng = tx.configure(model="micro", precision="fp32", max_batch=2, allow_synthetic_weights=True)
    try:
        audio = wo.varied_audio(4, 16000 * 65)
        out = tx.transcribe(audio, language="en", max_new_tokens=8)
        assert set(out) >= {"text", "confidence", "chunks_processed"}
        assert out["chunks_processed"] == 3 and isinstance(out["text"], str) and out["confidence"] == 0.0
        assert len(out["tokens"]) == 3 and all(len(t) <= 8 for t in out["tokens"])
        # a path gives the same ids as the array it was written from (up to PCM16 quantisation the ids may differ, so
        # compare against the quantised array)
        p = str(tmp_path / "clip.wav")
        wa.save_wav_pcm16(p, audio)
        q = wa.load_wav_pcm16(p)
        assert tx.transcribe(p, language="en", max_new_tokens=8)["tokens"] == tx.transcribe(q, language="en", max_new_tokens=8)["tokens"]
        # language auto-detection = argmax over the language tokens of one decoder step on [sot]
        om = wo.OracleWhisper(eng.dims, ck.synthetic_state_dict(eng.dims, 0))
        with torch.no_grad():
            # upstream detects on the first 30 s of the WHOLE-FILE mel (clamped with the whole-file maximum)
            mel = wo.log_mel_spectrogram(audio, eng.dims.n_mels, padding=wo.N_SAMPLES)[:, :3000]
            lg = om.logits(torch.tensor([[eng.tok.sot]]), om.encode(mel[None]))[0, 0]
        want = int(lg[eng.tok.sot + 1: eng.tok.sot + 1 + eng.tok.n_langs].argmax())
        from whisper_amd.tokenizer import LANGUAGE_CODES
        assert tx.transcribe(audio, max_new_tokens=2)["language"] == LANGUAGE_CODES[want]
        assert tx.transcribe(audio, mode="sequential", max_new_tokens=2, temperatures=(0.0,))["language"] == LANGUAGE_CODES[want]
        # ... also when the loud part of the file comes after the first window (the clamp of window 0 is the file's)
        late = np.concatenate([audio[:wo.N_SAMPLES] * 0.002, audio[wo.N_SAMPLES:]]).astype(np.float32)
        with torch.no_grad():
            mel = wo.log_mel_spectrogram(late, eng.dims.n_mels, padding=wo.N_SAMPLES)[:, :3000]
            lg = om.logits(torch.tensor([[eng.tok.sot]]), om.encode(mel[None]))[0, 0]
        want_late = int(lg[eng.tok.sot + 1: eng.tok.sot + 1 + eng.tok.n_langs].argmax())
        assert tx.transcribe(late, max_new_tokens=2)["language"] == LANGUAGE_CODES[want_late]
        # the drop-in module on top of the same engine
        import whisper_amd.speech_recognition as sr
        monkeypatch.setattr(sr, "whisper_model", eng)
        monkeypatch.setattr(sr, "LANGUAGE", "en")
        text, conf, lang = asyncio.run(sr.transcribe_audio(p, detailed=True))
        assert (conf, lang) == (0.0, "en") and isinstance(text, str)
        assert asyncio.run(sr.transcribe_audio(p, detailed=False))[0] == "Voice authentication required for full transcription"
        assert asyncio.run(sr.transcribe_audio(tmp_path / "missing.wav")) == ("Error during transcription", 0.0, "unknown")
    finally:
        tx.close()


@pytest.mark.parametrize("model", ["base", "small"])
def test_properties_at_scale_bf16(model):
    """Size-independent properties on base- and small-size models at batch 16 (BASELINE configs[1]): window results do not
    depend on the batch they were decoded in, nor on graph vs eager launches; rows are EOT padded; mel is window-local
    away from the clamp."""
    from whisper_amd.audio import synthetic_audio
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for(model)
    eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, 0, eot_boost=3.0), precision="bf16", max_batch=16)
    try:
        pcm = torch.from_numpy(np.concatenate([synthetic_audio(s) for s in range(16)])).cuda()
        mel = eng.log_mel(pcm)
        init = eng.initial_tokens(0)
        content = pcm.numel() // 160
        eng.encode(mel, 3000, mel.stride(0), content, 16)
        a = eng.decode_greedy(16, init, max_new_tokens=24, use_graph=True)
        ta, na = a["tokens"].cpu(), a["n_tokens"].cpu()
        b = eng.decode_greedy(16, init, max_new_tokens=24, use_graph=False)
        assert torch.equal(ta, b["tokens"].cpu()) and torch.equal(na, b["n_tokens"].cpu())
        # windows 4..7 decoded alone give the same ids as inside the batch of 16
        view = mel[:, 4 * 3000:]
        eng.encode(view, 3000, mel.stride(0), content - 4 * 3000, 4)
        c = eng.decode_greedy(4, init, max_new_tokens=24)
        assert torch.equal(c["tokens"].cpu(), ta[4:8])
        # ... and so does a single window (other encoder GEMM tile shape, one decode row): nothing in the engine may make a
        # window's ids depend on its batch -- the cross-attention key split is a function of max_batch, not of B
        eng.encode(mel[:, 9 * 3000:], 3000, mel.stride(0), content - 9 * 3000, 1)
        one = eng.decode_greedy(1, init, max_new_tokens=24)
        assert torch.equal(one["tokens"].cpu(), ta[9:10])
        for r in range(16):
            assert (ta[r, len(init) + int(na[r]):] == eng.tok.eot).all()
        # the same clip at a different position of the file has the same mel (only the clamp is global)
        m2 = eng.log_mel(pcm[: 2 * 480000])
        assert torch.allclose(m2[:, 100:2900], mel[:, 100:2900], atol=1e-6)
    finally:
        eng.close()


def test_pcm16_ingest_and_wav_path_bit_exact(tmp_path):
    from oracle import whisper_oracle as wo
    """Device-side PCM16 -> float32 (wx_pcm16_to_f32) equals the host /32768 conversion bit for bit, for aligned and
    ragged lengths; a wav path and the equivalent float array give the same tokens."""
    from whisper_amd import transcribe as tx
    from whisper_amd.audio import save_wav_pcm16
    eng = tx.configure(model="micro", precision="fp32", max_batch=2, allow_synthetic_weights=True)
    try:
        rng = np.random.default_rng(3)
        for n in (1, 7, 8, 4099, 480000):
            pcm = rng.integers(-32768, 32767, size=n, dtype=np.int16)
            got = eng.pcm16_to_f32(pcm).cpu().numpy()
            assert np.array_equal(got, pcm.astype(np.float32) / 32768.0)
        audio = wo.varied_audio(5, 16000 * 33)
        p = tmp_path / "clip.wav"
        save_wav_pcm16(p, audio)
        from whisper_amd.audio import load_wav_pcm16
        a = tx.transcribe(str(p), language="en", max_new_tokens=8)
        b = tx.transcribe(load_wav_pcm16(str(p)), language="en", max_new_tokens=8)
        assert a["tokens"] == b["tokens"] and a["chunks_processed"] == 2
    finally:
        tx.close()


def test_resample_ingest_matches_oracle(tmp_path):
    """wx_resample_pcm16 (channel average + polyphase resampling on the device, the reference's ffmpeg normalisation step)
    against the oracle's float64 restatement: within fp32 accumulation error before quantisation, at most one int16 step
    apart (and almost always equal) after it; a 44.1 kHz stereo WAV path transcribes like its 16 kHz mono equivalent."""
    import wave
    from oracle import whisper_oracle as wo
    from whisper_amd import transcribe as tx
    eng = tx.configure(model="micro", precision="fp32", max_batch=2, allow_synthetic_weights=True)
    try:
        rng = np.random.default_rng(11)
        for rate in (8000, 11025, 16000, 22050, 32000, 44100, 48000, 96000):
            for ch in (1, 2, 3):
                for n in (1, 5, 1000, rate + 13):
                    pcm = rng.integers(-30000, 30000, size=(n, ch), dtype=np.int16)
                    ref = wo.resample_pcm16(pcm, rate, quantize=False)
                    got = eng.ingest_pcm16(pcm, rate, quantize=False).cpu().numpy()
                    assert got.shape == ref.shape
                    assert np.abs(got - ref).max() < 4e-6, (rate, ch, n)
                    # quantisation = round-half-even of the kernel's own fp32 value to the int16 grid, clipped (bit-exact);
                    # against the float64 oracle that is at most one step, and only where fp32 error crosses a rounding edge
                    gotq = eng.ingest_pcm16(pcm, rate, quantize=True).cpu().numpy()
                    own = np.clip(np.rint(got * np.float32(32768.0)), -32768, 32767).astype(np.float32) / np.float32(32768.0)
                    assert np.array_equal(gotq, own), (rate, ch, n)
                    refq = wo.resample_pcm16(pcm, rate, quantize=True)
                    assert (np.abs(gotq - refq) * 32768.0).max() <= 1.0
        # end to end: a 44.1 kHz stereo file
        t = np.arange(44100 * 31) / 44100.0
        left = 0.3 * np.sin(2 * np.pi * 220 * t) + 0.02 * rng.standard_normal(t.size)
        right = 0.2 * np.sin(2 * np.pi * 1330 * t) + 0.02 * rng.standard_normal(t.size)
        pcm = np.clip(np.rint(np.stack([left, right], axis=1) * 32768), -32768, 32767).astype(np.int16)
        p = str(tmp_path / "stereo44.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
        a = tx.transcribe(p, language="en", max_new_tokens=8)
        b = tx.transcribe(eng.ingest_pcm16(pcm, 44100).cpu().numpy(), language="en", max_new_tokens=8)
        assert a["tokens"] == b["tokens"] and a["chunks_processed"] == 2
        with pytest.raises(Exception):
            eng.ingest_pcm16(pcm[:100], 16000 * 9)        # down > 8 * up: rejected by the C-ABI, not a fault
    finally:
        tx.close()


def test_cabi_rejects_bad_calls_and_stays_usable():
    """Error behaviour at the C-ABI on a live engine: every misuse returns a negative status with a message (nothing
    throws, nothing is launched), and the engine still decodes afterwards."""
    from oracle import whisper_oracle as wo
    from whisper_amd import _lib
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("micro")
    eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, 0, eot_boost=3.0), precision="fp32", max_batch=4)
    try:
        init = eng.initial_tokens(0)
        with pytest.raises(_lib.WxError, match="wx_encode"):          # decode before any encode
            eng.decode_greedy(1, init, max_new_tokens=4)
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(2)])
        with pytest.raises(_lib.WxError, match="max_batch"):
            eng.encode(torch.zeros((5, dims.n_mels, 3000), device="cuda"), dims.n_mels * 3000, 3000, 5 * 3000, 5)
        eng.encode_windows(mels)
        good = eng.decode_greedy(2, init, max_new_tokens=6)["tokens"].cpu()
        with pytest.raises(_lib.WxError):                              # more rows than were encoded
            eng.decode_greedy(3, init, max_new_tokens=4)
        with pytest.raises(_lib.WxError, match="out of range"):
            eng.decode_greedy(2, init[:-1] + [dims.n_vocab + 7], max_new_tokens=4)
        with pytest.raises(_lib.WxError):                              # prompt as long as the text context
            eng.decode_greedy(2, [eng.tok.sot] * dims.n_text_ctx, max_new_tokens=4)
        with pytest.raises((_lib.WxError, ValueError)):                # 2 windows x 5 beams > max_batch 4
            eng.decode_beam(2, init, beam_size=5, max_new_tokens=4)
        with pytest.raises(_lib.WxError):                              # negative temperature
            eng.decode_rows([init, init], temperature=-1.0, max_new_tokens=4)
        out = torch.empty((dims.n_mels, 10), device="cuda"); g = torch.empty(1, device="cuda")
        pcm = torch.zeros(16000, device="cuda")
        rc = eng.lib.wx_mel_stage1(eng.handle, pcm.data_ptr(), 0, 16000, 16000, 16000 + 480000, 0, 100, out.data_ptr(), 10, g.data_ptr(), None)
        assert rc < 0 and b"mel" in eng.lib.wx_last_error()           # ld < n_frames
        assert eng.lib.wx_decode_greedy(eng.handle, 2, None, None, None, None, None, None, None, None) < 0
        again = eng.decode_greedy(2, init, max_new_tokens=6)["tokens"].cpu()
        assert torch.equal(good, again)
    finally:
        eng.close()


def test_transcribe_modes_beam_and_sequential(tmp_path):
    """The north-star entry point with upstream's other decoding options: beam search and the sequential long-form loop
    return the same contract keys; chunks_processed counts decoded windows; an empty input is an empty result."""
    from oracle import whisper_oracle as wo
    from whisper_amd import transcribe as tx
    tx.configure(model="micro", precision="fp32", max_batch=10, allow_synthetic_weights=True)
    try:
        audio = wo.varied_audio(9, 16000 * 45)
        g = tx.transcribe(audio, language="en", max_new_tokens=10)
        b = tx.transcribe(audio, language="en", max_new_tokens=10, beam_size=5)
        s = tx.transcribe(audio, language="en", mode="sequential", max_new_tokens=10, temperatures=(0.0,))
        for r in (g, b, s):
            assert set(("text", "confidence", "chunks_processed", "language", "tokens")) <= set(r)
            assert r["confidence"] == 0.0 and r["language"] == "en"
        assert g["chunks_processed"] == b["chunks_processed"] == 2 and len(b["tokens"]) == 2
        assert s["chunks_processed"] >= 2 and all(sg["end"] >= sg["start"] for sg in s["segments"])
        assert [sg["seek"] for sg in s["segments"]] == sorted(sg["seek"] for sg in s["segments"])
        e = tx.transcribe(np.zeros(0, np.float32), language="en")
        assert e["text"] == "" and e["chunks_processed"] == 0
        with pytest.raises(ValueError):
            tx.transcribe(audio, language="en", mode="nope")
    finally:
        tx.close()


def _toy_vocab(path, n_ids=50257):
    """A tiktoken-format vocabulary for tests: the 256 byte tokens, then one word " w<i>" per id -- every id the synthetic
    model can emit decodes to real text (so "the last 20 words" exist), and text prompts encode to byte tokens."""
    import base64
    lines = [base64.b64encode(bytes([i])) + b" " + str(i).encode() for i in range(256)]
    lines += [base64.b64encode(f" w{i}".encode()) + b" " + str(i).encode() for i in range(256, n_ids)]
    with open(path, "wb") as f:
        f.write(b"\n".join(lines) + b"\n")


def test_large_file_path_end_to_end(tmp_path):
    """The reference's large-file path (ref:app/transcription/openai_whisper.py:268-337) against a LIVE engine: a 14-minute
    16 kHz PCM16 WAV (26.9 MB > the 25 MB limit) is cut into 20 MB pieces by duration, the pieces are transcribed in order
    with the last 20 words of the previous piece carried in the prompt, texts are joined with spaces, the piece files are
    removed; the result equals per-piece transcribe() calls made by hand with the same prompts."""
    import wave
    from oracle import whisper_oracle as wo
    from whisper_amd import large_audio as la
    from whisper_amd import transcribe as tx
    vocab = str(tmp_path / "toy.tiktoken")
    _toy_vocab(vocab)
    tx.configure(model="micro", precision="fp32", max_batch=8, allow_synthetic_weights=True, vocab=vocab)
    try:
        seconds = 14 * 60
        audio = np.concatenate([wo.varied_audio(200 + i, 16000 * 60) for i in range(seconds // 60)])
        pcm = np.clip(np.rint(audio * 32767.0), -32768, 32767).astype(np.int16)
        p = tmp_path / "long.wav"
        with wave.open(str(p), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
        assert p.stat().st_size / (1024 * 1024) > la.MAX_SINGLE_CALL_MB
        kw = dict(mode="windows", max_new_tokens=24)
        pieces = asyncio.run(la.chunk_large_audio(p))
        assert len(pieces) == 2 and pieces[0].parent == tmp_path / "long_chunks"
        r1 = tx.transcribe(str(pieces[0]), language="en", prompt="hello", **kw)
        assert len(r1["text"].split()) >= 20
        carried = "hello Previous context: " + " ".join(r1["text"].strip().split()[-20:])
        r2 = tx.transcribe(str(pieces[1]), language="en", prompt=carried, **kw)
        assert r2["tokens"] != tx.transcribe(str(pieces[1]), language="en", prompt="hello", **kw)["tokens"]   # the carry-over matters
        text, conf, lang, source = asyncio.run(la.transcribe_large_audio(p, True, "en", "hello", **kw))
        assert source == "chunked" and lang == "en" and conf == 0.0
        assert text == r1["text"].strip() + " " + r2["text"].strip()
        assert not (tmp_path / "long_chunks").exists() and p.exists()
        assert r1["chunks_processed"] + r2["chunks_processed"] == 21 + 8        # 624.3 s and 215.7 s pieces
        # under the limit: one local call, source "local"
        small = tmp_path / "small.wav"
        with wave.open(str(small), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm[: 16000 * 40].tobytes())
        out = asyncio.run(la.transcribe_large_audio(small, True, "en", None, **kw))
        assert out[3] == "local" and out[0] == tx.transcribe(str(small), language="en", **kw)["text"].strip()
        # the default piece mode is upstream's sequential loop
        seq = asyncio.run(la.transcribe_large_audio(small, True, "en", None, max_new_tokens=8, temperatures=(0.0,)))
        assert seq[3] == "local" and seq[0] == tx.transcribe(str(small), language="en", mode="sequential", max_new_tokens=8,
                                                             temperatures=(0.0,))["text"].strip()
    finally:
        tx.close()


@pytest.mark.parametrize("fmt", ["pt", "safetensors"])
def test_engine_from_a_checkpoint_file(tmp_path, fmt):
    """ref:app/transcription/speech_recognition.py:22 (whisper.load_model) -> configure(checkpoint=...): an engine built from
    a LOCAL openai-layout .pt / HF-layout .safetensors file decodes the ids of the engine built from the in-memory state
    dict, and the dimensions come from the file, not from the model name."""
    from oracle import whisper_oracle as wo
    from whisper_amd import transcribe as tx
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=5, eot_boost=3.0)
    path = str(tmp_path / f"model.{fmt}")
    if fmt == "pt":
        ck.save_checkpoint(path, dims, sd)
    else:
        from safetensors.torch import save_file
        hf = {ck.canonical_to_hf_name(k): v.contiguous() for k, v in sd.items()}
        hf["proj_out.weight"] = sd["decoder.token_embedding.weight"].clone()       # HF files carry the tied head too
        save_file(hf, path)
    audio = wo.varied_audio(12, 16000 * 50)
    ref_eng = WhisperEngine(dims, sd, precision="fp32", max_batch=2)
    try:
        want = ref_eng.transcribe_pcm(audio, ref_eng.initial_tokens(0), max_new_tokens=16)["tokens"]
    finally:
        ref_eng.close()
    eng = tx.configure(model="large-v3", checkpoint=path, precision="fp32", max_batch=2)     # the name is NOT what sizes the engine
    try:
        assert eng.dims == dims
        got = tx.transcribe(audio, language="en", max_new_tokens=16)
        assert got["tokens"] == want and got["chunks_processed"] == 2
        om = wo.OracleWhisper(dims, sd)
        with torch.no_grad():
            ref = wo.transcribe_windows(om, audio, eng.initial_tokens(0), eng.tok, eng.suppress_ids, max_new_tokens=16)
        assert got["tokens"] == ref["tokens"]
    finally:
        tx.close()
    with pytest.raises(FileNotFoundError):
        tx.configure(model="micro", checkpoint=str(tmp_path / "missing.pt"))


def test_confidence_mode_avg_logprob(tmp_path, monkeypatch):
    """confidence_mode="avg_logprob" returns exp(mean over windows of sum_logprob / (n_tokens + 1)) -- the
    oracle's number -- through transcribe() and through the drop-in (capped at 0.5 on the privacy branch); the default
    "reference" mode returns the reference's 0.0 (it reads a segment key upstream does not set, ref:speech_recognition.py:62-66)."""
    import math
    from oracle import whisper_oracle as wo
    from whisper_amd import audio as wa
    from whisper_amd import transcribe as tx
    eng = tx.configure(model="micro", precision="fp32", max_batch=4, allow_synthetic_weights=True, confidence_mode="avg_logprob", seed=0)
    try:
        p = str(tmp_path / "c.wav")
        wa.save_wav_pcm16(p, wo.varied_audio(15, 16000 * 70))
        audio = wa.load_wav_pcm16(p)                                  # the PCM16-quantised samples, as float32
        om = wo.OracleWhisper(eng.dims, ck.synthetic_state_dict(eng.dims, seed=0))
        with torch.no_grad():
            ref = wo.transcribe_windows(om, audio, eng.initial_tokens(0), eng.tok, eng.suppress_ids, max_new_tokens=20)
        avg = [float(s) / (len(t) + 1) for s, t in zip(ref["sum_logprobs"], ref["tokens"])]
        want = math.exp(sum(avg) / len(avg))
        got = tx.transcribe(p, language="en", max_new_tokens=20)
        assert got["tokens"] == ref["tokens"]
        assert 0.0 < got["confidence"] < 1.0 and abs(got["confidence"] - want) < 1e-3 * want
        # sequential mode: exp(mean of the segments' avg_logprob)
        s = tx.transcribe(p, language="en", mode="sequential", max_new_tokens=20, temperatures=(0.0,))
        from whisper_amd.tokenizer import Tokenizer
        with torch.no_grad():
            rs = wo.transcribe_sequential(om, audio, eng.tok, eng.suppress_ids, Tokenizer(eng.dims.n_vocab).decode, max_new_tokens=20)
        lps = [sg["avg_logprob"] for sg in rs["segments"]]
        assert lps and abs(s["confidence"] - math.exp(sum(lps) / len(lps))) < 1e-3 * s["confidence"]
        import whisper_amd.speech_recognition as sr
        monkeypatch.setattr(sr, "whisper_model", eng)
        monkeypatch.setattr(sr, "LANGUAGE", "en")
        monkeypatch.setattr(sr, "_AMD", {"decode_mode": "windows"})
        full = tx.transcribe(p, language="en")["confidence"]
        text, conf, lang = asyncio.run(sr.transcribe_audio(p, detailed=True))
        assert abs(conf - full) < 1e-12 and lang == "en"
        assert asyncio.run(sr.transcribe_audio(p, detailed=False))[1] == min(full, 0.5)
    finally:
        tx.close()
    tx.configure(model="micro", precision="fp32", max_batch=4, allow_synthetic_weights=True)
    try:
        assert tx.transcribe(p, language="en", max_new_tokens=4)["confidence"] == 0.0
    finally:
        tx.close()


def test_timing_harness_prints_the_reference_lines(tmp_path, capsys, monkeypatch):
    """ref:test_whisper.py:13-54: `python -m whisper_amd.harness clip.wav --model tiny` loads the model by name inside the
    timed region and prints the reference's summary block, in its order and format."""
    import re
    import sys
    from oracle import whisper_oracle as wo
    from whisper_amd import audio as wa
    from whisper_amd import harness
    p = str(tmp_path / "clip.wav")
    wa.save_wav_pcm16(p, wo.varied_audio(3, 16000 * 12))
    monkeypatch.setattr(sys, "argv", ["harness", p, "--model", "tiny", "--language", "en"])
    harness.main()
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "Loading Whisper model 'tiny'..." and out[1] == f"Transcribing {p} on cuda..."
    i = out.index("Transcription Results:")
    assert out[i - 1] == "" and out[i + 1] == "=" * 50
    assert out[i + 2].startswith("Text: ") and out[i + 3] == "Language: en"
    assert re.fullmatch(r"Confidence: \d+\.\d{4}", out[i + 4]) and re.fullmatch(r"Time taken: \d+\.\d{2} seconds", out[i + 5])
    assert out[i + 6] == "=" * 50
    monkeypatch.setattr(sys, "argv", ["harness", str(tmp_path / "nope.wav")])
    harness.main()
    assert capsys.readouterr().out.strip() == f"Error: Audio file '{tmp_path / 'nope.wav'}' not found"


def test_lanes_decode_the_same_ids():
    """Two and three batches in flight (whisper_amd/lanes.py: engine replicas on their own host threads and streams) decode a
    long file to exactly the ids, log-probabilities and no-speech probabilities of one engine taking the batches in turn --
    fp32 and bf16, greedy and beam search, a ragged last batch -- and through transcribe() with lanes configured."""
    from oracle import whisper_oracle as wo
    from whisper_amd import transcribe as tx
    from whisper_amd.engine import WhisperEngine
    from whisper_amd.lanes import EngineLanes
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    audio = wo.varied_audio(12, 16000 * 30 * 11 + 4321)            # 12 windows: batches of 4, 4, 4 (max_batch 4) or 5, 5, 2
    prompt = None
    for precision, max_batch, xattn in (("fp32", 4, False), ("bf16", 5, False), ("bf16", 5, True)):
        for n in (2, 3):
            # (the key split of the cross-attention stream depends on n_lanes: the single-engine reference is built with the same n_lanes)
            if xattn:
                os.environ["WX_XATTN"] = "1"            # cross-attention on the encoder output below 32 rows too
            try:
                one = WhisperEngine(dims, sd, precision=precision, max_batch=max_batch, n_lanes=n)
                lanes = EngineLanes.create(lambda k: WhisperEngine(dims, sd, precision=precision, max_batch=max_batch, n_lanes=k), n)
            finally:
                os.environ.pop("WX_XATTN", None)
            try:
                assert one.cross_form == ("encoder_output" if xattn else "kv")
                mel = one.log_mel(torch.from_numpy(audio).cuda())
                content = audio.size // 160
                prompt = one.initial_tokens(0)
                ref = one.transcribe_mel(mel, content, 0, 12, prompt, max_new_tokens=10)
                ref_beam = one.transcribe_mel(mel, content, 0, 12, prompt, beam_size=2, max_new_tokens=6)
                assert lanes.n_lanes == n and lanes.max_batch == max_batch and lanes.cross_form == one.cross_form
                for _ in range(2):                               # the second run replays the lanes' captured graphs
                    got = lanes.transcribe_mel(mel, content, 0, 12, prompt, max_new_tokens=10)
                    assert got["tokens"] == ref["tokens"] and got["steps"] == ref["steps"]
                    assert got["sum_logprobs"] == ref["sum_logprobs"] and got["no_speech_probs"] == ref["no_speech_probs"]
                got = lanes.transcribe_mel(mel, content, 0, 12, prompt, beam_size=2, max_new_tokens=6)
                assert got["tokens"] == ref_beam["tokens"] and got["sum_logprobs"] == ref_beam["sum_logprobs"]
                # an error inside a lane reaches the caller
                with pytest.raises(Exception):
                    lanes.submit(lambda e: e.decode_greedy(max_batch + 1, prompt)).result()
            finally:
                one.close()
                lanes.close()
    # the product entry point with lanes configured
    try:
        tx.configure(model="micro", precision="fp32", max_batch=4, allow_synthetic_weights=True)
        want = tx.transcribe(audio, language="en", max_new_tokens=8)
        tx.configure(model="micro", precision="fp32", max_batch=4, allow_synthetic_weights=True, lanes=2)
        got = tx.transcribe(audio, language="en", max_new_tokens=8)
        assert got["tokens"] == want["tokens"] and got["chunks_processed"] == want["chunks_processed"] == 12
        seq = tx.transcribe(audio[:16000 * 40], language="en", mode="sequential", max_new_tokens=8, temperatures=(0.0,))
        assert isinstance(seq["text"], str)
    finally:
        tx.close()


def test_nan_input_ends_rows_instead_of_indexing_with_garbage():
    """Audio that turns into NaN features (a NaN sample) makes every logit NaN: no candidate ever beats the initial one, whose
    id indexes nothing.  The sampler ends such a row with EOT (and the embedding lookup clamps any id it is handed), so the
    call returns -- empty transcripts -- instead of reading outside the token table."""
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("micro")
    eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, seed=0), precision="bf16", max_batch=2)
    try:
        mel = torch.full((2, dims.n_mels, 3000), float("nan"), device="cuda")
        eng.encode_windows(mel)
        assert torch.isnan(eng.encoder_output(2)).all()
        init = eng.initial_tokens(0)
        r = eng.decode_greedy(2, init, max_new_tokens=6)
        t = r["tokens"].cpu()
        assert (t[:, len(init):] == eng.tok.eot).all() and (r["n_tokens"].cpu() == 0).all()
        assert int(t.max()) < dims.n_vocab and int(t.min()) >= 0
    finally:
        eng.close()


@pytest.mark.parametrize("max_batch,n_lanes,splits,form", [(64, 1, 4, "encoder_output"), (48, 1, 5, "encoder_output"), (80, 1, 3, "encoder_output"),
                                                            (32, 1, 8, "encoder_output"), (64, 4, 1, "encoder_output"), (64, 3, 2, "encoder_output"),
                                                            (64, 2, 2, "encoder_output"), (128, 1, 4, "encoder_output"), (16, 1, None, "kv")])
def test_cross_attention_grid_is_a_function_of_max_batch_and_lanes(max_batch, n_lanes, splits, form):
    """wx_cross_attention_form / wx_cross_key_splits: the form and the key split of the decoder's cross-attention are fixed at
    engine creation from max_batch and n_lanes alone (batch invariance), one round of workgroups when the engine has the GPU to
    itself (128 rows run as two passes of 64)."""
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("micro")
    eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, seed=0), precision="bf16", max_batch=max_batch, n_lanes=n_lanes)
    try:
        assert eng.cross_form == form
        if splits is not None:
            assert eng.cross_key_splits == splits
    finally:
        eng.close()


def test_replicas_share_one_copy_of_the_weights():
    """wx_create_shared: a replica owns workspace, caches and graph but no weights (wx_weight_bytes 0); it decodes the ids of
    its parent, its own max_batch / suppress list are its own, and it outlives the parent (the weights are freed with the last
    engine that uses them)."""
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    for precision in ("fp32", "bf16"):
        parent = WhisperEngine(dims, sd, precision=precision, max_batch=4)
        rep = parent.replica()
        small = parent.replica(max_batch=2)
        try:
            assert parent.weight_bytes > 0 and rep.weight_bytes == 0 and small.weight_bytes == 0
            assert rep.workspace_bytes == parent.workspace_bytes and small.workspace_bytes < parent.workspace_bytes
            mels = _mels(dims, range(4))
            init = parent.initial_tokens(0)
            parent.encode_windows(mels)
            want = parent.decode_greedy(4, init, max_new_tokens=12)
            rep.encode_windows(mels)
            got = rep.decode_greedy(4, init, max_new_tokens=12)
            assert torch.equal(got["tokens"], want["tokens"]) and torch.equal(got["sum_logprobs"], want["sum_logprobs"])
            # a replica's suppress list is its own
            first = int(want["tokens"][0, len(init)])
            rep.set_suppress_ids(list(parent.suppress_ids) + [first])
            rep.encode_windows(mels)
            assert int(rep.decode_greedy(4, init, max_new_tokens=2)["tokens"][0, len(init)]) != first
            parent.encode_windows(mels)
            assert torch.equal(parent.decode_greedy(4, init, max_new_tokens=12)["tokens"], want["tokens"])
            # the parent goes first: the replicas keep the weights alive
            parent.close()
            small.encode_windows(mels[2:4])
            got2 = small.decode_greedy(2, init, max_new_tokens=12)
            assert torch.equal(got2["tokens"], want["tokens"][2:4])
            with pytest.raises(Exception):
                parent.replica()
        finally:
            parent.close(); rep.close(); small.close()
    # a replica must select the parent's cross-attention form (its weight layouts): bf16 below / above 32 rows differ
    parent = WhisperEngine(dims, sd, precision="bf16", max_batch=4)
    try:
        with pytest.raises(Exception, match="cross-attention form"):
            parent.replica(max_batch=64)
    finally:
        parent.close()


def test_reserve_cross_kv_before_the_first_beam_call():
    """wx_reserve_cross_kv: an encoder-output engine (bf16, max_batch >= 32) allocates its cross K/V cache at once instead of
    inside its first beam-search call; beam search then gives what it gives without the reservation, and the call is a no-op on
    engines that own the cache already."""
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    a = WhisperEngine(dims, sd, precision="bf16", max_batch=32)
    b = a.replica()
    try:
        assert a.cross_form == "encoder_output"
        w0 = b.workspace_bytes
        b.reserve_cross_kv()
        grown = b.workspace_bytes - w0
        assert grown == 2 * 32 * dims.n_text_layer * 1500 * dims.n_text_state * 2
        b.reserve_cross_kv()
        assert b.workspace_bytes - w0 == grown                       # second call: nothing more
        mels = _mels(dims, range(3))
        init = a.initial_tokens(0)
        a.encode_windows(mels); b.encode_windows(mels)
        ra = a.decode_beam(3, init, beam_size=5, max_new_tokens=8)
        assert a.workspace_bytes - w0 == grown                       # the lazy path allocated the same cache inside the call
        rb = b.decode_beam(3, init, beam_size=5, max_new_tokens=8)
        assert ra["tokens"] == rb["tokens"] and ra["sum_logprobs"] == rb["sum_logprobs"]
        f = WhisperEngine(dims, sd, precision="fp32", max_batch=2)
        try:
            w1 = f.workspace_bytes
            f.reserve_cross_kv()
            assert f.workspace_bytes == w1
        finally:
            f.close()
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("precision,xattn", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_compact_windows_then_decode_only_those(precision, xattn):
    """wx_compact_windows: after keeping windows (1, 3) of four, a 2-row decode gives exactly the ids, log-probabilities and
    no-speech probabilities those windows had in the batch of four (every form of the decoder's cross-attention moves its
    per-window state); bad index lists are refused and leave the engine usable."""
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    if xattn:
        os.environ["WX_XATTN"] = "1"
    try:
        eng = WhisperEngine(dims, sd, precision=precision, max_batch=4)
    finally:
        os.environ.pop("WX_XATTN", None)
    try:
        assert eng.cross_form == ("encoder_output" if xattn else "kv")
        mels = _mels(dims, range(4))
        init = eng.initial_tokens(0)
        eng.encode_windows(mels)
        want = eng.decode_greedy(4, init, max_new_tokens=12)
        enc4 = eng.encoder_output(4)
        for bad in ([3, 1], [1, 1], [0, 4], [-1], [1, 2, 3, 3]):
            with pytest.raises(Exception):
                eng.compact_windows(bad)
        eng.compact_windows([1, 3])
        got = eng.decode_greedy(2, init, max_new_tokens=12)
        for k in ("tokens", "sum_logprobs", "no_speech_probs", "n_tokens"):
            assert torch.equal(got[k], want[k][[1, 3]]), k
        assert torch.equal(eng.encoder_output(2), enc4[[1, 3]])
        with pytest.raises(Exception):
            eng.decode_greedy(3, init, max_new_tokens=2)          # the engine now holds two windows
        eng.compact_windows([1])
        one = eng.decode_greedy(1, init, max_new_tokens=12)
        assert torch.equal(one["tokens"][0], want["tokens"][3])
    finally:
        eng.close()


def test_temperature_fallback_redecodes_only_the_failed_files():
    """Sequential mode, three one-window files, a log-probability threshold placed between their T = 0 results so that exactly
    one of them fails: the second attempt (T = 0.2) decodes ONE row (the failed window, moved to slot 0), and the two files
    that passed keep the segments of the T = 0 decode exactly."""
    from oracle import whisper_oracle as wo
    from whisper_amd.engine import WhisperEngine
    from whisper_amd.longform import transcribe_sequential
    from whisper_amd.tokenizer import Tokenizer
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    eng = WhisperEngine(dims, sd, precision="fp32", max_batch=4)
    try:
        dec = Tokenizer(dims.n_vocab).decode
        audios = [wo.varied_audio(40 + i, 16000 * 20) for i in range(3)]
        base = transcribe_sequential(eng, audios, dec, language_index=0, temperatures=(0.0,), max_new_tokens=16, compression_ratio_threshold=None)
        lps = [f["segments"][0]["avg_logprob"] for f in base]
        order = sorted(range(3), key=lambda i: lps[i])
        thr = 0.5 * (lps[order[0]] + lps[order[1]])               # only the least likely file falls below
        assert lps[order[0]] < thr < lps[order[1]]
        calls = []
        orig = eng.decode_rows

        def spy(prompts, **kw):
            calls.append((len(prompts), kw.get("temperature")))
            return orig(prompts, **kw)
        eng.decode_rows = spy
        got = transcribe_sequential(eng, audios, dec, language_index=0, temperatures=(0.0, 0.2), max_new_tokens=16,
                                    compression_ratio_threshold=None, logprob_threshold=thr, no_speech_threshold=None)
        assert calls == [(3, 0.0), (1, 0.2)]
        for i in range(3):
            if i == order[0]:
                assert all(s["temperature"] == 0.2 for s in got[i]["segments"])
            else:
                assert [(s["tokens"], s["start"], s["end"], s["temperature"]) for s in got[i]["segments"]] == \
                       [(s["tokens"], s["start"], s["end"], 0.0) for s in base[i]["segments"]]
    finally:
        eng.close()


@pytest.mark.parametrize("hw_queues", ["8", "2"])
def test_lanes_get_a_hardware_queue_each(hw_queues):
    """Two lane streams on one hardware queue take turns (a third of the throughput at four lanes), and with GPU_MAX_HW_QUEUES = 8
    or 2 the runtime's mapping does put lanes together.  EngineLanes measures it (wx_stream_spin on two streams: 0.4 or 0.8 ms) and
    gives a lane that shares another stream, then one with a queue of its own (wx_stream_create_exclusive).  Own process: the
    variable is read when the HIP runtime starts."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES=hw_queues)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "lane_queue_check.py")], env=env, capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    print(rep)
    assert rep["pairs_still_taking_turns"] == []      # (rep["fixes"] says what it had to do: on this pool one or two lanes at 8, two at 2)
