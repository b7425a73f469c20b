"""The CPU oracle against the committed golden vectors (tests/golden, produced by oracle/make_golden.py from the
transformers 5.15.0 Whisper implementation).  The reference holds no fixtures for this path."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from oracle.make_golden import prefix_tokens
from whisper_amd import checkpoint as ck
from whisper_amd.audio import slaney_mel_filterbank

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("n_mels", [80, 128])
def test_mel_filters_and_log_mel_match_hf(n_mels):
    g = np.load(os.path.join(G, f"hf_mel_{n_mels}.npz"))
    assert np.abs(wo.mel_filters(n_mels) - g["filters"]).max() < 1e-7
    assert np.abs(slaney_mel_filterbank(n_mels) - g["filters"]).max() < 1e-7      # the product's own generator
    mel = wo.log_mel_spectrogram(wo.varied_audio(int(g["audio_seed"])), n_mels, padding=0).numpy()
    assert mel.shape == (n_mels, 3000)
    # north_star tolerance: 1e-4 relative (values span about [-0.7, 1.5])
    assert np.abs(mel[:, ::7] - g["frames_stride7"]).max() < 1e-4 * np.abs(g["frames_stride7"]).max()
    np.testing.assert_allclose(mel.astype(np.float64).sum(axis=1), g["row_sums"], rtol=1e-5)
    np.testing.assert_allclose(mel.astype(np.float64).sum(axis=0)[::3], g["col_sums"], rtol=1e-4, atol=1e-3)


def test_padded_mel_differs_from_hf_only_in_the_last_frames():
    """Upstream pads 30 s of zeros before the STFT, HF reflects at the clip end: same values except the frames whose
    window crosses the end."""
    a = wo.varied_audio(3)
    up = wo.log_mel_spectrogram(a, 80, padding=wo.N_SAMPLES)[:, :3000]
    hf = wo.log_mel_spectrogram(a, 80, padding=0)
    assert torch.allclose(up[:, :2998], hf[:, :2998], atol=1e-6)


@pytest.mark.parametrize("name", ["micro", "tiny"])
def test_encoder_and_logits_match_hf(name):
    g = np.load(os.path.join(G, f"hf_model_{name}.npz"))
    dims = ck.dims_for(name)
    sd = ck.synthetic_state_dict(dims, seed=0)
    tok = ck.special_tokens_for(dims.n_vocab)
    om = wo.OracleWhisper(dims, sd)
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in (0, 1)])
    toks = prefix_tokens(tok)
    assert np.array_equal(toks.numpy(), g["tokens"])
    with torch.no_grad():
        xa = om.encode(mels)
        lg = om.logits(toks, xa)
    scale = np.abs(g["enc_rows_stride50"]).max()
    assert np.abs(xa[:, ::50].numpy() - g["enc_rows_stride50"]).max() < 2e-4 * scale     # fp32, different op order
    np.testing.assert_allclose(xa.norm(dim=-1).numpy(), g["enc_row_norms"], rtol=2e-4)
    lscale = np.abs(g["logits_stride97"]).max()
    assert np.abs(lg[..., ::97].numpy() - g["logits_stride97"]).max() < 2e-4 * lscale
    top = lg.topk(8, dim=-1)
    assert np.array_equal(top.indices[..., 0].numpy(), g["logits_top_indices"][..., 0])     # argmax agrees with HF
    assert np.abs(top.values.numpy() - g["logits_top_values"]).max() < 2e-4 * lscale


@pytest.mark.parametrize("name", ["micro"])
def test_greedy_rules_regression(name):
    g = np.load(os.path.join(G, f"oracle_greedy_{name}.npz"))
    dims = ck.dims_for(name)
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    tok = ck.special_tokens_for(dims.n_vocab)
    om = wo.OracleWhisper(dims, sd)
    sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids())
    assert np.array_equal(np.asarray(sup, np.int32), g["suppress_ids"])
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(4)])
    with torch.no_grad():
        xa = om.encode(mels)
        for mode, wts, n_new in (("plain", True, 40), ("ts", False, 48)):
            init = wo.initial_tokens(tok, 0, without_timestamps=wts)
            assert np.array_equal(np.asarray(init, np.int32), g[f"{mode}_init"])
            r = wo.greedy_decode(om, xa, init, tok, sup, max_new_tokens=n_new, without_timestamps=wts)
            for b, t in enumerate(r["tokens"]):
                want = [int(v) for v in g[f"{mode}_ids"][b] if v >= 0]
                assert t == want
            np.testing.assert_allclose(r["sum_logprobs"], g[f"{mode}_sum_logprobs"], rtol=1e-4, atol=1e-3)
            assert g[f"{mode}_margins"].min() > 1e-4       # the fixture argmax is well conditioned against fp32 noise
    lens = [(g["plain_ids"][b] >= 0).sum() for b in range(4)]
    assert len(set(lens)) > 1, "fixture should contain sequences that end at different steps"


@pytest.mark.parametrize("n_vocab", [51865, 51866])
def test_logit_filters_match_hf_processors(n_vocab):
    """The oracle's decoding rules against transformers' own logits processors (hf:generation/logits_process.py:1816-2112)
    on seeded logit rows and token histories that reach every branch of the timestamp rules, with and without the
    "timestamp mass outweighs every text token" override: identical -inf masks, identical argmax, same log-probability;
    the no-speech probability against WhisperNoSpeechDetection.  Fixture: tests/golden/hf_rules.npz (oracle/make_golden.py)."""
    from oracle.make_golden import rule_cases, rule_logits
    g = np.load(os.path.join(G, "hf_rules.npz"))
    tok = ck.special_tokens_for(n_vocab)
    sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids(n_vocab))
    prompt = [tok.sot, tok.lang(0), tok.transcribe]
    cases = rule_cases(tok, n_vocab)
    assert len(cases) == g[f"mask_{n_vocab}"].shape[0]
    fired = 0
    for ci, (h, boost) in enumerate(cases):
        lg = rule_logits(1000 + ci, n_vocab, tok, boost)
        row = lg[None].clone()
        wo.apply_logit_filters(row, torch.tensor([prompt + h]), len(prompt), tok, sup, without_timestamps=False)
        want = np.unpackbits(g[f"mask_{n_vocab}"][ci])[:n_vocab].astype(bool)
        assert np.array_equal(torch.isinf(row[0]).numpy(), want), (ci, h, boost)
        assert int(row[0].argmax()) == int(g[f"argmax_{n_vocab}"][ci])
        lp = float(torch.log_softmax(row[0].float(), -1)[row[0].argmax()])
        assert abs(lp - float(g[f"logprob_{n_vocab}"][ci])) < 1e-5
        nsp = float(wo.no_speech_probability(lg[None], tok)[0])
        assert abs(nsp - float(g[f"no_speech_{n_vocab}"][ci])) < 1e-9 + 1e-5 * nsp
        fired += int(want[:tok.timestamp_begin].all())
    assert 0 < fired < len(cases)            # the override (all text masked) is exercised, and so is its absence


def test_greedy_loop_matches_hf_model_and_processors():
    """oracle greedy_decode == a greedy loop built from the HF model forward and the HF logits processors (ragged EOT, with
    and without timestamps): token ids equal, cumulative log-probabilities and no-speech probabilities within fp32 noise."""
    g = np.load(os.path.join(G, "hf_greedy_micro.npz"))
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    tok = ck.special_tokens_for(dims.n_vocab)
    om = wo.OracleWhisper(dims, sd)
    sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids(dims.n_vocab))
    assert np.array_equal(np.asarray(sup, np.int32), g["suppress_ids"])
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(4)])
    with torch.no_grad():
        xa = om.encode(mels)
        for mode, wts, n_new in (("plain", True, 40), ("ts", False, 48)):
            init = [int(i) for i in g[f"{mode}_init"]]
            assert init == wo.initial_tokens(tok, 0, without_timestamps=wts)
            r = wo.greedy_decode(om, xa, init, tok, sup, max_new_tokens=n_new, without_timestamps=wts)
            assert r["margins"].min() > 1e-4
            for b, t in enumerate(r["tokens"]):
                assert t == [int(v) for v in g[f"{mode}_ids"][b] if v >= 0]
            np.testing.assert_allclose(r["sum_logprobs"], g[f"{mode}_sum_logprobs"], rtol=1e-4, atol=1e-3)
            np.testing.assert_allclose(r["no_speech_probs"], g[f"{mode}_no_speech"], rtol=1e-3)


def test_bf16_emulation_is_a_small_perturbation_of_fp32():
    """The bf16-emulating oracle (the yardstick of the GPU bf16 tolerance) differs from the fp32 oracle by bf16 rounding
    noise only on smooth weights: more than fp32 noise, less than the smooth-weights bf16 bar of the GPU tests."""
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, attn_gain=1.0)
    f32, b16 = wo.OracleWhisper(dims, sd), wo.OracleWhisper(dims, sd, emulate_bf16=True)
    mel = wo.log_mel_spectrogram(wo.varied_audio(2), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000][None]
    toks = prefix_tokens(ck.special_tokens_for(dims.n_vocab))[:1]
    with torch.no_grad():
        a, b = f32.encode(mel), b16.encode(mel)
        la, lb = f32.logits(toks, a), b16.logits(toks, b)
    rel = lambda x, y: float((x - y).norm() / y.norm())
    assert 1e-4 < rel(b, a) < 1.5e-2 and 1e-4 < rel(lb, la) < 1.5e-2
    assert torch.equal(b, b.to(torch.bfloat16).float())          # the encoder output is stored in bf16


def test_timestamp_rules_unit():
    tok = ck.special_tokens_for(51865)
    V = 51865
    tb = tok.timestamp_begin
    # first sampled position: only timestamps up to max_initial_timestamp survive
    lg = torch.zeros(1, V)
    wo.apply_timestamp_rules(lg, torch.tensor([[tok.sot, tok.lang(0), tok.transcribe]]), 3, tok)
    assert torch.isinf(lg[0, :tb]).all() and torch.isfinite(lg[0, tb:tb + 51]).all() and torch.isinf(lg[0, tb + 51:]).all()
    # after <ts> <text>: a timestamp earlier than the last one is forbidden
    lg = torch.zeros(1, V)
    lg[0, 100] = 50.0                      # a strong text token so the "timestamp mass" rule does not fire
    wo.apply_timestamp_rules(lg, torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tb + 10, 500]]), 3, tok)
    assert torch.isinf(lg[0, tb:tb + 11]).all() and torch.isfinite(lg[0, tb + 11]).all() and torch.isfinite(lg[0, 500])
    # after <text> <ts>: the next token must be a timestamp or EOT
    lg = torch.zeros(1, V)
    wo.apply_timestamp_rules(lg, torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tb + 1, 500, tb + 20]]), 3, tok)
    assert torch.isinf(lg[0, :tok.eot]).all()
    # after <ts> <ts>: the next token must be text
    lg = torch.zeros(1, V)
    lg[0, 77] = 50.0
    wo.apply_timestamp_rules(lg, torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tb + 1, tb + 20]]), 3, tok)
    assert torch.isinf(lg[0, tb:]).all() and torch.isfinite(lg[0, 77])


def test_live_against_transformers_if_available():
    """Same check as the golden test but against the live transformers implementation (skipped where absent)."""
    pytest.importorskip("transformers")
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    from oracle.make_golden import build_hf
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=1)
    hf = build_hf(dims, sd)
    om = wo.OracleWhisper(dims, sd)
    mel = wo.log_mel_spectrogram(wo.varied_audio(9), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000][None]
    toks = prefix_tokens(ck.special_tokens_for(dims.n_vocab))[:1]
    with torch.no_grad():
        a, b = om.encode(mel), hf.model.encoder(mel).last_hidden_state
        la, lb = om.logits(toks, a), hf(input_features=mel, decoder_input_ids=toks).logits
    assert (a - b).abs().max() < 2e-4 * b.abs().max()
    assert (la - lb).abs().max() < 2e-4 * lb.abs().max()


@pytest.mark.parametrize("rate", [8000, 11025, 22050, 24000, 32000, 44100, 48000, 96000])
def test_resampler_restatement_matches_scipy(rate):
    """The oracle's polyphase resampler (the ffmpeg normalisation step of ref:app/audio/processor.py:136-147, restated
    with scipy.signal.resample_poly's arithmetic) against scipy itself on the same samples: within the rounding of its
    float32 taps.  Ragged lengths, one-sample input, mono and stereo."""
    import scipy.signal as ss
    rng = np.random.default_rng(rate)
    up, down = 16000 // np.gcd(16000, rate), rate // np.gcd(16000, rate)
    for n in (1, 2, rate // 3 + 11):
        pcm = rng.integers(-20000, 20000, size=(n, 2), dtype=np.int16)
        mono = pcm.astype(np.float64).mean(axis=1) / 32768.0
        ref = ss.resample_poly(mono, up, down)
        got = wo.resample_pcm16(pcm, rate, quantize=False)
        assert got.shape == ref.shape == (-(-n * up // down),)
        assert np.abs(got - ref).max() < 5e-7
        q = wo.resample_pcm16(pcm, rate, quantize=True)
        assert np.array_equal(q * 32768.0, np.rint(q * 32768.0)) and np.abs(q - ref).max() <= 0.5 / 32768 + 5e-7
        assert np.array_equal(wo.resample_pcm16(pcm[:, 0], rate, False), wo.resample_pcm16(pcm[:, :1], rate, False))



# ---- the sequential long-form loop's rules against transformers' helpers (tests/golden/hf_sequential.json) -----------------

def _hf_sequential():
    import json
    with open(os.path.join(G, "hf_sequential.json")) as f:
        return json.load(f)


def test_segment_cuts_and_seek_match_hf_retrieve_segment():
    """Segment boundaries, start / end times and the seek advance of BOTH restatements -- the oracle's ``split_segments`` and
    the product's ``whisper_amd.longform._split_segments`` -- equal ``WhisperGenerationMixin._retrieve_segment`` on crafted
    id sequences: no timestamps, one trailing timestamp, consecutive pairs with open text or a lone ending after them,
    <|0.00|> only, 30.00 s, a pair at the very start, timestamps only; at seek 0, mid-file, and in a short last window.
    The one designed difference: when the ids do not end in a lone timestamp, HF leaves the second timestamp of the closing
    pair in the last segment's ids (hf:models/whisper/generation_whisper.py:2004-2006), upstream does not."""
    from whisper_amd.longform import _split_segments
    fx = _hf_sequential()
    tb, eot = fx["timestamp_begin"], 50257
    assert len(fx["segments"]) >= 40
    for c in fx["segments"]:
        ids, seek, frames, hf = c["ids"], c["seek"], c["frames"], c["hf"]
        ora, ora_next = wo.split_segments(ids, tb, seek, frames)
        prod, prod_next = _split_segments(list(ids), tb, eot, seek, frames, lambda t: "x" * len(t), {})
        assert ora_next - seek == prod_next - seek == c["hf_advance"], c["case"]
        assert len(ora) == len(prod) == len(hf), c["case"]
        lone_end = len(ids) >= 2 and ids[-1] >= tb and ids[-2] < tb
        has_pair = any(a >= tb and b >= tb for a, b in zip(ids, ids[1:]))
        for i, (o, p, h) in enumerate(zip(ora, prod, hf)):
            np.testing.assert_allclose([o[0], o[1]], [h["start"], h["end"]], rtol=0, atol=1e-9, err_msg=c["case"])
            np.testing.assert_allclose([p["start"], p["end"]], [h["start"], h["end"]], rtol=0, atol=1e-9, err_msg=c["case"])
            want = h["tokens"]
            if has_pair and not lone_end and i == len(hf) - 1:
                assert want[-1] >= tb and want[-2] >= tb
                want = want[:-1]                      # HF's marker copy of the closing pair's second timestamp
            assert list(o[2]) == want, c["case"]
            # the product drops the ids of zero-length / text-less segments (upstream, not part of HF's helper)
            empty = o[0] == o[1] or not [t for t in want if t < eot]
            assert p["tokens"] == ([] if empty else want), c["case"]


def test_fallback_predicates_match_hf_need_fallback():
    """needs_fallback / should_skip of the oracle and of the product against ``_need_fallback`` on a 2 x 3 x 3 grid of
    (compression ratio, avg log-probability, no-speech probability) around the default thresholds 2.4 / -1.0 / 0.6.  Where
    upstream [upstream-recall] and HF differ by design -- a repetitive but LIKELY window that looks like silence: upstream
    cancels the fallback on the no-speech probability alone, HF also wants the log-probability below its threshold -- the
    restatements follow upstream and the test says so."""
    from whisper_amd.longform import _fallback_decision
    fx = _hf_sequential()
    diverged = 0
    for c in fx["fallback"]:
        cr, lp, ns = c["compression_ratio"], c["avg_logprob"], c["no_speech_prob"]
        got_o = wo.fallback_decision(cr, lp, ns)
        got_p = _fallback_decision(cr, lp, ns, 2.4, -1.0, 0.6)
        assert tuple(got_o) == tuple(got_p)
        if cr > 2.4 and lp >= -1.0 and ns > 0.6:
            assert c["hf_needs_fallback"] and not got_o[0]            # the designed difference
            diverged += 1
        else:
            assert got_o[0] == c["hf_needs_fallback"], c
        assert got_o[1] == c["hf_should_skip"], c
    assert diverged == 1 and len(fx["fallback"]) == 18
    # thresholds switched off one at a time
    assert _fallback_decision(9.0, -0.1, 0.0, None, -1.0, 0.6) == (False, False) == tuple(wo.fallback_decision(9.0, -0.1, 0.0, None, -1.0, 0.6))
    assert _fallback_decision(1.0, -5.0, 0.9, 2.4, None, 0.6) == (False, True) == tuple(wo.fallback_decision(1.0, -5.0, 0.9, 2.4, None, 0.6))
    assert _fallback_decision(1.0, -5.0, 0.9, 2.4, -1.0, None) == (True, False) == tuple(wo.fallback_decision(1.0, -5.0, 0.9, 2.4, -1.0, None))


def test_avg_logprob_counts_the_eot_step_like_hf():
    from whisper_amd.longform import _avg_logprob
    for c in _hf_sequential()["avg_logprob"]:
        assert abs(wo.avg_logprob(c["sum_logprob"], c["n_tokens"]) - c["hf"]) < 1e-5 * abs(c["hf"])
        assert abs(_avg_logprob(c["sum_logprob"], c["n_tokens"]) - c["hf"]) < 1e-5 * abs(c["hf"])
