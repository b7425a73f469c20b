"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Tolerances (north_star): mel within 1e-4 relative; encoder / logits within the fp32 and bf16 tolerances written
below; greedy token ids bit-exact in fp32 mode."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from whisper_amd import checkpoint as ck

pytestmark = pytest.mark.gpu

FP32_TOL = 2e-4     # max |hip - oracle| / max |oracle|, fp32 mode (different summation order only)
# bf16 mode (bf16 weights/activations, fp32 accumulate, fp32 residual), metric ||hip - ref|| / ||ref||:
BF16_RMS_TOL = 1.5e-2        # against the fp32 oracle on smooth-attention weights (attn_gain 1): the rounding noise of the bf16 pipeline
# Against the oracle's bf16-EMULATING mode (same rounding points as the engine), on any weights, the sharp default ones
# (attn_gain 3) included: bf16 rounding amplifies a summation-order difference of 1e-6 to a whole ulp wherever a rounding
# flips, so two correct evaluations of the same rounded model sit one NOISE FLOOR apart -- measured here as the distance
# between the emulation with fp32 and with fp64 accumulation.  The device must be within BF16_NOISE_FACTOR noise floors of
# the emulation.  Round 3: the emulation also restates the ONLINE softmax of the two MFMA attention kernels (flash.hip: 64-key
# tiles; xattn.hip: 32-key tiles per key split, per-split bf16 contexts) -- the softmax weights are rounded against the
# RUNNING maximum there, which alone moved a 2-layer model's encoder output by two floors -- and with that every model,
# weight set and key split measures 0.80-1.11 floors (round 2, weights rounded against the final maximum: 1.0-2.3, bar 2.0).
# A broken kernel is tens of floors away.  A flat 2e-2 holds on smooth weights and for tiny's encoder, but not for sharp
# 32-layer stacks whose own noise floor is 5-8e-2 -- and the old flat 2.5e-1 against the fp32 oracle would have passed a
# broken kernel.
BF16_NOISE_FACTOR = 1.5


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rms_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))


def check_close(got, ref, precision, sharp=True, emu=None, emu64=None):
    """fp32: against the fp32 oracle.  bf16: within BF16_NOISE_FACTOR noise floors (|emu64 - emu|) of the bf16-emulating
    oracle, and on smooth weights also against the fp32 oracle at BF16_RMS_TOL."""
    if precision == "fp32":
        assert rel_err(got, ref) < FP32_TOL
        return
    if emu is not None:
        e, floor = rms_err(got, emu), rms_err(emu64, emu)
        print(f"\nbf16 vs emulation {e:.3e} = {e / floor:.2f} noise floors ({floor:.3e}); vs fp32 oracle {rms_err(got, ref):.3e} "
              f"({'sharp' if sharp else 'smooth'} weights)")
        assert e < BF16_NOISE_FACTOR * floor
    if not sharp or emu is None:
        assert rms_err(got, ref) < BF16_RMS_TOL


@pytest.fixture(scope="module")
def models():
    from whisper_amd.engine import WhisperEngine
    cache = {}

    def get(name, precision, max_batch=4, xattn=False, split=0, **kw):
        """xattn=True: a bf16 engine below 32 rows that runs the decoder cross-attention on the encoder output all the same
        (WX_XATTN=1 at wx_create), so the small models cover both forms; split > 0: that many key splits per row of its stream
        kernel (WX_XATTN_SPLIT at wx_finalize) instead of the 8 a 4-row engine would take"""
        key = (name, precision, max_batch, xattn, split, tuple(sorted(kw.items())))
        if key not in cache:
            dims = ck.dims_for(name)
            sd = ck.synthetic_state_dict(dims, seed=0, **kw)
            if xattn:
                os.environ["WX_XATTN"] = "1"
            if split:
                os.environ["WX_XATTN_SPLIT"] = str(split)
            try:
                eng = WhisperEngine(dims, sd, precision=precision, max_batch=max_batch)
            finally:
                os.environ.pop("WX_XATTN", None)
                os.environ.pop("WX_XATTN_SPLIT", None)
            assert not split or eng.cross_key_splits == split
            assert eng.cross_form == ("encoder_output" if (precision == "bf16" and (xattn or max_batch >= 32)) else "kv")
            cache[key] = (dims, sd, eng)
        return cache[key]

    yield get
    for _, _, e in cache.values():
        e.close()


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("n_samples", [wo.N_SAMPLES, 100000, 16000 * 47 + 123])
def test_mel_parity(models, n_mels, n_samples):
    # the front end depends only on n_mels: micro has 80 bins, micro128 the large-v3 128-bin filterbank
    name = "micro" if n_mels == 80 else "micro128"
    dims, sd, eng = models(name, "fp32")
    audio = wo.synthetic_audio(3, n_samples)
    ref = wo.log_mel_spectrogram(audio, n_mels, padding=wo.N_SAMPLES).numpy()
    got = eng.log_mel(torch.from_numpy(audio)).cpu().numpy()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-4


def test_mel_with_a_dense_caller_supplied_filterbank():
    """A filterbank with more non-zero weights than the kernel keeps in LDS (any [n_mels, 201] matrix may be supplied, e.g.
    upstream's mel_filters.npz) takes the kernel's global-memory weight path: same result as the plain formula."""
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("micro")
    fb = np.abs(np.random.default_rng(2).standard_normal((dims.n_mels, 201))).astype(np.float32) * 0.01
    eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, seed=0), precision="fp32", max_batch=1, mel_filters=fb)
    try:
        audio = wo.synthetic_audio(4, 16000 * 7 + 5)
        x = torch.from_numpy(np.concatenate([audio, np.zeros(wo.N_SAMPLES, np.float32)]))
        st = torch.stft(x, 400, 160, window=torch.hann_window(400), return_complex=True)
        mel = torch.from_numpy(fb).double() @ (st[:, :-1].abs().double() ** 2)
        lg = torch.clamp(mel, min=1e-10).log10()
        ref = ((torch.maximum(lg, lg.max() - 8.0) + 4.0) / 4.0).numpy()
        got = eng.log_mel(torch.from_numpy(audio)).cpu().numpy()
        assert got.shape == ref.shape and rel_err(got, ref) < 1e-4
    finally:
        eng.close()


def test_mel_two_stage_matches_sharded_max(models):
    """Stage 1 on two halves + MAX of the two maxima + stage 2 == single pass (SURVEY.md §8e caveat 1)."""
    dims, sd, eng = models("micro", "fp32")
    audio = wo.synthetic_audio(5, 16000 * 75)
    audio[: 16000 * 30] *= 0.05            # make the global max live in the second half
    pcm = torch.from_numpy(audio).cuda()
    whole = eng.log_mel(pcm)
    n_file = pcm.numel(); n_total = n_file + wo.N_SAMPLES; n_frames = n_total // 160
    out = torch.empty((dims.n_mels, n_frames), device="cuda")
    g1 = torch.empty(1, device="cuda"); g2 = torch.empty(1, device="cuda")
    eng.reset_max(g1); eng.reset_max(g2)
    half = (n_frames // 2)
    s_lo = max(0, half * 160 - 200)
    eng.mel_stage1(pcm[: half * 160 + 400].contiguous(), 0, n_file, n_total, 0, half, out, g1)
    eng.mel_stage1(pcm[s_lo:].contiguous(), s_lo, n_file, n_total, half, n_frames - half, out[:, half:], g2)
    g = torch.maximum(g1, g2)
    eng.mel_stage2(out, n_frames, g)
    assert torch.equal(out, whole)


@pytest.mark.parametrize("name,precision,gain,xattn", [("micro", "fp32", 3.0, False), ("tiny", "fp32", 3.0, False),
                                                       ("micro", "bf16", 1.0, False), ("tiny", "bf16", 1.0, False), ("tiny", "bf16", 3.0, False),
                                                       ("micro", "bf16", 3.0, False), ("micro", "bf16", 1.0, True), ("tiny", "bf16", 1.0, True),
                                                       ("tiny", "bf16", 3.0, True), ("micro", "bf16", 3.0, True)])
def test_encoder_and_logits_parity(models, name, precision, gain, xattn):
    """bf16: both forms of the decoder cross-attention (on cached K/V; on the encoder output), each against the emulation of
    its own rounding points"""
    dims, sd, eng = models(name, precision, xattn=xattn, attn_gain=gain)
    om = wo.OracleWhisper(dims, sd)
    ks = eng.cross_key_splits if eng.cross_form == "encoder_output" else None      # the stream kernel's online softmax, restated
    oe = wo.OracleWhisper(dims, sd, emulate_bf16=True, flash_tiles=True, cross_form=eng.cross_form, cross_key_splits=ks) if precision == "bf16" else None
    o64 = wo.OracleWhisper(dims, sd, emulate_bf16=True, flash_tiles=True, acc_dtype=torch.float64, cross_form=eng.cross_form, cross_key_splits=ks) if precision == "bf16" else None
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in (0, 1)])
    with torch.no_grad():
        xa = om.encode(mels)
        xe = oe.encode(mels) if oe else None
        x64 = o64.encode(mels) if o64 else None
    eng.encode_windows(mels)
    got = eng.encoder_output(2).cpu()
    check_close(got, xa, precision, sharp=gain > 1.0, emu=xe, emu64=x64)
    tok = eng.tok
    toks = torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tok.no_timestamps, 100, 2000, 31337],
                         [tok.sot, tok.lang(3), tok.transcribe, tok.no_timestamps, 7, 50000, 421]])
    with torch.no_grad():
        ref = om.logits(toks, xa)
        ref_e = oe.logits(toks, xe) if oe else None
        ref_64 = o64.logits(toks, x64) if o64 else None
    lg = eng.decoder_logits(toks).cpu()
    assert lg.shape == ref.shape
    check_close(lg, ref, precision, sharp=gain > 1.0, emu=ref_e, emu64=ref_64)


@pytest.mark.parametrize("name,gain", [("micro", 3.0), ("tiny", 1.0), ("tiny", 3.0)])
@pytest.mark.parametrize("split", [1, 2, 3])
def test_cross_attention_stream_at_few_key_splits_bf16(models, name, gain, split):
    """The E-stream kernel at the key splits the benchmark's lanes use: 1 split = all 47 tiles of 32 keys in one workgroup (an odd
    count: 23 double iterations + the tail) and a single split through xattn_cproj_kernel's combine (S = 1); 2 splits = 23 + 24
    tiles (odd and even); 3 = 15 + 16 + 16.  Logits against the emulation of the same rounding points, greedy ids graph ==
    eager, and a window's ids do not depend on its batch."""
    dims, sd, eng = models(name, "bf16", xattn=True, split=split, attn_gain=gain, eot_boost=3.0)
    assert eng.cross_form == "encoder_output" and eng.cross_key_splits == split
    oe = wo.OracleWhisper(dims, sd, emulate_bf16=True, flash_tiles=True, cross_form="encoder_output", cross_key_splits=split)
    o64 = wo.OracleWhisper(dims, sd, emulate_bf16=True, flash_tiles=True, acc_dtype=torch.float64, cross_form="encoder_output", cross_key_splits=split)
    om = wo.OracleWhisper(dims, sd)
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in (0, 1, 2)])
    tok = eng.tok
    toks = torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tok.no_timestamps, 100, 2000, 31337]] * 3)
    with torch.no_grad():
        ref, ref_e, ref_64 = om.logits(toks, om.encode(mels)), oe.logits(toks, oe.encode(mels)), o64.logits(toks, o64.encode(mels))
    eng.encode_windows(mels)
    lg = eng.decoder_logits(toks).cpu()
    check_close(lg, ref, "bf16", sharp=gain > 1.0, emu=ref_e, emu64=ref_64)
    init = wo.initial_tokens(tok, 0)
    a = eng.decode_greedy(3, init, max_new_tokens=16, use_graph=True)["tokens"].cpu()
    b = eng.decode_greedy(3, init, max_new_tokens=16, use_graph=False)["tokens"].cpu()
    assert torch.equal(a, b)
    eng.encode_windows(mels[2:3])
    assert torch.equal(eng.decode_greedy(1, init, max_new_tokens=16)["tokens"].cpu()[0], a[2])


@pytest.mark.parametrize("name,split", [("micro", 1), ("tiny", 2), ("small1", 3), ("large1", 1), ("large1", 4)])
def test_stream_kernels_with_and_without_the_lds_image_are_bit_identical(monkeypatch, name, split):
    """Round 3's E-stream kernel (tile transposed by the matrix pipe from the registers the score product used, DMA staging, no
    LDS image) against round 2's (WX_XATTN_IMAGE=1: the tile through an LDS image and transposed LDS reads): same products in
    the same order, so logits are BIT-identical -- at every model width class (d = 128, 384, 768, 1280: 1, 3, 3 and 5 k-steps per
    wave, 4 or 8 waves holding E, one or two query tiles with the four heads past 16 packed into shared accumulators)."""
    import dataclasses
    from whisper_amd.engine import WhisperEngine
    if name == "small1":
        dims = dataclasses.replace(ck.dims_for("small"), n_audio_layer=1, n_text_layer=1)
    elif name == "large1":
        dims = dataclasses.replace(ck.dims_for("large-v3"), n_audio_layer=1, n_text_layer=2)
    else:
        dims = ck.dims_for(name)
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    monkeypatch.setenv("WX_XATTN", "1")
    monkeypatch.setenv("WX_XATTN_SPLIT", str(split))
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=3)
    try:
        assert eng.cross_form == "encoder_output" and eng.cross_key_splits == split
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in (0, 1, 2)])
        tok = eng.tok
        toks = torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tok.no_timestamps, 100, 2000, 31337]] * 3)
        eng.encode_windows(mels)
        new = eng.decoder_logits(toks).cpu()
        monkeypatch.setenv("WX_XATTN_IMAGE", "1")
        old = eng.decoder_logits(toks).cpu()
        monkeypatch.delenv("WX_XATTN_IMAGE")
        assert torch.isfinite(new).all() and torch.equal(new, old)
        init = wo.initial_tokens(tok, 0)
        a = eng.decode_greedy(3, init, max_new_tokens=10, use_graph=False)
        monkeypatch.setenv("WX_XATTN_IMAGE", "1")
        b = eng.decode_greedy(3, init, max_new_tokens=10, use_graph=False)
        assert torch.equal(a["tokens"], b["tokens"]) and torch.equal(a["sum_logprobs"], b["sum_logprobs"])
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["micro", "micro256", "tiny", "base1", "small1", "medium1", "large1"])
def test_key_split_combine_inside_the_output_projection_is_bit_identical_bf16(monkeypatch, name):
    """Up to four decode rows on the K/V form of the cross-attention: the key splits' states are combined in the prologue of the
    output projection (gemm.hip launch_gemm_acomb: one launch less per layer) instead of attn_decode_combine_kernel.  Same function,
    same rounding point, same MFMA order: logits, ids and log-probabilities are BIT-identical to the stand-alone combine
    (WX_NO_FUSE_COMBINE=1) at every width class and split count, eager and captured, and rows 0..3 of a 5-row batch (stand-alone
    combine by size) equal the same rows decoded as 4."""
    import dataclasses
    from whisper_amd.engine import WhisperEngine
    if name.endswith("1"):
        full = {"base1": "base", "small1": "small", "medium1": "medium", "large1": "large-v3"}[name]
        dims = dataclasses.replace(ck.dims_for(full), n_audio_layer=1, n_text_layer=3)
    else:
        dims = ck.dims_for(name)
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=5)
    try:
        assert eng.cross_form == "kv" and eng.cross_key_splits > 1
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(5)])
        tok = eng.tok
        toks = torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tok.no_timestamps, 100, 2000, 31337, 7, 50000]] * 5)
        init = wo.initial_tokens(tok, 0)
        eng.encode_windows(mels)
        five = eng.decoder_logits(toks).cpu()                        # 5 rows: the stand-alone combine launch
        for rows in (1, 2, 4):
            eng.encode_windows(mels[:rows])
            fused = eng.decoder_logits(toks[:rows]).cpu()
            a = eng.decode_greedy(rows, init, max_new_tokens=12, use_graph=False)
            g = eng.decode_greedy(rows, init, max_new_tokens=12, use_graph=True)
            monkeypatch.setenv("WX_NO_FUSE_COMBINE", "1")
            plain = eng.decoder_logits(toks[:rows]).cpu()
            b = eng.decode_greedy(rows, init, max_new_tokens=12, use_graph=False)
            monkeypatch.delenv("WX_NO_FUSE_COMBINE")
            assert torch.isfinite(fused).all() and torch.equal(fused, plain) and torch.equal(fused, five[:rows])
            for x in (b, g):
                assert torch.equal(a["tokens"], x["tokens"]) and torch.equal(a["sum_logprobs"], x["sum_logprobs"])
                assert torch.equal(a["no_speech_probs"], x["no_speech_probs"])
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["micro", "tiny", "large1"])
def test_self_attention_prefetch_depth_is_bit_identical_bf16(monkeypatch, name):
    """Few decode rows: the self-attention kernel keeps four key iterations' loads in flight per wave (attention.hip, R = 4) instead
    of one.  Same keys in the same order: logits at short and long positions, ids and log-probabilities are BIT-identical to
    R = 1 (WX_SELF_ATTN_R1=1), eager and captured."""
    import dataclasses
    from whisper_amd.engine import WhisperEngine
    dims = dataclasses.replace(ck.dims_for("large-v3"), n_audio_layer=1, n_text_layer=2) if name == "large1" else ck.dims_for(name)
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=3)
    try:
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(3)])
        tok = eng.tok
        rng = np.random.default_rng(3)
        toks = torch.tensor(np.concatenate([np.tile([tok.sot, tok.lang(0), tok.transcribe, tok.no_timestamps], (3, 1)),
                                            rng.integers(0, tok.eot, size=(3, 137))], axis=1))       # 141 positions: 1..5 iterations per wave
        init = wo.initial_tokens(tok, 0)
        eng.encode_windows(mels)
        deep = eng.decoder_logits(toks).cpu()
        a = eng.decode_greedy(3, init, max_new_tokens=40, use_graph=False)
        g = eng.decode_greedy(3, init, max_new_tokens=40, use_graph=True)
        monkeypatch.setenv("WX_SELF_ATTN_R1", "1")
        one = eng.decoder_logits(toks).cpu()
        b = eng.decode_greedy(3, init, max_new_tokens=40, use_graph=False)
        monkeypatch.delenv("WX_SELF_ATTN_R1")
        assert torch.isfinite(deep).all() and torch.equal(deep, one)
        for x in (b, g):
            assert torch.equal(a["tokens"], x["tokens"]) and torch.equal(a["sum_logprobs"], x["sum_logprobs"])
    finally:
        eng.close()


def test_encoder_parity_256_tiles_bf16(models):
    """3 windows of a 256-wide model: M = 4500 rows (a ragged last tile) takes the 256x256 LDS-DMA GEMM kernel; the decoder
    runs its split-K GEMMs at K = 256 / 1024."""
    dims, sd, eng = models("micro256", "bf16", attn_gain=1.0)
    om = wo.OracleWhisper(dims, sd)
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in (0, 1, 2)])
    with torch.no_grad():
        xa = om.encode(mels)
    eng.encode_windows(mels)
    check_close(eng.encoder_output(3).cpu(), xa, "bf16", sharp=False)
    tok = eng.tok
    toks = torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tok.no_timestamps, 100, 2000, 31337]] * 3)
    with torch.no_grad():
        ref = om.logits(toks, xa)
    check_close(eng.decoder_logits(toks).cpu(), ref, "bf16", sharp=False)


@pytest.mark.parametrize("name", ["micro", "tiny"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_greedy_tokens_bit_exact_fp32(models, name, use_graph):
    dims, sd, eng = models(name, "fp32", eot_boost=3.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(4)])
    init = wo.initial_tokens(tok, 0)
    with torch.no_grad():
        ref = wo.greedy_decode(om, om.encode(mels), init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()), max_new_tokens=40)
    eng.encode_windows(mels)
    r = eng.decode_greedy(4, init, max_new_tokens=40, use_graph=use_graph, check_interval=4)
    t, n = r["tokens"].cpu(), r["n_tokens"].cpu()
    got = [t[b, len(init):len(init) + int(n[b])].tolist() for b in range(4)]
    assert ref["margins"].min() > 1e-3, "fixture argmax is ill-conditioned; pick another seed"
    assert got == ref["tokens"]
    np.testing.assert_allclose(r["sum_logprobs"].cpu().numpy(), ref["sum_logprobs"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(r["no_speech_probs"].cpu().numpy(), ref["no_speech_probs"], rtol=1e-3)
    assert (t[:, :len(init)] == torch.tensor(init)).all()
    for b in range(4):                       # rows are EOT-padded after the end
        assert (t[b, len(init) + int(n[b]):] == tok.eot).all()


def test_english_only_vocabulary_greedy_fp32(models):
    """An English-only checkpoint (51 864 ids: eot 50256, sot 50257, no language tokens): start sequence [sot, no_timestamps],
    the GPT-2 non-speech suppress table, no language detection -- greedy ids, log-probabilities and no-speech probabilities
    equal the oracle's in fp32 mode, plain and with the timestamp rules."""
    dims, sd, eng = models("micro.en", "fp32", eot_boost=3.0)
    assert dims.n_vocab == 51864 and not ck.is_multilingual(dims.n_vocab)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    assert (tok.eot, tok.sot) == (50256, 50257)
    init = eng.initial_tokens(0)
    assert init == [tok.sot, tok.no_timestamps]
    sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids(dims.n_vocab))
    assert sorted(eng.suppress_ids) == sorted(sup)
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(60 + s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(3)])
    eng.encode_windows(mels)
    assert eng.detect_language(eng.log_mel(torch.from_numpy(wo.varied_audio(60))), 3000) == 0
    with torch.no_grad():
        xa = om.encode(mels)
        ref = wo.greedy_decode(om, xa, init, tok, sup, max_new_tokens=30)
    r = eng.decode_greedy(3, init, max_new_tokens=30)
    t, n = r["tokens"].cpu(), r["n_tokens"].cpu()
    assert ref["margins"].min() > 1e-3, "fixture argmax is ill-conditioned; pick another seed"
    assert [t[b, len(init):len(init) + int(n[b])].tolist() for b in range(3)] == ref["tokens"]
    np.testing.assert_allclose(r["sum_logprobs"].cpu().numpy(), ref["sum_logprobs"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(r["no_speech_probs"].cpu().numpy(), ref["no_speech_probs"], rtol=1e-3)
    init_ts = eng.initial_tokens(0, without_timestamps=False)
    assert init_ts == [tok.sot]
    with torch.no_grad():
        ref_ts = wo.greedy_decode(om, xa, init_ts, tok, sup, max_new_tokens=30, without_timestamps=False)
    r = eng.decode_greedy(3, init_ts, max_new_tokens=30, without_timestamps=False)
    t, n = r["tokens"].cpu(), r["n_tokens"].cpu()
    assert ref_ts["margins"].min() > 1e-3
    assert [t[b, len(init_ts):len(init_ts) + int(n[b])].tolist() for b in range(3)] == ref_ts["tokens"]


def test_greedy_with_timestamp_rules_fp32(models):
    dims, sd, eng = models("micro", "fp32", eot_boost=3.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(3)])
    init = wo.initial_tokens(tok, 0, without_timestamps=False)
    with torch.no_grad():
        ref = wo.greedy_decode(om, om.encode(mels), init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()),
                               max_new_tokens=48, without_timestamps=False)
    eng.encode_windows(mels)
    r = eng.decode_greedy(3, init, max_new_tokens=48, without_timestamps=False)
    t, n = r["tokens"].cpu(), r["n_tokens"].cpu()
    got = [t[b, len(init):len(init) + int(n[b])].tolist() for b in range(3)]
    assert got == ref["tokens"]
    assert all(g[0] >= tok.timestamp_begin for g in got if g)   # first sampled token must be a timestamp
    np.testing.assert_allclose(r["sum_logprobs"].cpu().numpy(), ref["sum_logprobs"], rtol=1e-4, atol=1e-3)


def test_long_form_matches_oracle_fp32(models):
    dims, sd, eng = models("micro", "fp32", max_batch=2, eot_boost=3.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    audio = wo.varied_audio(11, 16000 * 70 + 777)       # 3 windows, the last one ragged
    init = wo.initial_tokens(tok, 0)
    with torch.no_grad():
        ref = wo.transcribe_windows(om, audio, init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()), max_new_tokens=24)
    r = eng.transcribe_pcm(audio, init, max_new_tokens=24)
    assert r["n_windows"] == ref["n_windows"] == 3
    assert r["tokens"] == ref["tokens"]


@pytest.mark.parametrize("n_samples", [1, 150, 399, 401, wo.N_SAMPLES - 1, wo.N_SAMPLES, wo.N_SAMPLES + 1, wo.N_SAMPLES + 160])
def test_edge_lengths_match_oracle_fp32(models, n_samples):
    """Clips shorter than one FFT window (the reflect padding then reaches into the zero padding), of exactly one 30 s
    window, one sample longer (still one window: upstream counts whole 160-sample frames) and one frame longer (a second
    window that holds a single frame): mel, window count and token ids equal the oracle's."""
    dims, sd, eng = models("micro", "fp32", max_batch=2, eot_boost=3.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    audio = wo.varied_audio(23, n_samples)
    init = wo.initial_tokens(tok, 0)
    with torch.no_grad():
        ref = wo.transcribe_windows(om, audio, init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()), max_new_tokens=12)
    got_mel = eng.log_mel(torch.from_numpy(audio)).cpu().numpy()
    assert got_mel.shape == tuple(ref["mel"].shape)
    assert rel_err(got_mel, ref["mel"].numpy()) < 1e-4
    r = eng.transcribe_pcm(audio, init, max_new_tokens=12)
    assert r["n_windows"] == ref["n_windows"] == max(1, -(-(n_samples // 160) // 3000))
    assert r["tokens"] == ref["tokens"]


def test_long_prompt_fills_the_text_context_fp32(models):
    """A previous-text prompt longer than n_text_ctx/2 - 1 keeps its last 223 ids (upstream _get_initial_tokens), and
    decoding then runs until the text context is full: the longest decode the engine can be asked for."""
    dims, sd, eng = models("micro", "fp32", max_batch=2, eot_boost=0.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    rng = np.random.default_rng(5)
    prompt = [int(x) for x in rng.integers(0, tok.eot, size=300)]
    init = eng.initial_tokens(0, prompt_ids=prompt)
    assert init == wo.initial_tokens(tok, 0, prompt_ids=prompt, n_text_ctx=dims.n_text_ctx)
    assert len(init) == 1 + (dims.n_text_ctx // 2 - 1) + 4
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(2)])
    with torch.no_grad():
        ref = wo.greedy_decode(om, om.encode(mels), init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()))
    eng.encode_windows(mels)
    r = eng.decode_greedy(2, init)
    n = r["n_tokens"].cpu().tolist()
    t = r["tokens"].cpu()
    for b in range(2):
        # upstream's loop stops once the row holds n_text_ctx + 1 ids: it still samples from position n_text_ctx - 1
        assert len(ref["tokens"][b]) == dims.n_text_ctx + 1 - len(init)       # eot_boost 0: no EOT before the limit
        assert t[b, len(init):len(init) + n[b]].tolist() == ref["tokens"][b]


@pytest.mark.parametrize("precision,rows", [("fp32", 96), ("bf16", 72), ("bf16", 80), ("bf16", 96), ("bf16", 130)])
def test_large_decode_batches(models, precision, rows):
    """The split-K decode kernels serve up to 80 rows per pass (5 MFMA row tiles: 64 greedy windows, or 15 windows x 5
    beams); a larger bf16 batch runs as consecutive passes (96 rows = 2 x 48, 130 = 2 x 80 / 50), a larger fp32 batch on the
    general GEMM / attention kernels.  fp32 ids equal the oracle's, bf16 ids agree wherever the oracle's top-2 margin is
    comfortable, and a window of the second pass decodes to the same ids in a batch of its own."""
    dims, sd, eng = models("micro", precision, max_batch=rows, eot_boost=3.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(100 + s, 16000 * 3), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000]
                        for s in range(rows)])
    init = wo.initial_tokens(tok, 0)
    with torch.no_grad():
        ref = wo.greedy_decode(om, om.encode(mels), init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()), max_new_tokens=10)
    eng.encode_windows(mels)
    r = eng.decode_greedy(rows, init, max_new_tokens=10)
    t, n = r["tokens"].cpu(), r["n_tokens"].cpu()
    for b in range(rows):
        if precision == "fp32":
            assert t[b, len(init):len(init) + int(n[b])].tolist() == ref["tokens"][b]
        else:
            for i, want in enumerate(ref["tokens"][b]):
                if ref["margins"][i, b] < 0.15:
                    break
                assert int(t[b, len(init) + i]) == want
    if precision == "bf16" and rows > 80:
        assert eng.cross_form == "encoder_output"
        eng.encode_windows(mels[rows - 3:])
        small = eng.decode_greedy(3, init, max_new_tokens=10)["tokens"].cpu()
        assert torch.equal(small, t[rows - 3:])


@pytest.mark.parametrize("name,beam,boost,use_graph", [("micro", 5, 3.0, True), ("micro", 5, 6.0, False), ("tiny", 3, 3.0, True),
                                                     ("micro", 8, 6.0, True)])
def test_beam_search_matches_oracle_fp32(models, name, beam, boost, use_graph):
    """Beam search (BASELINE configs[3]) in fp32 mode: the selected ids, every finished candidate (order included) and their
    cumulative log-probabilities equal the oracle's restatement of upstream BeamSearchDecoder.  eot_boost 6 makes EOT a
    frequent proposal, so finished lists fill up and decoding stops early; boost 3 mostly ends at sample_len and
    exercises finalize()'s top-up from the live beams."""
    max_batch = 3 * beam
    dims, sd, eng = models(name, "fp32", max_batch=max_batch, eot_boost=boost)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(3)])
    init = wo.initial_tokens(tok, 0)
    with torch.no_grad():
        ref = wo.beam_search_decode(om, om.encode(mels), init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()),
                                    beam_size=beam, max_new_tokens=24)
    eng.encode_windows(mels)
    got = eng.decode_beam(3, init, beam_size=beam, max_new_tokens=24, use_graph=use_graph, check_interval=1)
    assert got["tokens"] == ref["tokens"]
    assert got["steps"] == ref["steps"]
    np.testing.assert_allclose(got["sum_logprobs"], ref["sum_logprobs"], rtol=1e-4, atol=2e-3)
    for b in range(3):
        assert [c[0] for c in got["candidates"][b]] == [c[0] for c in ref["candidates"][b]]
        np.testing.assert_allclose([c[1] for c in got["candidates"][b]], [c[1] for c in ref["candidates"][b]], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(got["no_speech_probs"], ref["no_speech_probs"], rtol=1e-3, atol=1e-6)
    # greedy afterwards on the same engine still works (the self-attention cache switches back to 16-token pages)
    r = eng.decode_greedy(3, init, max_new_tokens=8)
    with torch.no_grad():
        g = wo.greedy_decode(om, om.encode(mels), init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()), max_new_tokens=8)
    t, n = r["tokens"].cpu(), r["n_tokens"].cpu()
    assert [t[b, len(init):len(init) + int(n[b])].tolist() for b in range(3)] == g["tokens"]


@pytest.mark.parametrize("beam,patience", [(3, 1.5), (5, 1.5), (5, 1.7)])
def test_beam_search_patience_fp32(models, beam, patience):
    """patience != 1: max_candidates = round(beam * patience) with upstream's rounding (3 x 1.5 -> 4, 5 x 1.5 -> 8 and
    5 x 1.7 = 8.5 -> 8: ties to even) sizes the finished lists on both sides of the C-ABI; ids, candidates and scores equal
    the oracle's."""
    dims, sd, eng = models("micro", "fp32", max_batch=2 * beam, eot_boost=6.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in (5, 6)])
    init = wo.initial_tokens(tok, 0)
    with torch.no_grad():
        ref = wo.beam_search_decode(om, om.encode(mels), init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()),
                                    beam_size=beam, patience=patience, max_new_tokens=32)
    eng.encode_windows(mels)
    got = eng.decode_beam(2, init, beam_size=beam, patience=patience, max_new_tokens=32, check_interval=1)
    assert got["tokens"] == ref["tokens"] and got["steps"] == ref["steps"]
    for b in range(2):
        assert [c[0] for c in got["candidates"][b]] == [c[0] for c in ref["candidates"][b]]
        assert len(got["candidates"][b]) <= max(beam, int(round(beam * patience)))
        np.testing.assert_allclose([c[1] for c in got["candidates"][b]], [c[1] for c in ref["candidates"][b]], rtol=1e-4, atol=2e-3)


def test_beam_search_at_the_text_context_limit_fp32(models):
    """Beam search from a maximum-length previous-text prompt with no EOT in sight: like upstream, the beams take their last
    step from position n_text_ctx - 1 and end with n_text_ctx + 1 ids; finalize() then ranks the live beams."""
    dims, sd, eng = models("micro", "fp32", max_batch=4, eot_boost=0.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    rng = np.random.default_rng(6)
    init = eng.initial_tokens(0, prompt_ids=[int(x) for x in rng.integers(0, tok.eot, size=230)])
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(70 + s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(2)])
    with torch.no_grad():
        ref = wo.beam_search_decode(om, om.encode(mels), init, tok, wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()), beam_size=2)
    eng.encode_windows(mels)
    got = eng.decode_beam(2, init, beam_size=2)
    assert got["steps"] == ref["steps"] == dims.n_text_ctx + 1 - len(init)
    assert got["tokens"] == ref["tokens"] and all(len(t) == dims.n_text_ctx + 1 - len(init) for t in got["tokens"])
    np.testing.assert_allclose(got["sum_logprobs"], ref["sum_logprobs"], rtol=1e-4, atol=5e-3)


def test_beam_search_bf16_shares_cross_kv(models):
    """bf16 beam search at width 5 runs (rows = 15 <= 64: split-K decode GEMMs, grouped cross-attention) and beam width 1
    reproduces greedy decoding exactly."""
    dims, sd, eng = models("micro256", "bf16", max_batch=15, attn_gain=1.0, eot_boost=3.0)
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(3)])
    init = wo.initial_tokens(eng.tok, 0)
    eng.encode_windows(mels)
    g = eng.decode_greedy(3, init, max_new_tokens=16)
    t, n = g["tokens"].cpu(), g["n_tokens"].cpu()
    greedy = [t[b, len(init):len(init) + int(n[b])].tolist() for b in range(3)]
    b1 = eng.decode_beam(3, init, beam_size=1, max_new_tokens=16)
    assert b1["tokens"] == greedy
    b5 = eng.decode_beam(3, init, beam_size=5, max_new_tokens=16)
    assert len(b5["tokens"]) == 3 and all(len(c) >= 5 for c in b5["candidates"])
    # 15 windows x 5 beams = 75 rows (BASELINE configs[3] per GPU) in one batch: the windows that were also decoded in the
    # batch of 3 keep their result (batch invariance holds for beam search too: same key split, same GEMM k order)
    dims2, sd2, eng75 = models("micro256", "bf16", max_batch=75, attn_gain=1.0, eot_boost=3.0)
    mels15 = torch.cat([mels] + [torch.stack([wo.log_mel_spectrogram(wo.varied_audio(40 + s, 16000 * 5), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000]
                                              for s in range(12)])])
    eng75.encode_windows(mels15)
    b75 = eng75.decode_beam(15, init, beam_size=5, max_new_tokens=16)
    assert len(b75["tokens"]) == 15 and b75["tokens"][:3] == b5["tokens"]
    for b in range(3):     # the best beam can only score at least as well as the greedy path under the same model
        assert max(c[1] / max(len(c[0]), 1) for c in b5["candidates"][b]) >= g["sum_logprobs"].cpu()[b].item() / max(len(greedy[b]), 1) - 1e-2


@pytest.mark.parametrize("xattn", [False, True])
def test_prompt_pass_is_bit_identical_to_stepwise_feeding_bf16(models, monkeypatch, xattn):
    """bf16 decode feeds the forced part of the prompts several positions per pass (rows = sequences x positions).  With
    previous-text prompts of different lengths (SOT in the middle of the row, so the no-speech read falls inside a pass),
    ids, cumulative log-probabilities and no-speech probabilities are bit-identical to one-position-per-step feeding."""
    dims, sd, eng = models("micro256", "bf16", max_batch=16, xattn=xattn, attn_gain=1.0, eot_boost=3.0)
    tok = eng.tok
    rng = np.random.default_rng(8)
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(60 + s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(3)])
    eng.encode_windows(mels)
    prompts = [eng.initial_tokens(0, prompt_ids=[int(x) for x in rng.integers(0, tok.eot, size=n)]) for n in (37, 60, 19)]
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("WX_NO_PREFILL", flag)
        for temperature in (0.0, 0.7):
            r = eng.decode_rows(prompts, temperature=temperature, seed=5, max_new_tokens=12)
            outs.append((r["tokens"], [float(x) for x in r["sum_logprobs"]], [float(x) for x in r["no_speech_probs"]]))
    assert outs[0] == outs[2] and outs[1] == outs[3]
    assert all(len(t) > 0 for t in outs[0][0]) and all(0.0 < p < 1.0 for p in outs[0][2])
    # the same through the shared-prompt entry point (default 4-id prompt: three forced positions, one pass)
    g = []
    for flag in ("1", "0"):
        monkeypatch.setenv("WX_NO_PREFILL", flag)
        r = eng.decode_greedy(3, wo.initial_tokens(tok, 0), max_new_tokens=12)
        g.append((r["tokens"].cpu(), r["sum_logprobs"].cpu(), r["no_speech_probs"].cpu()))
    assert all(torch.equal(a, b) for a, b in zip(g[0], g[1]))


def test_sequential_long_form_matches_oracle_fp32(models):
    """Upstream's sequential loop (timestamps, previous-text prompt, timestamp-driven seek) at temperature 0: two files of
    different length decoded in lockstep (one batch per iteration, per-row prompts) give, file by file, the oracle's
    segments -- seek, start, end, ids -- and the same number of decoded windows."""
    from whisper_amd.longform import transcribe_sequential
    from whisper_amd.tokenizer import Tokenizer
    dims, sd, eng = models("micro", "fp32", eot_boost=3.0)
    om = wo.OracleWhisper(dims, sd)
    tok = eng.tok
    dec = Tokenizer(dims.n_vocab).decode
    sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids())
    audios = [wo.varied_audio(31, 16000 * 70), wo.varied_audio(32, 16000 * 35 + 777)]
    got = transcribe_sequential(eng, audios, dec, language_index=0, temperatures=(0.0,), max_new_tokens=20)
    for a, g in zip(audios, got):
        ref = wo.transcribe_sequential(om, a, tok, sup, dec, language_index=0, max_new_tokens=20)
        assert g["windows"] == ref["windows"] and g["tokens"] == ref["tokens"]
        assert [(s["seek"], s["tokens"]) for s in g["segments"]] == [(s["seek"], s["tokens"]) for s in ref["segments"]]
        np.testing.assert_allclose([s["start"] for s in g["segments"]], [s["start"] for s in ref["segments"]], atol=1e-9)
        np.testing.assert_allclose([s["end"] for s in g["segments"]], [s["end"] for s in ref["segments"]], atol=1e-9)
        np.testing.assert_allclose([s["avg_logprob"] for s in g["segments"]], [s["avg_logprob"] for s in ref["segments"]], rtol=1e-4, atol=1e-3)
        assert [s["needs_fallback"] for s in g["segments"]] == [s["needs_fallback"] for s in ref["segments"]]
    # every window looks like silence under an absurd threshold: all skipped, nothing emitted, same window count
    skipped = transcribe_sequential(eng, audios[1:], dec, temperatures=(0.0,), no_speech_threshold=1e-12, max_new_tokens=8)[0]
    ref = wo.transcribe_sequential(om, audios[1], tok, sup, dec, no_speech_threshold=1e-12, max_new_tokens=8)
    assert skipped["segments"] == [] and ref["segments"] == [] and skipped["windows"] == ref["windows"] == 2


def test_temperature_fallback_sampling(models):
    """T > 0 re-decodes (upstream decode_with_fallback): reproducible for a given seed, different across seeds, ids obey the
    logit filters, and a vanishing temperature reproduces greedy decoding."""
    dims, sd, eng = models("micro", "fp32", eot_boost=3.0)
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(2)])
    eng.encode_windows(mels)
    tok = eng.tok
    prompts = [eng.initial_tokens(0, without_timestamps=False), eng.initial_tokens(0, without_timestamps=False, prompt_ids=[100, 200, 300])]
    g = eng.decode_rows(prompts, temperature=0.0, max_new_tokens=16, without_timestamps=False)
    cold = eng.decode_rows(prompts, temperature=1e-4, seed=5, max_new_tokens=16, without_timestamps=False)
    assert cold["tokens"] == g["tokens"]
    a = eng.decode_rows(prompts, temperature=1.0, seed=7, max_new_tokens=16, without_timestamps=False)
    b = eng.decode_rows(prompts, temperature=1.0, seed=7, max_new_tokens=16, without_timestamps=False)
    c = eng.decode_rows(prompts, temperature=1.0, seed=8, max_new_tokens=16, without_timestamps=False)
    assert a["tokens"] == b["tokens"] and a["tokens"] != c["tokens"] and a["tokens"] != g["tokens"]
    sup = set(eng.suppress_ids)
    for row in a["tokens"] + c["tokens"]:
        assert all(0 <= t < dims.n_vocab and t not in sup and t != tok.no_timestamps for t in row)
        assert row[0] >= tok.timestamp_begin                      # timestamp rules: a window starts with a timestamp
        ts = [t for t in row if t >= tok.timestamp_begin]
        assert ts == sorted(ts)                                   # ... and timestamps never decrease
    assert all(lp < 0 for lp in a["sum_logprobs"])


@pytest.mark.parametrize("d", [512, 768, 1024])
def test_logits_gemm_wide_kernel_shapes_bf16(d):
    """The vocabulary projection (N = 51865, whole K per workgroup) runs on the persistent wide-N decode GEMM; K = 384 and
    1280 are covered by the tiny / large-v3 tests, the other model widths here with one-layer models: 80 rows (five row
    tiles) against the fp32 oracle, and bit-identical rows when the same windows are decoded two at a time (one row tile)."""
    import dataclasses
    from whisper_amd.engine import WhisperEngine
    dims = dataclasses.replace(ck.dims_for("base"), n_audio_state=d, n_audio_head=d // 64, n_audio_layer=1,
                               n_text_state=d, n_text_head=d // 64, n_text_layer=1)
    sd = ck.synthetic_state_dict(dims, seed=0, attn_gain=1.0)
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=80)
    try:
        om = wo.OracleWhisper(dims, sd)
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in (0, 1)])
        tok = eng.tok
        toks = torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tok.no_timestamps, 100, 2000],
                             [tok.sot, tok.lang(3), tok.transcribe, tok.no_timestamps, 7, 50000]])
        with torch.no_grad():
            ref = om.logits(toks, om.encode(mels))
        eng.encode_windows(mels.repeat(40, 1, 1))
        big = eng.decoder_logits(toks.repeat(40, 1)).cpu()
        check_close(big[:2], ref, "bf16", sharp=False)
        assert torch.equal(big[78:80], big[:2])
        eng.encode_windows(mels)
        assert torch.equal(eng.decoder_logits(toks).cpu(), big[:2])
    finally:
        eng.close()


def test_large_v3_full_size_bf16():
    """BASELINE configs[2] dimensions (large-v3: d 1280, 20 heads, 32 + 32 layers, 128 mel bins, vocab 51866), bf16: encoder
    output and teacher-forced logits of one window against the fp32 oracle, and the size-independent properties of the
    decode path at full size: a window decodes to the same ids alone and inside a batch, with and without hipGraph replay,
    rows are EOT padded, and the 256x256 encoder GEMM tiles / split-K decode GEMMs are the kernels that ran."""
    from whisper_amd.engine import WhisperEngine
    dims = ck.dims_for("large-v3")
    sd = ck.synthetic_state_dict(dims, seed=0, attn_gain=1.0, eot_boost=3.0)
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=4)
    try:
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in range(4)])
        om = wo.OracleWhisper(dims, sd)
        with torch.no_grad():
            xa = om.encode(mels[:1])
        eng.encode_windows(mels)
        got = eng.encoder_output(4).cpu()
        assert rms_err(got[:1], xa) < BF16_RMS_TOL
        tok = eng.tok
        toks = torch.tensor([[tok.sot, tok.lang(0), tok.transcribe, tok.no_timestamps, 100, 2000]])
        with torch.no_grad():
            ref = om.logits(toks, xa)
        lg = eng.decoder_logits(toks.repeat(4, 1)).cpu()
        assert rms_err(lg[:1], ref) < BF16_RMS_TOL
        init = eng.initial_tokens(0)
        a = eng.decode_greedy(4, init, max_new_tokens=12, use_graph=True)
        b = eng.decode_greedy(4, init, max_new_tokens=12, use_graph=False)
        ta = a["tokens"].cpu()
        assert torch.equal(ta, b["tokens"].cpu())
        eng.encode_windows(mels[2:3])
        c = eng.decode_greedy(1, init, max_new_tokens=12)
        assert torch.equal(c["tokens"].cpu()[0], ta[2])
        na = a["n_tokens"].cpu()
        for r in range(4):
            assert (ta[r, len(init) + int(na[r]):] == tok.eot).all()
    finally:
        eng.close()
