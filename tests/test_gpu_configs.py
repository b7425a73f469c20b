"""GPU tests at the BASELINE.json configurations themselves, through the C-ABI:

  configs[2]  large-v3 bf16 greedy, max_batch = 64, 64 windows -- the configuration bench.py times (same weights, same audio
              generator, same whole-buffer mel clamp);
  configs[3]  large-v3 beam = 5, 15 windows per GPU (75 rows = the per-GPU share of a 1 h file over 8 GPUs);
  configs[4]  large-v3-turbo (4 decoder layers), batch 64.
(configs[1], base / small at batch 16, is tests/test_gpu_api.py::test_properties_at_scale_bf16; configs[0] is the CPU path.)

The yardstick for bf16 is the oracle's bf16-EMULATING mode (same rounding points as the engine).  What is left between the two
is summation order, which bf16 rounding amplifies to the model's bf16 NOISE FLOOR (see tests/test_gpu_parity.py): the floor is
measured as the distance between the emulation with fp32 and with fp64 accumulation, and the device must be within
BF16_NOISE_FACTOR floors of the emulation on the sharp (attn_gain 3) weights the benchmark uses.  The distance to the plain
fp32 oracle is printed for the record, not asserted."""
import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from whisper_amd import checkpoint as ck
from whisper_amd.audio import synthetic_audio

pytestmark = pytest.mark.gpu

BF16_NOISE_FACTOR = 1.5   # ||hip - emulation|| <= this x ||emulation(fp64 acc) - emulation(fp32 acc)||, encoder output and logits
                          # (measured 0.98-1.08 at these sizes; the emulation restates the kernels' online softmax, tests/test_gpu_parity.py)


def rms_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))


_SD = {}
_EMU = {}      # emulation results per (weights key, probe windows): they do not depend on the engine's launch configuration


def weights(name, **kw):
    """the benchmark's weights: synthetic_state_dict(dims, seed=0) with its defaults (bench.py), or with the given overrides"""
    key = (name, tuple(sorted(kw.items())))
    if key not in _SD:
        _SD.clear()                      # one 6 GB state dict at a time
        dims = ck.dims_for(name)
        _SD[key] = (dims, ck.synthetic_state_dict(dims, seed=0, **kw))
    return _SD[key]


def bench_mel(eng, n_windows):
    """the benchmark's input: windows i = synthetic_audio(i) of one long buffer, whole-buffer clamp"""
    pcm = torch.from_numpy(np.concatenate([synthetic_audio(w) for w in range(n_windows)])).cuda()
    return eng.log_mel(pcm), pcm.numel() // 160


def check_windows_against_emulation(eng, dims, sd, mel, n_windows, probe, n_steps=12, emu_key=None, min_clear=0.0):
    """Encoder output, teacher-forced logits and margin-gated greedy ids of the windows in `probe`, decoded inside the full
    batch, against the bf16-emulating oracle run on those windows alone.  `emu_key`: cache the emulation's encoder outputs
    under this key (two engines that differ in launch configuration only share them; the decoder side is re-evaluated at each
    engine's own key split).  `min_clear`:
    the fraction of greedy steps whose emulated top-2 margin must exceed delta, i.e. where the arg-max itself is asserted."""
    tok = eng.tok
    assert eng.cross_form == "encoder_output"            # 32 rows and up: cross-attention on the encoder output
    # the emulation restates the stream kernel's online softmax at THIS engine's key split (oracle: _online_context)
    om = wo.OracleWhisper(dims, sd, emulate_bf16=True, flash_tiles=True, cross_form=eng.cross_form, cross_key_splits=eng.cross_key_splits)
    mw = torch.stack([mel[:, w * 3000:(w + 1) * 3000].cpu() for w in probe])
    init = eng.initial_tokens(0)
    toks = torch.tensor([init + [100, 2000, 31337]] * len(probe))
    o64 = wo.OracleWhisper(dims, sd, emulate_bf16=True, flash_tiles=True, acc_dtype=torch.float64, cross_form=eng.cross_form, cross_key_splits=eng.cross_key_splits)
    if emu_key is not None and emu_key in _EMU:          # the encoder side does not depend on the launch configuration
        xa, x64, mw0 = _EMU[emu_key]
        assert torch.equal(mw0, mw)                      # (nor does the log-mel)
    else:
        with torch.no_grad():
            xa = om.encode(mw)
            x64 = o64.encode(mw[:1])                     # the noise floor: one window is enough
        if emu_key is not None:
            _EMU[emu_key] = (xa, x64, mw)
    with torch.no_grad():
        ref_lg = om.logits(toks, xa)
        lg64 = o64.logits(toks[:1], x64)
    del o64
    f_enc, f_lg = rms_err(x64, xa[:1]), rms_err(lg64, ref_lg[:1])
    got_xa = eng.encoder_output(n_windows).cpu()[probe]
    e_enc = rms_err(got_xa, xa)
    lg = eng.decoder_logits(toks[:1].repeat(n_windows, 1)).cpu()[probe]
    e_lg = rms_err(lg, ref_lg)
    print(f"\n[{len(probe)} windows of {n_windows}] bf16 vs emulation: encoder {e_enc:.3e} = {e_enc / f_enc:.2f} noise floors, "
          f"logits {e_lg:.3e} = {e_lg / f_lg:.2f} noise floors")
    assert e_enc < BF16_NOISE_FACTOR * f_enc and e_lg < BF16_NOISE_FACTOR * f_lg
    # Greedy ids.  At these sizes the logit noise floor (a few percent of a logit spread of ~2) is of the order of the
    # top-1 / top-2 margins of random weights, so the emulation's own greedy path diverges from any other correct
    # evaluation within a few steps.  The check is therefore made ALONG THE DEVICE'S PATH: the emulation is teacher-forced
    # with the ids the device chose, its logits are filtered with the oracle's rules, and every device id must be an
    # arg-max up to the noise -- its emulated logit within `delta` of the row maximum -- and THE arg-max wherever the
    # emulation's margin exceeds `delta`.
    r = eng.decode_greedy(n_windows, init, max_new_tokens=n_steps)
    t = r["tokens"].cpu()
    delta = 6.0 * f_lg * float(ref_lg.double().pow(2).mean().sqrt())
    exact = clear = 0
    for j, w in enumerate(probe):
        seq = t[w, :len(init) + n_steps].long()[None]
        with torch.no_grad():
            lg_tf = om.logits(seq, xa[j:j + 1])[0]
        for i in range(n_steps):
            pos = len(init) - 1 + i
            row = lg_tf[pos][None].clone()
            wo.apply_logit_filters(row, seq[:, :pos + 1], len(init), tok, eng.suppress_ids)
            chosen = int(t[w, pos + 1])
            top2 = row[0].topk(2).values
            assert float(row[0, chosen]) >= float(top2[0]) - delta, (w, i, float(row[0, chosen]), float(top2[0]), delta)
            exact += int(chosen == int(row[0].argmax()))
            if float(top2[0] - top2[1]) > delta:
                clear += 1
                assert chosen == int(row[0].argmax()), (w, i)
    print(f"greedy ids along the device path: {exact} of {len(probe) * n_steps} are the emulation's arg-max, {clear} with a margin above "
          f"delta = {delta:.3f} (arg-max asserted there: {100.0 * clear / (len(probe) * n_steps):.0f} % of the steps); all within delta of the maximum")
    assert exact >= len(probe) * n_steps // 2
    assert clear >= min_clear * len(probe) * n_steps, (clear, len(probe) * n_steps)
    return t, init, (xa, ref_lg, lg, f_lg)


_LG_BY_LANES = {}


@pytest.mark.parametrize("n_lanes", [4, 1])
def test_large_v3_bf16_at_the_benchmarked_batch(n_lanes):
    """configs[2] as benchmarked: max_batch = 64 and n_lanes = 4 -- bench.py's default engine configuration, which sizes the
    cross-attention stream at ONE key split per row (64 workgroups, 47 tiles each, a single split through xattn_cproj_kernel) --
    and the one-lane configuration beside it (4 key splits: 256 workgroups); 64 windows in one batch."""
    from whisper_amd.engine import WhisperEngine
    dims, sd = weights("large-v3")
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=64, n_lanes=n_lanes)
    try:
        assert eng.cross_key_splits == (1 if n_lanes == 4 else 4)
        mel, content = bench_mel(eng, 64)
        eng.encode(mel, 3000, mel.stride(0), content, 64)
        t, init, (xa, ref_lg, lg, f_lg) = check_windows_against_emulation(eng, dims, sd, mel, 64, [0, 63], emu_key=("large-v3", 0, 63))
        # the two launch configurations round the partial contexts at different points (bf16 per key split): their logits
        # differ by bf16 noise, bounded like the distance to the emulation (lanes change bf16 results, never fp32 ones)
        _LG_BY_LANES[n_lanes] = lg
        if len(_LG_BY_LANES) == 2:
            e = rms_err(_LG_BY_LANES[1], _LG_BY_LANES[4])
            print(f"logits at 1 vs 4 key splits: {e:.3e} = {e / f_lg:.2f} noise floors")
            assert e < BF16_NOISE_FACTOR * f_lg
        if n_lanes == 4:
            # for the record: distance of the bf16 path from the plain fp32 oracle on the same two windows
            om32 = wo.OracleWhisper(dims, sd)
            with torch.no_grad():
                x32 = om32.encode(torch.stack([mel[:, w * 3000:(w + 1) * 3000].cpu() for w in (0, 63)]))
            print(f"bf16 emulation vs fp32 oracle, encoder output: {rms_err(xa, x32):.3e} (sharp weights)")
        # graph replay == eager launches at this batch
        e = eng.decode_greedy(64, init, max_new_tokens=12, use_graph=False)["tokens"].cpu()
        assert torch.equal(t, e)
        # window 17 decoded alone gives the ids it got inside the batch of 64 (batch invariance at the benchmarked max_batch)
        eng.encode(mel[:, 17 * 3000:], 3000, mel.stride(0), content - 17 * 3000, 1)
        one = eng.decode_greedy(1, init, max_new_tokens=12)["tokens"].cpu()
        assert torch.equal(one[0], t[17])
        tok = eng.tok
        assert (t[:, :len(init)] == torch.tensor(init)).all() and (t[:, len(init) + 12:] == tok.eot).all()
    finally:
        eng.close()


def test_large_v3_beam5_on_15_windows():
    """configs[3] per GPU: 15 windows x 5 beams = 75 rows in one batch.  Beam width 1 reproduces greedy decoding exactly;
    the first 3 windows decode to the same candidates in a 3-window run on the same engine."""
    from whisper_amd.engine import WhisperEngine
    dims, sd = weights("large-v3")
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=75)
    try:
        mel, content = bench_mel(eng, 15)
        init = eng.initial_tokens(0)
        eng.encode(mel, 3000, mel.stride(0), content, 15)
        g = eng.decode_greedy(15, init, max_new_tokens=10)
        tg, ng = g["tokens"].cpu(), g["n_tokens"].cpu()
        greedy = [tg[b, len(init):len(init) + int(ng[b])].tolist() for b in range(15)]
        b1 = eng.decode_beam(15, init, beam_size=1, max_new_tokens=10)
        assert b1["tokens"] == greedy
        b5 = eng.decode_beam(15, init, beam_size=5, max_new_tokens=10)
        assert len(b5["tokens"]) == 15 and all(len(c) >= 5 for c in b5["candidates"])
        for b in range(15):        # (the greedy path may fall out of a width-5 beam: no ordering against greedy is asserted)
            seqs = [tuple(c[0]) for c in b5["candidates"][b]]
            assert len(set(seqs)) == len(seqs) and all(np.isfinite(c[1]) and c[1] < 0 for c in b5["candidates"][b])
            assert all(len(q) == 10 for q in seqs)                      # no EOT with these weights: every hypothesis runs to the cap
        eng.encode(mel, 3000, mel.stride(0), content, 3)
        b3 = eng.decode_beam(3, init, beam_size=5, max_new_tokens=10)
        assert b3["tokens"] == b5["tokens"][:3]
        assert [[c[0] for c in cs] for cs in b3["candidates"]] == [[c[0] for c in cs] for cs in b5["candidates"][:3]]
    finally:
        eng.close()


def test_large_v3_bf16_ids_on_well_conditioned_weights():
    """The id-level check with teeth: large-v3 bf16 at the benchmarked engine configuration (max_batch 64, n_lanes 4 -> 1 key
    split) on weights whose bf16 noise floor is a few 1e-3 (attn_gain 1: the floor of the sharp benchmark weights is 8 %, of
    the order of their top-2 margins, so almost no step can be asserted there).  4 windows x 24 greedy steps inside the
    batch of 64: along the device's path every id is THE arg-max of the emulation wherever the emulation's margin exceeds
    delta = 6 floors x rms(logits) -- and that must hold for at least 80 % of the steps."""
    from whisper_amd.engine import WhisperEngine
    dims, sd = weights("large-v3", attn_gain=1.0)
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=64, n_lanes=4)
    try:
        assert eng.cross_key_splits == 1
        mel, content = bench_mel(eng, 64)
        eng.encode(mel, 3000, mel.stride(0), content, 64)
        check_windows_against_emulation(eng, dims, sd, mel, 64, [0, 21, 42, 63], n_steps=24, min_clear=0.8)
    finally:
        eng.close()


def test_large_v3_fp32_token_exact():
    """north_star: "fp32 greedy tokens identical to the reference" at the TARGET model.  fp32 engine, large-v3 dimensions
    (d 1280, 20 heads, 32 + 32 layers, 128 mel bins, vocab 51 866): gemm_f32_tile_kernel, attn_rows_kernel, the K/V-cache
    cross-attention and the sampler at full width.  2 windows, 16 greedy steps: ids bit-exact against the fp32 oracle,
    cumulative log-probabilities to 1e-4, encoder output and logits to the fp32 tolerance."""
    from whisper_amd.engine import WhisperEngine
    dims, sd = weights("large-v3", eot_boost=3.0)
    eng = WhisperEngine(dims, sd, precision="fp32", max_batch=2)
    try:
        assert eng.cross_form == "kv"
        tok = eng.tok
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :3000] for s in (0, 1)])
        init = wo.initial_tokens(tok, 0)
        om = wo.OracleWhisper(dims, sd)
        with torch.no_grad():
            xa = om.encode(mels)
            ref = wo.greedy_decode(om, xa, init, tok, eng.suppress_ids, max_new_tokens=16)
        eng.encode_windows(mels)
        got_xa = eng.encoder_output(2).cpu()
        e_enc = float((got_xa - xa).abs().max() / xa.abs().max())
        r = eng.decode_greedy(2, init, max_new_tokens=16)
        t, n = r["tokens"].cpu(), r["n_tokens"].cpu()
        got = [t[b, len(init):len(init) + int(n[b])].tolist() for b in range(2)]
        print(f"\nlarge-v3 fp32: encoder output max rel err {e_enc:.2e}; min top-2 margin {ref['margins'].min():.3e}; ids {got[0][:8]}...")
        assert e_enc < 2e-4
        assert ref["margins"].min() > 1e-3, "fixture argmax is ill-conditioned; pick another seed"
        assert got == ref["tokens"]
        np.testing.assert_allclose(r["sum_logprobs"].cpu().numpy(), ref["sum_logprobs"], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(r["no_speech_probs"].cpu().numpy(), ref["no_speech_probs"], rtol=1e-3)
        e2 = eng.decode_greedy(2, init, max_new_tokens=16, use_graph=False)["tokens"].cpu()
        assert torch.equal(e2, t)
    finally:
        eng.close()


def test_large_v3_turbo_bf16_batch_64():
    """configs[4] dimensions (32 encoder layers, 4 decoder layers), batch 64."""
    from whisper_amd.engine import WhisperEngine
    dims, sd = weights("large-v3-turbo")
    assert dims.n_text_layer == 4 and dims.n_audio_layer == 32
    eng = WhisperEngine(dims, sd, precision="bf16", max_batch=64)
    try:
        mel, content = bench_mel(eng, 64)
        eng.encode(mel, 3000, mel.stride(0), content, 64)
        t, init, _ = check_windows_against_emulation(eng, dims, sd, mel, 64, [41])
        eng.encode(mel[:, 41 * 3000:], 3000, mel.stride(0), content - 41 * 3000, 1)
        one = eng.decode_greedy(1, init, max_new_tokens=12)["tokens"].cpu()
        assert torch.equal(one[0], t[41])
    finally:
        eng.close()
