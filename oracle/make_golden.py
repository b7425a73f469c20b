#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  TEST INFRASTRUCTURE ONLY — run in the build container, never on the GPU box.

    python oracle/make_golden.py

Two kinds of vectors are written:
  * ``hf_*``     — outputs of the transformers 5.15.0 Whisper implementation (``WhisperFeatureExtractor``,
                   ``WhisperForConditionalGeneration`` built from a local ``WhisperConfig`` and loaded with the seeded
                   synthetic weights; no ``from_pretrained``, no network).  These pin oracle/whisper_oracle.py.
                   openai-whisper itself (the reference's dependency) is not installable here, so against it parity is
                   unpinned; transformers documents 1e-5 agreement of its feature extractor with the original.
  * ``hf_rules`` / ``hf_greedy_*`` — the decoding RULES from transformers' own logits processors
                   (``SuppressTokensAtBeginLogitsProcessor``, ``SuppressTokensLogitsProcessor``,
                   ``WhisperTimeStampLogitsProcessor``, ``WhisperNoSpeechDetection``, hf:generation/logits_process.py:
                   1816-2112): masks / argmax on seeded logit rows and token histories that reach every branch, and a
                   greedy loop (HF model forward + HF processors, argmax, EOT stickiness) on the seeded micro model.
                   They pin ``apply_logit_filters`` / ``greedy_decode`` of the oracle to an independent implementation.
  * ``hf_sequential.json`` — transformers' long-form helpers (``_retrieve_segment``, ``_need_fallback``,
                   ``_retrieve_avg_logprobs``) on crafted id sequences: they pin the segment / seek / fallback rules of the
                   sequential loop (oracle AND whisper_amd/longform.py) to an independent implementation.
  * ``oracle_*`` — outputs of the oracle itself (greedy ids, log-probs, margins); they pin the HIP path on the GPU box
                   and guard the oracle against regressions.  Beam search has no independent implementation in this
                   container (transformers' differs by design): [upstream-recall] only.
Inputs are regenerated from seeds at test time (whisper_amd.checkpoint.synthetic_state_dict, oracle.varied_audio), so
the fixtures hold outputs only and stay small.
"""
from __future__ import annotations

import os
import sys

os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

from oracle import whisper_oracle as wo          # noqa: E402
from whisper_amd import checkpoint as ck          # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PREFIX = [[0, 0, 0, 0, 100, 2000, 31337], [0, 3, 0, 0, 7, 50000, 421]]   # slots 0..3 are filled with sot/lang/task/notimestamps


def prefix_tokens(tok):
    rows = []
    for p in PREFIX:
        rows.append([tok.sot, tok.lang(p[1]), tok.transcribe, tok.no_timestamps] + p[4:])
    return torch.tensor(rows)


def build_hf(dims, sd):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    cfg = WhisperConfig(
        vocab_size=dims.n_vocab, num_mel_bins=dims.n_mels, encoder_layers=dims.n_audio_layer,
        encoder_attention_heads=dims.n_audio_head, decoder_layers=dims.n_text_layer,
        decoder_attention_heads=dims.n_text_head, decoder_ffn_dim=4 * dims.n_text_state,
        encoder_ffn_dim=4 * dims.n_audio_state, d_model=dims.n_audio_state, max_source_positions=dims.n_audio_ctx,
        max_target_positions=dims.n_text_ctx)
    m = WhisperForConditionalGeneration(cfg).eval()
    hsd = {ck.canonical_to_hf_name(k): v for k, v in sd.items()}
    hsd["proj_out.weight"] = sd["decoder.token_embedding.weight"]
    res = m.load_state_dict(hsd, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res
    return m


def rule_cases(tok, n_vocab):
    """Token histories (after the prompt) and logit shaping for the decoding-rule fixtures: every branch of the timestamp
    rules, with and without the "timestamps outweigh every text token" override."""
    tb = tok.timestamp_begin
    hist = [
        [],                                   # first sampled position: only timestamps <= max_initial survive; blank/eot masked
        [tb + 0],                             # <ts>: penultimate counts as timestamp -> must be text next
        [tb + 10, 500],                       # <ts> text: timestamps below the last one are forbidden
        [tb + 1, 500, tb + 20],               # text <ts>: next must be a timestamp (>= last) or EOT
        [tb + 1, tb + 20],                    # <ts><ts>: next must be text
        [tb + 3, 400, 401, tb + 40, tb + 40, 77],      # second segment running
        [tb + 3, 400, 401, tb + 40, tb + 40],          # second segment just opened
        [tb + 1499],                                    # last timestamp of the window
        [tb + 5, 300, tb + 1500],                       # text then <|30.00|>
        [600, 601],                                     # no timestamp yet past the first position (prompted continuation)
    ]
    cases = []
    for h in hist:
        for boost_ts in (0.0, 14.0):          # 14: the timestamp mass outweighs the best text token -> override fires
            cases.append((h, boost_ts))
    return cases


def rule_logits(seed: int, n_vocab: int, tok, boost_ts: float) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    lg = torch.randn(n_vocab, generator=g) * 3.0
    if boost_ts:
        lg[tok.timestamp_begin:] += boost_ts
    return lg


def hf_processors(tok, sup, begin_index, max_initial=50):
    import types
    from transformers.generation.logits_process import (LogitsProcessorList, SuppressTokensAtBeginLogitsProcessor,
                                                        SuppressTokensLogitsProcessor, WhisperTimeStampLogitsProcessor)
    gc = types.SimpleNamespace(eos_token_id=tok.eot, bos_token_id=tok.eot, no_timestamps_token_id=tok.no_timestamps,
                               max_initial_timestamp_index=max_initial)
    plain = LogitsProcessorList([SuppressTokensAtBeginLogitsProcessor([tok.blank, tok.eot], begin_index),
                                 SuppressTokensLogitsProcessor(sup)])
    with_ts = LogitsProcessorList(list(plain) + [WhisperTimeStampLogitsProcessor(gc, begin_index=begin_index)])
    return plain, with_ts


def make_rule_fixtures():
    """hf_rules.npz: transformers' processors on seeded logit rows (order of hf:models/whisper/generation_whisper.py:1774-1812)."""
    from transformers.generation.logits_process import WhisperNoSpeechDetection
    out = {}
    for n_vocab in (51865, 51866):
        tok = ck.special_tokens_for(n_vocab)
        sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids(n_vocab))
        prompt = [tok.sot, tok.lang(0), tok.transcribe]
        masks, argmax, lps, nsp = [], [], [], []
        for ci, (h, boost) in enumerate(rule_cases(tok, n_vocab)):
            lg = rule_logits(1000 + ci, n_vocab, tok, boost)
            ids = torch.tensor([prompt + h])
            _, procs = hf_processors(tok, sup, begin_index=len(prompt))
            sc = procs(ids, lg[None].clone())[0]
            masks.append(np.packbits(torch.isinf(sc).numpy()))
            argmax.append(int(sc.argmax()))
            lps.append(float(torch.log_softmax(sc.float(), -1)[sc.argmax()]))
            det = WhisperNoSpeechDetection(no_speech_token=tok.no_speech, begin_index=1)
            det(torch.tensor([[tok.sot]]), lg[None].clone())
            nsp.append(float(det.no_speech_prob[0]))
        out[f"mask_{n_vocab}"] = np.stack(masks)
        out[f"argmax_{n_vocab}"] = np.asarray(argmax, np.int32)
        out[f"logprob_{n_vocab}"] = np.asarray(lps, np.float32)
        out[f"no_speech_{n_vocab}"] = np.asarray(nsp, np.float32)
    np.savez_compressed(os.path.join(OUT, "hf_rules.npz"), **out)


def make_hf_greedy(name="micro", n_windows=4):
    """hf_greedy_<name>.npz: greedy decoding with the HF model forward and the HF logits processors; the loop around them
    (argmax, sum of log-probabilities while alive, EOT stickiness, stop when all rows ended) is upstream's GreedyDecoder."""
    from transformers.generation.logits_process import WhisperNoSpeechDetection
    dims = ck.dims_for(name)
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    tok = ck.special_tokens_for(dims.n_vocab)
    sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids(dims.n_vocab))
    hf = build_hf(dims, sd)
    mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :wo.N_FRAMES] for s in range(n_windows)])
    out = {"suppress_ids": np.asarray(sup, np.int32)}
    with torch.no_grad():
        enc = hf.model.encoder(mels)
        for mode, wts, n_new in (("plain", True, 40), ("ts", False, 48)):
            init = wo.initial_tokens(tok, 0, without_timestamps=wts)
            plain, with_ts = hf_processors(tok, sup, begin_index=len(init))
            procs = plain if wts else with_ts
            ids = torch.tensor([init] * n_windows)
            sums = torch.zeros(n_windows)
            nsp = None
            for step in range(n_new):
                lg_all = hf(encoder_outputs=enc, decoder_input_ids=ids).logits
                if step == 0:
                    det = WhisperNoSpeechDetection(no_speech_token=tok.no_speech, begin_index=1)
                    det(ids[:, :1], lg_all[:, init.index(tok.sot)].clone())
                    nsp = det.no_speech_prob.clone()
                sc = procs(ids, lg_all[:, -1].clone())
                nxt = sc.argmax(-1)
                lp = torch.log_softmax(sc.float(), -1)[torch.arange(n_windows), nxt]
                alive = ids[:, -1] != tok.eot
                sums = sums + lp * alive
                nxt = torch.where(alive, nxt, torch.full_like(nxt, tok.eot))
                ids = torch.cat([ids, nxt[:, None]], 1)
                if bool((ids[:, -1] == tok.eot).all()):
                    break
            body = ids[:, len(init):]
            L = body.shape[1]
            arr = np.full((n_windows, L), -1, np.int32)
            for b in range(n_windows):
                seq = body[b].tolist()
                seq = seq[:seq.index(tok.eot)] if tok.eot in seq else seq
                arr[b, :len(seq)] = seq
            out.update({f"{mode}_init": np.asarray(init, np.int32), f"{mode}_ids": arr,
                        f"{mode}_sum_logprobs": sums.numpy().astype(np.float32), f"{mode}_no_speech": nsp.numpy().astype(np.float32)})
    np.savez_compressed(os.path.join(OUT, f"hf_greedy_{name}.npz"), **out)


def sequential_cases(tb):
    """Crafted id sequences of one window (after the prompt, EOT stripped) x (seek, frames in the window)."""
    seqs = {
        "no_timestamps": [500, 501, 502],
        "one_trailing_timestamp": [tb, 500, 501, tb + 120],
        "pairs_then_open_text": [tb, 500, tb + 100, tb + 100, 600, 601, tb + 250, tb + 250, 700],
        "pairs_then_lone_ending": [tb, 500, tb + 100, tb + 100, 600, tb + 300],
        "only_zero_timestamp": [tb],
        "ends_at_30s_lone": [tb, 500, tb + 1500],
        "ends_at_30s_pair": [tb, 500, tb + 1500, tb + 1500],
        "timestamp_in_the_middle": [500, tb + 50, 501],
        "opens_with_a_pair": [tb + 10, tb + 10, 500, tb + 90, tb + 90],
        "pair_at_the_very_start_only": [tb, tb, 500, 501],
        "three_segments_lone_ending": [tb + 2, 400, tb + 40, tb + 40, 401, 402, tb + 99, tb + 99, 403, tb + 140],
        "text_then_pair_then_timestamp_text": [300, tb + 20, tb + 20, tb + 30, 301],
        "single_text_token": [777],
        "two_timestamps_only": [tb + 5, tb + 5],
    }
    windows = [(0, 3000), (4200, 3000), (9000, 1234)]
    return seqs, windows


def make_sequential_fixtures():
    """hf_sequential.json: transformers' long-form helpers on crafted inputs -- ``_retrieve_segment`` (segment cuts, times, seek
    advance; hf:models/whisper/generation_whisper.py:1977-2074), ``_need_fallback`` (:1243-1287) and ``_retrieve_avg_logprobs``
    (:1958-1974).  They pin oracle.split_segments / fallback_decision / avg_logprob and whisper_amd.longform's counterparts to an
    independent implementation.  Known differences between transformers and upstream openai-whisper [upstream-recall], kept
    in the fixture as `hf_differs`:
      * tokens of the LAST segment when the ids do not end in a lone timestamp: HF keeps the second timestamp of the closing
        pair in it (slices[-1] += 1, "to know it was no single ending"), upstream does not; times and seek agree;
      * HF computes the compression ratio on token BYTES, upstream on the decoded TEXT: only the predicate on the value is pinned;
      * HF cancels a fallback for silence only when the log-probability is ALSO below its threshold, upstream whenever
        no_speech_prob > threshold;
      * should_skip at avg_logprob == threshold exactly: HF needs `<`, upstream skips unless `>`."""
    import json
    import types
    from transformers.generation.logits_process import WhisperNoSpeechDetection
    from transformers.models.whisper.generation_whisper import WhisperGenerationMixin as G
    n_vocab = 51865
    tok = ck.special_tokens_for(n_vocab)
    tb = tok.timestamp_begin
    seqs, windows = sequential_cases(tb)
    out = {"timestamp_begin": tb, "time_precision": 0.02, "input_stride": 2, "segments": [], "fallback": [], "avg_logprob": []}
    for name, ids in seqs.items():
        for seek, frames in windows:
            segs, off = G._retrieve_segment(
                seek_sequence=torch.tensor(ids), seek_outputs=[None], time_offset=torch.tensor([seek * 0.02 / 2], dtype=torch.float64),
                timestamp_begin=tb, seek_num_frames=torch.tensor([frames]), time_precision=0.02, time_precision_features=0.01,
                input_stride=2, prev_idx=0, idx=0, return_token_timestamps=False, decoder_input_ids=torch.zeros((1, 4), dtype=torch.long))
            out["segments"].append({"case": name, "ids": ids, "seek": seek, "frames": frames,
                                    "hf": [{"start": float(sg["start"]), "end": float(sg["end"]), "tokens": sg["tokens"].tolist()} for sg in segs],
                                    "hf_advance": int(off)})
    # fallback predicates: a window of 30 ids whose token bytes compress badly (random ids) or well (one id repeated), each
    # id with log-probability L exactly (the scores are a normalised log-probability row), no-speech probability P
    host = type("Host", (G,), {})()
    g = torch.Generator().manual_seed(0)
    rand_ids = torch.randint(0, 50000, (30,), generator=g)
    same_ids = torch.full((30,), 1234)
    for ids_name, ids in (("incompressible", rand_ids), ("repetitive", same_ids)):
        for L in (-1.5, -1.0, -0.5):
            scores = []
            for t in ids.tolist():
                row = torch.full((n_vocab,), float(np.log((1.0 - np.exp(L)) / (n_vocab - 1))), dtype=torch.float64)
                row[t] = L
                scores.append(row)
            for P in (0.1, 0.6, 0.9):
                det = object.__new__(WhisperNoSpeechDetection)
                det._no_speech_prob = torch.tensor([P])       # what its __call__ leaves (logits_process.py:2110)
                gc = types.SimpleNamespace(compression_ratio_threshold=2.4, logprob_threshold=-1.0, no_speech_threshold=0.6)
                nf, skip = host._need_fallback(ids, [{"scores": scores}], 0, [det], gc, n_vocab, 0.0)
                out["fallback"].append({"ids": ids_name, "compression_ratio": float(G._retrieve_compression_ratio(ids, n_vocab)),
                                        "avg_logprob": float(G._retrieve_avg_logprobs(scores, ids, 0.0)), "no_speech_prob": P,
                                        "hf_needs_fallback": bool(nf), "hf_should_skip": bool(skip)})
    # avg_logprob: n sampled ids + EOT, seeded logits: HF divides the sum of ALL n + 1 log-probabilities by n + 1
    for n in (1, 7, 30):
        g = torch.Generator().manual_seed(100 + n)
        logits = torch.randn((n + 1, 997), generator=g, dtype=torch.float64) * 2.0
        ids = torch.cat([torch.randint(0, 996, (n,), generator=g), torch.tensor([996])])     # 996 plays EOT
        lp = torch.log_softmax(logits, -1)[torch.arange(n + 1), ids]
        out["avg_logprob"].append({"n_tokens": n, "sum_logprob": float(lp.sum()),
                                   "hf": float(G._retrieve_avg_logprobs([r for r in logits], ids, 0.0))})
    with open(os.path.join(OUT, "hf_sequential.json"), "w") as f:
        json.dump(out, f, indent=0)


def main():
    if sys.argv[1:] == ["sequential"]:
        make_sequential_fixtures()
        return
    from transformers import WhisperFeatureExtractor
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # ---- log-mel: HF feature extractor on one 30 s clip (reflect padding at the clip end = oracle with padding=0)
    for n_mels in (80, 128):
        audio = wo.varied_audio(3)
        fe = WhisperFeatureExtractor(feature_size=n_mels)
        mel = fe(audio, sampling_rate=16000, return_tensors="np").input_features[0].astype(np.float32)
        np.savez_compressed(os.path.join(OUT, f"hf_mel_{n_mels}.npz"),
                            audio_seed=3, frames_stride7=mel[:, ::7], row_sums=mel.astype(np.float64).sum(axis=1),
                            col_sums=mel.astype(np.float64).sum(axis=0)[::3], filters=fe.mel_filters.T.astype(np.float32))

    # ---- encoder output + teacher-forced logits: HF model with the seeded weights
    for name in ("micro", "tiny"):
        dims = ck.dims_for(name)
        sd = ck.synthetic_state_dict(dims, seed=0)
        tok = ck.special_tokens_for(dims.n_vocab)
        hf = build_hf(dims, sd)
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :wo.N_FRAMES] for s in (0, 1)])
        toks = prefix_tokens(tok)
        with torch.no_grad():
            xa = hf.model.encoder(mels).last_hidden_state
            lg = hf(input_features=mels, decoder_input_ids=toks).logits
        top = lg.topk(8, dim=-1)
        np.savez_compressed(os.path.join(OUT, f"hf_model_{name}.npz"),
                            enc_rows_stride50=xa[:, ::50].numpy(), enc_row_norms=xa.norm(dim=-1).numpy(),
                            logits_stride97=lg[..., ::97].numpy(), logits_top_values=top.values.numpy(),
                            logits_top_indices=top.indices.numpy().astype(np.int32), tokens=toks.numpy().astype(np.int32))

    # ---- decoding rules: oracle greedy (ragged EOT via eot_boost), with and without timestamp rules
    for name in ("micro", "tiny"):
        dims = ck.dims_for(name)
        sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
        tok = ck.special_tokens_for(dims.n_vocab)
        om = wo.OracleWhisper(dims, sd)
        sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids())
        mels = torch.stack([wo.log_mel_spectrogram(wo.varied_audio(s), dims.n_mels, padding=wo.N_SAMPLES)[:, :wo.N_FRAMES] for s in range(4)])
        out = {"suppress_ids": np.asarray(sup, np.int32)}
        with torch.no_grad():
            xa = om.encode(mels)
            for mode, wts, n_new in (("plain", True, 40), ("ts", False, 48)):
                init = wo.initial_tokens(tok, 0, without_timestamps=wts)
                r = wo.greedy_decode(om, xa, init, tok, sup, max_new_tokens=n_new, without_timestamps=wts)
                L = max(len(t) for t in r["tokens"])
                ids = np.full((4, max(L, 1)), -1, np.int32)
                for b, t in enumerate(r["tokens"]):
                    ids[b, :len(t)] = t
                margins = np.where(np.isfinite(r["margins"]), r["margins"], 1e9).astype(np.float32)
                out.update({f"{mode}_init": np.asarray(init, np.int32), f"{mode}_ids": ids,
                            f"{mode}_sum_logprobs": r["sum_logprobs"].astype(np.float32),
                            f"{mode}_no_speech": r["no_speech_probs"].astype(np.float32), f"{mode}_margins": margins,
                            f"{mode}_steps": r["steps"]})
        np.savez_compressed(os.path.join(OUT, f"oracle_greedy_{name}.npz"), **out)
    make_rule_fixtures()
    make_hf_greedy("micro")
    make_sequential_fixtures()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
