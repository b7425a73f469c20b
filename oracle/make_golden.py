#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  TEST INFRASTRUCTURE ONLY — run in the build container, never on the GPU box.

    python oracle/make_golden.py

Two kinds of vectors are written:
  * ``hf_*``     — outputs of the transformers 5.15.0 Whisper implementation (``WhisperFeatureExtractor``,
                   ``WhisperForConditionalGeneration`` built from a local ``WhisperConfig`` and loaded with the seeded
                   synthetic weights; no ``from_pretrained``, no network).  These pin oracle/whisper_oracle.py.
                   openai-whisper itself (the reference's dependency) is not installable here, so against it parity is
                   unpinned; transformers documents 1e-5 agreement of its feature extractor with the original.
  * ``oracle_*`` — outputs of the oracle itself for the decoding rules (greedy ids, log-probs, margins), which the HF
                   generate() stack does not reproduce one-to-one; they pin the HIP path on the GPU box and guard the
                   oracle against regressions.
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


def main():
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
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
