"""Model dimensions, special-token layout, synthetic checkpoints and local checkpoint loaders.

The reference resolves a model *name* to a download (ref:app/transcription/speech_recognition.py:22,
``whisper.load_model(WHISPER_MODEL, device=device)``).  There is no network here, so this module
resolves a name to dimensions only, and weights come either from a **local** file (openai ``.pt``
layout or HF safetensors layout, SURVEY.md §8c "Weight-name mapping") or from a seeded synthetic
generator (SURVEY.md §8d "Synthetic inputs").

Canonical tensor names follow the openai-whisper state dict (``encoder.blocks.N.attn.query.weight``,
``decoder.token_embedding.weight`` ...).  Everything in this file is host-side numpy/torch; nothing here
touches the GPU.
"""
from __future__ import annotations

import dataclasses
import math
import os
import re
from typing import Dict, Optional

import numpy as np
import torch


@dataclasses.dataclass(frozen=True)
class ModelDims:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int

    @property
    def head_dim(self) -> int:
        return self.n_audio_state // self.n_audio_head


# SURVEY.md §8 model table. n_audio_ctx = 1500, n_text_ctx = 448, FFN = 4 d, head_dim = 64.
_NAMED = {
    "tiny": dict(d=384, h=6, le=4, ld=4, mels=80, vocab=51865),
    "base": dict(d=512, h=8, le=6, ld=6, mels=80, vocab=51865),
    "small": dict(d=768, h=12, le=12, ld=12, mels=80, vocab=51865),
    "medium": dict(d=1024, h=16, le=24, ld=24, mels=80, vocab=51865),
    # English-only checkpoints (GPT-2 vocabulary, 51 864 ids: no language tokens; start sequence [sot] alone)
    "tiny.en": dict(d=384, h=6, le=4, ld=4, mels=80, vocab=51864),
    "base.en": dict(d=512, h=8, le=6, ld=6, mels=80, vocab=51864),
    "small.en": dict(d=768, h=12, le=12, ld=12, mels=80, vocab=51864),
    "medium.en": dict(d=1024, h=16, le=24, ld=24, mels=80, vocab=51864),
    "large-v3": dict(d=1280, h=20, le=32, ld=32, mels=128, vocab=51866),
    "large-v3-turbo": dict(d=1280, h=20, le=32, ld=4, mels=128, vocab=51866),
    # not an upstream model: a 2-layer, 2-head model with the real context sizes, for fast tests
    "micro": dict(d=128, h=2, le=2, ld=2, mels=80, vocab=51865),
    "micro128": dict(d=128, h=2, le=1, ld=1, mels=128, vocab=51866),
    "micro.en": dict(d=128, h=2, le=2, ld=2, mels=80, vocab=51864),
    # not an upstream model: widths that are multiples of 256, so a 3-window batch reaches the 256x256 encoder GEMM tiles
    "micro256": dict(d=256, h=4, le=2, ld=2, mels=80, vocab=51865),
}
_ALIASES = {"large": "large-v3", "turbo": "large-v3-turbo"}


def dims_for(name: str) -> ModelDims:
    name = _ALIASES.get(name, name)
    if name not in _NAMED:
        raise KeyError(f"unknown Whisper model name {name!r}; known: {sorted(_NAMED)}")
    c = _NAMED[name]
    return ModelDims(
        n_mels=c["mels"], n_audio_ctx=1500, n_audio_state=c["d"], n_audio_head=c["h"],
        n_audio_layer=c["le"], n_vocab=c["vocab"], n_text_ctx=448, n_text_state=c["d"],
        n_text_head=c["h"], n_text_layer=c["ld"],
    )


@dataclasses.dataclass(frozen=True)
class SpecialTokens:
    """Multilingual special-token ids (SURVEY.md §8 'Special tokens').  Not taken from HF defaults."""
    eot: int
    sot: int
    n_langs: int
    translate: int
    transcribe: int
    sot_lm: int
    sot_prev: int
    no_speech: int
    no_timestamps: int
    timestamp_begin: int
    blank: int = 220  # the " " token: begin-suppress pair is (220, eot)

    def lang(self, index: int) -> int:
        if not 0 <= index < self.n_langs:
            raise ValueError(f"language index {index} out of range 0..{self.n_langs - 1}")
        return self.sot + 1 + index


def special_tokens_for(n_vocab: int) -> SpecialTokens:
    """Special-token ids of the three upstream vocabulary layouts [upstream-recall: ``whisper.tokenizer.get_encoding``
    appends ``<|endoftext|>, <|startoftranscript|>, <|lang|> x num_languages, translate, transcribe, startoflm,
    startofprev, nospeech, notimestamps, <|0.00|> ...`` to the BPE ranks]:

    * 51865 -- multilingual (tiny ... large-v2): 50257 BPE ranks, 99 languages;
    * 51866 -- large-v3 family: the same ranks, 100 languages (adds Cantonese), so every id from translate on is +1;
    * 51864 -- English-only (``*.en``): the GPT-2 ranks (50256 of them), the same 99 language tokens in the table but
      never used: the start sequence of these models is ``[sot]`` alone (``is_multilingual`` False)."""
    if n_vocab not in (51864, 51865, 51866):
        raise ValueError(f"unsupported Whisper vocabulary size {n_vocab} (known: 51864 English-only, 51865, 51866)")
    n_langs = 100 if n_vocab == 51866 else 99
    eot = 50256 if n_vocab == 51864 else 50257
    sot = eot + 1
    translate = sot + 1 + n_langs
    return SpecialTokens(
        eot=eot, sot=sot, n_langs=n_langs, translate=translate, transcribe=translate + 1,
        sot_lm=translate + 2, sot_prev=translate + 3, no_speech=translate + 4,
        no_timestamps=translate + 5, timestamp_begin=translate + 6,
    )


def is_multilingual(n_vocab: int) -> bool:
    """Upstream ``ModelDimensions``-level rule: ``n_vocab >= 51865`` [upstream-recall]."""
    return n_vocab >= 51865


# Upstream ``Tokenizer.non_speech_tokens`` is derived from the BPE vocabulary (see ``non_speech_tokens_from_vocab``); the
# vocabulary files ship inside the openai-whisper package and are not available offline, so the two resulting id sets are
# tabulated here.  They are the ids below <|endoftext|> of transformers' generation defaults for the original checkpoints
# (hf:models/whisper/configuration_whisper.py:34-44 ``NON_SPEECH_TOKENS_MULTI`` for ``multilingual.tiktoken``, :23-33
# ``NON_SPEECH_TOKENS`` for ``gpt2.tiktoken``; the entries >= eot in those lists are the control tokens that
# ``default_suppress_ids`` adds itself).  tests/test_host_logic.py checks both tables against transformers.
_NON_SPEECH_MULTILINGUAL = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503, 522, 542,
    873, 893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536, 3846, 3961, 4183, 4667,
    6585, 6647, 7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562, 13793, 14157, 14635, 15265, 15618,
    16553, 16604, 18362, 18956, 20075, 21675, 22520, 26130, 26161, 26435, 28279, 29464, 31650, 32302, 32470,
    36865, 42863, 47425, 49870, 50254)
_NON_SPEECH_ENGLISH = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93,
    357, 366, 438, 532, 685, 705, 796, 930, 1058, 1220, 1267, 1279, 1303, 1343, 1377, 1391, 1635,
    1782, 1875, 2162, 2361, 2488, 3467, 4008, 4211, 4600, 4808, 5299, 5855, 6329, 7203, 9609, 9959,
    10563, 10786, 11420, 11709, 11907, 13163, 13697, 13700, 14808, 15306, 16410, 16791, 17992, 19203,
    19510, 20724, 22305, 22935, 27007, 30109, 30420, 33409, 34949, 40283, 40493, 40549, 47282, 49146)


def non_speech_suppress_ids(n_vocab: int = 51865) -> list:
    """Upstream's ``non_speech_tokens`` for the vocabulary the model was trained with: the multilingual table for the
    51865 / 51866 layouts (both use ``multilingual.tiktoken``), the GPT-2 table for English-only checkpoints.  When a
    vocabulary file is configured, ``transcribe.configure`` derives the set from it instead
    (``non_speech_tokens_from_vocab``) and passes it to the engine."""
    return list(_NON_SPEECH_MULTILINGUAL if is_multilingual(n_vocab) else _NON_SPEECH_ENGLISH)


def non_speech_tokens_from_vocab(encode) -> list:
    """Upstream ``Tokenizer.non_speech_tokens`` [upstream-recall] on a loaded vocabulary: ``encode(str) -> ids``.

    The ids of the annotation symbols that must not be sampled ("♪♪♪", "( SPEAKING FOREIGN LANGUAGE )", "[DAVID]" ...):
    every symbol of the list below, alone and with a leading space, is encoded; a one-token encoding is suppressed, and
    for the musical-note characters (multi-byte, possibly split over tokens) the first token is; " -" and " '" are
    always included, while plain hyphens and apostrophes inside words stay allowed."""
    symbols = list("\"#()*+/:;<=>@[\\]^_`{|}~「」『』")
    symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
    miscellaneous = set("♩♪♫♬♭♮♯")
    result = {encode(" -")[0], encode(" '")[0]}
    for symbol in symbols + sorted(miscellaneous):
        for tokens in (encode(symbol), encode(" " + symbol)):
            if len(tokens) == 1 or symbol in miscellaneous:
                result.add(tokens[0])
    return sorted(int(i) for i in result)


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Encoder positional table (hf:models/whisper/modeling_whisper.py:55-64)."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def tensor_shapes(dims: ModelDims) -> Dict[str, tuple]:
    """Every tensor of the canonical (openai) state dict with its shape."""
    d, dt = dims.n_audio_state, dims.n_text_state
    s: Dict[str, tuple] = {
        "encoder.conv1.weight": (d, dims.n_mels, 3), "encoder.conv1.bias": (d,),
        "encoder.conv2.weight": (d, d, 3), "encoder.conv2.bias": (d,),
        "encoder.positional_embedding": (dims.n_audio_ctx, d),
        "encoder.ln_post.weight": (d,), "encoder.ln_post.bias": (d,),
        "decoder.token_embedding.weight": (dims.n_vocab, dt),
        "decoder.positional_embedding": (dims.n_text_ctx, dt),
        "decoder.ln.weight": (dt,), "decoder.ln.bias": (dt,),
    }

    def attn(prefix: str, w: int):
        s[f"{prefix}.query.weight"] = (w, w); s[f"{prefix}.query.bias"] = (w,)
        s[f"{prefix}.key.weight"] = (w, w)  # K projection has no bias
        s[f"{prefix}.value.weight"] = (w, w); s[f"{prefix}.value.bias"] = (w,)
        s[f"{prefix}.out.weight"] = (w, w); s[f"{prefix}.out.bias"] = (w,)

    def block(prefix: str, w: int, cross: bool):
        attn(f"{prefix}.attn", w)
        s[f"{prefix}.attn_ln.weight"] = (w,); s[f"{prefix}.attn_ln.bias"] = (w,)
        if cross:
            attn(f"{prefix}.cross_attn", w)
            s[f"{prefix}.cross_attn_ln.weight"] = (w,); s[f"{prefix}.cross_attn_ln.bias"] = (w,)
        s[f"{prefix}.mlp.0.weight"] = (4 * w, w); s[f"{prefix}.mlp.0.bias"] = (4 * w,)
        s[f"{prefix}.mlp.2.weight"] = (w, 4 * w); s[f"{prefix}.mlp.2.bias"] = (w,)
        s[f"{prefix}.mlp_ln.weight"] = (w,); s[f"{prefix}.mlp_ln.bias"] = (w,)

    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", d, cross=False)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", dt, cross=True)
    return s


def synthetic_state_dict(dims: ModelDims, seed: int = 0, eot_boost: float = 1.0,
                         emb_std: float = 0.05, attn_gain: float = 3.0) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights of the named architecture.

    Linear/conv weights ~ N(0, 1/fan_in) (so activations keep unit scale through the pre-LN stack), biases
    ~ N(0, 0.02²), LayerNorm gains 1 + N(0, 0.05²).  ``emb_std`` sets the tied embedding scale and therefore the
    logit spread (std ≈ emb_std·sqrt(d)), which is what makes top-1/top-2 margins large against fp32 noise.
    ``attn_gain`` multiplies every query and key projection: plain 1/fan_in weights give near-uniform softmaxes whose
    output ignores both the audio and the token history (greedy decoding then repeats one token); a gain of 3 makes
    attention content dependent, so token sequences differ per clip and per step.
    ``eot_boost`` scales the EOT row of the embedding so that greedy decoding terminates at sequence-dependent
    lengths (used by the ragged-termination tests; 1.0 = plain)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in tensor_shapes(dims).items():
        if name == "encoder.positional_embedding":
            out[name] = sinusoids(shape[0], shape[1])
        elif name == "decoder.positional_embedding":
            out[name] = torch.randn(shape, generator=g) * 0.1
        elif name == "decoder.token_embedding.weight":
            out[name] = torch.randn(shape, generator=g) * emb_std
        elif name.endswith("_ln.weight") or name.endswith("ln_post.weight") or name.endswith("ln.weight"):
            out[name] = 1.0 + torch.randn(shape, generator=g) * 0.05
        elif name.endswith(".bias"):
            out[name] = torch.randn(shape, generator=g) * 0.02
        else:
            fan_in = int(np.prod(shape[1:]))
            out[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    if attn_gain != 1.0:
        for name in out:
            if name.endswith("query.weight") or name.endswith("key.weight"):
                out[name] = out[name] * attn_gain
    if eot_boost != 1.0:
        eot = special_tokens_for(dims.n_vocab).eot
        out["decoder.token_embedding.weight"][eot] *= eot_boost
    return out


# ---------------------------------------------------------------------------------------------
# Local checkpoint loading (never downloads)
# ---------------------------------------------------------------------------------------------

_HF_LAYER = [
    (r"self_attn\.q_proj", "attn.query"), (r"self_attn\.k_proj", "attn.key"),
    (r"self_attn\.v_proj", "attn.value"), (r"self_attn\.out_proj", "attn.out"),
    (r"self_attn_layer_norm", "attn_ln"),
    (r"encoder_attn\.q_proj", "cross_attn.query"), (r"encoder_attn\.k_proj", "cross_attn.key"),
    (r"encoder_attn\.v_proj", "cross_attn.value"), (r"encoder_attn\.out_proj", "cross_attn.out"),
    (r"encoder_attn_layer_norm", "cross_attn_ln"),
    (r"fc1", "mlp.0"), (r"fc2", "mlp.2"), (r"final_layer_norm", "mlp_ln"),
]


def hf_to_canonical_name(name: str) -> Optional[str]:
    """Map an HF ``WhisperForConditionalGeneration`` tensor name to the canonical name
    (hf:models/whisper/modeling_whisper.py:279-282,566-573,675-682,965).  Returns None for tensors that have no
    canonical counterpart (the tied ``proj_out.weight``)."""
    if name.startswith("model."):
        name = name[len("model."):]
    if name == "proj_out.weight":
        return None
    fixed = {
        "encoder.conv1.weight": "encoder.conv1.weight", "encoder.conv1.bias": "encoder.conv1.bias",
        "encoder.conv2.weight": "encoder.conv2.weight", "encoder.conv2.bias": "encoder.conv2.bias",
        "encoder.embed_positions.weight": "encoder.positional_embedding",
        "encoder.layer_norm.weight": "encoder.ln_post.weight", "encoder.layer_norm.bias": "encoder.ln_post.bias",
        "decoder.embed_tokens.weight": "decoder.token_embedding.weight",
        "decoder.embed_positions.weight": "decoder.positional_embedding",
        "decoder.layer_norm.weight": "decoder.ln.weight", "decoder.layer_norm.bias": "decoder.ln.bias",
    }
    if name in fixed:
        return fixed[name]
    m = re.match(r"(encoder|decoder)\.layers\.(\d+)\.(.+)\.(weight|bias)$", name)
    if not m:
        raise KeyError(f"unrecognised HF Whisper tensor name {name!r}")
    side, idx, mid, leaf = m.groups()
    for pat, repl in _HF_LAYER:
        if re.fullmatch(pat, mid):
            return f"{side}.blocks.{idx}.{repl}.{leaf}"
    raise KeyError(f"unrecognised HF Whisper tensor name {name!r}")


def canonical_to_hf_name(name: str) -> str:
    """Inverse of :func:`hf_to_canonical_name` (used to build the HF stand-in model in tests/oracle tooling)."""
    fixed = {
        "encoder.positional_embedding": "model.encoder.embed_positions.weight",
        "encoder.ln_post.weight": "model.encoder.layer_norm.weight",
        "encoder.ln_post.bias": "model.encoder.layer_norm.bias",
        "decoder.token_embedding.weight": "model.decoder.embed_tokens.weight",
        "decoder.positional_embedding": "model.decoder.embed_positions.weight",
        "decoder.ln.weight": "model.decoder.layer_norm.weight", "decoder.ln.bias": "model.decoder.layer_norm.bias",
    }
    if name in fixed:
        return fixed[name]
    if name.startswith("encoder.conv"):
        return "model." + name
    m = re.match(r"(encoder|decoder)\.blocks\.(\d+)\.(.+)\.(weight|bias)$", name)
    if not m:
        raise KeyError(name)
    side, idx, mid, leaf = m.groups()
    inv = {
        "attn.query": "self_attn.q_proj", "attn.key": "self_attn.k_proj", "attn.value": "self_attn.v_proj",
        "attn.out": "self_attn.out_proj", "attn_ln": "self_attn_layer_norm",
        "cross_attn.query": "encoder_attn.q_proj", "cross_attn.key": "encoder_attn.k_proj",
        "cross_attn.value": "encoder_attn.v_proj", "cross_attn.out": "encoder_attn.out_proj",
        "cross_attn_ln": "encoder_attn_layer_norm", "mlp.0": "fc1", "mlp.2": "fc2", "mlp_ln": "final_layer_norm",
    }
    return f"model.{side}.layers.{idx}.{inv[mid]}.{leaf}"


def dims_from_state_dict(sd: Dict[str, torch.Tensor]) -> ModelDims:
    d = sd["encoder.conv1.weight"].shape[0]
    dt = sd["decoder.token_embedding.weight"].shape[1]
    n_enc = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.blocks."))
    n_dec = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.blocks."))
    return ModelDims(
        n_mels=sd["encoder.conv1.weight"].shape[1], n_audio_ctx=sd["encoder.positional_embedding"].shape[0],
        n_audio_state=d, n_audio_head=d // 64, n_audio_layer=n_enc,
        n_vocab=sd["decoder.token_embedding.weight"].shape[0],
        n_text_ctx=sd["decoder.positional_embedding"].shape[0], n_text_state=dt, n_text_head=dt // 64,
        n_text_layer=n_dec,
    )


def load_checkpoint(path: str):
    """Load weights from a LOCAL file.  Returns ``(dims, state_dict)`` with canonical names, fp32.

    * ``*.safetensors`` – HF layout, read with safetensors (no code execution).
    * ``*.pt`` / ``*.bin`` – openai layout ``{"dims": {...}, "model_state_dict": {...}}`` or a bare state dict,
      read with ``torch.load(weights_only=True)``.
    """
    if not os.path.isfile(path):
        raise FileNotFoundError(f"checkpoint {path!r} not found (this build never downloads weights)")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        raw = load_file(path)
        sd = {}
        for k, v in raw.items():
            ck = hf_to_canonical_name(k)
            if ck is not None:
                sd[ck] = v.float()
        return dims_from_state_dict(sd), sd
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "model_state_dict" in obj:
        raw = obj["model_state_dict"]
        dims = ModelDims(**{k: int(v) for k, v in obj["dims"].items()}) if "dims" in obj else None
    else:
        raw, dims = obj, None
    if any(k.startswith("model.") for k in raw):
        raw = {hf_to_canonical_name(k): v for k, v in raw.items() if hf_to_canonical_name(k) is not None}
    sd = {k: v.float() for k, v in raw.items()}
    return dims or dims_from_state_dict(sd), sd


def save_checkpoint(path: str, dims: ModelDims, sd: Dict[str, torch.Tensor]) -> None:
    """Write the openai ``.pt`` layout (used by tests to exercise :func:`load_checkpoint`)."""
    torch.save({"dims": dataclasses.asdict(dims), "model_state_dict": sd}, path)
