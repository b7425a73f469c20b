"""Token-id <-> text for the Whisper vocabulary, from LOCAL files only.

Upstream bundles its BPE vocabulary inside the openai-whisper package (``assets/multilingual.tiktoken``); that package
is not available here, so:
  * with a vocabulary file (tiktoken format: ``base64(token) rank`` per line, or HF ``vocab.json`` byte-level BPE) the
    tokenizer decodes ids to real text and can encode prompts (byte-pair merges by rank);
  * without one, ids are rendered as ``<|id|>`` placeholders (SURVEY.md §8c "known weaknesses (ii)").
``LANGUAGES`` is what the reference returns from ``get_supported_languages()``
(ref:app/transcription/speech_recognition.py:82-89 -> ``whisper.tokenizer.LANGUAGES``).
"""
from __future__ import annotations

import base64
import json
import os
from typing import Dict, Iterable, List, Optional

from .checkpoint import SpecialTokens, special_tokens_for

# code -> name, upstream order (index = language token offset from <|startoftranscript|> + 1)
LANGUAGES: Dict[str, str] = {
    "en": "english", "zh": "chinese", "de": "german", "es": "spanish", "ru": "russian", "ko": "korean",
    "fr": "french", "ja": "japanese", "pt": "portuguese", "tr": "turkish", "pl": "polish", "ca": "catalan",
    "nl": "dutch", "ar": "arabic", "sv": "swedish", "it": "italian", "id": "indonesian", "hi": "hindi",
    "fi": "finnish", "vi": "vietnamese", "he": "hebrew", "uk": "ukrainian", "el": "greek", "ms": "malay",
    "cs": "czech", "ro": "romanian", "da": "danish", "hu": "hungarian", "ta": "tamil", "no": "norwegian",
    "th": "thai", "ur": "urdu", "hr": "croatian", "bg": "bulgarian", "lt": "lithuanian", "la": "latin",
    "mi": "maori", "ml": "malayalam", "cy": "welsh", "sk": "slovak", "te": "telugu", "fa": "persian",
    "lv": "latvian", "bn": "bengali", "sr": "serbian", "az": "azerbaijani", "sl": "slovenian", "kn": "kannada",
    "et": "estonian", "mk": "macedonian", "br": "breton", "eu": "basque", "is": "icelandic", "hy": "armenian",
    "ne": "nepali", "mn": "mongolian", "bs": "bosnian", "kk": "kazakh", "sq": "albanian", "sw": "swahili",
    "gl": "galician", "mr": "marathi", "pa": "punjabi", "si": "sinhala", "km": "khmer", "sn": "shona",
    "yo": "yoruba", "so": "somali", "af": "afrikaans", "oc": "occitan", "ka": "georgian", "be": "belarusian",
    "tg": "tajik", "sd": "sindhi", "gu": "gujarati", "am": "amharic", "yi": "yiddish", "lo": "lao",
    "uz": "uzbek", "fo": "faroese", "ht": "haitian creole", "ps": "pashto", "tk": "turkmen", "nn": "nynorsk",
    "mt": "maltese", "sa": "sanskrit", "lb": "luxembourgish", "my": "myanmar", "bo": "tibetan", "tl": "tagalog",
    "mg": "malagasy", "as": "assamese", "tt": "tatar", "haw": "hawaiian", "ln": "lingala", "ha": "hausa",
    "ba": "bashkir", "jw": "javanese", "su": "sundanese", "yue": "cantonese",
}
LANGUAGE_CODES: List[str] = list(LANGUAGES.keys())


def supported_languages(n_vocab: int) -> Dict[str, str]:
    """99 languages for the 51865-vocab models, 100 (adds Cantonese) for the large-v3 family."""
    n = special_tokens_for(n_vocab).n_langs
    return {c: LANGUAGES[c] for c in LANGUAGE_CODES[:n]}


def language_index(code: str, n_vocab: int) -> int:
    code = code.lower()
    names = {v: k for k, v in LANGUAGES.items()}
    code = names.get(code, code)
    langs = LANGUAGE_CODES[:special_tokens_for(n_vocab).n_langs]
    if code not in langs:
        raise ValueError(f"unsupported language {code!r}")
    return langs.index(code)


def _gpt2_byte_decoder() -> Dict[str, int]:
    """Inverse of GPT-2's bytes_to_unicode table (HF vocab.json stores tokens in that alphabet)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return {chr(c): b for b, c in zip(bs, cs)}


class Tokenizer:
    def __init__(self, n_vocab: int, vocab_path: Optional[str] = None):
        self.n_vocab = n_vocab
        self.special: SpecialTokens = special_tokens_for(n_vocab)
        self.id_to_bytes: Optional[List[bytes]] = None
        self.ranks: Optional[Dict[bytes, int]] = None
        if vocab_path:
            self._load(vocab_path)

    @property
    def has_vocab(self) -> bool:
        return self.id_to_bytes is not None

    def _load(self, path: str) -> None:
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        table: Dict[int, bytes] = {}
        if path.endswith(".json"):
            dec = _gpt2_byte_decoder()
            with open(path, "r", encoding="utf-8") as f:
                vocab = json.load(f)
            for tok, idx in vocab.items():
                try:
                    table[int(idx)] = bytes(dec[ch] for ch in tok)
                except KeyError:
                    table[int(idx)] = tok.encode("utf-8")
        else:
            with open(path, "rb") as f:
                for line in f:
                    if not line.strip():
                        continue
                    tok, rank = line.split()
                    table[int(rank)] = base64.b64decode(tok)
        size = max(table) + 1
        self.id_to_bytes = [table.get(i, b"") for i in range(size)]
        self.ranks = {b: i for i, b in table.items() if b}

    # -- decode ---------------------------------------------------------------------------------------------
    def decode(self, ids: Iterable[int]) -> str:
        """ids -> text, skipping every id >= eot (special and timestamp tokens), as upstream ``decode`` does."""
        ids = [int(i) for i in ids if int(i) < self.special.eot]
        if self.id_to_bytes is None:
            return "".join(f"<|{i}|>" for i in ids)
        return b"".join(self.id_to_bytes[i] if i < len(self.id_to_bytes) else b"" for i in ids).decode("utf-8", errors="replace")

    # -- encode (prompts) -------------------------------------------------------------------------------------
    def encode(self, text: str) -> List[int]:
        if self.ranks is None:
            raise RuntimeError("no vocabulary loaded: pass prompts as token ids")
        import regex
        pat = regex.compile(r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+""")
        out: List[int] = []
        for piece in pat.findall(text):
            parts = [bytes([b]) for b in piece.encode("utf-8")]
            while len(parts) > 1:
                best, best_rank = None, None
                for i in range(len(parts) - 1):
                    rk = self.ranks.get(parts[i] + parts[i + 1])
                    if rk is not None and (best_rank is None or rk < best_rank):
                        best, best_rank = i, rk
                if best is None:
                    break
                parts[best:best + 2] = [parts[best] + parts[best + 1]]
            out.extend(self.ranks[p] for p in parts)
        return out
