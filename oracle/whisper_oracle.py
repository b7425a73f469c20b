"""CPU oracle for the local-Whisper hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch fp32 restatement of the algorithm behind the reference's
``whisper_model.transcribe(...)`` call (ref:app/transcription/speech_recognition.py:48-55).  That algorithm
lives in the third-party package ``openai-whisper`` (declared ``>=20231117``, ref:requirements.txt:26, no lock
file), which is NOT present in /root/reference, in this container or on the GPU box.  Each function below
restates the published algorithm and cites the independent port that *is* readable here (``hf:`` =
transformers 5.15.0) line by line; where the two differ the upstream behaviour is followed and the
difference is noted.

PARITY PINNING: the reference holds no golden vector, known-answer test or fixture for this path (every
reference test mocks it) and the dependency itself cannot be run: against openai-whisper
itself **parity is unpinned**.  What pins this oracle instead: tests/golden/*.npz, produced by
oracle/make_golden.py from the transformers 5.15.0 Whisper implementation (feature extractor, encoder,
decoder logits) on seeded weights and seeded audio, and re-checked by tests/test_oracle_vs_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(whisper_amd/) never does.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE      # 480000 samples per 30 s window
N_FRAMES = N_SAMPLES // HOP_LENGTH          # 3000 mel frames per window


# -------------------------------------------------------------------------------------------------
# audio ingest: the reference's ffmpeg normalisation step (any PCM16 -> 16 kHz mono s16le)
# -------------------------------------------------------------------------------------------------

def resample_taps(up: int, down: int):
    """Default low-pass of scipy.signal.resample_poly (scipy 1.15.3 signal/_signaltools.py, ``resample_poly``:
    ``firwin(2*half_len + 1, 1/max_rate, window=('kaiser', 5.0))`` with ``half_len = 10*max_rate``, times ``up``),
    rounded to float32 as the device table is."""
    q = max(up, down)
    half_len = 10 * q
    n = np.arange(2 * half_len + 1, dtype=np.float64) - half_len
    cutoff = 1.0 / q
    win = np.i0(5.0 * np.sqrt(np.clip(1.0 - (n / half_len) ** 2, 0.0, 1.0))) / np.i0(5.0)      # Kaiser window, beta 5
    h = cutoff * np.sinc(cutoff * n) * win
    h = h / h.sum()                                                                               # firwin(scale=True): unit DC gain
    return (h * up).astype(np.float32), half_len


def resample_pcm16(pcm: np.ndarray, src_rate: int, quantize: bool = True) -> np.ndarray:
    """int16 ``[n_frames]`` / ``[n_frames, n_channels]`` at ``src_rate`` -> mono float32 at 16 kHz.

    What ``ffmpeg -acodec pcm_s16le -ar 16000 -ac 1`` does in the reference (ref:app/audio/processor.py:136-147) followed
    by upstream load_audio's ``/ 32768``: mix the channels down by averaging, resample, round to the int16 grid.  ffmpeg is
    absent here and its resampling filter is its own, so parity with ffmpeg is UNPINNED; the arithmetic restated here is
    scipy.signal.resample_poly's (zero-stuffed up-sampling by ``up``, the centred FIR above, decimation by ``down``,
    zeros beyond both ends, ``ceil(n*up/down)`` outputs), which tests/test_oracle_vs_golden.py pins against scipy itself."""
    x = np.asarray(pcm)
    if x.ndim == 1:
        x = x[:, None]
    g = math.gcd(SAMPLE_RATE, int(src_rate))
    up, down = SAMPLE_RATE // g, int(src_rate) // g
    mono = x.astype(np.float64).sum(axis=1) / (32768.0 * x.shape[1])
    y = resample_poly_f64(mono, up, down)
    if quantize:
        y = np.clip(np.rint(y * 32768.0), -32768, 32767) / 32768.0
    return y.astype(np.float32)


def resample_poly_f64(x: np.ndarray, up: int, down: int) -> np.ndarray:
    """y[n] = sum_m x[m] * h[half_len + n*down - m*up] over the m that fall under the filter, float64 accumulation."""
    h32, half_len = resample_taps(up, down)
    h = h32.astype(np.float64)
    n_in = x.shape[0]
    n_out = -(-n_in * up // down)
    n = np.arange(n_out, dtype=np.int64)
    first = -((half_len - n * down) // up)              # ceil((n*down - half_len) / up)
    y = np.zeros(n_out, np.float64)
    for j in range(2 * half_len // up + 1):
        m = first + j
        idx = half_len + n * down - m * up
        ok = (m >= 0) & (m < n_in) & (idx >= 0)
        y[ok] += x[m[ok]] * h[idx[ok]]
    return y


# -------------------------------------------------------------------------------------------------
# log-mel front end
# -------------------------------------------------------------------------------------------------

def _hz_to_mel_slaney(f):
    """hf:audio_utils.py:470-481 (slaney branch)."""
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * logstep, mels)


def _mel_to_hz_slaney(m):
    """hf:audio_utils.py:506-517 (slaney branch)."""
    m = np.asarray(m, dtype=np.float64)
    freq = 200.0 * m / 3.0
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), freq)


def mel_filters(n_mels: int) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular filterbank, ``[n_mels, 201]`` float32.

    Upstream ships this matrix as ``assets/mel_filters.npz`` (librosa.filters.mel(sr=16000, n_fft=400,
    n_mels=...)); it is regenerated here, following hf:audio_utils.py:638-729 (centres linear in mel space,
    triangles in Hz, ``enorm = 2 / (f[i+2] - f[i])``)."""
    n_freq = 1 + N_FFT // 2
    fft_freqs = np.linspace(0.0, SAMPLE_RATE // 2, n_freq)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(8000.0), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    slopes = hz_pts[None, :] - fft_freqs[:, None]                  # [n_freq, n_mels + 2]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))                     # [n_freq, n_mels]
    fb *= (2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels]))[None, :]
    return np.ascontiguousarray(fb.T).astype(np.float32)


def log_mel_spectrogram(audio, n_mels: int, padding: int = 0) -> torch.Tensor:
    """``[n_mels, n_frames]`` float32, n_frames = (len(audio) + padding) // 160.

    Same steps as hf:models/whisper/feature_extraction_whisper.py:140-165: periodic Hann(400),
    ``torch.stft(n_fft=400, hop=160, center=True)`` (reflect padding), drop the last frame, |.|², filterbank,
    ``log10(clamp(1e-10))``, ``max(x, x.max() - 8)``, ``(x + 4) / 4``.  As upstream does (and unlike HF's
    batched path, :160-162) the max is taken over the WHOLE input, not per 30 s window."""
    audio = torch.as_tensor(np.asarray(audio), dtype=torch.float32)
    if padding > 0:
        audio = F.pad(audio, (0, padding))
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    filters = torch.from_numpy(mel_filters(n_mels))
    mel_spec = filters @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def log_mel_unnormalised(audio, n_mels: int) -> torch.Tensor:
    """The stage-1 quantity ``log10(clamp(mel, 1e-10))`` before the global clamp (used to test the two-stage
    sharded mel path, SURVEY.md §8e caveat 1)."""
    audio = torch.as_tensor(np.asarray(audio), dtype=torch.float32)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=torch.hann_window(N_FFT), return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    return torch.clamp(torch.from_numpy(mel_filters(n_mels)) @ magnitudes, min=1e-10).log10()


def pad_or_trim_frames(mel: torch.Tensor, length: int = N_FRAMES) -> torch.Tensor:
    if mel.shape[-1] > length:
        return mel[..., :length]
    if mel.shape[-1] < length:
        return F.pad(mel, (0, length - mel.shape[-1]))
    return mel


def synthetic_audio(seed: int, n_samples: int = N_SAMPLES) -> np.ndarray:
    """SURVEY.md §8d generator: two sines plus Gaussian noise, float32, ``default_rng(seed)``."""
    n = np.arange(n_samples, dtype=np.float64)
    rng = np.random.default_rng(seed)
    x = 0.3 * np.sin(2 * np.pi * 220.0 * n / SAMPLE_RATE) + 0.2 * np.sin(2 * np.pi * 1330.0 * n / SAMPLE_RATE)
    x = x + 0.05 * rng.standard_normal(n_samples)
    return x.astype(np.float32)


def varied_audio(seed: int, n_samples: int = N_SAMPLES) -> np.ndarray:
    """Test signal whose content depends strongly on the seed: three gliding tones with per-second random
    frequency (80-3500 Hz) and amplitude, plus light noise.  Used by the parity tests so that encoder output and
    token sequences differ from clip to clip (the §8d benchmark signal differs only in its noise)."""
    rng = np.random.default_rng(1000 + seed)
    t = np.arange(n_samples) / float(SAMPLE_RATE)
    nseg = int(np.ceil(n_samples / SAMPLE_RATE))
    x = np.zeros(n_samples)
    for _ in range(3):
        f = rng.uniform(80.0, 3500.0, nseg + 1)
        a = rng.uniform(0.0, 0.4, nseg + 1)
        fi = np.interp(t, np.arange(nseg + 1), f)
        ai = np.interp(t, np.arange(nseg + 1), a)
        x += ai * np.sin(2 * np.pi * np.cumsum(fi) / SAMPLE_RATE)
    x += 0.02 * rng.standard_normal(n_samples)
    return x.astype(np.float32)


# -------------------------------------------------------------------------------------------------
# model
# -------------------------------------------------------------------------------------------------

def _layer_norm(x, w, b):
    return F.layer_norm(x.to(w.dtype), (x.shape[-1],), w, b, 1e-5)


class OracleWhisper:
    """fp32 encoder/decoder forward on a canonical (openai-named) state dict.

    Encoder: hf:models/whisper/modeling_whisper.py:612-646 (stem :566-567,618-619; layer :391-407).
    Decoder: hf:...modeling_whisper.py:733-795 (layer :469-503); logits via the tied embedding (:965,1080).
    Attention: q and k are each scaled by head_dim**-0.25 before the product, the softmax runs in fp32 (upstream
    ``MultiHeadAttention.qkv_attention`` non-SDPA path; HF scales q only, :304-309 — same value in exact
    arithmetic).  The key projection has no bias (hf:...:279).

    ``emulate_bf16=True`` restates the SAME forward with the roundings of the engine's bf16 mode: matrix
    weights, the token / decoder-position tables, LayerNorm outputs, scaled q / k, v, attention outputs, GELU outputs, the
    conv-stem activations, the encoder output and the cross K/V cache are rounded to bf16 (round-to-nearest-even, what
    ``(bf16)x`` does on the device); every product accumulates in fp32 and the residual stream stays fp32.  The encoder's
    flash kernel also rounds the un-normalised softmax weights to bf16 for the P.V product (the row sum uses the
    unrounded weights); the decoder's self-attention kernel keeps them in fp32.  The engine's bf16 decoder CROSS-attention
    works on the encoder output E itself (whisper_amd/csrc/xattn.hip: q~_h = s Wk_h^T bf16(q_h s), scores q~_h . e_j,
    context c_h = softmax . E, output Wv_h bf16(c_h) + bv -- algebraically the same attention with the key / value
    projections moved to the query side), so its rounding points are q (bf16), q~ (bf16), the softmax weights (bf16 for
    the P.E product) and the context (bf16); ``_cross_attention_absorbed`` restates exactly that.  It is the yardstick for the bf16 tolerance:
    what is left between it and the device is summation order -- which bf16 rounding amplifies: two evaluations that
    differ by 1e-6 before a rounding point differ by a whole bf16 ulp after it wherever the rounding flips, and after a few
    layers their distance saturates at the bf16 noise floor of the model.  ``acc_dtype=torch.float64`` gives a second,
    independent evaluation of the same rounded model (same rounding points, float64 accumulation): the distance between
    the two emulations MEASURES that noise floor, and the GPU tests bound the device's distance by a small multiple of it."""

    def __init__(self, dims, sd: Dict[str, torch.Tensor], emulate_bf16: bool = False, acc_dtype: torch.dtype = torch.float32,
                 cross_form: str = "encoder_output", cross_key_splits: Optional[int] = None, flash_tiles: bool = False):
        """``cross_form`` (bf16 emulation only): "encoder_output" = the engine's cross-attention on E itself (bf16 engines with
        max_batch >= 32), "kv" = on cached K/V (smaller engines, beam search); WhisperEngine.cross_form says which one runs.
        ``cross_key_splits`` (encoder_output form): also restate the stream kernel's ONLINE softmax -- tiles of 32 keys, the
        softmax weights rounded to bf16 against the RUNNING maximum of their key split, each split's context normalised by its
        own sum and rounded to bf16, the splits combined in fp32 and rounded again (WhisperEngine.cross_key_splits says how
        many) -- instead of rounding the weights once against the final maximum.  ``flash_tiles``: the same for the ENCODER's
        flash-attention kernel (whisper_amd/csrc/flash.hip: tiles of 64 keys, weights rounded to bf16 against the running
        maximum, row sum of the unrounded weights)."""
        assert cross_form in ("encoder_output", "kv")
        self.dims = dims
        self.bf16 = bool(emulate_bf16)
        self.absorbed = self.bf16 and cross_form == "encoder_output"
        self.cross_key_splits = int(cross_key_splits) if cross_key_splits else None
        self.flash_tiles = bool(flash_tiles) and self.bf16
        self.dt = acc_dtype
        self.sd = {k: v.to(acc_dtype) for k, v in sd.items()}
        if self.bf16:
            for k, v in self.sd.items():
                if (k.endswith(".weight") and v.dim() >= 2) or k == "decoder.positional_embedding":
                    self.sd[k] = self._r(v)

    def _r(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(torch.bfloat16).to(self.dt) if self.bf16 else x

    # -- attention ------------------------------------------------------------------------------
    def _heads(self, x, n_head):
        B, T, D = x.shape
        return x.view(B, T, n_head, D // n_head).permute(0, 2, 1, 3)

    def _qkv_attention(self, q, k, v, n_head, mask=None, round_p=False):
        D = q.shape[-1]
        scale = (D // n_head) ** -0.25
        q = self._r(self._heads(q, n_head) * scale)
        k = self._r(self._heads(k, n_head) * scale)
        v = self._r(self._heads(v, n_head))
        qk = q @ k.transpose(-1, -2)
        if mask is not None:
            qk = qk + mask
        if self.bf16 and round_p and self.flash_tiles:
            out = self._online_attention(qk, v, 64)
        elif self.bf16 and round_p:
            p = torch.exp(qk - qk.amax(dim=-1, keepdim=True))
            out = (self._r(p) @ v) / p.sum(dim=-1, keepdim=True)
        else:
            out = F.softmax(qk.to(self.dt), dim=-1) @ v
        return self._r(out.permute(0, 2, 1, 3).flatten(start_dim=2))

    def _cross_attention_absorbed(self, h, xa, p, n_head):
        """bf16-mode decoder cross-attention on the encoder output (the engine's xattn.hip path): ``h [B, t, d]`` the
        normalised decoder stream, ``xa [B, 1500, d]`` the (bf16-rounded) encoder output."""
        sd = self.sd
        B, t, D = h.shape
        hd = D // n_head
        s = hd ** -0.25
        q = self._r((F.linear(h, sd[p + ".query.weight"], sd[p + ".query.bias"]) * s)).view(B, t, n_head, hd)
        wk = sd[p + ".key.weight"].view(n_head, hd, D)                   # Wk_h: rows h*64 .. of the key projection
        qt = self._r(torch.einsum("bthc,hcd->bhtd", q, wk) * s)          # q~ [B, H, t, D]
        sc = qt @ xa.transpose(-1, -2)[:, None]                          # [B, H, t, 1500]
        if self.cross_key_splits:
            ctx = self._online_context(sc, xa, self.cross_key_splits)
        else:
            pr = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
            ctx = self._r((self._r(pr) @ xa[:, None]) / pr.sum(dim=-1, keepdim=True))     # [B, H, t, D]
        wv = sd[p + ".value.weight"].view(n_head, hd, D)
        out = torch.einsum("bhtd,hcd->bthc", ctx, wv).reshape(B, t, D) + sd[p + ".value.bias"]
        return self._r(out)

    def _online_attention(self, qk, v, tile):
        """softmax(qk) v the way a flash kernel with `tile` keys per step evaluates it: weights exp(score - running maximum)
        rounded to bf16 for the P.V product, unrounded in the row sum, earlier tiles rescaled in fp32 (flash.hip)."""
        *lead, T = qk.shape
        n_t = (T + tile - 1) // tile
        pad = n_t * tile - T
        s_p = torch.cat([qk, torch.full((*lead, pad), float("-inf"), dtype=qk.dtype)], dim=-1) if pad else qk
        s_t = s_p.view(*lead, n_t, tile)
        m_run = torch.cummax(s_t.amax(dim=-1), dim=-1).values
        p = torch.exp(s_t - m_run[..., None])
        resc = torch.exp(m_run - m_run[..., -1:])
        l = (p.sum(dim=-1) * resc).sum(dim=-1)
        pe = (self._r(p) * resc[..., None]).reshape(*lead, n_t * tile)[..., :T]
        return (pe @ v) / l[..., None]

    def _online_context(self, sc, xa, n_split, tile=32):
        """The stream kernel's softmax-weighted context, rounding point for rounding point (whisper_amd/csrc/xattn.hip,
        xattn_estream_kernel + the combine of xattn_cproj_kernel): key split s takes the 32-key tiles [s nt / S, (s+1) nt / S)
        of the nt = ceil(T / 32) tiles (T padded to a multiple of 32: zero rows, scores -inf); per tile the weights are
        exp(score - running maximum), rounded to bf16 for the P.E product and left unrounded in the row sum, earlier tiles
        rescaled by exp(old maximum - new maximum) in fp32; the split's context is divided by its own sum and rounded to
        bf16; the splits are combined with w_s = l_s exp(m_s - M) / sum in fp32 and the result rounded to bf16."""
        B, H, t, T = sc.shape
        nt = (((T + 15) // 16 + 1) // 2 * 2) // 2                          # the engine's x_nt16 / 2
        parts, ms, ls = [], [], []
        for s_i in range(n_split):
            t0, t1 = s_i * nt // n_split, (s_i + 1) * nt // n_split
            k0, k1 = t0 * tile, min(t1 * tile, T)
            s_sp = sc[..., k0:k1]
            n_t = t1 - t0
            pad = n_t * tile - (k1 - k0)
            if pad:
                s_sp = torch.cat([s_sp, torch.full((B, H, t, pad), float("-inf"), dtype=sc.dtype)], dim=-1)
            s_t = s_sp.view(B, H, t, n_t, tile)
            m_run = torch.cummax(s_t.amax(dim=-1), dim=-1).values          # [B, H, t, n_t] running maximum after each tile
            m_fin = m_run[..., -1:]
            p = torch.exp(s_t - m_run[..., None])                          # weights against the running maximum of their tile
            resc = torch.exp(m_run - m_fin)                                # product of the later tiles' rescale factors
            l = (p.sum(dim=-1) * resc).sum(dim=-1)                         # row sum: unrounded weights
            pe = (self._r(p) * resc[..., None]).reshape(B, H, t, n_t * tile)[..., :k1 - k0]
            c = pe @ xa[:, None, k0:k1]
            parts.append(self._r(c / l[..., None]).to(self.dt)); ms.append(m_fin[..., 0]); ls.append(l)
        m_all = torch.stack(ms, dim=-1)
        w = torch.stack(ls, dim=-1) * torch.exp(m_all - m_all.amax(dim=-1, keepdim=True))
        w = w / w.sum(dim=-1, keepdim=True)
        ctx = sum(w[..., i, None] * parts[i] for i in range(n_split))
        return self._r(ctx)

    def _lin(self, x, prefix, bias=True):
        return F.linear(x, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"] if bias else None)

    def _ln(self, x, prefix):
        return self._r(_layer_norm(x, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"]))

    def _mlp(self, x, p):
        return self._lin(self._r(F.gelu(self._lin(x, p + ".mlp.0"))), p + ".mlp.2")

    # -- encoder --------------------------------------------------------------------------------
    def encode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel ``[B, n_mels, 3000]`` → ``[B, 1500, d]``."""
        sd, dims = self.sd, self.dims
        x = self._r(F.gelu(F.conv1d(self._r(mel.to(self.dt)), sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=1)))
        x = F.gelu(F.conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], stride=2, padding=1))
        x = x.permute(0, 2, 1)
        x = x + sd["encoder.positional_embedding"]
        for i in range(dims.n_audio_layer):
            p = f"encoder.blocks.{i}"
            h = self._ln(x, p + ".attn_ln")
            a = self._qkv_attention(self._lin(h, p + ".attn.query"), self._lin(h, p + ".attn.key", bias=False),
                                    self._lin(h, p + ".attn.value"), dims.n_audio_head, round_p=True)
            x = x + self._lin(a, p + ".attn.out")
            x = x + self._mlp(self._ln(x, p + ".mlp_ln"), p)
        return self._ln(x, "encoder.ln_post")

    # -- decoder --------------------------------------------------------------------------------
    def cross_kv(self, xa: torch.Tensor):
        """Per decoder layer (K, V) of the encoder output, computed once per window
        (hf:...modeling_whisper.py:311-335 cross-KV reuse)."""
        if self.absorbed:       # the engine attends the encoder output itself: nothing to project
            return [xa] * self.dims.n_text_layer
        out = []
        for i in range(self.dims.n_text_layer):
            p = f"decoder.blocks.{i}.cross_attn"
            out.append((self._lin(xa, p + ".key", bias=False), self._lin(xa, p + ".value")))
        return out

    def decode_step(self, tokens: torch.Tensor, offset: int, cross, self_cache: List) -> torch.Tensor:
        """Run ``tokens [B, t]`` occupying positions offset..offset+t-1; append to ``self_cache`` (a list of
        [K, V] per layer, created empty by the caller).  Returns logits ``[B, t, n_vocab]`` float32."""
        sd, dims = self.sd, self.dims
        B, t = tokens.shape
        x = sd["decoder.token_embedding.weight"][tokens] + sd["decoder.positional_embedding"][offset:offset + t]
        total = offset + t
        mask = torch.full((total, total), float("-inf"), dtype=self.dt).triu_(1)[offset:total, :total]
        for i in range(dims.n_text_layer):
            p = f"decoder.blocks.{i}"
            h = self._ln(x, p + ".attn_ln")
            k_new = self._lin(h, p + ".attn.key", bias=False)
            v_new = self._lin(h, p + ".attn.value")
            if len(self_cache) <= i:
                self_cache.append([k_new, v_new])
            else:
                self_cache[i][0] = torch.cat([self_cache[i][0], k_new], dim=1)
                self_cache[i][1] = torch.cat([self_cache[i][1], v_new], dim=1)
            a = self._qkv_attention(self._lin(h, p + ".attn.query"), self_cache[i][0], self_cache[i][1],
                                    dims.n_text_head, mask)
            x = x + self._lin(a, p + ".attn.out")
            h = self._ln(x, p + ".cross_attn_ln")
            if self.absorbed:
                a = self._cross_attention_absorbed(h, cross[i], p + ".cross_attn", dims.n_text_head)
            else:
                ck, cv = cross[i]
                a = self._qkv_attention(self._lin(h, p + ".cross_attn.query"), ck, cv, dims.n_text_head)
            x = x + self._lin(a, p + ".cross_attn.out")
            x = x + self._mlp(self._ln(x, p + ".mlp_ln"), p)
        x = self._ln(x, "decoder.ln")
        return (x @ sd["decoder.token_embedding.weight"].t()).float()        # logits are fp32 for every accumulation dtype

    def logits(self, tokens: torch.Tensor, xa: torch.Tensor) -> torch.Tensor:
        """Teacher-forced logits for every position of ``tokens [B, t]``."""
        return self.decode_step(tokens, 0, self.cross_kv(xa), [])


# -------------------------------------------------------------------------------------------------
# decoding rules
# -------------------------------------------------------------------------------------------------

def suppress_token_ids(tok, non_speech: Sequence[int]) -> List[int]:
    """Upstream ``DecodingTask._get_suppress_tokens`` with the default ``suppress_tokens="-1"``: the non-speech
    symbol ids plus [transcribe, translate, sot, sot_prev, sot_lm, no_speech]
    (hf:generation/logits_process.py:1902-1906 applies the resulting list)."""
    ids = list(non_speech) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech]
    return sorted(set(int(i) for i in ids))


def initial_tokens(tok, language_index: int, task: str = "transcribe", without_timestamps: bool = True,
                   prompt_ids: Optional[Sequence[int]] = None, n_text_ctx: int = 448) -> List[int]:
    """``[sot_prev, prompt…] + [sot, lang, task, (no_timestamps)]`` (hf:models/whisper/generation_whisper.py:1487-1608).
    The prompt keeps its last ``n_ctx // 2 - 1`` ids, as upstream ``_get_initial_tokens`` does."""
    seq = [tok.sot, tok.lang(language_index), tok.transcribe if task == "transcribe" else tok.translate]
    if without_timestamps:
        seq.append(tok.no_timestamps)
    if prompt_ids:
        keep = list(prompt_ids)[-(n_text_ctx // 2 - 1):]
        seq = [tok.sot_prev] + keep + seq
    return seq


def apply_timestamp_rules(logits: torch.Tensor, tokens: torch.Tensor, sample_begin: int, tok,
                          max_initial_timestamp_index: Optional[int] = 50) -> None:
    """In-place timestamp rules, hf:generation/logits_process.py:2000-2047 (same steps as upstream
    ``ApplyTimestampRules.apply``)."""
    logits[:, tok.no_timestamps] = -np.inf
    tb = tok.timestamp_begin
    for k in range(tokens.shape[0]):
        seq = tokens[k, sample_begin:].tolist()
        last_was_ts = len(seq) >= 1 and seq[-1] >= tb
        penult_was_ts = len(seq) < 2 or seq[-2] >= tb
        if last_was_ts:
            if penult_was_ts:
                logits[k, tb:] = -np.inf
            else:
                logits[k, :tok.eot] = -np.inf
        ts = [s for s in seq if s >= tb]
        if ts:
            last = ts[-1] if (last_was_ts and not penult_was_ts) else ts[-1] + 1
            logits[k, tb:last] = -np.inf
    if tokens.shape[1] == sample_begin:
        logits[:, :tb] = -np.inf
        if max_initial_timestamp_index is not None:
            logits[:, tb + max_initial_timestamp_index + 1:] = -np.inf
    logprobs = F.log_softmax(logits.float(), dim=-1)
    for k in range(tokens.shape[0]):
        ts_lp = logprobs[k, tb:].logsumexp(dim=-1)
        max_text = logprobs[k, :tb].max()
        if ts_lp > max_text:
            logits[k, :tb] = -np.inf


def apply_logit_filters(logits: torch.Tensor, tokens: torch.Tensor, sample_begin: int, tok, suppress_ids: Sequence[int],
                        without_timestamps: bool = True, suppress_blank: bool = True,
                        max_initial_timestamp_index: Optional[int] = 50) -> None:
    """Upstream ``DecodingTask.logit_filters`` applied in upstream's order, in place: SuppressBlank at the first sampled
    position (hf:generation/logits_process.py:1859-1866 ``SuppressTokensAtBeginLogitsProcessor`` with (blank, eot)),
    SuppressTokens (:1902-1906), ApplyTimestampRules (:2000-2047) when timestamps are enabled.  The HF generate stack
    orders its processors the same way (hf:models/whisper/generation_whisper.py:1774-1812).
    tests/golden/hf_rules.npz pins this function against those three HF classes."""
    if suppress_blank and tokens.shape[1] == sample_begin:
        logits[:, [tok.blank, tok.eot]] = -np.inf
    logits[:, list(suppress_ids)] = -np.inf
    if not without_timestamps:
        apply_timestamp_rules(logits, tokens, sample_begin, tok, max_initial_timestamp_index)


def no_speech_probability(logits_at_sot: torch.Tensor, tok) -> torch.Tensor:
    """softmax of the UNFILTERED logits at the <|startoftranscript|> position, read at <|nospeech|>
    (hf:generation/logits_process.py:2091-2112 ``WhisperNoSpeechDetection``)."""
    return logits_at_sot.float().softmax(dim=-1)[:, tok.no_speech]


def greedy_decode(model: OracleWhisper, xa: torch.Tensor, init_tokens: Sequence[int], tok,
                  suppress_ids: Sequence[int], max_new_tokens: Optional[int] = None,
                  without_timestamps: bool = True, suppress_blank: bool = True) -> Dict[str, object]:
    """Upstream ``DecodingTask._main_loop`` + ``GreedyDecoder.update`` at temperature 0, no fallback.

    Loop (≤ ``sample_len = n_text_ctx // 2`` steps): logits of the last position → at step 0 the no-speech
    probability is read from the *unfiltered* softmax at the SOT position (hf:generation/logits_process.py:
    2091-2112) → SuppressBlank at the first sampled step (hf:...:1859-1866) → SuppressTokens (:1902-1906) →
    timestamp rules when enabled → argmax; ``sum_logprob`` accumulates log_softmax(filtered)[token] while the
    sequence has not emitted EOT; finished sequences keep emitting EOT; stop when all are finished.

    Returns tokens (list per sequence, sampled part up to but excluding EOT), sum_logprobs, avg_logprobs
    (sum / (len + 1), hf:models/whisper/generation_whisper.py:1958-1974), no_speech_probs, and per-step top-1/top-2
    margins of the filtered logits (used by the fixtures to prove the argmax is well-conditioned)."""
    dims = model.dims
    B = xa.shape[0]
    sample_len = dims.n_text_ctx // 2 if max_new_tokens is None else max_new_tokens
    init = list(init_tokens)
    sample_begin = len(init)
    sot_index = init.index(tok.sot)
    tokens = torch.tensor([init] * B, dtype=torch.long)
    cross = model.cross_kv(xa)
    cache: List = []
    sum_logprobs = torch.zeros(B)
    no_speech = torch.zeros(B)
    margins = []
    steps = 0
    for i in range(sample_len):
        if i == 0:
            lg_all = model.decode_step(tokens, 0, cross, cache)
            no_speech = no_speech_probability(lg_all[:, sot_index], tok).clone()
            lg = lg_all[:, -1].clone()
        else:
            lg = model.decode_step(tokens[:, -1:], tokens.shape[1] - 1, cross, cache)[:, -1].clone()
        apply_logit_filters(lg, tokens, sample_begin, tok, suppress_ids, without_timestamps, suppress_blank)
        top2 = lg.topk(2, dim=-1).values
        margins.append((top2[:, 0] - top2[:, 1]).numpy().copy())
        nxt = lg.argmax(dim=-1)
        logprobs = F.log_softmax(lg.float(), dim=-1)
        cur = logprobs[torch.arange(B), nxt]
        alive = tokens[:, -1] != tok.eot
        sum_logprobs = sum_logprobs + cur * alive
        nxt = torch.where(alive, nxt, torch.full_like(nxt, tok.eot))
        tokens = torch.cat([tokens, nxt[:, None]], dim=-1)
        steps += 1
        if bool((tokens[:, -1] == tok.eot).all()) or tokens.shape[-1] > dims.n_text_ctx:
            break
    out_tokens = []
    for b in range(B):
        seq = tokens[b, sample_begin:].tolist()
        if tok.eot in seq:
            seq = seq[:seq.index(tok.eot)]
        out_tokens.append(seq)
    avg = [float(sum_logprobs[b]) / (len(out_tokens[b]) + 1) for b in range(B)]
    return {"tokens": out_tokens, "sum_logprobs": sum_logprobs.numpy(), "avg_logprobs": np.array(avg, np.float32),
            "no_speech_probs": no_speech.numpy(), "margins": np.stack(margins, axis=0), "steps": steps,
            "raw_tokens": tokens.numpy()}


def beam_search_decode(model: OracleWhisper, xa: torch.Tensor, init_tokens: Sequence[int], tok,
                       suppress_ids: Sequence[int], beam_size: int = 5, patience: float = 1.0,
                       max_new_tokens: Optional[int] = None, without_timestamps: bool = True,
                       suppress_blank: bool = True) -> Dict[str, object]:
    """Upstream ``DecodingTask`` with ``BeamSearchDecoder`` (temperature 0, ``beam_size`` given) [upstream-recall: the
    openai-whisper source is not in this container; transformers' beam search is a different algorithm, so this
    function is NOT pinned by the HF golden vectors -- parity of the beam path is unpinned, the GPU path is checked
    against this restatement only].

    Per step, per audio: every beam proposes its ``beam_size + 1`` best tokens of log_softmax(filtered logits);
    candidates are keyed by the full sequence (at the first step all beams hold the same prefix, so they collapse to
    one beam's proposals); candidates are visited by cumulative log-probability, best first: a candidate ending in EOT
    goes to the audio's finished list, any other becomes a beam, until ``beam_size`` beams are kept.  Newly finished
    sequences are appended best-first while the list has fewer than ``max_candidates = round(beam_size * patience)``
    entries; decoding stops when every audio has that many (or after ``sample_len`` steps / ``n_text_ctx`` tokens).  The
    self-attention cache follows the beams (``rearrange_kv_cache``).  ``finalize`` tops short lists up with the live
    beams (best first, EOT appended); the result is the finished sequence with the highest sum_logprob / length
    (``MaximumLikelihoodRanker``, ``length_penalty=None``).

    Returns tokens (per audio, without EOT), sum_logprobs of the selected sequence, avg_logprobs, no_speech_probs,
    steps, and ``candidates`` (per audio: list of (tokens, sum_logprob) in finished-list order)."""
    dims = model.dims
    n_audio = xa.shape[0]
    n = beam_size
    max_candidates = round(beam_size * patience)
    sample_len = dims.n_text_ctx // 2 if max_new_tokens is None else max_new_tokens
    init = list(init_tokens)
    sample_begin = len(init)
    sot_index = init.index(tok.sot)
    tokens = torch.tensor([init] * (n_audio * n), dtype=torch.long)
    xa_rep = xa.repeat_interleave(n, dim=0)
    cross = model.cross_kv(xa_rep)
    cache: List = []
    sum_logprobs = torch.zeros(n_audio * n)
    finished_sequences: List[Dict] = [{} for _ in range(n_audio)]
    no_speech = torch.zeros(n_audio * n)
    steps = 0
    for i in range(sample_len):
        if i == 0:
            lg_all = model.decode_step(tokens, 0, cross, cache)
            no_speech = no_speech_probability(lg_all[:, sot_index], tok).clone()
            lg = lg_all[:, -1].clone()
        else:
            lg = model.decode_step(tokens[:, -1:], tokens.shape[1] - 1, cross, cache)[:, -1].clone()
        apply_logit_filters(lg, tokens, sample_begin, tok, suppress_ids, without_timestamps, suppress_blank)
        logprobs = F.log_softmax(lg.float(), dim=-1)
        next_tokens, source_indices, newly = [], [], []
        new_sums = sum_logprobs.clone()
        for a in range(n_audio):
            scores, sources, finished = {}, {}, {}
            for j in range(n):
                idx = a * n + j
                prefix = tokens[idx].tolist()
                vals, ids = logprobs[idx].topk(n + 1)
                for logprob, token in zip(vals, ids):
                    seq = tuple(prefix + [int(token)])
                    scores[seq] = float(sum_logprobs[idx] + logprob)
                    sources[seq] = idx
            saved = 0
            for seq in sorted(scores, key=scores.get, reverse=True):
                if seq[-1] == tok.eot:
                    finished[seq] = scores[seq]
                else:
                    new_sums[len(next_tokens)] = scores[seq]
                    next_tokens.append(seq)
                    source_indices.append(sources[seq])
                    saved += 1
                    if saved == n:
                        break
            newly.append(finished)
        sum_logprobs = new_sums
        tokens = torch.tensor(next_tokens, dtype=torch.long)
        src = torch.tensor(source_indices)
        for layer in cache:                       # rearrange_kv_cache
            layer[0] = layer[0][src]
            layer[1] = layer[1][src]
        for prev, new in zip(finished_sequences, newly):
            for seq in sorted(new, key=new.get, reverse=True):
                if len(prev) >= max_candidates:
                    break
                prev[seq] = new[seq]
        steps += 1
        completed = all(len(f) >= max_candidates for f in finished_sequences)
        if completed or tokens.shape[-1] > dims.n_text_ctx:
            break
    # finalize
    tok3 = tokens.reshape(n_audio, n, -1)
    sums2 = sum_logprobs.reshape(n_audio, n)
    for a, seqs in enumerate(finished_sequences):
        if len(seqs) < n:
            for j in list(np.argsort(sums2[a].numpy()))[::-1]:
                seq = tok3[a, j].tolist() + [tok.eot]
                seqs[tuple(seq)] = float(sums2[a, j])
                if len(seqs) >= n:
                    break
    out_tokens, out_sums, cands = [], [], []
    for a, seqs in enumerate(finished_sequences):
        items = []
        for seq, sc in seqs.items():
            body = list(seq[sample_begin:])
            body = body[:body.index(tok.eot)]
            items.append((body, sc))
        best = int(np.argmax([sc / len(body) if len(body) > 0 else -np.inf for body, sc in items])) if any(len(b) for b, _ in items) else 0
        out_tokens.append(items[best][0]); out_sums.append(items[best][1]); cands.append(items)
    avg = [out_sums[a] / (len(out_tokens[a]) + 1) for a in range(n_audio)]
    return {"tokens": out_tokens, "sum_logprobs": np.array(out_sums, np.float32), "avg_logprobs": np.array(avg, np.float32),
            "no_speech_probs": no_speech.reshape(n_audio, n)[:, 0].numpy(), "steps": steps, "candidates": cands}


def transcribe_windows(model: OracleWhisper, audio: np.ndarray, init_tokens: Sequence[int], tok,
                       suppress_ids: Sequence[int], max_new_tokens: Optional[int] = None,
                       without_timestamps: bool = True, batch: int = 8) -> Dict[str, object]:
    """Independent-window long-form variant (a declared divergence from upstream's sequential
    seek loop): one log-mel over the whole file (+30 s zero padding, global clamp), fixed 3000-frame windows,
    each window decoded from the same initial tokens with no carry-over."""
    mel = log_mel_spectrogram(audio, model.dims.n_mels, padding=N_SAMPLES)
    content_frames = mel.shape[-1] - N_FRAMES
    n_win = max(1, math.ceil(content_frames / N_FRAMES))
    # upstream slices mel[:, seek : seek + min(N_FRAMES, content_frames - seek)] and zero-pads to 3000 frames
    segs = torch.stack([pad_or_trim_frames(mel[:, w * N_FRAMES:min((w + 1) * N_FRAMES, content_frames)])
                        for w in range(n_win)])
    toks, sums, nsp = [], [], []
    for s in range(0, n_win, batch):
        r = greedy_decode(model, model.encode(segs[s:s + batch]), init_tokens, tok, suppress_ids,
                          max_new_tokens, without_timestamps)
        toks += r["tokens"]; sums += list(r["sum_logprobs"]); nsp += list(r["no_speech_probs"])
    return {"tokens": toks, "sum_logprobs": np.array(sums, np.float32), "no_speech_probs": np.array(nsp, np.float32),
            "n_windows": n_win, "mel": mel}


# -------------------------------------------------------------------------------------------------
# sequential long-form loop
# -------------------------------------------------------------------------------------------------

def compression_ratio(text: str) -> float:
    import zlib
    b = text.encode("utf-8")
    return len(b) / len(zlib.compress(b))


def avg_logprob(sum_logprob: float, n_tokens: int) -> float:
    """Upstream ``DecodingResult.avg_logprob = sum_logprob / (len(tokens) + 1)``: the EOT step counts
    (hf:models/whisper/generation_whisper.py:1958-1974 ``_retrieve_avg_logprobs``: "don't remove the eos token logprob")."""
    return float(sum_logprob) / (int(n_tokens) + 1)


def fallback_decision(compression_ratio_value: float, avg_logprob_value: float, no_speech_prob: float,
                      compression_ratio_threshold: Optional[float] = 2.4, logprob_threshold: Optional[float] = -1.0,
                      no_speech_threshold: Optional[float] = 0.6):
    """(needs_fallback, should_skip) of one decoded window [upstream-recall: ``decode_with_fallback`` and the no-speech skip of
    ``Whisper.transcribe``]; HF's counterpart is hf:models/whisper/generation_whisper.py:1243-1287 ``_need_fallback``.
    Where the two differ, upstream is followed (tests/golden/hf_sequential.json records the cases): upstream cancels the
    fallback whenever no_speech_prob > threshold, HF only if the log-probability is ALSO below its threshold; upstream skips
    a silent window unless avg_logprob > threshold, HF only if avg_logprob < threshold (they differ at equality)."""
    needs_fallback = False
    if compression_ratio_threshold is not None and compression_ratio_value > compression_ratio_threshold:
        needs_fallback = True
    if logprob_threshold is not None and avg_logprob_value < logprob_threshold:
        needs_fallback = True
    if no_speech_threshold is not None and no_speech_prob > no_speech_threshold:
        needs_fallback = False
    should_skip = False
    if no_speech_threshold is not None:
        should_skip = no_speech_prob > no_speech_threshold
        if logprob_threshold is not None and avg_logprob_value > logprob_threshold:
            should_skip = False
    return needs_fallback, should_skip


def split_segments(tokens: Sequence[int], timestamp_begin: int, seek: int, segment_size: int):
    """The ids one window decoded to (after the prompt, EOT stripped) -> ``([(start s, end s, ids)], next seek)``
    [upstream-recall: the segment loop of ``Whisper.transcribe``]; HF's counterpart, which tests/golden/hf_sequential.json
    pins this against, is hf:models/whisper/generation_whisper.py:1977-2074 ``_retrieve_segment``.  Ids are cut after every
    pair of consecutive timestamps; a single timestamp at the very end means "no speech after it" (seek moves on by the
    whole window), otherwise seek moves to the last timestamp of a pair (x 2 mel frames per timestamp unit) and everything
    after it is decoded again; with no pair at all the window is one segment that ends at its last timestamp, if it has one
    other than <|0.00|>, else at the window's end."""
    tokens = list(tokens)
    tb = timestamp_begin
    time_offset = seek * HOP_LENGTH / SAMPLE_RATE
    is_ts = [t >= tb for t in tokens]
    single_timestamp_ending = is_ts[-2:] == [False, True]
    consecutive = [i + 1 for i in range(len(tokens) - 1) if is_ts[i] and is_ts[i + 1]]
    out = []
    if consecutive:
        slices = list(consecutive)
        if single_timestamp_ending:
            slices.append(len(tokens))
        last_slice = 0
        for cur in slices:
            sl = tokens[last_slice:cur]
            out.append((time_offset + (sl[0] - tb) * 0.02, time_offset + (sl[-1] - tb) * 0.02, sl))
            last_slice = cur
        if single_timestamp_ending:
            seek_next = seek + segment_size
        else:
            seek_next = seek + (tokens[last_slice - 1] - tb) * 2
    else:
        duration = segment_size * HOP_LENGTH / SAMPLE_RATE
        ts = [t for t in tokens if t >= tb]
        if ts and ts[-1] != tb:
            duration = (ts[-1] - tb) * 0.02
        out.append((time_offset, time_offset + duration, tokens))
        seek_next = seek + segment_size
    return out, seek_next


def transcribe_sequential(model: OracleWhisper, audio: np.ndarray, tok, suppress_ids: Sequence[int], decode_text,
                          language_index: int = 0, initial_prompt_ids: Optional[Sequence[int]] = None,
                          compression_ratio_threshold: Optional[float] = 2.4, logprob_threshold: Optional[float] = -1.0,
                          no_speech_threshold: Optional[float] = 0.6, condition_on_previous_text: bool = True,
                          max_new_tokens: Optional[int] = None) -> Dict[str, object]:
    """Upstream ``Whisper.transcribe`` main loop at temperature 0 [upstream-recall; the same loop in the HF port is
    hf:models/whisper/generation_whisper.py:785-903 (loop), :1853-1918 (prompt), :1977-2074 (segments / seek), :1243-1287
    (fallback checks)]: whole-file log-mel with 30 s of padding; ``while seek < content_frames``: cut 3000 frames at
    ``seek``, decode with timestamps and the previous text as prompt (``[sot_prev] + all_tokens[-223:] + sot sequence``),
    skip the window if it looks like silence (no_speech_prob > threshold and avg_logprob <= logprob threshold), split the
    ids at consecutive timestamp pairs into segments, advance ``seek`` by the last timestamp (x2 frames) or by the whole
    window.  Only temperature 0 is restated: the fallback decision is computed and reported (``needs_fallback``) but the
    sampled re-decodes at T > 0 depend on torch's global generator and are not part of the parity surface."""
    dims = model.dims
    mel = log_mel_spectrogram(audio, dims.n_mels, padding=N_SAMPLES)
    content_frames = mel.shape[-1] - N_FRAMES
    seek = 0
    all_tokens: List[int] = list(initial_prompt_ids or [])
    prompt_reset_since = 0
    segments: List[Dict[str, object]] = []
    windows = 0
    tb = tok.timestamp_begin
    while seek < content_frames:
        segment_size = min(N_FRAMES, content_frames - seek)
        mel_segment = pad_or_trim_frames(mel[:, seek:seek + segment_size])
        prompt = all_tokens[prompt_reset_since:]
        init = initial_tokens(tok, language_index, without_timestamps=False, prompt_ids=prompt, n_text_ctx=dims.n_text_ctx)
        with torch.no_grad():
            r = greedy_decode(model, model.encode(mel_segment[None]), init, tok, suppress_ids, max_new_tokens=max_new_tokens,
                              without_timestamps=False)
        windows += 1
        tokens = r["tokens"][0]
        avg_lp = avg_logprob(float(r["sum_logprobs"][0]), len(tokens)); no_speech_prob = float(r["no_speech_probs"][0])
        text = decode_text(tokens).strip()
        cr = compression_ratio(text) if text else 0.0
        needs_fallback, should_skip = fallback_decision(cr, avg_lp, no_speech_prob, compression_ratio_threshold, logprob_threshold,
                                                        no_speech_threshold)
        if should_skip:
            seek += segment_size
            continue
        pieces, seek_next = split_segments(tokens, tb, seek, segment_size)
        current: List[Dict[str, object]] = [
            {"seek": seek, "start": start, "end": end, "tokens": list(toks), "text": decode_text([t for t in toks if t < tok.eot]),
             "temperature": 0.0, "avg_logprob": avg_lp, "compression_ratio": cr, "no_speech_prob": no_speech_prob,
             "needs_fallback": needs_fallback} for start, end, toks in pieces]
        for sg in current:
            if sg["start"] == sg["end"] or sg["text"].strip() == "":
                sg["text"] = ""; sg["tokens"] = []
        segments.extend(current)
        for sg in current:
            all_tokens.extend(sg["tokens"])
        if not condition_on_previous_text:
            prompt_reset_since = len(all_tokens)
        seek = seek_next
    return {"segments": segments, "windows": windows, "tokens": all_tokens[len(initial_prompt_ids or []):],
            "text": decode_text([t for t in all_tokens[len(initial_prompt_ids or []):] if t < tok.eot])}
