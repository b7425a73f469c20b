"""Python host side of the MI355X Whisper engine: owns one ``wx_engine`` handle (C-ABI, include/whisper_amd.h) and
drives mel -> encoder -> greedy decode for batches of 30 s windows.  PyTorch is used for device buffers and streams
only; all arithmetic runs in the HIP library.

Replaces what the reference obtains from ``whisper.load_model(...)`` + ``model.transcribe(...)``
(ref:app/transcription/speech_recognition.py:22,55).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .audio import HOP_LENGTH, N_FRAMES, N_SAMPLES, n_windows, slaney_mel_filterbank
from .checkpoint import ModelDims, SpecialTokens, is_multilingual, non_speech_suppress_ids, special_tokens_for

PRECISIONS = {"fp32": 0, "bf16": 1}


def default_suppress_ids(tok: SpecialTokens, n_vocab: int = 51865, non_speech: Optional[Sequence[int]] = None) -> List[int]:
    """Upstream ``_get_suppress_tokens`` for ``suppress_tokens="-1"``: the vocabulary's non-speech symbols (derived from a
    loaded vocabulary by the caller, else the table of ``checkpoint.non_speech_suppress_ids``) + task/control tokens."""
    ns = non_speech_suppress_ids(n_vocab) if non_speech is None else non_speech
    ids = list(ns) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech]
    return sorted(set(int(i) for i in ids))


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class WhisperEngine:
    def __init__(self, dims: ModelDims, state_dict: Dict[str, torch.Tensor], precision: str = "bf16",
                 max_batch: int = 8, device: int = 0, suppress_ids: Optional[Sequence[int]] = None,
                 mel_filters: Optional[np.ndarray] = None, n_lanes: int = 1, share_weights_with: Optional["WhisperEngine"] = None):
        """``mel_filters``: optional ``[n_mels, 201]`` float32 filterbank (e.g. the array of upstream's
        ``assets/mel_filters.npz``); default: the Slaney bank computed by ``audio.slaney_mel_filterbank``.
        ``n_lanes``: how many engines will share this GPU at the same time (whisper_amd/lanes.py); sizes the grid of the
        decoder's cross-attention stream (wx_config.n_lanes).
        ``share_weights_with``: an engine of the same model, precision and device whose device weights this one uses instead
        of uploading its own (``wx_create_shared``; ``state_dict`` is then ignored and may be None) -- see ``replica()``."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        if not torch.cuda.is_available():
            raise RuntimeError("whisper_amd needs a ROCm GPU (no CPU fallback)")
        self.lib = _lib.load()
        self.dims, self.precision, self.max_batch, self.device = dims, precision, int(max_batch), int(device)
        self.tok = special_tokens_for(dims.n_vocab)
        self.torch_device = torch.device("cuda", self.device)
        t = self.tok
        cfg = _lib.WxConfig(
            n_mels=dims.n_mels, n_audio_ctx=dims.n_audio_ctx, n_audio_state=dims.n_audio_state,
            n_audio_head=dims.n_audio_head, n_audio_layer=dims.n_audio_layer, n_vocab=dims.n_vocab,
            n_text_ctx=dims.n_text_ctx, n_text_state=dims.n_text_state, n_text_head=dims.n_text_head,
            n_text_layer=dims.n_text_layer, precision=PRECISIONS[precision], max_batch=self.max_batch, device=self.device,
            tok_eot=t.eot, tok_sot=t.sot, tok_translate=t.translate, tok_transcribe=t.transcribe, tok_sot_prev=t.sot_prev,
            tok_sot_lm=t.sot_lm, tok_no_speech=t.no_speech, tok_no_timestamps=t.no_timestamps,
            tok_timestamp_begin=t.timestamp_begin, tok_blank=t.blank, n_lanes=int(n_lanes))
        self.n_lanes = int(n_lanes)
        self.encode_gate = None           # a lock shared by the lanes of a GPU (EngineLanes sets it); None: encode() never waits
        handle = ctypes.c_void_p()
        if share_weights_with is not None:
            p = share_weights_with
            if (p.dims, p.precision, p.device) != (dims, precision, self.device) or not p.handle:
                raise ValueError("share_weights_with: a live engine of the same model, precision and device is needed")
            _lib.check(self.lib.wx_create_shared(p.handle, ctypes.byref(cfg), ctypes.byref(handle)), "wx_create_shared")
            self.handle = handle
            self.suppress_ids = list(p.suppress_ids)
            if suppress_ids is not None:
                self.set_suppress_ids(suppress_ids)
            return
        _lib.check(self.lib.wx_create(ctypes.byref(cfg), ctypes.byref(handle)), "wx_create")
        self.handle = handle
        try:
            for name, w in state_dict.items():
                a = np.ascontiguousarray(w.detach().cpu().float().numpy())
                shape = (ctypes.c_int64 * a.ndim)(*a.shape)
                _lib.check(self.lib.wx_set_weight(self.handle, name.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim),
                           f"wx_set_weight({name})")
            fb = np.ascontiguousarray(slaney_mel_filterbank(dims.n_mels) if mel_filters is None else mel_filters, dtype=np.float32)
            if fb.shape != (dims.n_mels, 201):
                raise ValueError(f"mel_filters must be [{dims.n_mels}, 201]")
            _lib.check(self.lib.wx_set_mel_filters(self.handle, fb.ctypes.data_as(ctypes.c_void_p)), "wx_set_mel_filters")
            self.suppress_ids = list(default_suppress_ids(t, dims.n_vocab) if suppress_ids is None else suppress_ids)
            sup = np.asarray(self.suppress_ids, np.int32)
            _lib.check(self.lib.wx_set_suppress_tokens(self.handle, sup.ctypes.data_as(ctypes.c_void_p), len(sup)),
                       "wx_set_suppress_tokens")
            _lib.check(self.lib.wx_finalize(self.handle), "wx_finalize")
        except Exception:
            self.close()
            raise

    def replica(self, max_batch: Optional[int] = None, n_lanes: Optional[int] = None) -> "WhisperEngine":
        """Another engine on this engine's device weights (own workspace, caches, decode graph): what a lane of
        whisper_amd/lanes.py runs on.  The weights stay resident until the last engine that uses them is closed."""
        return WhisperEngine(self.dims, None, precision=self.precision, max_batch=self.max_batch if max_batch is None else max_batch,
                             device=self.device, n_lanes=self.n_lanes if n_lanes is None else n_lanes, share_weights_with=self)

    def set_suppress_ids(self, ids: Sequence[int]) -> None:
        self.suppress_ids = [int(i) for i in ids]
        sup = np.asarray(self.suppress_ids, np.int32)
        _lib.check(self.lib.wx_set_suppress_tokens(self.handle, sup.ctypes.data_as(ctypes.c_void_p), len(sup)), "wx_set_suppress_tokens")

    # -- lifetime ---------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.wx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(torch.cuda.current_stream(self.torch_device).cuda_stream)

    @property
    def weight_bytes(self) -> int:
        return int(self.lib.wx_weight_bytes(self.handle))

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.wx_workspace_bytes(self.handle))

    @property
    def cross_form(self) -> str:
        """"kv": decoder cross-attention on cached per-layer K/V; "encoder_output": on the encoder output itself (bf16 engines
        with max_batch >= 32; include/whisper_amd.h wx_cross_attention_form)."""
        return "encoder_output" if int(self.lib.wx_cross_attention_form(self.handle)) == 1 else "kv"

    @property
    def cross_key_splits(self) -> int:
        return int(self.lib.wx_cross_key_splits(self.handle))

    # -- audio ingest ---------------------------------------------------------------------------------------
    def pcm16_to_f32(self, pcm16) -> torch.Tensor:
        """int16 PCM (numpy array or tensor) -> float32 samples in [-1, 1) on the device (x / 32768, converted by the
        HIP kernel: the int16 samples cross PCIe, not float32)."""
        t = torch.from_numpy(np.array(pcm16, copy=True)) if not torch.is_tensor(pcm16) else pcm16
        if t.dtype != torch.int16:
            raise ValueError("pcm16_to_f32 expects int16 samples")
        t = t.to(self.torch_device).contiguous()
        out = torch.empty(t.numel(), dtype=torch.float32, device=self.torch_device)
        _lib.check(self.lib.wx_pcm16_to_f32(self.handle, _ptr(t), t.numel(), _ptr(out), self._stream()), "wx_pcm16_to_f32")
        return out

    def ingest_pcm16(self, pcm16, src_rate: int, quantize: bool = True) -> torch.Tensor:
        """Interleaved int16 PCM ``[n_frames]`` or ``[n_frames, n_channels]`` at ``src_rate`` -> mono float32 at 16 kHz on
        the device: channel average + polyphase resampling in one HIP kernel (``wx_resample_pcm16``), the reference's
        ffmpeg normalisation (ref:app/audio/processor.py:136-147) without the subprocess.  ``quantize`` rounds to the
        int16 grid like the intermediate s16le WAV of the reference does."""
        from .audio import resample_filter, resample_out_len, resample_ratio
        t = torch.from_numpy(np.array(pcm16, copy=True)) if not torch.is_tensor(pcm16) else pcm16
        if t.dtype != torch.int16 or t.dim() not in (1, 2):
            raise ValueError("ingest_pcm16 expects int16 samples, [n_frames] or [n_frames, n_channels]")
        n_frames, ch = int(t.shape[0]), (int(t.shape[1]) if t.dim() == 2 else 1)
        if n_frames == 0:
            return torch.empty(0, dtype=torch.float32, device=self.torch_device)
        up, down = resample_ratio(int(src_rate))
        if up == 1 and down == 1 and ch == 1:
            return self.pcm16_to_f32(t.reshape(-1))
        taps, half_len = resample_filter(up, down)
        n_out = resample_out_len(n_frames, up, down)
        t = t.to(self.torch_device).contiguous()
        out = torch.empty(n_out, dtype=torch.float32, device=self.torch_device)
        _lib.check(self.lib.wx_resample_pcm16(self.handle, _ptr(t), n_frames, ch, up, down, taps.ctypes.data, half_len,
                                              1 if quantize else 0, _ptr(out), n_out, self._stream()), "wx_resample_pcm16")
        return out

    # -- mel ----------------------------------------------------------------------------------------------
    def mel_stage1(self, pcm: torch.Tensor, pcm_offset: int, n_file: int, n_total: int, frame0: int, n_frames: int,
                   out: torch.Tensor, gmax: torch.Tensor) -> None:
        """log10(max(mel, 1e-10)) for file frames [frame0, frame0 + n_frames) into ``out[:, :n_frames]`` and fold the
        slice maximum into ``gmax`` (1-element fp32 tensor, reset beforehand)."""
        assert pcm.dtype == torch.float32 and pcm.is_cuda and pcm.is_contiguous() and out.dtype == torch.float32
        _lib.check(self.lib.wx_mel_stage1(self.handle, _ptr(pcm), pcm_offset, pcm.numel(), n_file, n_total, frame0, n_frames,
                                          _ptr(out), out.stride(0), _ptr(gmax), self._stream()), "wx_mel_stage1")

    def log_mel(self, pcm: torch.Tensor, padding: int = N_SAMPLES, all_reduce_max=None) -> torch.Tensor:
        """Whole-buffer log-mel ``[n_mels, (len + padding) // 160]`` (global clamp).  ``all_reduce_max`` is an optional
        callable applied to the 1-element max tensor between the stages (multi-GPU: all_reduce MAX)."""
        pcm = pcm.to(self.torch_device, torch.float32).contiguous()
        n_file = pcm.numel()
        n_total = n_file + padding
        n_frames = n_total // HOP_LENGTH
        out = torch.empty((self.dims.n_mels, n_frames), dtype=torch.float32, device=self.torch_device)
        gmax = torch.empty(1, dtype=torch.float32, device=self.torch_device)
        _lib.check(self.lib.wx_mel_reset_max(self.handle, _ptr(gmax), self._stream()), "wx_mel_reset_max")
        self.mel_stage1(pcm, 0, n_file, n_total, 0, n_frames, out, gmax)
        if all_reduce_max is not None:
            all_reduce_max(gmax)
        self.mel_stage2(out, n_frames, gmax)
        return out

    def log_mel_static(self, pcm: torch.Tensor, out: torch.Tensor, padding: int = N_SAMPLES) -> None:
        """``log_mel`` through the allocation-free entry point (``wx_log_mel_static``): the caller provides the device input
        and the ``[n_mels, >= (len + padding) // 160]`` output; nothing is allocated, no workspace is passed."""
        assert pcm.dtype == torch.float32 and pcm.is_cuda and pcm.is_contiguous() and out.dtype == torch.float32 and out.is_cuda
        _lib.check(self.lib.wx_log_mel_static(self.handle, _ptr(pcm), pcm.numel(), int(padding), _ptr(out), out.stride(0),
                                              self._stream()), "wx_log_mel_static")

    def mel_stage2(self, mel: torch.Tensor, n_frames: int, gmax: torch.Tensor) -> None:
        _lib.check(self.lib.wx_mel_stage2(self.handle, _ptr(mel), mel.stride(0), n_frames, _ptr(gmax), self._stream()),
                   "wx_mel_stage2")

    def reset_max(self, gmax: torch.Tensor) -> None:
        _lib.check(self.lib.wx_mel_reset_max(self.handle, _ptr(gmax), self._stream()), "wx_mel_reset_max")

    # -- encoder ------------------------------------------------------------------------------------------
    def encode(self, mel: torch.Tensor, win_stride: int, row_stride: int, n_valid_frames: int, B: int) -> None:
        assert mel.dtype == torch.float32 and mel.is_cuda
        gate = self.encode_gate
        if gate is None:
            _lib.check(self.lib.wx_encode(self.handle, _ptr(mel), win_stride, row_stride, n_valid_frames, B, self._stream()),
                       "wx_encode")
            return
        # one encoder at a time among the lanes of a GPU (whisper_amd/lanes.py): held until its kernels have finished
        with gate:
            _lib.check(self.lib.wx_encode(self.handle, _ptr(mel), win_stride, row_stride, n_valid_frames, B, self._stream()),
                       "wx_encode")
            torch.cuda.current_stream(self.torch_device).synchronize()

    def encode_windows(self, mel_windows: torch.Tensor) -> None:
        """``mel_windows [B, n_mels, 3000]`` fp32 on device."""
        mel_windows = mel_windows.to(self.torch_device, torch.float32).contiguous()
        B = mel_windows.shape[0]
        assert mel_windows.shape[1:] == (self.dims.n_mels, N_FRAMES)
        self.encode(mel_windows, self.dims.n_mels * N_FRAMES, N_FRAMES, B * N_FRAMES, B)
        self._keep = mel_windows

    def encoder_output(self, B: int) -> torch.Tensor:
        out = torch.empty((B, self.dims.n_audio_ctx, self.dims.n_audio_state), dtype=torch.float32, device=self.torch_device)
        _lib.check(self.lib.wx_get_encoder_output(self.handle, B, _ptr(out), self._stream()), "wx_get_encoder_output")
        return out

    def reserve_cross_kv(self) -> None:
        """Allocate the per-layer cross K/V cache now instead of inside the first beam-search call (bf16 engines with
        max_batch >= 32 do not own one otherwise; 15.7 GB at large-v3 x 64 rows): ``wx_reserve_cross_kv``."""
        _lib.check(self.lib.wx_reserve_cross_kv(self.handle), "wx_reserve_cross_kv")

    def compact_windows(self, keep: Sequence[int]) -> None:
        """Keep the windows ``keep`` (strictly ascending slots of the last encode) as slots 0..len(keep)-1, so that the next
        decode call runs on them alone: upstream's temperature fallback re-decodes only the windows that failed its checks."""
        src = np.asarray([int(k) for k in keep], np.int32)
        _lib.check(self.lib.wx_compact_windows(self.handle, src.ctypes.data_as(ctypes.c_void_p), len(src), self._stream()), "wx_compact_windows")

    # -- decoder ------------------------------------------------------------------------------------------
    def decode_greedy(self, B: int, prompt: Sequence[int], max_new_tokens: int = 0, without_timestamps: bool = True,
                      suppress_blank: bool = True, max_initial_timestamp_index: int = 50, use_graph: bool = True,
                      check_interval: int = 8, profile_stride: int = 0) -> Dict[str, object]:
        prompt = [int(p) for p in prompt]
        opts = _lib.WxDecodeOpts(
            n_prompt=len(prompt), sot_index=prompt.index(self.tok.sot), max_new_tokens=int(max_new_tokens),
            without_timestamps=int(without_timestamps), suppress_blank=int(suppress_blank),
            max_initial_timestamp_index=int(max_initial_timestamp_index), use_graph=int(use_graph),
            check_interval=int(check_interval), profile_stride=int(profile_stride))
        dev = self.torch_device
        tokens = torch.empty((B, self.dims.n_text_ctx + 1), dtype=torch.int32, device=dev)    # ids per row: n_text_ctx + 1
        n_tokens = torch.empty(B, dtype=torch.int32, device=dev)
        sum_lp = torch.empty(B, dtype=torch.float32, device=dev)
        nsp = torch.empty(B, dtype=torch.float32, device=dev)
        steps = ctypes.c_int32(0)
        h_prompt = (ctypes.c_int32 * len(prompt))(*prompt)
        _lib.check(self.lib.wx_decode_greedy(self.handle, B, h_prompt, ctypes.byref(opts), _ptr(tokens), _ptr(n_tokens),
                                             _ptr(sum_lp), _ptr(nsp), ctypes.byref(steps), self._stream()), "wx_decode_greedy")
        return {"tokens": tokens, "n_tokens": n_tokens, "sum_logprobs": sum_lp, "no_speech_probs": nsp,
                "steps": int(steps.value), "n_prompt": len(prompt)}

    def decode_rows(self, prompts: Sequence[Sequence[int]], temperature: float = 0.0, seed: int = 0, max_new_tokens: int = 0,
                    without_timestamps: bool = True, suppress_blank: bool = True, max_initial_timestamp_index: int = 50,
                    use_graph: bool = True, check_interval: int = 8) -> Dict[str, object]:
        """Sampling decode with one prompt per window of the last encode (``len(prompts)`` rows); ``temperature`` 0 is
        greedy.  Returns per-row token lists (after each row's own prompt, EOT excluded), sum_logprobs, no_speech_probs."""
        B, ctx = len(prompts), self.dims.n_text_ctx + 1                        # ids per row (include/whisper_amd.h)
        rows = np.full((B, ctx), self.tok.eot, np.int32)
        n_prompt = np.zeros(B, np.int32); sot = np.zeros(B, np.int32)
        for b, p in enumerate(prompts):
            p = [int(x) for x in p]
            rows[b, :len(p)] = p
            n_prompt[b] = len(p); sot[b] = p.index(self.tok.sot)
        opts = _lib.WxDecodeOpts(
            n_prompt=0, sot_index=0, max_new_tokens=int(max_new_tokens), without_timestamps=int(without_timestamps),
            suppress_blank=int(suppress_blank), max_initial_timestamp_index=int(max_initial_timestamp_index),
            use_graph=int(use_graph), check_interval=int(check_interval), profile_stride=0)
        dev = self.torch_device
        tokens = torch.empty((B, ctx), dtype=torch.int32, device=dev)
        n_tokens = torch.empty(B, dtype=torch.int32, device=dev)
        sum_lp = torch.empty(B, dtype=torch.float32, device=dev)
        nsp = torch.empty(B, dtype=torch.float32, device=dev)
        steps = ctypes.c_int32(0)
        _lib.check(self.lib.wx_decode_rows(self.handle, B, rows.ctypes.data_as(ctypes.c_void_p), n_prompt.ctypes.data_as(ctypes.c_void_p),
                                           sot.ctypes.data_as(ctypes.c_void_p), ctypes.byref(opts), ctypes.c_float(temperature),
                                           ctypes.c_uint32(seed & 0xFFFFFFFF), _ptr(tokens), _ptr(n_tokens), _ptr(sum_lp), _ptr(nsp),
                                           ctypes.byref(steps), self._stream()), "wx_decode_rows")
        t_cpu, n_cpu = tokens.cpu().numpy(), n_tokens.cpu().numpy()
        out = [t_cpu[b, n_prompt[b]:n_prompt[b] + int(n_cpu[b])].tolist() for b in range(B)]
        return {"tokens": out, "sum_logprobs": sum_lp.cpu().tolist(), "no_speech_probs": nsp.cpu().tolist(), "steps": int(steps.value)}

    def decode_beam(self, B: int, prompt: Sequence[int], beam_size: int = 5, patience: float = 1.0, max_new_tokens: int = 0,
                    without_timestamps: bool = True, suppress_blank: bool = True, max_initial_timestamp_index: int = 50,
                    use_graph: bool = True, check_interval: int = 8) -> Dict[str, object]:
        """Beam search for the B windows of the last encode (upstream ``DecodingTask.run`` with ``beam_size``).  The device
        runs the per-step update; ``finalize`` (short finished lists are topped up with the live beams, best first) and
        the ``sum_logprob / length`` ranking run here on the few candidates that come back."""
        prompt = [int(p) for p in prompt]
        opts = _lib.WxDecodeOpts(
            n_prompt=len(prompt), sot_index=prompt.index(self.tok.sot), max_new_tokens=int(max_new_tokens),
            without_timestamps=int(without_timestamps), suppress_blank=int(suppress_blank),
            max_initial_timestamp_index=int(max_initial_timestamp_index), use_graph=int(use_graph),
            check_interval=int(check_interval), profile_stride=0)
        dev, n, ctx = self.torch_device, int(beam_size), self.dims.n_text_ctx + 1   # ids per row
        maxc = int(round(n * patience))     # upstream's rule, computed once here: it sizes fin_t / fin_s and goes to the device
        fin_t = torch.empty((B, maxc, ctx), dtype=torch.int32, device=dev)
        fin_s = torch.empty((B, maxc), dtype=torch.float32, device=dev)
        fin_c = torch.empty(B, dtype=torch.int32, device=dev)
        beam_t = torch.empty((B * n, ctx), dtype=torch.int32, device=dev)
        beam_s = torch.empty(B * n, dtype=torch.float32, device=dev)
        nsp = torch.empty(B, dtype=torch.float32, device=dev)
        steps = ctypes.c_int32(0)
        h_prompt = (ctypes.c_int32 * len(prompt))(*prompt)
        _lib.check(self.lib.wx_decode_beam(self.handle, B, n, maxc, h_prompt, ctypes.byref(opts), _ptr(fin_t),
                                           _ptr(fin_s), _ptr(fin_c), _ptr(beam_t), _ptr(beam_s), _ptr(nsp), ctypes.byref(steps),
                                           self._stream()), "wx_decode_beam")
        fin_t, fin_s, fin_c = fin_t.cpu().numpy(), fin_s.cpu().numpy(), fin_c.cpu().numpy()
        beam_t, beam_s = beam_t.cpu().numpy().reshape(B, n, ctx), beam_s.cpu().numpy().reshape(B, n)
        eot, n_prompt = self.tok.eot, len(prompt)

        def body(row):
            seq = row[n_prompt:].tolist()
            return seq[:seq.index(eot)] if eot in seq else seq
        live_len = n_prompt + int(steps.value)          # tokens every live beam holds (prompt + one per sampling step)
        tokens, sums, cands = [], [], []
        for b in range(B):
            seqs: Dict[tuple, float] = {}
            for k in range(int(fin_c[b])):
                seqs[tuple(body(fin_t[b, k]))] = float(fin_s[b, k])
            if len(seqs) < n:                             # finalize(): not enough finished sequences
                for j in list(np.argsort(beam_s[b]))[::-1]:
                    seq = beam_t[b, j, n_prompt:live_len].tolist()
                    seqs[tuple(seq[:seq.index(eot)] if eot in seq else seq)] = float(beam_s[b, j])
                    if len(seqs) >= n:
                        break
            items = list(seqs.items())
            best = int(np.argmax([sc / len(sq) if len(sq) else -np.inf for sq, sc in items]))
            tokens.append(list(items[best][0])); sums.append(items[best][1]); cands.append([(list(sq), sc) for sq, sc in items])
        return {"tokens": tokens, "sum_logprobs": sums, "no_speech_probs": nsp.cpu().tolist(), "steps": int(steps.value),
                "n_prompt": n_prompt, "candidates": cands}

    def last_profile(self):
        """(total_ms, launches) of the cross-attention launches bracketed by the last profiled eager decode."""
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        _lib.check(self.lib.wx_get_profile(self.handle, ctypes.byref(ms), ctypes.byref(n)), "wx_get_profile")
        return float(ms.value), int(n.value)

    def decoder_logits(self, tokens: torch.Tensor) -> torch.Tensor:
        tokens = tokens.to(self.torch_device, torch.int32).contiguous()
        B, n = tokens.shape
        out = torch.empty((B, n, self.dims.n_vocab), dtype=torch.float32, device=self.torch_device)
        _lib.check(self.lib.wx_decoder_logits(self.handle, B, _ptr(tokens), n, _ptr(out), self._stream()), "wx_decoder_logits")
        return out

    def detect_language(self, mel: torch.Tensor, n_content_frames: int) -> int:
        """Upstream ``detect_language`` on the first 30 s segment of a (whole-file, globally clamped) mel ``[n_mels, ld]``:
        one decoder step on ``[sot]``, argmax over the language tokens (hf:models/whisper/generation_whisper.py:1610-1673).
        English-only checkpoints have no language tokens to choose from: upstream's ``transcribe`` fixes "en" for them."""
        if not is_multilingual(self.dims.n_vocab):
            return 0
        self.encode(mel, N_FRAMES, mel.stride(0), n_content_frames, 1)
        tok = self.tok
        logits = self.decoder_logits(torch.tensor([[tok.sot]], dtype=torch.int32))[0, 0]
        return int(torch.argmax(logits[tok.sot + 1: tok.sot + 1 + tok.n_langs]).item())

    # -- whole path ---------------------------------------------------------------------------------------
    def initial_tokens(self, language_index: int = 0, task: str = "transcribe", without_timestamps: bool = True,
                       prompt_ids: Optional[Sequence[int]] = None) -> List[int]:
        t = self.tok
        if is_multilingual(self.dims.n_vocab):
            seq = [t.sot, t.lang(language_index), t.transcribe if task == "transcribe" else t.translate]
        else:
            seq = [t.sot]           # English-only checkpoints: upstream's sot_sequence carries no language / task token
        if without_timestamps:
            seq.append(t.no_timestamps)
        if prompt_ids:
            seq = [t.sot_prev] + list(prompt_ids)[-(self.dims.n_text_ctx // 2 - 1):] + seq
        return seq

    def transcribe_mel(self, mel: torch.Tensor, n_content_frames: int, window_lo: int, window_hi: int,
                       prompt: Sequence[int], beam_size: Optional[int] = None, patience: float = 1.0, **decode_kw) -> Dict[str, object]:
        """Decode windows [window_lo, window_hi) of a whole-file mel ``[n_mels, ld]`` in batches of ``max_batch`` windows
        (greedy) or ``max_batch // beam_size`` windows (beam search: rows = windows x beams)."""
        toks: List[List[int]] = []
        sums: List[float] = []
        nsp: List[float] = []
        steps = 0
        ld = mel.stride(0)
        per = self.max_batch if not beam_size else max(1, self.max_batch // int(beam_size))
        if beam_size and int(beam_size) > self.max_batch:
            raise ValueError("beam_size exceeds max_batch")
        for w0 in range(window_lo, window_hi, per):
            B = min(per, window_hi - w0)
            view = mel[:, w0 * N_FRAMES:]
            self.encode(view, N_FRAMES, ld, n_content_frames - w0 * N_FRAMES, B)
            if beam_size:
                r = self.decode_beam(B, prompt, beam_size=int(beam_size), patience=patience, **decode_kw)
                toks += r["tokens"]; sums += r["sum_logprobs"]; nsp += r["no_speech_probs"]
            else:
                r = self.decode_greedy(B, prompt, **decode_kw)
                t_cpu, n_cpu = r["tokens"].cpu(), r["n_tokens"].cpu()
                for b in range(B):
                    toks.append(t_cpu[b, r["n_prompt"]:r["n_prompt"] + int(n_cpu[b])].tolist())
                sums += r["sum_logprobs"].cpu().tolist()
                nsp += r["no_speech_probs"].cpu().tolist()
            steps += r["steps"]
        return {"tokens": toks, "sum_logprobs": sums, "no_speech_probs": nsp, "steps": steps}

    def transcribe_pcm(self, audio, prompt: Sequence[int], **decode_kw) -> Dict[str, object]:
        """Single-process long-form path: whole-file log-mel (+30 s zero padding, global clamp), independent 30 s
        windows, greedy decode.  ``audio``: float32 mono 16 kHz array or tensor."""
        pcm = torch.as_tensor(np.asarray(audio, dtype=np.float32) if not torch.is_tensor(audio) else audio)
        pcm = pcm.to(self.torch_device, torch.float32).contiguous()
        mel = self.log_mel(pcm)
        content = pcm.numel() // HOP_LENGTH
        nw = n_windows(pcm.numel())
        out = self.transcribe_mel(mel, content, 0, nw, prompt, **decode_kw)
        out["n_windows"] = nw
        return out
