/*
 * whisper_amd.h — C-ABI of the MI355X-native local-Whisper engine (libwhisper_amd.so).
 *
 * The reference (alexk136/whisper) has NO native interface for this path: its boundary is the Python call
 *   whisper_model = whisper.load_model(WHISPER_MODEL, device=device)   ref:app/transcription/speech_recognition.py:22
 *   result = whisper_model.transcribe(str(audio_path), **options)      ref:app/transcription/speech_recognition.py:55
 * into the third-party openai-whisper package.  The entry points below are what a binding for that path binds
 * instead (ctypes stub: INTEGRATION.md §2); each one names the reference call it replaces.  All functions
 * return 0 on success and a negative wx_status otherwise, never throw, and record a message retrievable with
 * wx_last_error().  Pointers named d_* are DEVICE pointers (HBM, on the engine's device); h_* are host
 * pointers.  `stream` is a hipStream_t passed as void* (NULL = the default stream).  The engine owns weights,
 * workspace and KV caches; the caller owns every buffer it passes in.  Calls on one engine must not overlap, with one exception: wx_mel_reset_max,
 * wx_mel_stage1, wx_mel_stage2, wx_pcm16_to_f32 and wx_resample_pcm16 only read the engine's tables and work on the caller's
 * buffers and stream, so they may run (from another host thread, on another stream) while the engine encodes or decodes.
 * Engines are independent of one another: several of them, on one GPU or on several, may be driven from different host threads
 * at the same time (whisper_amd/lanes.py keeps several batches in flight per GPU that way, on one copy of the weights:
 * wx_create_shared).
 */
#ifndef WHISPER_AMD_H
#define WHISPER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wx_engine wx_engine;

typedef enum {
    WX_OK = 0,
    WX_ERR_INVALID = -1,   /* bad argument / shape / state */
    WX_ERR_HIP = -2,       /* a HIP runtime call failed */
    WX_ERR_NOMEM = -3,     /* device allocation failed */
    WX_ERR_STATE = -4      /* call order violated (e.g. weights not finalized) */
} wx_status;

enum { WX_PREC_F32 = 0, WX_PREC_BF16 = 1 };

/* Model + runtime configuration.  Dimensions are openai-whisper's ModelDimensions (what
 * whisper.load_model reads from the checkpoint's "dims", ref:speech_recognition.py:22). */
typedef struct {
    int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
    int32_t precision;      /* WX_PREC_F32: every tensor and product in fp32 (token-exact mode);
                               WX_PREC_BF16: bf16 weights/activations on MFMA, fp32 accumulate + fp32 residual */
    int32_t max_batch;      /* max 30 s windows per wx_encode / wx_decode call (sizes workspace + caches) */
    int32_t device;         /* HIP device ordinal */
    /* special tokens (multilingual layout) */
    int32_t tok_eot, tok_sot, tok_translate, tok_transcribe, tok_sot_prev, tok_sot_lm;
    int32_t tok_no_speech, tok_no_timestamps, tok_timestamp_begin, tok_blank;
    int32_t n_lanes;        /* engines that will share this GPU at the same time (whisper_amd/lanes.py keeps several batches in flight);
                               0 or 1 = this engine has the GPU to itself.  Sizes the grid of the decoder's cross-attention stream
                               (key splits per row = 256 CUs / (n_lanes x rows per pass), rounded down alone and up when sharing, 1 to 8): a launch that fills every CU
                               finishes one batch sooner, one that leaves CUs to the other lanes finishes all of them sooner.  A
                               property of the engine like max_batch: results do not depend on the batch at hand. */
} wx_config;

/* Decode options: the subset of openai-whisper DecodingOptions the reference's call can reach
 * (ref:speech_recognition.py:48-52 passes language/task/fp16 only, so upstream defaults apply). */
typedef struct {
    int32_t n_prompt;            /* length of the initial token sequence [.., sot, lang, task, (no_timestamps)] */
    int32_t sot_index;           /* index of <|startoftranscript|> inside the prompt */
    int32_t max_new_tokens;      /* upstream sample_len = n_text_ctx / 2 when <= 0 */
    int32_t without_timestamps;  /* 1: no timestamp rules (prompt ends with <|notimestamps|>) */
    int32_t suppress_blank;      /* 1 = upstream default */
    int32_t max_initial_timestamp_index; /* upstream default 50 (1.0 s / 0.02 s); < 0 disables */
    int32_t use_graph;           /* 1: replay one captured hipGraph per decode step; 0: eager launches */
    int32_t check_interval;      /* host polls the device-side finished counter every this many steps (>=1) */
    int32_t profile_stride;      /* > 0 (eager mode only): bracket every cross-attention launch of every
                                    profile_stride-th sampling step with HIP events on the launch stream */
} wx_decode_opts;

const char* wx_last_error(void);
const char* wx_version(void);

/* replaces whisper.load_model (ref:speech_recognition.py:22): create, upload tensors by canonical
 * openai name (fp32 host data; converted to the engine precision on device), finalize. */
int wx_create(const wx_config* cfg, wx_engine** out);
void wx_destroy(wx_engine* e);
int wx_set_weight(wx_engine* e, const char* name, const float* h_data, const int64_t* shape, int32_t ndim);
int wx_set_mel_filters(wx_engine* e, const float* h_filters /* [n_mels][201] */);
/* the ids DecodingTask._get_suppress_tokens resolves (upstream suppress_tokens="-1": the vocabulary's non-speech symbols
 * + the task / control tokens).  May be called again after wx_finalize: the device mask is replaced once the engine's
 * device is idle (blocking); it must not be called while a decode call on this engine is running. */
int wx_set_suppress_tokens(wx_engine* e, const int32_t* h_ids, int32_t n);
int wx_finalize(wx_engine* e);

/* a second engine on the SAME weights (still whisper.load_model, ref:speech_recognition.py:22: the reference holds ONE model object
 * however many calls use it).  `parent` must be finalized; cfg's model dimensions, precision and device must equal the parent's,
 * max_batch and n_lanes are the new engine's own (but must select the same cross-attention form, wx_cross_attention_form).  The
 * new engine comes back finalized: it owns workspace, caches, decode state and graph, shares the parent's weight allocations
 * (they are freed with the LAST engine that uses them, whatever the order of the wx_destroy calls), starts from the parent's
 * suppress list (wx_set_suppress_tokens changes its own copy only) and reports wx_weight_bytes() == 0.  This is how
 * whisper_amd/lanes.py keeps several batches in flight per GPU on one copy of the weights. */
int wx_create_shared(const wx_engine* parent, const wx_config* cfg, wx_engine** out);

/* replaces the tail of whisper.load_audio (inside model.transcribe, ref:speech_recognition.py:55): the reference's audio
 * front end hands over 16 kHz mono s16le PCM (ref:app/audio/processor.py:141-143) and upstream divides by 32768 on the
 * host; here the int16 samples are uploaded as they are (half the PCIe bytes) and converted on the device.
 * d_pcm16 and d_out must be 16-byte aligned. */
int wx_pcm16_to_f32(wx_engine* e, const int16_t* d_pcm16, int64_t n, float* d_out, void* stream);

/* replaces the reference's ffmpeg normalisation step (ref:app/audio/processor.py:136-147: `acodec=pcm_s16le, ar=16000,
 * ac=1`) for PCM16 input of any rate and channel count: channel average, then a polyphase windowed-sinc resampler
 * y[n] = sum_m x[m] * h[half_len + n*down - m*up]  (up/down = 16000/src_rate reduced; x is 0 outside the file), i.e.
 * scipy.signal.resample_poly's arithmetic with the caller's centred taps h_taps[2*half_len + 1] (host, already scaled by
 * `up`).  d_pcm16: interleaved [n_frames][n_channels] int16 on the device.  d_out: n_out = ceil(n_frames*up/down) mono
 * float32 samples in [-1, 1).  quantize != 0 rounds every output to the int16 grid (round-half-even, clipped), as the
 * reference's intermediate s16le WAV does.  ffmpeg's own filter differs (and is absent here): parity with ffmpeg is
 * unpinned; the oracle is a numpy restatement pinned against scipy.signal.resample_poly.
 * Limits: 1 <= n_channels <= 8, down <= 8*up, 2*half_len + 1 <= 65536. */
int wx_resample_pcm16(wx_engine* e, const int16_t* d_pcm16, int64_t n_frames, int32_t n_channels, int32_t up, int32_t down,
                      const float* h_taps, int32_t half_len, int32_t quantize, float* d_out, int64_t n_out, void* stream);

/* replaces whisper.log_mel_spectrogram inside Whisper.transcribe (ref:speech_recognition.py:55).
 * Stage 1: frames [frame0, frame0+n_frames) of the file -> d_logmel[m * ld + (f - frame0)] =
 *          log10(max(mel, 1e-10)), and atomically folds the slice maximum into *d_max (init with wx_mel_reset_max).
 *          d_pcm holds n_valid samples starting at file sample pcm_offset (halo for a sharded slice); n_total =
 *          file length + zero padding: samples in [n_file, n_total) read as 0, reflect padding applies at 0 and n_total.
 * Stage 2: in place max(x, *d_max - 8), (x + 4) / 4 over n_mels x n_frames.
 * Between the stages a multi-GPU caller all-reduces *d_max with MAX (SURVEY.md §8e caveat 1). */
int wx_mel_reset_max(wx_engine* e, float* d_max, void* stream);
int wx_mel_stage1(wx_engine* e, const float* d_pcm, int64_t pcm_offset, int64_t n_valid, int64_t n_file,
                  int64_t n_total, int64_t frame0, int64_t n_frames, float* d_logmel, int64_t ld,
                  float* d_max, void* stream);
int wx_mel_stage2(wx_engine* e, float* d_logmel, int64_t ld, int64_t n_frames, const float* d_max, void* stream);

/* NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate device memory or hand over a workspace: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  Whole-buffer log-mel (what wx_mel_reset_max +
 * wx_mel_stage1 + wx_mel_stage2 compute for frame0 = 0, pcm_offset = 0: (n_samples + padding) / 160 frames, clamp against
 * the buffer's own maximum) with NO workspace argument: the maximum and the per-workgroup partial maxima live in a 256 KiB
 * file-scope device array of the library (one copy per GPU, held until the process ends).  The call allocates nothing --
 * no hipMalloc / hipFree, no host containers -- and launches three kernels on `stream`.  Limits: at most
 * (65536 - 64) * 8 = 523776 frames (87 min of audio including padding), rejected with WX_ERR_INVALID beyond that.  Every
 * call on one GPU shares that GPU's array: calls that overlap on one GPU (two streams, two engines on the same device)
 * race; serialise them.  Bit-identical to the three-call form (max is order-independent). */
int wx_log_mel_static(wx_engine* e, const float* d_pcm, int64_t n_samples, int64_t padding, float* d_logmel, int64_t ld,
                      void* stream);

/* replaces AudioEncoder.forward + the per-layer cross-attention K/V projection (inside model.transcribe,
 * ref:speech_recognition.py:55).  d_mel: fp32 features, element (w, m, t) at d_mel[w*win_stride + m*row_stride + t]
 * (contiguous windows: win_stride = n_mels*3000, row_stride = 3000; a whole-file mel [n_mels][ld]: win_stride =
 * 3000, row_stride = ld).  Frames with w*3000 + t >= n_valid_frames read as 0 (upstream zero-pads the last
 * segment).  Results stay in the engine
 * (encoder output + cross K/V cache for slots 0..B-1).  wx_get_encoder_output copies [B][n_audio_ctx][d] fp32. */
int wx_encode(wx_engine* e, const float* d_mel, int64_t win_stride, int64_t row_stride, int64_t n_valid_frames,
              int32_t B, void* stream);
int wx_get_encoder_output(wx_engine* e, int32_t B, float* d_out, void* stream);

/* bf16 engines with max_batch >= 32 run greedy / sampling decodes on the encoder output and own NO per-layer cross K/V cache;
 * beam search at width > 1 needs one and allocates it inside its first call (2 x max_batch x n_text_layer x 1500 x d x 2 bytes:
 * 15.7 GB at large-v3 x 64 rows), where an out-of-memory failure would surface mid-request and the allocation stalls other
 * engines on the device.  Call this once after wx_finalize / wx_create_shared when beam search is expected (upstream
 * DecodingOptions.beam_size; the reference's own call never sets it, ref:speech_recognition.py:48-52): the cache is allocated
 * now (WX_ERR_NOMEM now), never freed before wx_destroy, and projected lazily as before.  A no-op on engines that own the
 * cache already. */
int wx_reserve_cross_kv(wx_engine* e);

/* serves upstream's decode_with_fallback (Whisper.transcribe inside ref:speech_recognition.py:55 re-decodes ONLY the windows whose
 * compression-ratio / log-probability checks failed): keep n windows of the last wx_encode -- slot i takes the encoder results of
 * slot h_src[i]; h_src strictly ascending, i <= h_src[i] < windows of the last wx_encode -- so that the next decode call runs
 * on those windows alone (rows 0..n-1).  Device-to-device copies on `stream`; afterwards the engine holds n windows. */
int wx_compact_windows(wx_engine* e, const int32_t* h_src, int32_t n, void* stream);

/* replaces DecodingTask.run greedy (temperature 0) for the B windows of the last wx_encode.
 * h_prompt: n_prompt ids shared by all windows.  Outputs (device): d_tokens [B][n_text_ctx + 1] int32 (prompt +
 * sampled ids, EOT-padded after the end), d_n_tokens [B] (count of sampled ids before EOT), d_sum_logprob [B],
 * d_no_speech_prob [B].  *h_steps (optional) receives the number of sampling steps executed.  Token rows hold
 * n_text_ctx + 1 ids because upstream's loop still samples from the last position (it stops once a row is LONGER than
 * n_text_ctx). */
int wx_decode_greedy(wx_engine* e, int32_t B, const int32_t* h_prompt, const wx_decode_opts* opts,
                     int32_t* d_tokens, int32_t* d_n_tokens, float* d_sum_logprob, float* d_no_speech_prob,
                     int32_t* h_steps, void* stream);

/* wx_decode_greedy with a prompt PER WINDOW and an optional temperature: what upstream's sequential long-form loop needs
 * (Whisper.transcribe inside ref:speech_recognition.py:55: every window is conditioned on the text decoded before it,
 * "condition_on_previous_text", and re-decoded at temperatures 0.2 ... 1.0 when the compression-ratio / log-probability
 * checks fail, "decode_with_fallback").  h_prompts [B][n_text_ctx + 1] holds each row's initial ids (h_n_prompt[b] of them,
 * <|startoftranscript|> at h_sot_index[b]); opts->n_prompt / sot_index are ignored.  Rows run in lockstep over positions,
 * each starts sampling after its own prompt and stops at EOT, after max_new_tokens ids or at n_text_ctx.  temperature > 0
 * samples from softmax(logits / T) with a counter-based generator seeded by `seed` (upstream draws from torch's global
 * generator, so sampled ids agree with it in distribution only); log-probabilities are those of the unscaled logits, as
 * upstream.  Outputs as wx_decode_greedy; d_n_tokens counts ids after each row's own prompt. */
int wx_decode_rows(wx_engine* e, int32_t B, const int32_t* h_prompts, const int32_t* h_n_prompt, const int32_t* h_sot_index,
                   const wx_decode_opts* opts, float temperature, uint32_t seed, int32_t* d_tokens, int32_t* d_n_tokens,
                   float* d_sum_logprob, float* d_no_speech_prob, int32_t* h_steps, void* stream);

/* replaces DecodingTask.run with BeamSearchDecoder (upstream beam_size / patience options; the reference itself never
 * sets them -- ref:speech_recognition.py:48-52 -- but BASELINE.json configs[3] does) for the B windows of the last
 * wx_encode.  Rows = B x beam_size must fit max_batch.  The device runs upstream's update() every step: per window the
 * beam_size+1 best continuations of every beam, visited best first, EOT candidates appended to the window's finished
 * list while it holds fewer than max_candidates entries, the first beam_size others kept as beams; decoding stops when
 * every window's list is full (or at sample_len).  max_candidates is upstream's round(beam_size * patience), computed by
 * the CALLER (Python's round: double arithmetic, ties to even) because it also sizes the caller's buffers below; it must
 * lie in [beam_size, 2 * beam_size] (patience 1 ... 2).  The self-attention cache follows the
 * beams through index rows (no K/V copy); the beams of a window share one pass over its cross-attention K/V.
 * Outputs (device): d_fin_tokens [B][max_candidates][n_text_ctx + 1] (prompt + ids + EOT padding, in finished order),
 * d_fin_sum_logprob [B][max_candidates], d_fin_count [B], and the live beams d_beam_tokens [B*beam_size][n_text_ctx + 1],
 * d_beam_sum_logprob [B*beam_size] -- upstream finalize() (top up short lists from the beams) and the
 * sum_logprob / length ranking are a few dozen host operations and are left to the caller (whisper_amd/engine.py). */
int wx_decode_beam(wx_engine* e, int32_t B, int32_t beam_size, int32_t max_candidates, const int32_t* h_prompt,
                   const wx_decode_opts* opts, int32_t* d_fin_tokens, float* d_fin_sum_logprob, int32_t* d_fin_count,
                   int32_t* d_beam_tokens, float* d_beam_sum_logprob, float* d_no_speech_prob, int32_t* h_steps,
                   void* stream);

/* replaces TextDecoder.forward (teacher forced): logits of every position of d_tokens [B][n] against the
 * last wx_encode, written to d_logits [B][n][n_vocab] fp32.  Parity/debug entry point. */
int wx_decoder_logits(wx_engine* e, int32_t B, const int32_t* d_tokens, int32_t n, float* d_logits, void* stream);

/* result of the last profiled wx_decode_greedy: summed HIP-event time of the bracketed cross-attention launches */
int wx_get_profile(wx_engine* e, double* total_ms, int64_t* launches);

/* introspection for benchmarks.  wx_weight_bytes counts what wx_finalize left resident: the weights in the engine dtype and,
 * on bf16 engines, a second copy of the decoder projections and the token embedding in the load order of the decode GEMMs
 * (about half as much again); 0 for an engine made by wx_create_shared (it owns none). */
int64_t wx_weight_bytes(wx_engine* e);
int64_t wx_workspace_bytes(wx_engine* e);
/* which form of the decoder cross-attention this engine's greedy / sampling decode runs: 0 = on cached per-layer K/V (upstream's
 * form: openai-whisper MultiHeadAttention.forward with kv_cache), 1 = on the encoder output itself with the key / value
 * projections moved to the query side (bf16 engines with max_batch >= 32; same attention, different bf16 rounding points:
 * DESIGN.md section 4).  Fixed at wx_create; the parity tests pick the matching oracle form with it. */
int32_t wx_cross_attention_form(wx_engine* e);
/* key splits per row of that cross-attention's stream launch (workgroups per launch = rows x this); fixed at wx_finalize from
 * max_batch and n_lanes only */
int32_t wx_cross_key_splits(wx_engine* e);

/* Streams for batches in flight (whisper_amd/lanes.py; no counterpart in the reference, whose model serves one call at a time).
 * Two streams that the HIP runtime maps onto ONE hardware queue run one after the other: with four lanes that costs a third of
 * the throughput (profiles/r03_cu_share_and_hw_queues.txt, sections 2-3), and the mapping is the runtime's (GPU_MAX_HW_QUEUES, the
 * streams made before).  wx_stream_spin launches one wave that idles for `microseconds` on `stream` (the current device), so a caller
 * can time two streams against each other; wx_stream_create_exclusive makes a stream with a hardware queue of its own (a compute-unit
 * mask naming every compute unit of `device`: such streams never share a queue); wx_stream_destroy destroys it. */
int wx_stream_spin(void* stream, int32_t microseconds);
int wx_stream_create_exclusive(int32_t device, void** out_stream);
int wx_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_AMD_H */
