"""Batches in flight: N replicas of one engine ("lanes") on one GPU, each driven by its own host thread on its own HIP stream.

Why.  One batch of 30 s windows keeps the GPU busy only in patches: the decode loop is a chain of short, dependent launches (a
dozen per decoder layer and token, 4-12 us each, most of them waiting on one memory round trip) around one bandwidth-bound
stream per layer, and the encoder runs alone in front of it.  A second, independent batch fills the holes -- its short launches
run beside the other batch's stream, its encoder beside the other's decode loop.  Measured on one MI355X, large-v3 bf16, 64
windows per batch, 224 tokens: 1053 ms per batch with 1 lane, 816 with 2, 742 with 3, 705 with 4 (1823 -> 2723 audio-s/s; each
engine sizes its cross-attention grid for its share of the GPU, wx_config.n_lanes), no further gain at 6 or 8 (the runtime maps
streams onto 4 hardware queues).  Every lane runs the
same engine configuration (same kernels, same max_batch, same n_lanes and therefore the same key split of the cross-attention
stream), so a window's ids do not depend on the lane it ran on (tests/test_gpu_api.py::test_lanes_decode_the_same_ids).  The
number of lanes itself is part of that configuration: it sets the key split, and each split's partial context is rounded to bf16
before the splits are combined -- bf16 results of configure(lanes=4) differ from lanes=1 by rounding noise (measured 0.09 noise
floors on the logits at large-v3, tests/test_gpu_configs.py), fp32 results do not change at all.

What it costs: one more workspace per lane (large-v3 bf16, max_batch 64: 8 GB of 288).  The weights exist once: the lanes after
the first are made with ``WhisperEngine.replica()`` (``wx_create_shared``), which shares the first engine's device weights.

The reference has no counterpart: its model is one global object behind an asyncio lock-free call
(ref:app/transcription/speech_recognition.py:20-26,55), one file at a time.  Independent 30 s windows are what
makes batches, and therefore lanes, possible.
"""
from __future__ import annotations

import logging
import os
import queue
import threading
import time
from typing import Callable, Dict, List, Optional, Sequence

import torch


log = logging.getLogger(__name__)


class _Lane:
    def __init__(self, engine, index: int):
        self.engine = engine
        self.index = index
        self.exclusive_stream = None      # raw handle of a wx_stream_create_exclusive stream, when the lane needed one
        # (a duck-typed engine on a CPU device -- the host-logic tests' stand-in -- gets a thread and no stream)
        self.stream = torch.cuda.Stream(device=engine.torch_device) if engine.torch_device.type == "cuda" else None
        self.jobs: "queue.Queue" = queue.Queue()
        self.thread = threading.Thread(target=self._run, name=f"whisper-amd-lane-{index}", daemon=True)
        self.thread.start()

    def _run(self):
        if self.stream is not None:
            torch.cuda.set_device(self.engine.torch_device)
        while True:
            job = self.jobs.get()
            if job is None:
                return
            fn, args, box, done = job
            try:
                if self.stream is None:
                    box["value"] = fn(self.engine, *args)
                else:
                    with torch.cuda.stream(self.stream):
                        box["value"] = fn(self.engine, *args)
                        self.stream.synchronize()        # results are complete when the job is reported done
            except BaseException as exc:                 # handed to the caller of result()
                box["error"] = exc
            finally:
                done.set()


class _Pending:
    def __init__(self, box: Dict[str, object], done: threading.Event):
        self._box, self._done = box, done

    def result(self):
        self._done.wait()
        if "error" in self._box:
            raise self._box["error"]
        return self._box["value"]


class EngineLanes:
    """``engines``: replicas of ONE configuration on ONE device (same weights, precision, max_batch).  Jobs of a lane run in
    submission order on the lane's own stream; different lanes run concurrently.  Attribute access falls through to lane 0's
    engine, so an ``EngineLanes`` can stand wherever a ``WhisperEngine`` is used (whisper_amd.dist.transcribe_sharded); only
    ``transcribe_mel`` is different: it deals the batches of a file out over the lanes."""

    def __init__(self, engines: Sequence, one_encoder_at_a_time: bool = True):
        if not engines:
            raise ValueError("at least one engine")
        e0 = engines[0]
        for e in engines[1:]:
            if (e.dims, e.precision, e.max_batch, e.torch_device) != (e0.dims, e0.precision, e0.max_batch, e0.torch_device):
                raise ValueError("lanes must be replicas of one engine configuration on one device")
        # One encoder at a time: the encoder is the one stage that fills the chip by itself (MFMA-bound 256x256 tiles on every CU),
        # so two of them side by side only take turns and evict each other's panels from L2, while the decode loops of the other
        # lanes -- which do fill the holes of an encoder -- wait behind both.  A host lock around encode(), held until its kernels
        # have finished, keeps the lanes' encoders apart: 672.6 instead of 681.0 ms per step at 4 lanes x 64 windows of large-v3
        # (profiles/r03_cu_share_and_hw_queues.txt, section 4).  Results do not change (same kernels, same order per lane).
        if one_encoder_at_a_time and len(engines) > 1 and hasattr(e0, "encode_gate"):
            gate = threading.Lock()
            for e in engines:
                e.encode_gate = gate
        self._lanes = [_Lane(e, i) for i, e in enumerate(engines)]
        self._next = 0
        self.queue_fixes: List[str] = []
        if len(engines) > 1 and e0.torch_device.type == "cuda" and os.environ.get("WX_LANES_NO_QUEUE_CHECK") != "1":
            try:
                self._separate_hardware_queues()
            except Exception as exc:                       # a check, not a requirement: the lanes work on the streams they have
                log.warning("hardware-queue check of the lane streams did not finish (%s); lanes keep their streams", exc)

    # Two lane streams that the HIP runtime put on ONE hardware queue run one after the other -- a third of the throughput at four
    # lanes -- and which streams share is the runtime's business (GPU_MAX_HW_QUEUES = 7 or 8 pairs lanes 0/2 and 1/3; the default 4
    # does not; profiles/r03_cu_share_and_hw_queues.txt sections 2-3).  So it is measured: a wave that idles for 400 us on each of
    # two streams takes 0.4 ms when they have a queue each and 0.8 ms when they share one.  A lane that shares gets another stream
    # of the runtime's pool, then one with a queue of its own (wx_stream_create_exclusive).
    def _serialised(self, a, b, lib, us: int = 400) -> bool:
        import ctypes
        best = None
        for _ in range(2):                                   # (the shorter of two trials: a host hiccup is not a shared queue)
            a.synchronize(); b.synchronize()
            t0 = time.perf_counter()
            for st in (a, b):
                rc = lib.wx_stream_spin(ctypes.c_void_p(st.cuda_stream), us)
                if rc != 0:
                    return False
            a.synchronize(); b.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best > 1.6 * us * 1e-6

    def _separate_hardware_queues(self) -> None:
        import ctypes
        from . import _lib
        lib = _lib.load()
        dev = self._lanes[0].engine.torch_device
        with torch.cuda.device(dev):
            for j in range(1, len(self._lanes)):
                lane = self._lanes[j]
                for attempt in range(3):
                    if not any(self._serialised(self._lanes[i].stream, lane.stream, lib) for i in range(j)):
                        break
                    if attempt == 0:
                        lane.stream = torch.cuda.Stream(device=dev)
                        self.queue_fixes.append(f"lane {j}: shared a hardware queue with an earlier lane; took another stream")
                    elif attempt == 1:
                        raw = ctypes.c_void_p()
                        _lib.check(lib.wx_stream_create_exclusive(dev.index or 0, ctypes.byref(raw)), "wx_stream_create_exclusive")
                        lane.exclusive_stream = raw
                        lane.stream = torch.cuda.ExternalStream(raw.value, device=dev)
                        self.queue_fixes.append(f"lane {j}: still shared; made a stream with a hardware queue of its own")
                    else:
                        log.warning("lane %d still shares a hardware queue with another lane: its batches will take turns with that lane's", j)
        for msg in self.queue_fixes:
            log.info(msg)

    @classmethod
    def create(cls, make_engine: Callable[[int], object], n_lanes: int, share_weights: bool = True,
               one_encoder_at_a_time: bool = True) -> "EngineLanes":
        """``make_engine(n_lanes)`` builds the first engine; it should pass ``n_lanes`` on to ``WhisperEngine`` so that the
        engine sizes its cross-attention grid for its share of the GPU.  The other lanes are ``replica()``s of it -- same
        configuration, the same device weights (``wx_create_shared``) -- unless ``share_weights`` is False or the engine has
        no ``replica`` (a stand-in), in which case ``make_engine`` is called once per lane."""
        if n_lanes < 1:
            raise ValueError("n_lanes must be >= 1")
        first = make_engine(n_lanes)
        engines = [first]
        try:
            for _ in range(n_lanes - 1):
                engines.append(first.replica() if share_weights and hasattr(first, "replica") else make_engine(n_lanes))
        except BaseException:
            for e in engines:
                e.close()
            raise
        return cls(engines, one_encoder_at_a_time=one_encoder_at_a_time)

    @property
    def n_lanes(self) -> int:
        return len(self._lanes)

    @property
    def engines(self) -> List:
        return [ln.engine for ln in self._lanes]

    def __getattr__(self, name):                          # everything else: lane 0's engine
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._lanes[0].engine, name)

    def submit(self, fn: Callable, *args, lane: Optional[int] = None) -> _Pending:
        """Queue ``fn(engine, *args)`` on a lane (round robin unless ``lane`` is given)."""
        if lane is None:
            lane, self._next = self._next, (self._next + 1) % len(self._lanes)
        box: Dict[str, object] = {}
        done = threading.Event()
        self._lanes[lane].jobs.put((fn, args, box, done))
        return _Pending(box, done)

    def map(self, fn: Callable, items: Sequence) -> List:
        """``[fn(engine, item) for item in items]``, item i on lane i mod n_lanes, results in order."""
        pend = [self.submit(fn, item, lane=i % len(self._lanes)) for i, item in enumerate(items)]
        out, first_error = [], None
        for p in pend:                                     # every job has ended before an error leaves this function: the
            try:                                           # others may still be reading what the caller passed in
                out.append(p.result())
            except BaseException as exc:
                first_error = first_error or exc
        if first_error is not None:
            raise first_error
        return out

    def transcribe_mel(self, mel: torch.Tensor, n_content_frames: int, window_lo: int, window_hi: int,
                       prompt: Sequence[int], beam_size: Optional[int] = None, patience: float = 1.0, **decode_kw) -> Dict[str, object]:
        """Same contract as ``WhisperEngine.transcribe_mel``; consecutive batches of ``max_batch`` windows (``max_batch //
        beam_size`` with beam search) go to consecutive lanes."""
        e0 = self._lanes[0].engine
        per = e0.max_batch if not beam_size else max(1, e0.max_batch // int(beam_size))
        if e0.torch_device.type == "cuda":
            torch.cuda.current_stream(e0.torch_device).synchronize()  # the mel was produced on the caller's stream
        spans = [(w0, min(w0 + per, window_hi)) for w0 in range(window_lo, window_hi, per)]

        def one(engine, span):
            return engine.transcribe_mel(mel, n_content_frames, span[0], span[1], prompt, beam_size=beam_size, patience=patience, **decode_kw)

        parts = self.map(one, spans)
        out: Dict[str, object] = {"tokens": [], "sum_logprobs": [], "no_speech_probs": [], "steps": 0}
        for p in parts:
            out["tokens"] += p["tokens"]; out["sum_logprobs"] += p["sum_logprobs"]; out["no_speech_probs"] += p["no_speech_probs"]
            out["steps"] += p["steps"]
        return out

    def close(self):
        for ln in self._lanes:
            ln.jobs.put(None)
        for ln in self._lanes:
            waited = 0
            ln.thread.join(timeout=60)
            while ln.thread.is_alive() and waited < 10:    # a long decode or a first graph capture is still running: destroying
                log.warning("lane %d is still busy; waiting for its job before its engine is destroyed", ln.index)
                ln.thread.join(timeout=60); waited += 1    # the engine under it would free memory its kernels are using
            if ln.thread.is_alive():
                log.error("lane %d did not finish; its engine is left alive (leaked) rather than destroyed under a running job", ln.index)
                continue
            ln.engine.close()
            if ln.exclusive_stream is not None:
                from . import _lib
                _lib.load().wx_stream_destroy(ln.exclusive_stream)
                ln.exclusive_stream = None
        self._lanes = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


__all__ = ["EngineLanes"]
