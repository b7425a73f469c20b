#!/usr/bin/env python3
"""Timing of upstream's sequential long-form mode (whisper_amd/longform.py) on one synthetic file: every window carries
up to 223 ids of previous text as its prompt, so prompt feeding is half of the work.  Compares the prompt pass (several
forced positions per pass) with one-position-per-step feeding (WX_NO_PREFILL=1).
usage: python tools/bench_sequential.py [model] [minutes] [files]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisper_amd import checkpoint as ck
from whisper_amd.audio import synthetic_audio
from whisper_amd.engine import WhisperEngine
from whisper_amd.longform import transcribe_sequential
from whisper_amd.tokenizer import Tokenizer

model = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
files = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dims = ck.dims_for(model)
eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, seed=0), precision="bf16", max_batch=16)
dec = Tokenizer(dims.n_vocab).decode
audios = [np.concatenate([synthetic_audio(100 * f + w) for w in range(int(minutes * 2))]) for f in range(files)]
for flag in ("1", "0", "1", "0"):
    os.environ["WX_NO_PREFILL"] = flag
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = transcribe_sequential(eng, audios, dec, temperatures=(0.0,), compression_ratio_threshold=None, logprob_threshold=None,
                                no_speech_threshold=None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{model} sequential, {files} file(s) x {minutes:.0f} min, prompt pass {'off' if flag == '1' else 'on '}: {dt:7.2f} s  "
          f"RTFx {files * minutes * 60 / dt:7.1f}  windows {[o['windows'] for o in out]} tokens {sum(len(o['tokens']) for o in out)}", flush=True)
eng.close()
