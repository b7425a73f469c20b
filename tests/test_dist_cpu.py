"""N>1 path on CPU: world_size-2 gloo run of whisper_amd.dist.transcribe_sharded with a CPU stand-in engine built on the
oracle (tests only).  Checks the window partition, the PCM halo, the MAX all-reduce of the mel clamp and the token
all-gather against the single-process oracle result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import whisper_oracle as wo
from whisper_amd import checkpoint as ck
from whisper_amd import dist as wdist
from whisper_amd.audio import N_FRAMES


class OracleEngine:
    """Duck-typed stand-in for WhisperEngine (mel_stage1/2, reset_max, transcribe_mel) on the CPU oracle."""

    def __init__(self, dims, sd, max_new, max_batch=8, name="engine"):
        self.dims, self.tok = dims, ck.special_tokens_for(dims.n_vocab)
        self.torch_device = torch.device("cpu")
        self.om = wo.OracleWhisper(dims, sd)
        self.sup = wo.suppress_token_ids(self.tok, ck.non_speech_suppress_ids())
        self.max_new = max_new
        self.max_batch, self.precision, self.name = max_batch, "fp32", name
        self.batches = []                                  # (window_lo, window_hi, thread name) of every transcribe_mel call

    def close(self):
        pass

    def initial_tokens(self, language_index=0, prompt_ids=None, **kw):
        return wo.initial_tokens(self.tok, language_index, prompt_ids=prompt_ids)

    def reset_max(self, g):
        g.fill_(float("-inf"))

    def mel_stage1(self, pcm, pcm_offset, n_file, n_total, frame0, n_frames, out, gmax):
        full = torch.zeros(n_total)
        full[pcm_offset:pcm_offset + pcm.numel()] = pcm          # only this rank's samples (+halo) are known
        lm = wo.log_mel_unnormalised(full.numpy(), self.dims.n_mels)[:, frame0:frame0 + n_frames]
        out[:, :n_frames] = lm
        gmax.copy_(torch.maximum(gmax, lm.max().reshape(1)))

    def mel_stage2(self, mel, n_frames, gmax):
        mel[:, :n_frames] = (torch.maximum(mel[:, :n_frames], gmax - 8.0) + 4.0) / 4.0

    def detect_language(self, mel, n_content):
        seg = wo.pad_or_trim_frames(mel[:, :min(N_FRAMES, max(n_content, 0))])
        with torch.no_grad():
            lg = self.om.logits(torch.tensor([[self.tok.sot]]), self.om.encode(seg[None]))[0, 0]
        return int(lg[self.tok.sot + 1: self.tok.sot + 1 + self.tok.n_langs].argmax())

    def transcribe_mel(self, mel, n_content, w_lo, w_hi, prompt, beam_size=None, patience=1.0, **kw):
        import threading
        self.batches.append((w_lo, w_hi, threading.current_thread().name))
        segs = []
        for w in range(w_lo, w_hi):
            seg = mel[:, w * N_FRAMES:min((w + 1) * N_FRAMES, max(n_content, 0))]
            segs.append(wo.pad_or_trim_frames(seg))
        with torch.no_grad():
            xa = self.om.encode(torch.stack(segs))
            if beam_size:
                assert (w_hi - w_lo) * beam_size <= self.max_batch
                r = wo.beam_search_decode(self.om, xa, prompt, self.tok, self.sup, beam_size=beam_size, patience=patience, max_new_tokens=self.max_new)
            else:
                r = wo.greedy_decode(self.om, xa, prompt, self.tok, self.sup, self.max_new)
        return {"tokens": r["tokens"], "sum_logprobs": list(r["sum_logprobs"]), "no_speech_probs": list(r["no_speech_probs"]),
                "steps": r["steps"]}


def _worker(rank, world, port, n_samples, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    eng = OracleEngine(dims, sd, max_new=12)
    audio = wo.varied_audio(21, n_samples)
    audio[: n_samples // 3] *= 0.02          # the global mel maximum lives in rank 1's share
    # a callable prompt: the language is detected on rank 0's window 0 (whole-file clamp, known after the MAX all-reduce)
    # and broadcast, then every rank builds the same prompt
    out = wdist.transcribe_sharded(eng, audio, lambda li: wo.initial_tokens(eng.tok, li))
    q.put((rank, out["tokens"], out["sum_logprobs"], out["n_windows"], out["local_windows"], out["language_index"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    n_samples = 16000 * 95 + 321           # 4 windows, the last one ragged
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_samples, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    tok = ck.special_tokens_for(dims.n_vocab)
    audio = wo.varied_audio(21, n_samples)
    audio[: n_samples // 3] *= 0.02
    om = wo.OracleWhisper(dims, sd)
    with torch.no_grad():
        mel = wo.log_mel_spectrogram(audio, dims.n_mels, padding=wo.N_SAMPLES)          # whole file, whole-file clamp
        lg = om.logits(torch.tensor([[tok.sot]]), om.encode(mel[None, :, :N_FRAMES]))[0, 0]
        lang = int(lg[tok.sot + 1: tok.sot + 1 + tok.n_langs].argmax())
        ref = wo.transcribe_windows(om, audio, wo.initial_tokens(tok, lang), tok,
                                    wo.suppress_token_ids(tok, ck.non_speech_suppress_ids()), max_new_tokens=12)
    assert ref["n_windows"] == 4
    for rank, toks, sums, nw, local, li in sorted(got):
        assert li == lang
        assert nw == 4 and local == ((0, 2) if rank == 0 else (2, 4))
        assert toks == ref["tokens"]                      # every rank holds the full, window-ordered result
        np.testing.assert_allclose(sums, ref["sum_logprobs"], rtol=1e-4, atol=1e-3)


def _lanes_worker(rank, world, port, q):
    """configs[3]'s real host path: transcribe() -> transcribe_sharded -> EngineLanes.transcribe_mel -> beam search, two ranks,
    two lanes per rank (stand-in engines on the CPU oracle)."""
    import threading
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from whisper_amd import transcribe as tx
    from whisper_amd.lanes import EngineLanes
    from whisper_amd.tokenizer import Tokenizer
    # every collective of the run, with the thread that issued it
    issued = []
    for name in ("all_reduce", "all_gather", "broadcast"):
        real = getattr(dist, name)

        def wrapped(*a, _real=real, _name=name, **k):
            issued.append((_name, threading.current_thread() is threading.main_thread()))
            return _real(*a, **k)
        setattr(dist, name, wrapped)
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    engines = [OracleEngine(dims, sd, max_new=6, max_batch=10, name=f"lane{i}") for i in range(2)]
    lanes = EngineLanes(engines)
    tx._STATE.update(engine=lanes, tokenizer=Tokenizer(dims.n_vocab), settings={"model": "micro", "confidence_mode": "avg_logprob"})
    out = {}
    for label, n_samples in (("five", 16000 * 125 + 77), ("one", 16000 * 9)):       # 5 windows (3 + 2: not divisible by world x 2); 1 window < 2 ranks
        del issued[:]
        for e in engines:
            del e.batches[:]
        audio = wo.varied_audio(33, n_samples)
        audio[: n_samples // 2] *= 0.03                      # the whole-file mel maximum lives in the later windows
        r = tx.transcribe(audio, language=None, beam_size=5, max_new_tokens=6)
        out[label] = {"tokens": r["tokens"], "language": r["language"], "chunks": r["chunks_processed"], "avg_logprob": r["avg_logprob"],
                      "collectives": list(issued), "batches": [list(e.batches) for e in engines]}
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()
    lanes.close()


def test_two_ranks_two_lanes_beam_search_through_transcribe():
    """BASELINE configs[3]'s path on the host side, rehearsed on CPU: transcribe(language=None, beam_size=5) on two gloo ranks,
    each with an EngineLanes of two engines (max_batch 10 rows = 2 windows x 5 beams per batch).  Windows come back in file
    order and equal a single-process beam search of the whole file; the language detected by rank 0 reaches rank 1; a rank's
    batches are dealt to its lanes in order, on the lanes' threads; and every collective -- the MAX all-reduce of the mel
    clamp, the language broadcast, the three result all-gathers -- is issued by the main thread, the same sequence on both
    ranks, also when a rank has no window at all."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lanes_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0, eot_boost=3.0)
    tok = ck.special_tokens_for(dims.n_vocab)
    om = wo.OracleWhisper(dims, sd)
    sup = wo.suppress_token_ids(tok, ck.non_speech_suppress_ids())
    from whisper_amd.tokenizer import LANGUAGE_CODES
    for label, n_samples, n_win in (("five", 16000 * 125 + 77, 5), ("one", 16000 * 9, 1)):
        audio = wo.varied_audio(33, n_samples)
        audio[: n_samples // 2] *= 0.03
        with torch.no_grad():
            mel = wo.log_mel_spectrogram(audio, dims.n_mels, padding=wo.N_SAMPLES)
            content = mel.shape[-1] - N_FRAMES
            lg = om.logits(torch.tensor([[tok.sot]]), om.encode(wo.pad_or_trim_frames(mel[:, :min(N_FRAMES, content)])[None]))[0, 0]
            lang = int(lg[tok.sot + 1: tok.sot + 1 + tok.n_langs].argmax())
            segs = torch.stack([wo.pad_or_trim_frames(mel[:, w * N_FRAMES:min((w + 1) * N_FRAMES, content)]) for w in range(n_win)])
            ref = wo.beam_search_decode(om, om.encode(segs), wo.initial_tokens(tok, lang), tok, sup, beam_size=5, max_new_tokens=6)
        for rank in (0, 1):
            g = got[rank][label]
            assert g["chunks"] == n_win and g["language"] == LANGUAGE_CODES[lang]
            assert g["tokens"] == ref["tokens"]                                   # whole file, window order, on every rank
            np.testing.assert_allclose(g["avg_logprob"], [sl / (len(t) + 1) for sl, t in zip(ref["sum_logprobs"], ref["tokens"])], rtol=1e-4, atol=1e-3)
            assert [c[0] for c in g["collectives"]] == ["all_reduce", "broadcast", "all_gather", "all_gather", "all_gather"]
            assert all(main for _, main in g["collectives"])
        # rank 0 owns windows [0, 3) of five: batches (0, 2) -> lane 0, (2, 3) -> lane 1, each on its lane's own thread; rank 1 owns
        # [3, 5): one batch on lane 0.  With one window rank 1 decodes nothing and still takes part in every collective.
        b0, b1 = got[0][label]["batches"], got[1][label]["batches"]
        if label == "five":
            assert [(lo, hi) for lo, hi, _ in b0[0]] == [(0, 2)] and [(lo, hi) for lo, hi, _ in b0[1]] == [(2, 3)]
            assert [(lo, hi) for lo, hi, _ in b1[0]] == [(0, 2)] and b1[1] == []
            assert b0[0][0][2] == "whisper-amd-lane-0" and b0[1][0][2] == "whisper-amd-lane-1"
        else:
            assert [(lo, hi) for lo, hi, _ in b0[0]] == [(0, 1)] and b0[1] == [] and b1 == [[], []]


def test_single_process_path_needs_no_process_group():
    dims = ck.dims_for("micro")
    sd = ck.synthetic_state_dict(dims, seed=0)
    eng = OracleEngine(dims, sd, max_new=4)
    audio = wo.varied_audio(2, 16000 * 31)
    out = wdist.transcribe_sharded(eng, audio, wo.initial_tokens(eng.tok, 0))
    assert out["n_windows"] == 2 and len(out["tokens"]) == 2 and out["local_windows"] == (0, 2)
    assert wdist.local_sample_range(10_000_000, 2, 4) == (2 * 480000 - 200, 4 * 480000 + 200)
    assert wdist.local_sample_range(500_000, 0, 2) == (0, 500_000)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher and no rank environment starts N rank processes itself (rendezvous on
    127.0.0.1), relays rank 0's line and fails when any rank fails; under a launcher (WORLD_SIZE set) it does not re-launch
    and refuses a --gpus that disagrees with the world size.  --dry-run: the ranks stop before touching the GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rank"] == 0 and line["master_addr"] == "127.0.0.1" and int(line["master_port"]) > 0
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, BENCH_DRY_RUN_FAIL_RANK="1"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "ranks failed" in bad.stderr
    under = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                           env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert under.returncode == 0 and json.loads(under.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    wrong = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                           env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert wrong.returncode != 0
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["n_gpus"] == 1
