#!/usr/bin/env python3
"""bench.py — RTFx (audio-seconds per wall-second) of the local Whisper path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic 30 s windows that is already resident in
HBM: log-mel (whole-buffer clamp) -> encoder -> cross-K/V projection -> greedy decode to EOT or upstream's
sample_len (224 new tokens).  N=1 workload = BASELINE.json configs[2]: Whisper large-v3, bf16, greedy, batch 64
windows on one MI355X.  With N>1 every rank runs the same per-GPU workload on its own windows of one long virtual
file (weak scaling); the only collectives are the 4-byte MAX all-reduce of the mel clamp and the final all-gather of
token ids.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — the dominant kernel (decoder cross-attention, HBM-bound): algorithmic bytes per launch divided by
                 its mean launch duration measured with HIP events on the launch stream in an eager, event-bracketed
                 decode pass over the same batch (the timed region itself replays a hipGraph per step, which events
                 cannot look inside; profiles/ holds the rocprofv3 kernel stats of the same command for comparison);
  cpu_baseline — the CPU oracle (oracle/whisper_oracle.py, PyTorch fp32) timed on the host cores on one window.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from whisper_amd import checkpoint as ck                      # noqa: E402
from whisper_amd.audio import N_FRAMES, N_SAMPLES, synthetic_audio  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--model", default="large-v3")
    p.add_argument("--batch", type=int, default=64, help="30 s windows per GPU per step")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--max-new-tokens", type=int, default=0, help="0 = upstream sample_len (n_text_ctx // 2)")
    p.add_argument("--beam", type=int, default=0, help="beam width (0 = greedy); beam search keeps the split-K decode path up to batch x beam = 80 rows")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real multi-GPU path) | gloo (rehearsal of the N>1 code path)")
    p.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    p.add_argument("--cpu-decode-steps", type=int, default=0, help="0 = as many greedy steps as the GPU run took (no extrapolation)")
    p.add_argument("--no-profile-pass", action="store_true", help="skip the event-bracketed eager pass (PMC runs)")
    p.add_argument("--lanes", type=int, default=4, help="batches in flight per GPU: engine replicas, each on its own host thread and stream "
                   "(whisper_amd/lanes.py); step i runs on lane i mod lanes.  1 = one batch at a time (per-kernel profiles)")
    p.add_argument("--separate-weights", action="store_true", help="A/B: every lane uploads its own copy of the weights (round 2's layout) "
                   "instead of sharing lane 0's (wx_create_shared)")
    p.add_argument("--no-single-batch", action="store_true", help="skip the single_batch leg (a few steps of ONE batch at a time on an engine "
                   "sized for the whole GPU, after the timed region)")
    p.add_argument("--overlap-encoders", action="store_true", help="A/B: lanes may run their encoders side by side (round 2's behaviour); default: "
                   "one encoder at a time per GPU (EngineLanes' encode gate)")
    p.add_argument("--dry-run", action="store_true", help="launcher rehearsal: every rank reports its rank environment and exits before touching the GPU")
    return p.parse_args()


def host_cores():
    """(threads to use, physical cores, how they were found).  Physical cores = distinct (physical id, core id) pairs of
    /proc/cpuinfo; the threads actually used are bounded by this process's CPU share: its affinity mask and the harness's
    OMP_NUM_THREADS (the GPU box shows the whole machine in nproc but grants a 16-CPU share per GPU)."""
    phys, how = None, ""
    try:
        pairs, cur = set(), {}
        with open("/proc/cpuinfo") as f:
            for line in f:
                if ":" in line:
                    k, v = [x.strip() for x in line.split(":", 1)]
                    cur[k] = v
                elif not line.strip():
                    if "core id" in cur:
                        pairs.add((cur.get("physical id", "0"), cur["core id"]))
                    cur = {}
        if "core id" in cur:
            pairs.add((cur.get("physical id", "0"), cur["core id"]))
        if pairs:
            phys, how = len(pairs), "/proc/cpuinfo distinct (physical id, core id)"
    except OSError:
        pass
    if not phys:
        phys, how = os.cpu_count() or 1, "os.cpu_count() (no core ids in /proc/cpuinfo)"
    limit = [phys]
    try:
        limit.append(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        limit.append(int(os.environ["OMP_NUM_THREADS"]))
    return max(1, min(limit)), phys, how + f"; threads = min(physical, affinity, OMP_NUM_THREADS) = {max(1, min(limit))}"


def cpu_baseline(dims, sd, tok, suppress_ids, gpu_steps_per_window: int, decode_steps: int):
    """Oracle (kind 'port') on the host cores.  (1) one window of the SAME workload as the GPU line: mel, encoder and the
    full greedy decode (as many steps as the GPU run took per window: nothing is extrapolated unless --cpu-decode-steps
    shortens it).  (2) BASELINE.json configs[0] exactly: Whisper tiny fp32 greedy on one 30 s synthetic clip (seed 0),
    sample_len steps -- the reference's own CPU-runnable case."""
    from oracle import whisper_oracle as wo
    cores, phys, how = host_cores()
    torch.set_num_threads(cores)
    audio = synthetic_audio(0)

    def run(dims_, sd_, sup_, n_steps, warm):
        om = wo.OracleWhisper(dims_, sd_)
        tok_ = ck.special_tokens_for(dims_.n_vocab)
        init = wo.initial_tokens(tok_, 0)
        with torch.no_grad():
            if warm:        # one untimed pass (thread pool, allocator); skipped for the large model, where it would double the run
                xw = om.encode(wo.log_mel_spectrogram(audio, dims_.n_mels, padding=wo.N_SAMPLES)[None, :, :wo.N_FRAMES])
                wo.greedy_decode(om, xw, init, tok_, sup_, max_new_tokens=4)
            t0 = time.perf_counter()
            mel = wo.log_mel_spectrogram(audio, dims_.n_mels, padding=wo.N_SAMPLES)[:, :wo.N_FRAMES]
            t1 = time.perf_counter()
            xa = om.encode(mel[None])
            t2 = time.perf_counter()
            r = wo.greedy_decode(om, xa, init, tok_, sup_, max_new_tokens=n_steps)
            t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2, int(r["steps"])

    n = decode_steps if decode_steps > 0 else gpu_steps_per_window
    t_mel, t_enc, t_dec, steps = run(dims, sd, suppress_ids, n, warm=dims.n_audio_state <= 768)
    per_step = t_dec / max(1, steps)
    total = t_mel + t_enc + per_step * gpu_steps_per_window
    extrap = "" if steps == gpu_steps_per_window else f", decode extrapolated linearly from {steps} to {gpu_steps_per_window} steps"
    out = {
        "value": 30.0 / total, "unit": "audio-sec/sec", "cores": cores, "physical_cores": phys, "cores_source": how, "kind": "port",
        "sample": (f"oracle/whisper_oracle.py (PyTorch fp32) on 1 window of the same workload: mel {t_mel:.2f}s, encoder "
                   f"{t_enc:.2f}s, {steps} greedy steps {t_dec:.2f}s{extrap}; weight load excluded"),
    }
    d0 = ck.dims_for("tiny")
    sd0 = ck.synthetic_state_dict(d0, seed=0)
    from whisper_amd.engine import default_suppress_ids
    tok0 = ck.special_tokens_for(d0.n_vocab)
    m0, e0, c0, s0 = run(d0, sd0, default_suppress_ids(tok0, d0.n_vocab), d0.n_text_ctx // 2, warm=True)
    out["config0"] = {
        "workload": "BASELINE.json configs[0]: Whisper tiny fp32 greedy, one 30 s synthetic clip (seed 0), CPU oracle, full decode",
        "value": 30.0 / (m0 + e0 + c0), "unit": "audio-sec/sec", "seconds": m0 + e0 + c0, "mel_s": m0, "encoder_s": e0,
        "decode_s": c0, "decode_steps": s0, "cores": cores,
    }
    return out


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N fresh rank processes of this script (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relay rank 0's JSON line, and fail if any rank
    fails.  This parent never touches the GPU (no process that has initialised HIP is ever replaced or forked)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]          # (released before the ranks bind it: a clash shows up as a failed rank, reported below)
    import tempfile
    import threading
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        log = tempfile.TemporaryFile(mode="w+")           # per-rank stderr (and stdout of ranks > 0): its tail goes into the failure message
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else log, stderr=log, text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "3000"))
    failed = None
    while True:                                           # watch EVERY rank: one that dies leaves the others in a collective forever
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or all(rc == 0 for rc in rcs):
            failed = bad
            break
        if time.monotonic() > deadline:
            failed = [(r, "timeout") for r, rc in enumerate(rcs) if rc is None]
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if failed:
        print(f"bench.py: ranks failed: {failed}", file=sys.stderr)
        for r, _ in failed:
            logs[r].seek(0)
            tail = logs[r].read()[-2000:]
            if tail.strip():
                print(f"--- rank {r} (tail) ---\n{tail}", file=sys.stderr)
        return 1
    for r, log in enumerate(logs):                        # a clean run: pass the ranks' stderr on
        log.seek(0)
        txt = log.read()
        if txt.strip():
            sys.stderr.write(txt)
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        if os.environ.get("BENCH_DRY_RUN_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rank": rank, "local_rank": local_rank, "master_addr": os.environ.get("MASTER_ADDR"),
                              "master_port": os.environ.get("MASTER_PORT"), "dry_run": True}), flush=True)
        return
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from whisper_amd import dist as wdist
    from whisper_amd.dist import sharded_mel
    from whisper_amd.engine import WhisperEngine

    dims = ck.dims_for(args.model)
    tok = ck.special_tokens_for(dims.n_vocab)
    B = args.batch
    sd = ck.synthetic_state_dict(dims, seed=0)
    from whisper_amd.lanes import EngineLanes
    if args.lanes < 1:
        raise SystemExit("--lanes must be >= 1")
    lanes = EngineLanes.create(lambda n: WhisperEngine(dims, sd, precision=args.precision, max_batch=B * max(1, args.beam), device=local_rank, n_lanes=n),
                               args.lanes, share_weights=not args.separate_weights, one_encoder_at_a_time=not args.overlap_encoders)
    eng = lanes.engines[0]
    prompt = eng.initial_tokens(0)

    # this rank's windows of one long virtual file of world*B windows; clip i of the file uses seed i
    w_lo, w_hi = rank * B, (rank + 1) * B
    n_file = world * B * N_SAMPLES
    host = np.concatenate([synthetic_audio(w) for w in range(w_lo, w_hi)])
    s_lo = w_lo * N_SAMPLES
    pcm = torch.from_numpy(host).to(dev)          # resident in HBM before the timed region
    del host
    torch.cuda.synchronize(dev)

    stream = torch.cuda.Stream(device=dev)

    # One step = one batch of B windows through the whole hot path.  Front: log-mel on the main thread, in step order -- with
    # N > 1 it holds the path's one exchange in front of the decode (the 4-byte MAX all-reduce of the mel clamp), and every
    # collective of this program is issued by this thread in the same order on every rank.  (The mel entry points only read the
    # engine's tables, so lane 0's engine serves them while its lane decodes.)  Back: encoder + decode loop on the step's lane
    # (engine replica, host thread, stream).  Then the result gather, main thread again.
    def front(marks=None):
        def mark():
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
        mark()
        mel = sharded_mel(eng, pcm, s_lo, n_file, w_lo, w_hi)
        mark()
        torch.cuda.current_stream(dev).synchronize()       # the lane's stream reads it
        return mel

    def back(eng, mel, use_graph=True, profile_stride=0, marks=None):
        def mark():
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
        mark()
        eng.encode(mel, N_FRAMES, mel.stride(0), n_file // 160 - w_lo * N_FRAMES, B)
        mark()
        if args.beam:
            rb = eng.decode_beam(B, prompt, beam_size=args.beam, max_new_tokens=args.max_new_tokens, use_graph=use_graph)
            toks = torch.full((B, dims.n_text_ctx + 1), eng.tok.eot, dtype=torch.int32)
            for b, t in enumerate(rb["tokens"]):
                toks[b, :len(t)] = torch.tensor(t, dtype=torch.int32)
            r = {"tokens": toks.to(dev), "n_tokens": torch.tensor([len(t) for t in rb["tokens"]], dtype=torch.int32, device=dev),
                 "steps": rb["steps"]}
        else:
            r = eng.decode_greedy(B, prompt, max_new_tokens=args.max_new_tokens, use_graph=use_graph,
                                  profile_stride=profile_stride)
        mark()
        return r

    def gather(r):             # result gather: padded ids + lengths, ~115 KB per rank
        if world > 1:
            wdist.all_gather(r["tokens"])
            wdist.all_gather(r["n_tokens"])
        return r

    def run_steps(n_steps, lane_of):
        """n_steps batches; batch i on lane lane_of(i); at most 2 x lanes log-mels ahead of the slowest lane"""
        pending, r, done = [], None, 0
        for i in range(n_steps):
            pending.append(lanes.submit(back, front(), not args.no_graph, lane=lane_of(i)))
            while len(pending) - done > 2 * lanes.n_lanes:
                r = gather(pending[done].result()); done += 1
        while done < len(pending):
            r = gather(pending[done].result()); done += 1
        return r

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        # warm-up: W steps, at least one per lane (its graph is captured there), one lane at a time
        n_warm = max(args.warmup, lanes.n_lanes)
        for i in range(n_warm):
            gather(lanes.submit(back, front(), not args.no_graph, lane=i % lanes.n_lanes).result())
        sync_all()
        t0 = time.perf_counter()
        r = run_steps(args.steps, lambda i: i % lanes.n_lanes)
        sync_all()
        elapsed = time.perf_counter() - t0
        steps_per_window = int(r["steps"])
        n_tokens = r["n_tokens"].float().mean().item()

        # per-stage device time of one more (untimed) step, alone on the GPU: events on the stream of each stage
        m_front, m_back = [], []
        lanes.submit(back, front(m_front), not args.no_graph, 0, m_back, lane=0).result()
        torch.cuda.synchronize(dev)
        st_ms = [m_front[0].elapsed_time(m_front[1]), m_back[0].elapsed_time(m_back[1]), m_back[1].elapsed_time(m_back[2])]
        # dominant kernel: HIP events around every eighth position's launches of it, on the launching stream, in an eager pass of
        # one batch PER LANE, all lanes at once -- the conditions of the timed region (which replays graphs, where events cannot go)
        prof_ms, prof_n = 0.0, 0
        if not args.no_profile_pass and not args.beam:
            mels = [front() for _ in range(lanes.n_lanes)]
            for p in [lanes.submit(back, mels[i], False, 8, lane=i) for i in range(lanes.n_lanes)]:
                p.result()
            torch.cuda.synchronize(dev)
            for e_i in lanes.engines:
                ms_i, n_i = e_i.last_profile()
                prof_ms += ms_i; prof_n += n_i

        # single_batch leg: ONE batch at a time on an engine sized for the whole GPU (n_lanes = 1: the stream kernel at one full
        # round of workgroups), same weights (a replica of lane 0's engine).  Not part of `value`; it is the latency-side figure.
        single = None
        if not args.no_single_batch and not args.beam and lanes.n_lanes > 1 and rank == 0:
            solo = eng.replica(n_lanes=1)
            try:
                back(solo, front())                                    # capture its graph
                torch.cuda.synchronize(dev)
                n_solo = 3
                ts = time.perf_counter()
                for _ in range(n_solo):
                    back(solo, front())
                torch.cuda.synchronize(dev)
                solo_s = (time.perf_counter() - ts) / n_solo
                ms_f, ms_b = [], []
                back(solo, front(ms_f), True, 0, ms_b)
                torch.cuda.synchronize(dev)
                back(solo, front(), False, 8)                          # eager, event-bracketed stream launches, alone on the GPU
                torch.cuda.synchronize(dev)
                solo_prof_ms, solo_prof_n = solo.last_profile()
                single = {"lanes": 1, "key_splits": solo.cross_key_splits, "steps": n_solo, "ms_per_step": solo_s * 1e3,
                          "value": B * 30.0 / solo_s, "unit": "audio-sec/sec",
                          "stages_ms": {"mel": ms_f[0].elapsed_time(ms_f[1]), "encoder": ms_b[0].elapsed_time(ms_b[1]), "decode": ms_b[1].elapsed_time(ms_b[2])},
                          "kernel_avg_launch_ms": solo_prof_ms / max(1, solo_prof_n), "kernel_launches_timed": solo_prof_n}
            finally:
                solo.close()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        wdist.all_reduce_max(t)
    elapsed = float(t.item())

    if rank == 0:
        es = 2 if args.precision == "bf16" else 4
        audio_sec = world * B * 30.0 * args.steps
        value = audio_sec / elapsed
        # bf16 greedy: the cross-attention streams the encoder output itself, 1500 x d elements per window and layer (xattn.hip);
        # fp32 / beam search: the K and V caches of every (b, h), twice that
        estream = args.precision == "bf16" and not args.beam and os.environ.get("WX_NO_XATTN") is None
        bytes_per_launch = B * dims.n_audio_ctx * dims.n_audio_state * es * (1 if estream else 2)
        avg_ms = prof_ms / max(1, prof_n)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if prof_n else 0.0
        # traffic: HBM bytes per launch from a rocprofv3 --pmc pass of THIS command (TCC FETCH_SIZE with the guide's gfx950
        # corrections), committed under profiles/ -- a replayed constant, not something this run measured; other shapes
        # than the default workload's report none
        traffic, traffic_source = None, None
        for name in (("r03_estream_pmc.json", "r02_estream_pmc.json") if estream else ("r01_cross_attn_pmc.json",)):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc) and (args.model, B, args.precision) == ("large-v3", 64, "bf16"):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/{name} (separate rocprofv3 --pmc pass of this command; not measured in this run)"
                    break
                except Exception:
                    traffic = None
        out = {
            "metric": "audio-sec/sec (RTFx) whole node, large-v3 greedy, 1/2/4/8 MI355X",
            "value": value, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {
                "workload": f"Whisper {args.model} {args.precision} {'beam=%d' % args.beam if args.beam else 'greedy'}, batch {B} x 30 s windows per GPU: "
                            f"log-mel + encoder + KV-cache decode"
                            + (" (BASELINE.json configs[2])" if (args.model, args.beam, B) == ("large-v3", 0, 64) else ""),
                "windows_per_gpu": B, "decode_steps_per_window": steps_per_window, "mean_new_tokens": n_tokens,
                "prompt": "[sot, <|en|>, transcribe, notimestamps]", "weights": "seeded random init (no checkpoints offline)",
                "hip_graph": not args.no_graph, "parallelism": f"window-dp{world}",
                "batches_in_flight_per_gpu": lanes.n_lanes,
            },
            "roofline": {
                "kernel": ("xattn_estream_kernel (decoder cross-attention on the encoder output)" if estream else
                           "attn_decode_kernel (decoder cross-attention over the K/V cache)"),
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms, "launches_timed": prof_n,
                "measured": "HIP events around the kernel's launches on their own streams, eager pass of one batch per lane, all lanes at once",
                "launches_in_flight": lanes.n_lanes, "workgroups_per_launch": B * eng.cross_key_splits if estream else None,
                "frac_of_share": achieved / (HBM_PEAK_GBS / lanes.n_lanes),      # against 1 / launches_in_flight of the peak
                "note": ("one workgroup owns a CU; with several batches in flight a launch is sized for its share of the 256 CUs "
                         "(wx_config.n_lanes) and up to launches_in_flight of them share HBM: frac is per launch against the whole "
                         "GPU's peak; alone on the GPU at 4 key splits the kernel runs 48 us = 5.1 TB/s = 0.64 "
                         "(profiles/r02_kernel_stats_bench_lanes1.txt)") if estream and lanes.n_lanes > 1 else None,
            },
        }
        # stage rooflines (SURVEY §8d formulas; encoder FLOPs include the cross-K/V projection)
        d_, L_, Ld_, nm_ = dims.n_audio_state, dims.n_audio_layer, dims.n_text_layer, dims.n_mels
        mel_bytes = B * (N_SAMPLES * 4 + 2 * nm_ * N_FRAMES * 4 + nm_ * N_FRAMES * 4)      # stage 1 read+write, stage 2 read+write
        enc_flop = B * (6 * nm_ * d_ * 3000 + 6 * d_ * d_ * 1500 + L_ * (24 * d_ * d_ * 1500 + 4 * 1500 * 1500 * d_)
                        + (0 if estream else 4 * d_ * d_ * 1500 * Ld_))          # no cross-K/V projection on the E-stream path
        rows = B * max(1, args.beam)
        t_avg = len(prompt) + steps_per_window / 2
        dec_bytes = (steps_per_window + len(prompt) - 1) * (B * Ld_ * (1 if estream else 2) * 1500 * d_ * es + rows * Ld_ * 2 * t_avg * d_ * es
                                                             + (16 * d_ * d_ * Ld_ + dims.n_vocab * d_) * es)
        out["stages"] = {
            "mel": {"ms": st_ms[0], "achieved_GBs": mel_bytes / st_ms[0] / 1e6, "peak_GBs": HBM_PEAK_GBS, "bound": "hbm"},
            "encoder": {"ms": st_ms[1], "achieved_TFLOPs": enc_flop / st_ms[1] / 1e9, "peak_TFLOPs": 2500.0, "bound": "mfma"},
            "decode": {"ms": st_ms[2], "achieved_GBs": dec_bytes / st_ms[2] / 1e6, "peak_GBs": HBM_PEAK_GBS, "bound": "hbm",
                       "ms_per_token_step": st_ms[2] / max(1, steps_per_window + len(prompt) - 1)},
        }
        out["stages"]["conditions"] = (f"one more batch after the timed region, ALONE on the GPU, on lane 0's engine (the benchmarked engine shape: "
                                       f"{eng.cross_key_splits} key split(s) per row of the stream kernel); in the timed region {lanes.n_lanes} batches overlap")
        # the whole step against the HBM roofline: algorithmic bytes of one batch (mel + every decode position's weights, encoder
        # output and self-attention cache; the encoder's own traffic is not counted: it is the MFMA-bound stage) over the measured
        # time per batch, and the encoder's FLOPs that run inside the same interval
        step_bytes = mel_bytes + dec_bytes
        out["roofline"]["step"] = {"bytes": step_bytes, "flops": enc_flop, "ms_per_step": elapsed / args.steps * 1e3,
                                   "hbm_GBs": step_bytes / (elapsed / args.steps) / 1e9,
                                   "hbm_frac": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                   "mfma_TFLOPs_in_the_same_interval": enc_flop / (elapsed / args.steps) / 1e12}
        if single is not None:
            single["kernel_achieved_GBs"] = bytes_per_launch / (single["kernel_avg_launch_ms"] * 1e-3) / 1e9 if single["kernel_launches_timed"] else 0.0
            single["kernel_frac"] = single["kernel_achieved_GBs"] / HBM_PEAK_GBS
            out["single_batch"] = single
        out["config"]["weights_shared_by_lanes"] = not args.separate_weights
        out["config"]["one_encoder_at_a_time"] = not args.overlap_encoders and args.lanes > 1
        out["config"]["weight_bytes_per_gpu"] = sum(e_i.weight_bytes for e_i in lanes.engines)
        out["config"]["workspace_bytes_per_gpu"] = sum(e_i.workspace_bytes for e_i in lanes.engines)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(dims, sd, tok, eng.suppress_ids, steps_per_window, args.cpu_decode_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    lanes.close()


if __name__ == "__main__":
    main()
