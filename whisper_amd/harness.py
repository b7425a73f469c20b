#!/usr/bin/env python3
"""Timing harness with the reference's CLI and output (ref:test_whisper.py:13-74, identical copy at
ref:cli_tests/test_whisper.py): load a model by name, transcribe one file, print text / language / confidence / seconds.

    python -m whisper_amd.harness audio.wav --model base --language en [--checkpoint model.pt] [--vocab vocab.tiktoken]

As in the reference the timer starts BEFORE the model is loaded (ref:test_whisper.py:15), so "Time taken" includes weight
upload; ``--repeat N`` additionally prints the steady-state time of N further calls.  Without ``--checkpoint`` the weights
are seeded random numbers of the named architecture (there is nothing to download offline) and the text is a rendering
of token ids.
"""
import argparse
import os
import time


def transcribe_audio(audio_file, model_name="base", language=None, checkpoint=None, vocab=None, precision="bf16", repeat=0,
                     mode="sequential"):
    start_time = time.time()
    print(f"Loading Whisper model '{model_name}'...")
    import torch
    from whisper_amd import transcribe as tx
    device = "cuda" if torch.cuda.is_available() else "cpu"
    tx.configure(model=model_name, checkpoint=checkpoint, vocab=vocab, precision=precision, max_batch=16,
                 allow_synthetic_weights=checkpoint is None)
    print(f"Transcribing {audio_file} on {device}...")
    # mode "sequential" = upstream's own long-form loop, what the reference's model.transcribe(audio_file, **options) runs
    # (ref:test_whisper.py:31); "windows" = the batched independent-window mode
    result = tx.transcribe(audio_file, language=language, mode=mode)          # language None = auto-detection
    torch.cuda.synchronize()
    text = result["text"].strip()
    language = result.get("language", "unknown")
    confidence = float(result.get("confidence", 0.0))
    elapsed = time.time() - start_time
    print("\nTranscription Results:")
    print("=" * 50)
    print(f"Text: {text}")
    print(f"Language: {language}")
    print(f"Confidence: {confidence:.4f}")
    print(f"Time taken: {elapsed:.2f} seconds")
    if repeat > 0:
        t0 = time.time()
        for _ in range(repeat):
            tx.transcribe(audio_file, language=language, mode=mode)
        torch.cuda.synchronize()
        print(f"Steady state: {(time.time() - t0) / repeat:.3f} seconds per call ({result['chunks_processed']} x 30 s windows)")
    print("=" * 50)
    tx.close()
    return text, language, confidence


def main():
    parser = argparse.ArgumentParser(description="Test Whisper speech recognition (MI355X engine)")
    parser.add_argument("audio_file", help="Path to a 16 kHz mono PCM16 WAV file")
    parser.add_argument("--model", default="base", choices=["tiny", "base", "small", "medium", "large", "large-v3", "turbo"],
                        help="Whisper model to use")
    parser.add_argument("--language", help='Language code (e.g., "en", "ru") or None for auto-detection')
    parser.add_argument("--checkpoint", help="local openai .pt or HF .safetensors file (default: seeded random weights)")
    parser.add_argument("--vocab", help="local tiktoken vocabulary file")
    parser.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    parser.add_argument("--repeat", type=int, default=0)
    parser.add_argument("--mode", default="sequential", choices=["sequential", "windows"])
    args = parser.parse_args()
    if not os.path.exists(args.audio_file):
        print(f"Error: Audio file '{args.audio_file}' not found")
        return
    transcribe_audio(args.audio_file, args.model, args.language, args.checkpoint, args.vocab, args.precision, args.repeat, args.mode)


if __name__ == "__main__":
    main()
