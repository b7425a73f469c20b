"""whisper_amd — MI355X-native (gfx950) implementation of the local Whisper path of alexk136/whisper.

Public surface:
    whisper_amd.transcribe.transcribe(audio, language=None, prompt=None) -> {"text", "confidence", "chunks_processed"}
    whisper_amd.speech_recognition  — drop-in for app/transcription/speech_recognition.py
    whisper_amd.engine.WhisperEngine — handle over the C-ABI library (include/whisper_amd.h)
"""
__version__ = "0.1.0"
