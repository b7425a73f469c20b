"""The allocation-free entry point (wx_log_mel_static, include/whisper_amd.h) against the existing three-call form
(wx_mel_reset_max + wx_mel_stage1 + wx_mel_stage2) through the C-ABI: bit-identical output on two calls in a row, the
second with a smaller input (the library's static device scratch keeps the first call's partial maxima), the size limit is
an error and not a fault, and torch's allocator sees no allocation during the calls."""
import numpy as np
import pytest
import torch

from whisper_amd import checkpoint as ck

pytestmark = pytest.mark.gpu


def test_log_mel_static_matches_the_three_call_form_twice():
    from oracle import whisper_oracle as wo
    from whisper_amd import _lib
    from whisper_amd.engine import WhisperEngine
    for name in ("micro", "micro128"):                      # 80 and 128 mel bins
        dims = ck.dims_for(name)
        eng = WhisperEngine(dims, ck.synthetic_state_dict(dims, 0), precision="fp32", max_batch=1)
        try:
            big = torch.from_numpy(wo.varied_audio(7, 16000 * 95 + 321)).cuda()
            big[: 16000 * 40] *= 0.01                        # the maximum lives late in the first buffer ...
            small = torch.from_numpy(wo.varied_audio(8, 16000 * 11 + 57) * 0.03).cuda()     # ... and the second is quieter
            for pcm, padding in ((big, wo.N_SAMPLES), (small, wo.N_SAMPLES), (small, 0), (big, wo.N_SAMPLES)):
                want = eng.log_mel(pcm, padding=padding)
                out = torch.full((dims.n_mels, want.shape[1] + 5), 7.0, device="cuda")      # ld > frame count: padding columns stay
                torch.cuda.synchronize()
                before = torch.cuda.memory_stats()["allocation.all.allocated"]
                eng.log_mel_static(pcm, out, padding=padding)
                torch.cuda.synchronize()
                assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
                assert torch.equal(out[:, :want.shape[1]], want)
                assert (out[:, want.shape[1]:] == 7.0).all()
            # against the oracle too (1e-4 relative, the mel bar)
            ref = wo.log_mel_spectrogram(small.cpu().numpy(), dims.n_mels, padding=wo.N_SAMPLES).numpy()
            out = torch.empty((dims.n_mels, ref.shape[1]), device="cuda")
            eng.log_mel_static(small, out)
            assert np.abs(out.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
            # beyond the static scratch: a clear error, nothing launched, and the entry point still works afterwards
            n_big = (65536 - 64) * 8 * 160 + 160
            huge = torch.zeros(n_big, device="cuda")
            sink = torch.empty((dims.n_mels, n_big // 160), device="cuda")
            with pytest.raises(_lib.WxError, match="static scratch"):
                eng.log_mel_static(huge, sink, padding=0)
            del huge, sink
            eng.log_mel_static(small, out)
            assert np.abs(out.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
        finally:
            eng.close()
