#!/usr/bin/env python3
"""MDCT / IMDCT kernel timing at the literal shape (B clips of T = 196608, N = 512, hop = 256): HIP events over `reps`
launches, algorithmic bytes (4T + 4 n_frames N forward, 4 n_frames N + 4 out_len inverse) over the time.
MFC_MDCT_GENERIC=1 selects the generic power-of-two kernels of csrc/mdct.hip for an A/B in a second process."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meanflow_audio_codec_amd.preprocessing import imdct, mdct   # noqa: E402


def main(B=128, T=196608, N=512, hop=256, reps=30):
    x = 0.1 * torch.randn(B, T, device="cuda")
    X = mdct(x, N, hop)
    y = imdct(X, N, hop)
    nf = X.shape[1]
    out = {"B": B, "T": T, "N": N, "hop": hop, "generic": os.environ.get("MFC_MDCT_GENERIC") == "1"}
    for name, fn, nbytes in (("fwd", lambda: mdct(x, N, hop), 4 * B * (T + nf * N)),
                             ("inv", lambda: imdct(X, N, hop), 4 * B * (nf * N + y.shape[1]))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        out[name] = {"ms": round(ms, 4), "GB_per_s": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 4)}
    err = (y[:, 1024:766 * hop] - 2 * x[:, 1024:766 * hop]).abs().max().item()
    out["roundtrip_err"] = err
    print(json.dumps(out))


if __name__ == "__main__":
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 128)
