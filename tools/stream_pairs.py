"""Counts repeated starting (state, increment) pairs of the samples of a frame, as rtx_debug_streams reports them
(torch sorts them on the GPU; torch is imported before the library, whose HIP runtime it then shares).
Prints one JSON object.  usage: python tools/stream_pairs.py WIDTH HEIGHT SPP RNG_STREAMS [STEP]"""
import json
import sys

import torch  # noqa: E402  (first: see above)

from raytracer_amd.capi import Config, Renderer


def count_repeats(t):
    """Elements of the int64 tensor t that repeat an earlier one: the tensor is split into 64 buckets by its top
    six bits, slice by slice, and every bucket sorted on its own (a sort of the whole 34 GB tensor, or of an
    eighth of it, is more than torch.sort takes)."""
    repeats, slice_len = 0, 1 << 27
    for b in range(64):
        parts = []
        for a in range(0, t.numel(), slice_len):
            sl = t[a:a + slice_len]
            parts.append(sl[((sl >> 58) & 63) == b])
        part, _ = torch.sort(torch.cat(parts))
        repeats += int((part[1:] == part[:-1]).sum().item())
        del part, parts
    return repeats


W, H, spp, mode = (int(a) for a in sys.argv[1:5])
step = int(sys.argv[5]) if len(sys.argv) > 5 else 20
with Renderer(Config(width=W, height=H, rng_streams=mode, pool_paths=1 << 12)) as r:
    pairs = torch.empty(W * H * spp, dtype=torch.int64, device="cuda")
    for base in range(0, spp, step):
        r.debug_streams(pairs[W * H * base:], min(step, spp - base), sample_base=base, seed=0x5EED)
    torch.cuda.synchronize()
    inc = (pairs >> 32) & 0xffffffff
    out = {"samples": W * H * spp, "rng_streams": mode, "all_increments_odd": bool(int((inc & 1).min().item()) == 1),
           "distinct_increments_in_first_million": int(torch.unique(inc[:1 << 20]).numel()),
           "first_increment": int(inc[0].item()), "first_pairs": [int(v) & (2**64 - 1) for v in pairs[:4].cpu()],
           "at": {str(k): int(pairs[k].item()) & (2**64 - 1) for k in (0, 1, W * H - 1, W * H, min(2 * W * H + 1234, W * H * spp - 1))},
           "repeats": count_repeats(pairs)}
print(json.dumps(out))
