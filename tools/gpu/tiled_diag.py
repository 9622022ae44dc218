"""GPU box diagnostic: decode the golden multi-frame chunk streams and compare sample by sample with the oracle."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _lib as L
from tests.test_codec_gpu import api_decode, api_encode

t = json.load(open(os.path.join(L.GOLDEN, "tiled.json")))
inp = np.load(os.path.join(L.GOLDEN, "tiled_inputs.npz"))
for name in sorted(t["frames"]):
    if not name.startswith("odd_"):
        continue
    c = t["frames"][name]
    want = bytes.fromhex(c["stream_hex"])
    x = inp[c["input"]]
    d = api_decode(want).reshape(x.shape)
    o = np.asarray(L.orc_decode(want)).reshape(x.shape)
    bad = np.argwhere(d != o)
    print(name, "decode mismatches:", len(bad), "first", bad[:3].tolist(), "max", float(np.abs(d - o).max()), flush=True)
    if len(bad):
        rows = sorted(set((int(b[0]), int(b[1])) for b in bad))
        print("   rows (tile,y):", rows[:40], flush=True)
