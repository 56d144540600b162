"""One-off (this container only: reads /root/reference/tests/data): is any *signal.tensor the reference's tests hold the
signal of a read in one of its POD5 fixtures?  If so that tensor pins the decoded VALUES of SURVEY row f2.

Decoding here is the CPU oracle's svb16 (oracle_py.svb16_decode) after zstd — the same stages pod5.py runs
(pod5_get_read_complete_signal, data_loader/DataLoader.cpp:163-170)."""
import glob
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
from dorado_amd import pod5  # noqa: E402
import oracle_py  # noqa: E402

ROOT = "/root/reference/tests/data"


def load_tensor(path):
    try:
        t = torch.load(path, weights_only=False)
    except Exception:
        t = torch.jit.load(path)
    if hasattr(t, "parameters") and not isinstance(t, torch.Tensor):
        ps = list(t.parameters()) + list(t.buffers())
        t = ps[0]
    return t


def main():
    tensors = {}
    for p in sorted(glob.glob(ROOT + "/**/*signal*.tensor", recursive=True)):
        t = load_tensor(p)
        a = t.detach().cpu().numpy().reshape(-1)
        tensors[p] = a
        print(f"{p[len(ROOT) + 1:]}: dtype {t.dtype} n {a.size} first {a[:6].tolist()}")
    by_len = {}
    for p, a in tensors.items():
        by_len.setdefault(a.size, []).append(p)
    hits = 0
    nreads = 0
    for f in sorted(glob.glob(ROOT + "/**/*.pod5", recursive=True)):
        try:
            pf = pod5.Pod5File(f)
        except Exception as e:  # noqa: BLE001
            print("skip", f[len(ROOT) + 1:], type(e).__name__)
            continue
        for r in pf.reads():
            nreads += 1
            # candidate by length (exact, or the tensor is a trimmed piece: check containment for short tensors only)
            cands = by_len.get(r.num_samples, [])
            if not cands:
                continue
            streams, ns = pf.inflated_rows(r.signal_rows)
            raw = np.concatenate([oracle_py.svb16_decode(np.frombuffer(s, np.uint8), n) for s, n in zip(streams, ns)])
            for p in cands:
                a = tensors[p]
                same = np.array_equal(raw.astype(np.float64), a.astype(np.float64))
                print(f"length match: read {r.read_id} of {f[len(ROOT) + 1:]} vs {p[len(ROOT) + 1:]}: values equal = {same}")
                hits += same
    print(f"{nreads} reads in the POD5 fixtures, {len(tensors)} signal tensors, {hits} value-identical pairs")


if __name__ == "__main__":
    main()
