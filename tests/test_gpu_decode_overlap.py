"""Goes to tests/ together with decode_overlap.patch (tools/next_round/README.md).  pytest -m gpu.
The decoder on its own stream (mibc_set_decode_overlap) must not change a byte: several consecutive batches with DIFFERENT inputs
are run back to back without a sync in between — so the decoder of batch i really runs beside the network of batch i + 1 and
every buffer hazard (scores parity, decoder scratch, output planes) is live — once with the switch off, once with it on."""
import numpy as np
import pytest

from dorado_amd import capi, config, synth

pytestmark = pytest.mark.gpu


def _planes(eng, batches, t_in, overlap):
    n = batches[0].shape[0]
    T = eng.output_steps(t_in)
    eng.reserve(n, t_in)
    eng.set_decode_overlap(overlap)
    d_in = [eng.device_alloc(b.nbytes) for b in batches]
    d_out = [eng.device_alloc(3 * n * T) for _ in batches]
    for d, b in zip(d_in, batches):
        eng.h2d(d, b)
    eng.sync()
    for d, o in zip(d_in, d_out):                 # no sync between the calls
        eng.call_device(d, n, t_in, o)
    eng.sync()
    outs = []
    for o in d_out:
        h = np.zeros((3, n, T), np.int8)
        eng.d2h(h, o)
        outs.append(h)
    for p in d_in + d_out:
        eng.device_free(p)
    eng.set_decode_overlap(False)
    return outs


@pytest.mark.parametrize("model", ["lstm_one_launch", "lstm_sub_batches", "tx"])
def test_decode_overlap_changes_nothing(model):
    if model == "tx":
        cfg, n, t_in = config.tiny_tx(), 64, 1536
    else:
        cfg = config.tiny(256 if model == "lstm_sub_batches" else 128, 4)
        n, t_in = (512 if model == "lstm_sub_batches" else 128), 1200
    ws = synth.make_weights(cfg, seed=5)
    eng = capi.Engine(cfg, ws)
    batches = [synth.make_signal(n, t_in, seed=100 + i) for i in range(5)]
    a = _planes(eng, batches, t_in, False)
    b = _planes(eng, batches, t_in, True)
    c = _planes(eng, batches[::-1], t_in, True)[::-1]      # another order: parity of the scores buffers flips per batch
    for i in range(len(batches)):
        assert (a[i] == b[i]).all() and (a[i] == c[i]).all(), (model, i)
    assert sum(int(x[0].sum()) for x in a) > 1000          # bases were called
    # the synchronous host call and the two-slot path go through the same switch
    eng.set_decode_overlap(True)
    got = eng.call(batches[0])
    eng.set_decode_overlap(False)
    want = eng.call(batches[0])
    for (s1, q1, m1), (s2, q2, m2) in zip(got, want):
        assert s1 == s2 and q1 == q2 and (m1 == m2).all()
    # the synchronous int16 host call, and `call_device; d2h(out)` WITHOUT a sync in between, join the decoder stream too
    raw = np.round(batches[1].astype(np.float32) * 200.0).astype(np.int16)
    ss = np.tile(np.array([[0.0, 200.0]], np.float32), (n, 1))
    want16 = eng.call_i16(raw, ss)
    eng.set_decode_overlap(True)
    got16 = eng.call_i16(raw, ss)
    T = eng.output_steps(t_in)
    d_in, d_out = eng.device_alloc(batches[2].nbytes), eng.device_alloc(3 * n * T)
    eng.h2d(d_in, batches[2])
    for _ in range(3):                                      # decoders of earlier calls still in flight on the decoder stream
        eng.call_device(d_in, n, t_in, d_out)
    h = np.zeros((3, n, T), np.int8)
    eng.d2h(h, d_out)                                       # no eng.sync() before the copy
    eng.set_decode_overlap(False)
    eng.device_free(d_in)
    eng.device_free(d_out)
    for (s1, q1, m1), (s2, q2, m2) in zip(got16, want16):
        assert s1 == s2 and q1 == q2 and (m1 == m2).all()
    assert (h == a[2]).all()
    eng.close()
