"""The two contracts of ss_process_device that let a caller do without ss_sync, against the reference's own code (oracle/_ref):

  SS_FLAG_STREAM_ORDERED   every stage of a call is on ss_stream before the call returns: a producer that refills the ONE input
                           buffer and a consumer that copies the ONE output set away, both enqueued on ss_stream, nothing else
  ss_input_wait            the default, overlapped path with a producer on a stream of its own that rotates three input buffers and
                           learns from the library when a buffer is dead

Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from parity import BAND, cand_set, check_plane, dont_care_limit, floor_tolerance

pytestmark = pytest.mark.gpu

CENTER = 145_000_000


def _ref(ref_mod, n, fs, iq, t):
    ref_mod.ref().orc_set_fft_backend(0)
    r = ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t)
    off = np.zeros(len(r["cands"]) + 1, np.int32)
    off[1:] = np.cumsum([len(c) for c in r["cands"]])
    return {"psd": r["psd"], "avg": r["avg"], "cand_off": off, "cand_idx": np.concatenate(r["cands"]).astype(np.int32)}


def _compare(got_psd, got_off, got_idx, ref, what):
    check_plane("psd", got_psd, ref["psd"], floor_tolerance(ref["psd"]))
    a, b = cand_set(got_off, got_idx), cand_set(ref["cand_off"], ref["cand_idx"])
    near = np.abs(ref["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in a ^ b if not near[f, i]]
    assert not outside, (what, sorted(outside)[:10])
    assert len(b) > 5_000 and len(a ^ b) <= dont_care_limit(len(b)), (what, len(b), len(a ^ b))
    print(f"\n[{what}] {len(b)} reference candidates, {len(a ^ b)} inside the {BAND} dB band")


@pytest.mark.parametrize("n,fs,nb,ncalls,fmt", [(8192, 2_048_000, 256, 8, "cf32"), (65536, 20_000_000, 32, 5, "cs8")])
def test_stream_ordered_producer_and_consumer_on_the_chains_stream(ref_mod, n, fs, nb, ncalls, fmt):
    import torch
    import hipapi
    dev = torch.device("cuda", 0)
    total = nb * ncalls
    learn, step_ms = (101, 20) if total > 1000 else (41, 50)  # the reference's 2000 ms of learning at 50 / 20 frames per second
    band = pkg.synth.SyntheticBand(n, seed=61, on_frame=learn + 29, off_frame=total - 10)
    if fmt == "cs8":
        iq8 = band.frames_cs8(total)
        iq = (iq8[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (iq8[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
        host = torch.from_numpy(iq8).pin_memory()
    else:
        iq = band.frames_cf32(total)
        host = torch.from_numpy(iq.view(np.float32)).pin_memory()
    t = (10_000 + step_ms * np.arange(total)).astype(np.int64)
    ref = _ref(ref_mod, n, fs, iq, t)
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, max_batch=nb, learn_frames=learn, flags=pkg.abi.SS_FLAG_STREAM_ORDERED,
                             in_format=pkg.abi.SS_FMT_CS8 if fmt == "cs8" else pkg.abi.SS_FMT_CF32)
    s = eng.stream_handle
    d_iq = torch.empty_like(host[:nb], device=dev)  # ONE input buffer, ONE output set: refilled / copied away on the chain's stream
    d_psd = torch.empty((nb, n), dtype=torch.float32, device=dev)
    d_off = torch.zeros(nb + 1, dtype=torch.int32, device=dev)
    d_idx = torch.empty(nb * 1024, dtype=torch.int32, device=dev)
    h_psd = torch.empty((ncalls, nb, n), dtype=torch.float32).pin_memory()
    h_off = torch.empty((ncalls, nb + 1), dtype=torch.int32).pin_memory()
    h_idx = torch.empty((ncalls, nb * 1024), dtype=torch.int32).pin_memory()
    torch.cuda.synchronize()
    in_bytes = host[:nb].numel() * host.element_size()
    for k in range(ncalls):
        hipapi.copy_async(d_iq.data_ptr(), host[k * nb:(k + 1) * nb].data_ptr(), in_bytes, hipapi.H2D, s)  # producer: ordered behind the previous call's last read
        eng.process_device(d_iq, nb, psd=d_psd, cand_off=d_off, cand_idx=d_idx)
        hipapi.copy_async(h_psd[k].data_ptr(), d_psd.data_ptr(), nb * n * 4, hipapi.D2H, s)               # consumer: ordered behind every stage of this call
        hipapi.copy_async(h_off[k].data_ptr(), d_off.data_ptr(), (nb + 1) * 4, hipapi.D2H, s)
        hipapi.copy_async(h_idx[k].data_ptr(), d_idx.data_ptr(), nb * 1024 * 4, hipapi.D2H, s)
    hipapi.stream_sync(s)  # (not ss_sync: a plain wait for the stream)
    counts = np.concatenate([np.diff(h_off[k].numpy()) for k in range(ncalls)])
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    idx = np.concatenate([h_idx[k].numpy()[:h_off[k].numpy()[-1]] for k in range(ncalls)])
    _compare(h_psd.numpy().reshape(total, n), off, idx, ref, f"SS_FLAG_STREAM_ORDERED, {n} points, {ncalls} x {nb} frames through one buffer set")
    st = eng.stats()
    assert st["calls_overlapped"] == 0 and st["calls_in_order"] == ncalls and not st["overlap"], st
    eng.close()


def test_input_wait_lets_a_producer_rotate_three_buffers_under_overlapped_calls(ref_mod):
    import torch
    import hipapi
    dev = torch.device("cuda", 0)
    n, fs, nb, ncalls, m = 8192, 2_048_000, 256, 14, 3
    band = pkg.synth.SyntheticBand(n, seed=62, on_frame=130, off_frame=nb * ncalls - 40)
    total = nb * ncalls
    iq = band.frames_cf32(total)
    t = (10_000 + 20 * np.arange(total)).astype(np.int64)
    ref = _ref(ref_mod, n, fs, iq, t)
    host = torch.from_numpy(iq.view(np.float32)).pin_memory()
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, max_batch=nb, learn_frames=101)
    chain = eng.stream_handle
    prod, ev = hipapi.stream_create(), hipapi.event_create()
    bufs = [torch.empty_like(host[:nb], device=dev) for _ in range(m)]
    outs = [dict(psd=torch.empty((nb, n), dtype=torch.float32, device=dev), off=torch.zeros(nb + 1, dtype=torch.int32, device=dev),
                 idx=torch.empty(nb * 1024, dtype=torch.int32, device=dev)) for _ in range(ncalls)]
    torch.cuda.synchronize()
    in_bytes = host[:nb].numel() * host.element_size()
    for k in range(ncalls):
        if k >= m:
            eng.input_wait(prod, m - 1)  # buffer k mod m held call k - m: dead once call k - m + 1's launch has read its tail
        hipapi.copy_async(bufs[k % m].data_ptr(), host[k * nb:(k + 1) * nb].data_ptr(), in_bytes, hipapi.H2D, prod)
        hipapi.stream_wait_stream(chain, prod, ev)  # work on ss_stream before a call (the producer of d_iq) is waited for by the call
        o = outs[k]
        eng.process_device(bufs[k % m], nb, psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"])
    eng.sync()
    hipapi.stream_sync(prod)
    st = eng.stats()
    assert st["calls_overlapped"] >= ncalls - 2 and not st["demoted"], st  # (the learning call runs in order)
    offs = [o["off"].cpu().numpy() for o in outs]
    counts = np.concatenate([np.diff(x) for x in offs])
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    idx = np.concatenate([o["idx"].cpu().numpy()[:x[-1]] for o, x in zip(outs, offs)])
    psd = np.concatenate([o["psd"].cpu().numpy() for o in outs])
    _compare(psd, off, idx, ref, f"ss_input_wait: {ncalls} x {nb} frames, {m} input buffers refilled in flight")
    eng.close()
    hipapi.event_destroy(ev)
    hipapi.stream_destroy(prod)


def test_stats_and_input_wait_say_what_they_are_asked():
    """ss_get_stats honours the caller's idea of the struct's size (fields are only ever appended); ss_input_wait refuses
    calls_back < 1 (the latest call's last frames are read once more by the call after it) and is a no-op behind the chain's own
    stream on a context whose calls run in order."""
    import ctypes as C
    eng = pkg.SpectrumEngine(512_000, CENTER, fft_size=2048, decim=1, max_batch=32)
    lib = eng._lib
    st = pkg.abi.SsStats()
    st.size = 24  # size + state + calls + calls_overlapped
    st.drains = 777
    assert lib.ss_get_stats(eng._h, C.byref(st)) == 0 and st.size == 24 and st.drains == 777  # nothing behind the 24 bytes is touched
    st.size = 4
    assert lib.ss_get_stats(eng._h, C.byref(st)) == pkg.abi.SS_ERR_INVALID
    with pytest.raises(pkg.abi.SpecscanError, match="calls_back"):
        eng.input_wait(None, 0)
    eng.input_wait(None, 1)
    full = eng.stats()
    assert full["calls"] == 0 and not full["overlap"] and not full["culling"], full
    eng.close()
