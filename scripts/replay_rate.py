"""End-to-end (file + PCIe inclusive) scan rate of the replay front end on this box, next to the HBM-resident rate
bench.py reports. Writes a synthetic dump under /tmp, replays it through ss_feed_* and through ss_process.
    python scripts/replay_rate.py [--frames 4096] [--fft 8192] [--fmt cf32|cs8] [--depth 3]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtl_sdr_scanner_cpp_amd as pkg  # noqa: E402
from rtl_sdr_scanner_cpp_amd import replay  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--fft", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--fmt", default="cf32")
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--read-threads", type=int, default=4)
    a = ap.parse_args()
    fs, center = 2_048_000, 145_000_000
    band = pkg.synth.SyntheticBand(a.fft, seed=1, on_frame=150, off_frame=a.frames - 50)
    chunk = 256
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        ext = "fc" if a.fmt == "cf32" else "cs8"
        path = os.path.join(d, replay.make_raw_file_name("full", ext, center, fs)[2:])
        sink = replay.RawFileSink(8 if a.fmt == "cf32" else 2)
        sink.start_recording(path)
        whole = []
        for f0 in range(0, a.frames, chunk):
            band2 = pkg.synth.SyntheticBand(a.fft, seed=1 + f0, on_frame=0, off_frame=chunk)
            fr = band2.frames_cf32(chunk) if a.fmt == "cf32" else band2.frames_cs8(chunk)
            sink.work(fr)
            whole.append(fr)
        sink.close()
        whole = np.concatenate(whole)
        info = replay.parse_raw_file_name(path)
        kw = dict(fft_size=a.fft, decim=1, learn_frames=100, max_batch=a.batch, **replay.engine_overrides_for(info))
        res = {"file_bytes": os.path.getsize(path), "frames": a.frames, "fft": a.fft, "fmt": a.fmt, "batch": a.batch, "depth": a.depth}
        best = 0.0
        for _ in range(a.repeat):
            eng = pkg.SpectrumEngine(fs, center, **kw)
            st = replay.ReplayStats()
            for _r in replay.replay_file(eng, path, batch=a.batch, depth=a.depth, stats=st, read_threads=a.read_threads):
                pass
            best = max(best, st.msamples_per_sec(a.fft))
            res["replay_read_seconds"] = round(st.read_seconds, 4)
            res["replay_seconds"] = round(st.seconds, 4)
            eng.close()
        res["replay_feed_MSps"] = round(best, 1)
        res["read_threads"] = a.read_threads
        # the feed alone: slots are submitted as they are (no file read), i.e. a source that writes into pinned memory itself
        best = 0.0
        for _ in range(a.repeat):
            eng = pkg.SpectrumEngine(fs, center, **kw)
            feed = eng.feed(depth=a.depth, cand_cap=1 << 20)
            nb = a.frames // a.batch
            for k in range(a.depth):
                feed.acquire()[:] = whole[: a.batch]
                feed.submit(a.batch)
            for k in range(a.depth):
                feed.collect()
            t0 = time.perf_counter()
            sub = done = 0
            while done < nb:
                while sub < nb and feed.pending < a.depth:
                    feed.acquire()
                    feed.submit(a.batch)
                    sub += 1
                feed.collect()
                done += 1
            dt = time.perf_counter() - t0
            best = max(best, nb * a.batch * a.fft / dt / 1e6)
            feed.close()
            eng.close()
        res["feed_only_MSps"] = round(best, 1)
        best = 0.0
        for _ in range(a.repeat):
            eng = pkg.SpectrumEngine(fs, center, **kw)
            t0 = time.perf_counter()
            for f0 in range(0, a.frames, a.batch):
                eng.process(whole[f0:f0 + a.batch], want=())
            dt = time.perf_counter() - t0
            best = max(best, a.frames * a.fft / dt / 1e6)
            eng.close()
        res["sync_ss_process_MSps"] = round(best, 1)
        bps = 8 if a.fmt == "cf32" else 2
        res["replay_feed_GBps_in"] = round(res["replay_feed_MSps"] * bps / 1e3, 2)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
