"""Nothing inside a launch waits without bound (csrc/scan_step.h, csrc/detect_fused.h). The one in-launch hand-over of the chain —
the tile-culling plan of the detect stage that rides on an 8192-point launch, from four plan workgroups to the workgroups that
evaluate the listed tiles — has a bounded wait: a consumer that has polled StepArgs::wait_limit times makes the plan of its list
itself. Here (1) the plan workgroups of the diagnostics build never publish anything (SS_HINT_MODE=3), so EVERY consumer has to
help itself, and the results must be those of the ordinary run, list by list; (2) the timed path of config 2 runs against the
reference with a handful of CUs (ROC_GLOBAL_CU_MASK / HSA_CU_MASK: fewer slots than a launch has waiting consumers), in processes
of their own. Needs an MI355X: run with -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CENTER = 145_000_000


def _run(eng, batches, n):
    import torch
    dev = torch.device("cuda", 0)
    d_iq = [torch.from_numpy(b.view(np.float32)).to(dev) for b in batches]
    outs = [dict(off=torch.zeros(b.shape[0] + 1, dtype=torch.int32, device=dev), idx=torch.empty(b.shape[0] * 1024, dtype=torch.int32, device=dev),
                 avg=torch.empty(b.shape[0] * 1024, dtype=torch.float32, device=dev)) for b in batches]
    torch.cuda.synchronize()
    for d, o in zip(d_iq, outs):
        eng.process_device(d, d.shape[0], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])
    eng.sync()
    res = []
    for o in outs:
        off = o["off"].cpu().numpy()
        res.append((off, o["idx"].cpu().numpy()[:off[-1]], o["avg"].cpu().numpy()[:off[-1]]))
    return res


def test_consumers_that_never_hear_from_the_plan_workgroups_make_the_plan_themselves(monkeypatch):
    n, fs, nb, ncalls = 8192, 2_048_000, 256, 9
    band = pkg.synth.SyntheticBand(n, seed=51, on_frame=300, off_frame=1500, period=1900)
    iq = band.frames_cf32(nb * ncalls)
    batches = [iq[k * nb:(k + 1) * nb] for k in range(ncalls)]
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, max_batch=nb, learn_frames=100)
    want = _run(eng, batches, n)
    st = eng.stats()
    assert st["culling"] and st["tiles_culled"] > 0 and st["wait_fallbacks"] == 0, st
    eng.close()
    monkeypatch.setenv("SS_HINT_MODE", "3")  # plan workgroups leave at once: nobody publishes a list
    monkeypatch.setenv("SS_WAIT_LIMIT", "6")
    pkg.engine.use_diag_library(True)
    try:
        eng2 = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, max_batch=nb, learn_frames=100)
        got = _run(eng2, batches, n)
        st2 = eng2.stats()
        eng2.close()
    finally:
        pkg.engine.use_diag_library(False)
    assert st2["wait_fallbacks"] > 100, st2  # (every workgroup that serves a list, in every launch that carries a planned stage)
    assert st2["tiles_tested"] == 0  # the plan role counted nothing: it never ran
    total = 0
    for k, ((o1, i1, a1), (o2, i2, a2)) in enumerate(zip(want, got)):
        assert (o1 == o2).all() and (i1 == i2).all() and (a1 == a2).all(), k
        total += int(o1[-1])
    assert total > 10_000
    print(f"\n[wait bound] {st2['wait_fallbacks']} workgroups made their list's plan themselves; {total} candidates identical")


@pytest.mark.parametrize("mask_env", [{"ROC_GLOBAL_CU_MASK": "0xf"}, {"ROC_GLOBAL_CU_MASK": "0xffffffff"}, {"HSA_CU_MASK": "0:0-7"}])
def test_the_timed_path_with_a_handful_of_cus(ref_mod, mask_env):
    """Config 2's timed path against the reference on a few CUs only — fewer residency slots than a launch has consumers that wait
    for a plan: completes (a hang would run into the timeout) and meets the contract."""
    env = dict(os.environ)
    env.update(mask_env)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_stated_configs.py") + "::test_config2_the_timed_path_against_the_reference"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-1000:]
