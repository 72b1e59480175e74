#!/bin/bash
# round 5, session 11: where the 8192-point transform's extra distance to fp64 comes from (engine / reference rms 1.24 over all bins):
# the parity sample with the twiddle factors as they ship (TW = 2: wave-uniform x per-lane products), with the pass-3 factors from
# global tables of single-rounded entries (TW = 1) and with every table from global memory (TW = 0)
OUT=gpurun_out/r05_s11
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
for tw in 2 1 0; do
  SS_FFT_TW=$tw timeout 300 python bench.py --diag-lib --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-live-pmc --sub > $OUT/tw$tw.json 2> $OUT/tw$tw.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/tw$tw.json').read().strip().splitlines()[-1])
    p = j['parity']
    print('TW=$tw', j['ms_per_step'], p.get('failed'), p.get('all_bins_vs_fp64_fft_dB'), p['abs_err_dB']['psd'])
except Exception as e:
    print('TW=$tw ERR', e, open('$OUT/tw$tw.err').read()[-500:])
PY
done
